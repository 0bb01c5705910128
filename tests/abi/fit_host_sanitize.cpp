// Host-only build of the optimiser of gpcc_grid_loglik (gpcc.jl_amd/csrc/gpcc_fit.h) for AddressSanitizer / UBSan on
// the CPU (GPU sanitizers are not available on the pool).  Built and run by tests/test_host_cpu.py.
#include "../../gpcc.jl_amd/csrc/gpcc_fit.h"

#include <cstdio>

static int rosen(void *, long K, const long *pidx, const double *X, double *f)
{
    for (long i = 0; i < K; ++i) {
        const double x = X[2 * i], y = X[2 * i + 1];
        f[i] = (pidx[i] % 7 == 3 && x < -1.0) ? std::numeric_limits<double>::quiet_NaN()   // a rejected region for some problems
                                                : 100.0 * (y - x * x) * (y - x * x) + (1.0 - x) * (1.0 - x);
    }
    return 0;
}

static int bowl5(void *, long K, const long *, const double *X, double *f)
{
    for (long i = 0; i < K; ++i) {
        double s = 0.0;
        for (int d = 0; d < 5; ++d) s += (d + 1) * (X[5 * i + d] - 0.1 * d) * (X[5 * i + d] - 0.1 * d);
        f[i] = s;
    }
    return 0;
}

int main()
{
    {
        const long P = 257;
        std::vector<double> x0(P * 2), xmin(P * 2), fmin(P);
        gpccfit::Rng rg(5);
        for (auto &v : x0) v = 4.0 * rg.uniform() - 2.0;
        gpccfit::BatchedNelderMead nm(P, 2, 4000, 1e-10);
        if (nm.run(rosen, nullptr, x0.data(), xmin.data(), fmin.data())) return 1;
        long good = 0;
        for (long p = 0; p < P; ++p) good += fmin[p] < 1e-6;
        std::printf("rosenbrock: %ld of %ld converged, %lld evaluations in %lld rounds\n", good, P, nm.f_calls, nm.rounds);
        if (good < P * 9 / 10) return 2;
    }
    {   // speculative rounds (round 3): the same trajectories as the plain rounds, bit for bit, in fewer rounds
        const long P = 61;
        std::vector<double> x0(P * 2), xa(P * 2), fa(P), xb(P * 2), fb(P);
        gpccfit::Rng rg(9);
        for (auto &v : x0) v = 4.0 * rg.uniform() - 2.0;
        gpccfit::BatchedNelderMead plain(P, 2, 500, 1e-10), spec(P, 2, 500, 1e-10);
        spec.speculate_max = 1024;
        if (plain.run(rosen, nullptr, x0.data(), xa.data(), fa.data()) || spec.run(rosen, nullptr, x0.data(), xb.data(), fb.data())) return 8;
        for (long p = 0; p < P; ++p)
            if (fa[p] != fb[p] || xa[2 * p] != xb[2 * p] || xa[2 * p + 1] != xb[2 * p + 1] || plain.it[p] != spec.it[p]) return 9;
        std::printf("speculative rounds: %lld rounds instead of %lld, %lld evaluations instead of %lld\n", spec.rounds, plain.rounds,
                    spec.f_calls, plain.f_calls);
        if (!(spec.rounds < plain.rounds && spec.f_calls > plain.f_calls)) return 10;
    }
    {
        const long P = 31;
        std::vector<double> x0(P * 5, 1.0), xmin(P * 5), fmin(P);
        gpccfit::BatchedNelderMead nm(P, 5, 3000, 1e-9);
        if (nm.run(bowl5, nullptr, x0.data(), xmin.data(), fmin.data())) return 3;
        for (long p = 0; p < P; ++p)
            if (!(fmin[p] < 1e-6)) return 4;
        std::printf("bowl5: ok\n");
    }
    {
        double vary[3] = {1.0, 2.5, 0.3}, out[4 * 6 * 4];
        gpccfit::initial_params(3, 4, 6, 0.1, 300.0, 42, vary, out);
        for (double v : out)
            if (!std::isfinite(v)) return 5;
        for (int i = 0; i < 4 * 6; ++i) {
            const double rho = gpccfit::transformbetween(out[i * 4 + 3], 0.1, 300.0);
            if (!(rho > 0.1 && rho < 300.0)) return 6;
            for (int l = 0; l < 3; ++l) {
                const double a = gpccfit::makepositive(out[i * 4 + l]);
                if (!(a > 0.79 * vary[l] && a < 1.21 * vary[l])) return 7;
            }
        }
        std::printf("initial_params: ok\n");
    }
    {   // the round-robin deal of a grid over n devices and its reassembly from the gathered blocks (multi_grid_loglik): every grid
        // point comes back exactly once, in its own slot, with its own values -- G not divisible by n, G < n, G = 0, one device
        long cases = 0;
        for (int L = 1; L <= 3; ++L)
            for (int n = 1; n <= 5; ++n)
                for (long G = 0; G <= 17; ++G) {
                    const int W = L + 4;
                    const long blk = (G + n - 1) / n;
                    std::vector<double> gathered((size_t)n * (blk > 0 ? blk : 1) * W, -777.0);
                    long total = 0;
                    for (int i = 0; i < n; ++i) {
                        const long Gi = gpccfit::deal_count(G, n, i);
                        total += Gi;
                        if (Gi > blk) return 11;
                        std::vector<double> ll(Gi), rho(Gi), alpha((size_t)Gi * L);
                        std::vector<int> info(Gi), its(Gi);
                        for (long j = 0; j < Gi; ++j) {
                            const long g = gpccfit::deal_global(n, i, j);
                            if (g < 0 || g >= G) return 12;
                            ll[j] = -1000.0 - g; info[j] = (int)(g % 3); its[j] = (int)(100 + g); rho[j] = 0.5 + g;
                            for (int l = 0; l < L; ++l) alpha[(size_t)j * L + l] = 10.0 * g + l;
                        }
                        gpccfit::deal_pack_rows(blk, L, Gi, ll.data(), info.data(), its.data(), rho.data(), alpha.data(), gathered.data() + (size_t)i * blk * W);
                        for (long j = Gi; j < blk; ++j)
                            if (!std::isnan(gathered[((size_t)i * blk + j) * W])) return 13;   // padding rows are marked
                    }
                    if (total != G) return 14;
                    std::vector<double> lo(G + 1, 7.0), ro(G + 1, 7.0), ao((size_t)(G + 1) * L, 7.0);
                    std::vector<int> io(G + 1, 7), to(G + 1, 7);
                    gpccfit::deal_scatter(G, n, L, blk, gathered.data(), lo.data(), io.data(), to.data(), ro.data(), ao.data());
                    for (long g = 0; g < G; ++g) {
                        if (lo[g] != -1000.0 - g || io[g] != (int)(g % 3) || to[g] != (int)(100 + g) || ro[g] != 0.5 + g) return 15;
                        for (int l = 0; l < L; ++l)
                            if (ao[(size_t)g * L + l] != 10.0 * g + l) return 16;
                    }
                    if (lo[G] != 7.0 || io[G] != 7 || ao[(size_t)G * L] != 7.0) return 17;   // nothing written past the grid
                    ++cases;
                }
        std::printf("round-robin deal: %ld cases ok\n", cases);
    }
    return 0;
}
