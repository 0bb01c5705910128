// chain_queue_check.cpp -- host check of the job order of the persistent few-evaluation launch (gpcc.jl_amd/csrc/gpcc_chain_queue.h):
// for every matrix size (nt = 2 .. 64 tile rows: N <= 8192, all the default policy ever gives to the launch), with and without helper workgroups, with whole-tile and quarter-tile "next step"
// updates, with bulk jobs of at most 1, 2, 4 and 8 columns:
//   (1) coverage: every tile (I,k), I > k, is solved exactly once per quarter (by queue jobs, or -- tile (k+1,k) with helpers -- by the
//       chain's helpers); every trailing tile (I,J), k < J <= I, except the chain's own (k+1,k+1), is updated by column k exactly
//       once (a whole-tile job -- of one column, or of an aligned block of 2, 4 or 8 columns at once) or by exactly its four row quarters;
//   (2) order: every input of a queue job is produced by a job EARLIER in the order, or by the chain -- and whatever the chain
//       itself still needs from the queue at that point is earlier too.  That is all the launch's freedom from deadlock rests on:
//       the oldest unfinished job can always run, however few workgroups are resident.
// g++ -std=c++17 -O1 tests/abi/chain_queue_check.cpp -o chain_queue_check && ./chain_queue_check
#include <cstdio>
#include <cstdlib>
#include <map>
#include <tuple>
#include <vector>

#include "../../gpcc.jl_amd/csrc/gpcc_chain_queue.h"

static int fails = 0;
#define CHECK(cond, ...)                         \
    do {                                         \
        if (!(cond)) {                           \
            if (fails < 20) { std::printf("FAIL: " __VA_ARGS__); std::printf("\n"); } \
            ++fails;                             \
        }                                        \
    } while (0)

// nt = 64 is the largest matrix the default policy gives to the persistent launch (one evaluation x 64^2 = chain_work_max)
#ifndef NT_MAX
#define NT_MAX 64
#endif

int main()
{
    long jobs_total = 0, pair_jobs = 0;
    for (int nt = 2; nt <= NT_MAX; ++nt)
        for (int helpers = 0; helpers < 2; ++helpers)
            for (int quarters = 0; quarters < 2; ++quarters)
            for (int pairs = 1; pairs <= GPCC_CHAIN_MAX_BATCH; pairs *= 2) {   // (the widest column block of a bulk job)
                // order index of every queue job
                std::map<std::tuple<int, int, int>, long> solve_at;        // (I, k, q) -> index
                std::map<std::tuple<int, int, int, int>, long> upd_at;     // (I, J, k, q or -1 for a whole tile) -> index
                long idx = 0;
                struct Rec { GpccChainJob jb; long at; };
                std::vector<Rec> all;
                for (int ks = 0; ks < nt; ++ks) {
                    const int len = gpcc_chain_list_len(nt, ks, helpers, quarters, pairs);
                    CHECK(len >= 0, "nt %d ks %d: negative list length", nt, ks);
                    for (int jj = 0; jj < len; ++jj, ++idx) {
                        const GpccChainJob jb = gpcc_chain_decode(nt, ks, jj, helpers, quarters, pairs);
                        all.push_back({jb, idx});
                        CHECK(jb.kind >= 1 && jb.kind <= 4 && (pairs > 1 || jb.kind != 4), "nt %d ks %d jj %d: kind %d", nt, ks, jj, jb.kind);
                        if (jb.kind == 4) ++pair_jobs;
                        CHECK(jb.k >= 0 && jb.k < nt && jb.I > jb.k && jb.I < nt, "nt %d ks %d jj %d: tile row %d of step %d", nt, ks, jj, jb.I, jb.k);
                        if (jb.kind == 1) {
                            CHECK(jb.q >= 0 && jb.q < 4 && !(helpers && jb.I == jb.k + 1), "nt %d: solve (%d,%d,%d)", nt, jb.I, jb.k, jb.q);
                            CHECK(solve_at.emplace(std::make_tuple(jb.I, jb.k, jb.q), idx).second, "nt %d: solve (%d,%d,%d) twice", nt, jb.I, jb.k, jb.q);
                        } else if (jb.kind == 4) {   // the aligned block of q columns k .. k + q - 1 in one job, queued when its last column is solved
                            CHECK((jb.q == 2 || jb.q == 4 || jb.q == 8) && jb.q <= pairs && jb.k % jb.q == 0 && jb.k + jb.q == ks, "nt %d ks %d: block (%d, %d)", nt, ks, jb.k, jb.q);
                            CHECK(jb.J > jb.k + jb.q - 1 && jb.J <= jb.I && jb.I < nt, "nt %d: %d-column update (%d,%d) from %d", nt, jb.q, jb.I, jb.J, jb.k);
                            for (int col = jb.k; col < jb.k + jb.q; ++col)
                                CHECK(upd_at.emplace(std::make_tuple(jb.I, jb.J, col, -1), idx).second, "nt %d: update (%d,%d) by %d twice", nt, jb.I, jb.J, col);
                        } else {
                            CHECK(jb.J > jb.k && jb.J <= jb.I && !(jb.I == jb.k + 1 && jb.J == jb.k + 1), "nt %d: update (%d,%d) by %d", nt, jb.I, jb.J, jb.k);
                            const int q = jb.kind == 3 ? jb.q : -1;
                            CHECK(upd_at.emplace(std::make_tuple(jb.I, jb.J, jb.k, q), idx).second, "nt %d: update (%d,%d) by %d quarter %d twice", nt, jb.I, jb.J, jb.k, q);
                        }
                    }
                }
                jobs_total += idx;
                // (1) coverage
                auto solve_done_at = [&](int I, int k) -> long {   // index of the last of the four quarter solves; -1: the chain's helpers do it
                    if (helpers && I == k + 1) return -1;
                    long last = -2;
                    for (int q = 0; q < 4; ++q) {
                        auto it = solve_at.find(std::make_tuple(I, k, q));
                        if (it == solve_at.end()) return -2;
                        if (it->second > last) last = it->second;
                    }
                    return last;
                };
                auto upd_done_at = [&](int I, int J, int k) -> long {   // index of the job (or the last quarter) that applies column k to tile (I,J); -2: missing
                    auto w = upd_at.find(std::make_tuple(I, J, k, -1));
                    long last = -2;
                    int nq = 0;
                    for (int q = 0; q < 4; ++q) {
                        auto it = upd_at.find(std::make_tuple(I, J, k, q));
                        if (it != upd_at.end()) { ++nq; if (it->second > last) last = it->second; }
                    }
                    if (w != upd_at.end()) return nq == 0 ? w->second : -3;   // whole AND quarters: wrong
                    return nq == 4 ? last : -2;
                };
                for (int k = 0; k < nt; ++k)
                    for (int I = k + 1; I < nt; ++I) {
                        CHECK(solve_done_at(I, k) != -2, "nt %d helpers %d: tile (%d,%d) not solved by four quarters", nt, helpers, I, k);
                        for (int J = k + 1; J <= I; ++J) {
                            if (I == k + 1 && J == k + 1) continue;   // the chain's own tile
                            CHECK(upd_done_at(I, J, k) >= 0, "nt %d helpers %d quarters %d: tile (%d,%d) by column %d: %ld", nt, helpers, quarters, I, J, k, upd_done_at(I, J, k));
                        }
                    }
                // (2) order.  What the chain needs from the queue, transitively, as the largest order index:
                //     H(k): the helpers' solve of tile (k+1,k) [or, without helpers, nothing: those solves are queue jobs];  D(k): diagonal step k
                std::vector<long> D(nt, -1), H(nt, -1);
                for (int k = 0; k < nt; ++k) {
                    long d = k ? D[k - 1] : -1;
                    if (k >= 1) {
                        // tile (k,k): its updates by columns 0 .. k-2 are queue jobs (column k-1 is folded in by the chain itself), L(k,k-1) comes
                        // from the solves of step k-1
                        for (int col = 0; col + 1 < k; ++col) d = std::max(d, upd_done_at(k, k, col));
                        const long s = solve_done_at(k, k - 1);
                        d = std::max(d, s == -1 ? H[k - 1] : s);
                    }
                    D[k] = d;
                    if (k + 1 < nt) {   // helpers: tile (k+1,k) wants its updates by columns < k and the diagonal step k
                        long h = d;
                        for (int col = 0; col < k; ++col) h = std::max(h, upd_done_at(k + 1, k, col));
                        H[k] = h;
                    }
                }
                for (const Rec &r : all) {
                    const GpccChainJob &jb = r.jb;
                    if (jb.kind == 1) {   // solve (I,k,q): tile (I,k) updated by every column < k; row blocks of L_kk from diagonal step k
                        for (int col = 0; col < jb.k; ++col)
                            CHECK(upd_done_at(jb.I, jb.k, col) < r.at, "nt %d: solve (%d,%d) at %ld before its update by column %d (%ld)", nt, jb.I, jb.k, r.at, col, upd_done_at(jb.I, jb.k, col));
                        CHECK(D[jb.k] < r.at, "nt %d: solve (%d,%d) at %ld, but diagonal step %d needs queue job %ld", nt, jb.I, jb.k, r.at, jb.k, D[jb.k]);
                    } else {              // update (I,J) by column k (kind 4: and k + 1): both column tiles solved, the tile updated by every column < k
                        for (int col = jb.k; col < jb.k + (jb.kind == 4 ? jb.q : 1); ++col)
                        for (int T : {jb.I, jb.J}) {
                            const long s = solve_done_at(T, col);
                            CHECK((s == -1 ? H[col] : s) < r.at, "nt %d helpers %d: update (%d,%d) by %d at %ld before the solve of (%d,%d)", nt, helpers, jb.I, jb.J, col, r.at, T, col);
                        }
                        for (int col = 0; col < jb.k; ++col)
                            CHECK(upd_done_at(jb.I, jb.J, col) < r.at, "nt %d: update (%d,%d) by %d at %ld before column %d", nt, jb.I, jb.J, jb.k, r.at, col);
                    }
                }
            }
    std::printf("chain queue: %ld jobs (%ld of several columns) over nt = 2 .. %d x helpers x quarters x widest block 1, 2, 4, 8 checked, %d failures\n", jobs_total, pair_jobs, NT_MAX, fails);
    return fails ? 1 : 0;
}
