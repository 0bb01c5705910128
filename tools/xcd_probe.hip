// xcd_probe.hip -- what the persistent few-evaluation kernel (gpcc_chain.hip.h) relies on, measured on the box:
//   (1) which CONSUMER load forms see a tile another workgroup has just published with 16-byte `sc1` (write-through) stores +
//       drained flag (MI355X_MICROARCH.md, inter-workgroup visibility): plain loads, `sc1` register loads, LDS-DMA with and without
//       `sc1`, LDS-DMA behind an agent-scope acquire -- every word checked, consumer L1-warm (it pre-reads the lines with plain
//       loads before every wait), uneven load (odd workgroups stream a private buffer between rounds);
//   (2) the price of one hop (flag only; flag + 16 KiB payload) between two workgroups on different XCDs and on the same XCD;
//   (3) a returning agent-scope atomicAdd on one head word (the job queue) with 1, 64, 256 pullers.
// Every spin is bounded (a timeout word ends the kernel).  hipcc --offload-arch=gfx950 -O3 tools/xcd_probe.hip -o tools/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define TILE_BYTES 16384
#define TILE_U4 (TILE_BYTES / 16)
#define SPIN_LIMIT (1u << 22)

__device__ __forceinline__ unsigned ld_sc1_u32(const unsigned *p)
{
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st_sc1_u32(unsigned *p, unsigned v)
{
    asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_sc1_u4(u4 *p, u4 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ u4 ld_sc1_u4(const u4 *p)
{
    u4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// one lane polls *p until it is >= want (relaxed sc1 loads + s_sleep); false = timed out / aborted
__device__ __forceinline__ bool wait_ge(const unsigned *p, unsigned want, unsigned *tmo)
{
    for (unsigned spins = 0;; ++spins) {
        if (ld_sc1_u32(p) >= want) return true;
        __builtin_amdgcn_s_sleep(2);
        if ((spins & 63) == 63 && ld_sc1_u32(tmo) != 0) return false;
        if (spins > SPIN_LIMIT) {
            st_sc1_u32(tmo, 1u);
            return false;
        }
    }
}
__device__ __forceinline__ unsigned pattern(unsigned round, unsigned wg, unsigned idx) { return round * 2654435761u + wg * 40503u + idx * 97u + 12345u; }

// mode: 0 plain register loads, 1 sc1 register loads, 2 LDS-DMA sc1, 3 acquire fence + plain LDS-DMA, 4 plain LDS-DMA, 5 acquire + plain register loads
__global__ __launch_bounds__(256) void coherence(u4 *tiles, unsigned *flag, unsigned *ack, unsigned *tmo, unsigned *bad, const u4 *stream, int rounds, int mode)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int ok_s;
    const int w = blockIdx.x, G = gridDim.x, tid = threadIdx.x, src = (w + 1) % G;
    u4 *mine = tiles + (size_t)w * TILE_U4;
    const u4 *theirs = tiles + (size_t)src * TILE_U4;
    unsigned nbad = 0, sink = 0;
    for (int r = 1; r <= rounds; ++r) {
        // ---- produce round r into my tile (after my consumer has read round r-1)
        if (tid == 0) ok_s = wait_ge(&ack[w], (unsigned)(r - 1), tmo) ? 1 : 0;
        __syncthreads();
        if (!ok_s) return;
        for (int i = tid; i < TILE_U4; i += 256) {
            u4 v;
            for (int e = 0; e < 4; ++e) v[e] = pattern(r, w, 4 * i + e);
            st_sc1_u4(mine + i, v);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains
        __syncthreads();
        if (tid == 0) st_sc1_u32(&flag[w], (unsigned)r);
        // ---- uneven load: odd workgroups stream a private buffer
        if (w & 1)
            for (int i = tid; i < 16384; i += 256) { const u4 v = stream[(size_t)w * 16384 + i]; sink += v[0]; }
        // ---- L1-warm: pre-read the producer's lines with plain loads (old contents or new, whatever is there)
        for (int i = tid; i < TILE_U4; i += 256) { const u4 v = theirs[i]; sink += v[1]; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) ok_s = wait_ge(&flag[src], (unsigned)r, tmo) ? 1 : 0;
        if (mode == 3 || mode == 5) {
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __syncthreads();
        if (!ok_s) return;
        // ---- consume
        if (mode == 0 || mode == 1 || mode == 5) {
            for (int i = tid; i < TILE_U4; i += 256) {
                u4 v = (mode == 1) ? ld_sc1_u4(theirs + i) : theirs[i];
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                for (int e = 0; e < 4; ++e) nbad += (v[e] != pattern(r, src, 4 * i + e));
            }
        } else {
            const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) void *)smem;
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
            for (int pc = wave; pc < TILE_BYTES / 1024; pc += 4) {   // 1 KiB pieces
                const unsigned la = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds + pc * 1024));
                const char *gb = (const char *)theirs + pc * 1024;
                if (mode == 2)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1" : : "s"(la), "v"((unsigned)lane * 16u), "s"(gb) : "memory", "m0");
                else
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(la), "v"((unsigned)lane * 16u), "s"(gb) : "memory", "m0");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const unsigned *sw = (const unsigned *)smem;
            for (int i = tid; i < TILE_BYTES / 4; i += 256) nbad += (sw[i] != pattern(r, src, i));
            __syncthreads();
        }
        if (tid == 0) st_sc1_u32(&ack[src], (unsigned)r);
    }
    if (nbad) atomicAdd(bad, nbad);
    if (sink == 0x12345678u) bad[1] = sink;
}

// ping-pong between workgroup 0 and workgroup `peer`: flag only, or flag + a 16 KiB payload (sc1 stores, sc1 register loads)
__global__ __launch_bounds__(256) void pingpong(u4 *tiles, unsigned *flag, unsigned *tmo, unsigned long long *ticks, int peer, int iters, int payload)
{
    __shared__ int ok_s;
    const int w = blockIdx.x, tid = threadIdx.x;
    if (w != 0 && w != peer) return;
    const int me = (w == 0) ? 0 : 1, other = 1 - me;
    u4 *mine = tiles + (size_t)me * TILE_U4;
    const u4 *theirs = tiles + (size_t)other * TILE_U4;
    unsigned sink = 0;
    const unsigned long long t0 = wall_clock64();
    for (int it = 1; it <= iters; ++it) {
        const bool my_turn = ((it & 1) == me);
        if (my_turn) {
            if (payload) {
                for (int i = tid; i < TILE_U4; i += 256) { u4 v = {(unsigned)it, (unsigned)i, 0u, 0u}; st_sc1_u4(mine + i, v); }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            if (tid == 0) st_sc1_u32(&flag[me], (unsigned)it);
        } else {
            if (tid == 0) ok_s = wait_ge(&flag[other], (unsigned)it, tmo) ? 1 : 0;
            __syncthreads();
            if (!ok_s) return;
            if (payload) {
                u4 v[4];
                for (int u = 0; u < 4; ++u) v[u] = ld_sc1_u4(theirs + tid + 256 * u);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                for (int u = 0; u < 4; ++u) sink += v[u][0];
                __syncthreads();
            }
        }
    }
    if (tid == 0 && w == 0) ticks[0] = wall_clock64() - t0;
    if (sink == 0x12345678u) ticks[1] = sink;
}

__global__ __launch_bounds__(64) void dequeue(unsigned *head, unsigned long long *ticks, int iters)
{
    unsigned s = 0;
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0)
        for (int i = 0; i < iters; ++i) s += __hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = wall_clock64() - t0;
    if (s == 0x12345678u) ticks[1] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
    const int G = 128, rounds = 300;
    u4 *tiles, *stream; unsigned *words; unsigned long long *ticks;
    CK(hipMalloc(&tiles, (size_t)G * TILE_BYTES));
    CK(hipMalloc(&stream, (size_t)G * 16384 * 16));
    CK(hipMalloc(&words, 4096 * 4));
    CK(hipMalloc(&ticks, 64));
    CK(hipMemset(stream, 1, (size_t)G * 16384 * 16));
    int clk_khz = 100000;
    (void)hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0);
    const char *names[] = {"plain register loads, no acquire", "sc1 register loads", "LDS-DMA sc1", "agent acquire + plain LDS-DMA", "plain LDS-DMA, no acquire", "agent acquire + plain register loads"};
    for (int mode = 0; mode < 6; ++mode) {
        CK(hipMemset(tiles, 0, (size_t)G * TILE_BYTES));
        CK(hipMemset(words, 0, 4096 * 4));
        unsigned *flag = words, *ack = words + 1024, *tmo = words + 2048, *bad = words + 2056;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(coherence, G, 256, 96 * 1024, 0, tiles, flag, ack, tmo, bad, stream, rounds, mode);   // 96 KiB LDS: one workgroup per CU
        hipEventRecord(b, 0);
        CK(hipDeviceSynchronize());
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        unsigned h[16]; CK(hipMemcpy(h, tmo, 64, hipMemcpyDeviceToHost));
        printf("coherence  %-40s: %u stale words of %lld, timeout %u, %.2f us per round\n", names[mode], h[8], (long long)G * rounds * (TILE_BYTES / 4), h[0], ms * 1e3 / rounds);
    }
    for (int payload = 0; payload < 2; ++payload)
        for (int peer : {1, 8, 3}) {
            CK(hipMemset(words, 0, 4096 * 4));
            const int iters = 2000;
            hipLaunchKernelGGL(pingpong, 16, 256, 96 * 1024, 0, tiles, words, words + 2048, ticks, peer, iters, payload);
            CK(hipDeviceSynchronize());
            unsigned long long t; CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
            unsigned tm; CK(hipMemcpy(&tm, words + 2048, 4, hipMemcpyDeviceToHost));
            printf("pingpong   workgroups 0 <-> %d (%s), %s: %.2f us per hop (timeout %u)\n", peer, peer == 8 ? "same XCD by b %% 8" : "other XCD", payload ? "flag + 16 KiB sc1 payload" : "flag only",
                   (double)t / clk_khz * 1e3 / iters, tm);
        }
    for (int pullers : {1, 64, 256}) {
        CK(hipMemset(words, 0, 64));
        const int iters = 2000;
        hipLaunchKernelGGL(dequeue, pullers, 64, 96 * 1024, 0, words, ticks, iters);
        CK(hipDeviceSynchronize());
        unsigned long long t; CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
        printf("dequeue    %3d pullers: %.3f us per returning atomicAdd (per puller)\n", pullers, (double)t / clk_khz * 1e3 / iters);
    }
    return 0;
}
