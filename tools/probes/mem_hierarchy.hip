// Memory-hierarchy probe: what does a streaming read reach when its working set sits in the XCD's L2, in the
// memory-side cache (256 MB), or in HBM -- and does a buffer that one kernel has just written come back faster to the next
// kernel while it still fits the memory-side cache?  The causal pipeline hands four summary sets from kernel to kernel; this
// is the measurement behind the decision how to batch those launches.
//   hipcc --offload-arch=gfx950 -O3 -o mem_hierarchy mem_hierarchy.hip && ./mem_hierarchy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define GAS __attribute__((address_space(1)))

// every workgroup reads `per_wg` 8 KB tiles per repetition; the tile index walks the whole buffer with a stride that changes
// with the repetition, so a working set larger than a cache level is never re-read from it
template <bool NT, int DEPTH>
__global__ __launch_bounds__(512) void k_read(const char* buf, long tiles, int per_wg, int reps, unsigned* sink) {
    const int t = threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        long base = ((long)blockIdx.x * per_wg + (long)r * 7919 * per_wg) % tiles;
        u32x4 v[DEPTH];
        for (int s = 0; s < per_wg; s += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                long tile = base + s + d; if (tile >= tiles) tile -= tiles;
                const GAS u32x4* g = (const GAS u32x4*)(buf + (tile * 512 + t) * 16);
                v[d] = NT ? __builtin_nontemporal_load(g) : *g;
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
        }
    }
    if (acc.x == 0x12345u) sink[t] = acc.y;
}

template <bool NT>
__global__ __launch_bounds__(512) void k_write(char* buf, long tiles, int per_wg, unsigned seed) {
    const int t = threadIdx.x;
    for (int s = 0; s < per_wg; ++s) {
        long tile = (long)blockIdx.x * per_wg + s;
        if (tile >= tiles) return;
        u32x4 v = {seed, (unsigned)t, (unsigned)s, 1u};
        GAS u32x4* g = (GAS u32x4*)(buf + (tile * 512 + t) * 16);
        if (NT) __builtin_nontemporal_store(v, g); else *g = v;
    }
}

int main() {
    const size_t maxb = (size_t)4096 << 20;
    char* buf; unsigned* sink;
    CK(hipMalloc(&buf, maxb)); CK(hipMalloc(&sink, 4096));
    CK(hipMemset(buf, 1, maxb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("== repeated reads of a working set of W MB (8 KB tiles, 512 threads, 8 tiles in flight per workgroup)\n");
    const int sizes[] = {1, 2, 4, 8, 16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 4096};
    for (int mb : sizes) {
        const long tiles = ((long)mb << 20) / 8192;
        const int per_wg = 8;
        const int wgs = 2048;                          // 8 per CU
        // total bytes per launch >= 4 GB
        int reps = (int)(((size_t)4 << 30) / ((size_t)wgs * per_wg * 8192));
        if (reps < 1) reps = 1;
        for (int nt = 0; nt < 2; ++nt) {
            float best = 1e9f;
            for (int it = 0; it < 4; ++it) {
                CK(hipEventRecord(e0));
                if (nt) hipLaunchKernelGGL((k_read<true, 8>), dim3(wgs), dim3(512), 0, 0, buf, tiles, per_wg, reps, sink);
                else hipLaunchKernelGGL((k_read<false, 8>), dim3(wgs), dim3(512), 0, 0, buf, tiles, per_wg, reps, sink);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (it > 0 && ms < best) best = ms;
            }
            const double bytes = (double)wgs * per_wg * 8192 * reps;
            printf("W = %5d MB  %s loads: %7.2f TB/s\n", mb, nt ? "nontemporal" : "regular    ", bytes / best * 1e-9);
        }
    }
    printf("== kernel A writes W MB, kernel B reads the same W MB (each once), alternating; time of B alone and of A alone\n");
    const int sizes2[] = {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048};
    for (int mb : sizes2) {
        const long tiles = ((long)mb << 20) / 8192;
        const int per_wg = 8;
        const int wgs = (int)(tiles / per_wg);
        for (int ntw = 0; ntw < 2; ++ntw) {
            float tw = 0, tr = 0; const int iters = 10;
            for (int it = 0; it < iters + 2; ++it) {
                hipEvent_t a0, a1, b1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b1));
                CK(hipEventRecord(a0));
                if (ntw) hipLaunchKernelGGL((k_write<true>), dim3(wgs), dim3(512), 0, 0, buf, tiles, per_wg, (unsigned)it);
                else hipLaunchKernelGGL((k_write<false>), dim3(wgs), dim3(512), 0, 0, buf, tiles, per_wg, (unsigned)it);
                CK(hipEventRecord(a1));
                hipLaunchKernelGGL((k_read<true, 8>), dim3(wgs), dim3(512), 0, 0, buf, tiles, per_wg, 1, sink);
                CK(hipEventRecord(b1)); CK(hipEventSynchronize(b1));
                float m1, m2; CK(hipEventElapsedTime(&m1, a0, a1)); CK(hipEventElapsedTime(&m2, a1, b1));
                if (it >= 2) { tw += m1; tr += m2; }
                CK(hipEventDestroy(a0)); CK(hipEventDestroy(a1)); CK(hipEventDestroy(b1));
            }
            const double bytes = (double)mb * 1048576.0;
            printf("W = %5d MB  %s stores: write %6.2f TB/s (%7.1f us)   read-back %6.2f TB/s (%7.1f us)\n", mb,
                   ntw ? "nontemporal" : "regular    ", bytes / (tw / iters) * 1e-9, tw / iters * 1e3, bytes / (tr / iters) * 1e-9, tr / iters * 1e3);
        }
    }
    return 0;
}
