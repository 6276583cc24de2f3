// Ceiling probe for the fast path's streaming kernels: what does a kernel with k_fs_state_fwd's traffic shape
// (three 8 KB tile reads + one 8 KB write per 64-token block, C2 sizes) reach when it does nothing else?
//   hipcc --offload-arch=gfx950 -O3 -o stream_ceiling stream_ceiling.hip && ./stream_ceiling
// Variants: workgroup count / size, tiles in flight per workgroup, nontemporal loads, an LDS hop with barriers per tile (the
// structure of the real kernel), reads : writes = 3:1, 3:0 (read only), 2:1, 1:1 and 2:2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define GAS __attribute__((address_space(1)))

template <bool NT> __device__ inline u32x4 ld16(const void* p) {
    const GAS u32x4* g = (const GAS u32x4*)p;
    if (NT) return __builtin_nontemporal_load(g);
    return *g;
}
template <bool NT> __device__ inline void st16(void* p, u32x4 v) {
    GAS u32x4* g = (GAS u32x4*)p;
    if (NT) __builtin_nontemporal_store(v, g); else *g = v;
}

struct Args { const char* in[3]; char* out[2]; long tiles; int tiles_per_wg; };

// one "tile" = T threads x 16 B per input; a workgroup walks tiles_per_wg consecutive tiles, DEPTH tiles' loads in flight
template <int T, int NR, int NW, int DEPTH, bool NT, bool LDSHOP>
__global__ __launch_bounds__(T) void k_stream(Args a) {
    __shared__ u32x4 lds[LDSHOP ? T * NR : 1];
    const long tile0 = (long)blockIdx.x * a.tiles_per_wg;
    const int t = threadIdx.x;
    u32x4 r[DEPTH][NR];
    auto issue = [&](int d, long tile) {
#pragma unroll
        for (int i = 0; i < NR; ++i) r[d][i] = ld16<NT>(a.in[i] + (tile * T + t) * 16);
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) issue(d, tile0 + (d < a.tiles_per_wg ? d : 0));
    u32x4 acc = {0, 0, 0, 0};
    for (int s = 0; s < a.tiles_per_wg; s += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int nx = s + d + DEPTH - 1;
            issue((d + DEPTH - 1) % DEPTH, tile0 + (nx < a.tiles_per_wg ? nx : 0));
            u32x4 v = r[d][0];
#pragma unroll
            for (int i = 1; i < NR; ++i) v ^= r[d][i];
            if (LDSHOP) {
#pragma unroll
                for (int i = 0; i < NR; ++i) lds[i * T + t] = r[d][i];
                __syncthreads();
#pragma unroll
                for (int i = 0; i < NR; ++i) v += lds[i * T + ((t * 17 + 5) % T)];
                __syncthreads();
            }
            if (NW == 0) acc ^= v;
#pragma unroll
            for (int i = 0; i < NW; ++i) st16<NT>(a.out[i] + ((tile0 + s + d) * T + t) * 16, v);
        }
    }
    if (NW == 0 && acc.x == 0x12345u) st16<false>(a.out[0] + t * 16, acc);
}

template <int T, int NR, int NW, int DEPTH, bool NT, bool LDSHOP>
static void run(const char* name, std::vector<char*>& bufs, size_t bytes_per_tensor, int tiles_per_wg) {
    const long tiles = bytes_per_tensor / (T * 16);
    const int wgs = (int)(tiles / tiles_per_wg);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nsets = (int)bufs.size() / 5, iters = 24;
    auto launch = [&](int it) {
        Args a;
        char** b = &bufs[(it % nsets) * 5];
        a.in[0] = b[0]; a.in[1] = b[1]; a.in[2] = b[2]; a.out[0] = b[3]; a.out[1] = b[4];
        a.tiles = tiles; a.tiles_per_wg = tiles_per_wg;
        hipLaunchKernelGGL((k_stream<T, NR, NW, DEPTH, NT, LDSHOP>), dim3(wgs), dim3(T), 0, 0, a);
    };
    for (int i = 0; i < 4; ++i) launch(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch(i);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, gb = (double)(NR + NW) * bytes_per_tensor / 1e9;
    printf("%-44s T=%4d wgs=%6d tiles/wg=%3d depth=%d nt=%d lds=%d  R%d:W%d  %7.1f us  %6.2f TB/s\n", name, T, wgs, tiles_per_wg, DEPTH,
           (int)NT, (int)LDSHOP, NR, NW, us, gb / us * 1e3);
}

// the real state kernels' shape: 8 tiles of 3 inputs per workgroup, ONE 64 KB store burst at the end (all workgroups resident at
// once -> the whole chip reads, then the whole chip writes)
template <int T, int NR, int DEPTH, bool NT, bool LDSHOP>
__global__ __launch_bounds__(T) void k_stream_defer(Args a) {
    __shared__ u32x4 lds[LDSHOP ? T * NR : 1];
    const long tile0 = (long)blockIdx.x * 8;
    const int t = threadIdx.x;
    u32x4 r[DEPTH][NR], acc[8];
    auto issue = [&](int d, long tile) {
#pragma unroll
        for (int i = 0; i < NR; ++i) r[d][i] = ld16<NT>(a.in[i] + (tile * T + t) * 16);
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) issue(d, tile0 + d);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int nx = s + DEPTH - 1;
        issue((s + DEPTH - 1) % DEPTH, tile0 + (nx < 8 ? nx : 0));
        u32x4 v = r[s % DEPTH][0];
#pragma unroll
        for (int i = 1; i < NR; ++i) v ^= r[s % DEPTH][i];
        if (LDSHOP) {
#pragma unroll
            for (int i = 0; i < NR; ++i) lds[i * T + t] = r[s % DEPTH][i];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NR; ++i) v += lds[i * T + ((t * 17 + 5) % T)];
            __syncthreads();
        }
        acc[s] = v;
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) st16<false>(a.out[0] + ((tile0 + s) * T + t) * 16, acc[s]);
}

template <int T, int NR, int DEPTH, bool NT, bool LDSHOP>
static void run_defer(const char* name, std::vector<char*>& bufs, size_t bytes_per_tensor) {
    const long tiles = bytes_per_tensor / (T * 16);
    const int wgs = (int)(tiles / 8);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nsets = (int)bufs.size() / 5, iters = 24;
    auto launch = [&](int it) {
        Args a;
        char** b = &bufs[(it % nsets) * 5];
        a.in[0] = b[0]; a.in[1] = b[1]; a.in[2] = b[2]; a.out[0] = b[3]; a.out[1] = b[4];
        a.tiles = tiles; a.tiles_per_wg = 8;
        hipLaunchKernelGGL((k_stream_defer<T, NR, DEPTH, NT, LDSHOP>), dim3(wgs), dim3(T), 0, 0, a);
    };
    for (int i = 0; i < 4; ++i) launch(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch(i);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, gb = (double)(NR + 1) * bytes_per_tensor / 1e9;
    printf("%-44s T=%4d wgs=%6d deferred 64 KB store depth=%d nt=%d lds=%d  R%d:W1  %7.1f us  %6.2f TB/s\n", name, T, wgs, DEPTH, (int)NT,
           (int)LDSHOP, NR, us, gb / us * 1e3);
}

int main() {
    const size_t bytes = 8ull * 16 * 4096 * 64 * 2;      // one C2 token tensor: 67 MB
    const int nsets = 4;                                  // 4 x 5 x 67 MB = 1.34 GB footprint: nothing survives in the 256 MB MALL
    std::vector<char*> bufs(nsets * 5);
    for (auto& p : bufs) { CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 1, bytes)); }
    // the real kernel: 512 threads, 8 tiles (blocks) per workgroup, 1024 workgroups
    run<512, 3, 1, 1, false, false>("3r1w depth1", bufs, bytes, 8);
    run<512, 3, 1, 2, false, false>("3r1w depth2", bufs, bytes, 8);
    run<512, 3, 1, 2, true, false>("3r1w depth2 nt", bufs, bytes, 8);
    run<512, 3, 1, 2, false, true>("3r1w depth2 ldshop", bufs, bytes, 8);
    run<512, 3, 1, 2, true, true>("3r1w depth2 ldshop nt", bufs, bytes, 8);
    run<512, 3, 1, 4, true, false>("3r1w depth4 nt", bufs, bytes, 8);
    run<512, 3, 1, 2, true, false>("3r1w depth2 nt tiles/wg=2", bufs, bytes, 2);
    run<512, 3, 1, 1, true, false>("3r1w depth1 nt tiles/wg=1", bufs, bytes, 1);
    run<256, 3, 1, 1, true, false>("3r1w depth1 nt T=256 tiles/wg=1", bufs, bytes, 1);
    run<256, 3, 1, 1, false, false>("3r1w depth1 T=256 tiles/wg=1", bufs, bytes, 1);
    run<256, 3, 1, 2, true, false>("3r1w depth2 nt T=256 tiles/wg=4", bufs, bytes, 4);
    run<1024, 3, 1, 2, true, false>("3r1w depth2 nt T=1024 tiles/wg=4", bufs, bytes, 4);
    run<512, 3, 1, 2, true, false>("3r1w depth2 nt tiles/wg=32", bufs, bytes, 32);
    run<512, 3, 1, 2, true, false>("3r1w depth2 nt tiles/wg=16", bufs, bytes, 16);
    run_defer<512, 3, 2, true, false>("3r1w deferred store depth2 nt", bufs, bytes);
    run_defer<512, 3, 3, true, false>("3r1w deferred store depth3 nt", bufs, bytes);
    run_defer<512, 3, 3, true, true>("3r1w deferred store depth3 nt ldshop", bufs, bytes);
    run_defer<512, 3, 3, false, true>("3r1w deferred store depth3 ldshop", bufs, bytes);
    run_defer<256, 3, 3, true, true>("3r1w deferred store depth3 nt ldshop T=256", bufs, bytes);
    // other mixes
    run<512, 3, 0, 2, true, false>("3r0w depth2 nt", bufs, bytes, 8);
    run<512, 3, 0, 2, false, false>("3r0w depth2", bufs, bytes, 8);
    run<256, 3, 0, 1, false, false>("3r0w depth1 T=256 tiles/wg=1", bufs, bytes, 1);
    run<512, 2, 1, 2, true, false>("2r1w depth2 nt", bufs, bytes, 8);
    run<512, 1, 1, 2, true, false>("1r1w depth2 nt", bufs, bytes, 8);
    run<256, 1, 1, 1, false, false>("1r1w depth1 T=256 tiles/wg=1 (copy)", bufs, bytes, 1);
    run<512, 2, 2, 2, true, false>("2r2w depth2 nt", bufs, bytes, 8);
    run<512, 3, 2, 2, true, false>("3r2w depth2 nt", bufs, bytes, 8);
    run<512, 3, 2, 2, false, false>("3r2w depth2", bufs, bytes, 8);
    return 0;
}
