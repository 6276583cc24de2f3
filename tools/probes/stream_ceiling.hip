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

// the operator's real addressing: tensors are (B, N, H, D) -- a (b,h)'s token rows are 128-byte pieces H * 128 bytes apart; tile =
// 64 rows of one head; workgroup = 8 consecutive blocks of one (b,h) (STRIDED) vs the same bytes as contiguous 8 KB tiles
template <int NR, int DEPTH, bool NT, int STRIDED>
__global__ __launch_bounds__(512) void k_stream_layout(Args a) {
    constexpr int T = 512, H = 16, NB = 64;          // heads, 64-token blocks per (b,h)
    constexpr int HP = STRIDED > 1 ? STRIDED : 1;    // adjacent heads read together: rows of HP * 128 contiguous bytes
    const int wg = blockIdx.x, jg = wg & 7, bh = wg >> 3, b = bh / H, h = bh % H;
    const int t = threadIdx.x, row = t >> 3, piece = t & 7;
    u32x4 r[DEPTH][NR], acc[8];
    auto addr = [&](int blk) -> long {
        if (STRIDED == 1) return (((long)b * NB * 64 + (long)blk * 64 + row) * H + h) * 128 + piece * 16;
        if (STRIDED > 1) {   // workgroup = (head group of HP, 8 / HP blocks): step s covers 64 / HP rows of HP heads
            const int hg = (bh % H) / HP * HP, sub = bh % HP;                 // head group; which share of the blocks
            const int step = blk - jg * 8, blk2 = jg * 8 + sub * (8 / HP) + step / HP, part = step % HP;
            const int rr = part * (64 / HP) + t / (8 * HP), pc = t % (8 * HP);
            return (((long)b * NB * 64 + (long)blk2 * 64 + rr) * H + hg) * 128 + pc * 16;
        }
        return (((long)bh * NB + blk) * 64 + row) * 128 + piece * 16;
    };
    auto issue = [&](int d, int blk) {
#pragma unroll
        for (int i = 0; i < NR; ++i) r[d][i] = ld16<NT>(a.in[i] + addr(blk));
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) issue(d, jg * 8 + d);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int nx = s + DEPTH - 1;
        issue((s + DEPTH - 1) % DEPTH, jg * 8 + (nx < 8 ? nx : 0));
        u32x4 v = r[s % DEPTH][0];
#pragma unroll
        for (int i = 1; i < NR; ++i) v ^= r[s % DEPTH][i];
        acc[s] = v;
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) st16<false>(a.out[0] + (((long)wg * 8 + s) * T + t) * 16, acc[s]);
}
template <int NR, int DEPTH, bool NT, int STRIDED>
static void run_layout(const char* name, std::vector<char*>& bufs, size_t bytes_per_tensor) {
    const int wgs = (int)(bytes_per_tensor / (512 * 16) / 8);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nsets = (int)bufs.size() / 5, iters = 24;
    auto launch = [&](int it) {
        Args a;
        char** b = &bufs[(it % nsets) * 5];
        a.in[0] = b[0]; a.in[1] = b[1]; a.in[2] = b[2]; a.out[0] = b[3]; a.out[1] = b[4];
        a.tiles = 0; a.tiles_per_wg = 8;
        hipLaunchKernelGGL((k_stream_layout<NR, DEPTH, NT, STRIDED>), dim3(wgs), dim3(512), 0, 0, a);
    };
    for (int i = 0; i < 4; ++i) launch(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch(i);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, gb = (double)(NR + 1) * bytes_per_tensor / 1e9;
    printf("%-52s wgs=%5d depth=%d nt=%d strided=%d R%d:W1  %7.1f us  %6.2f TB/s\n", name, wgs, DEPTH, (int)NT, (int)STRIDED, NR, us, gb / us * 1e3);
}

// two adjacent heads per 1024-thread workgroup, each half (8 waves) streaming its own head as above: the two halves' requests for
// neighbouring 128-byte pieces are issued within the same few hundred cycles
template <int NR, int DEPTH, bool NT>
__global__ __launch_bounds__(1024) void k_stream_pair(Args a) {
    constexpr int T = 512, H = 16, NB = 64;
    const int wg = blockIdx.x, jg = wg & 7, bhp = wg >> 3, half = threadIdx.x >> 9, bh = bhp * 2 + half, b = bh / H, h = bh % H;
    const int t = threadIdx.x & 511, row = t >> 3, piece = t & 7;
    u32x4 r[DEPTH][NR], acc[8];
    auto addr = [&](int blk) -> long { return (((long)b * NB * 64 + (long)blk * 64 + row) * H + h) * 128 + piece * 16; };
    auto issue = [&](int d, int blk) {
#pragma unroll
        for (int i = 0; i < NR; ++i) r[d][i] = ld16<NT>(a.in[i] + addr(blk));
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) issue(d, jg * 8 + d);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int nx = s + DEPTH - 1;
        issue((s + DEPTH - 1) % DEPTH, jg * 8 + (nx < 8 ? nx : 0));
        u32x4 v = r[s % DEPTH][0];
#pragma unroll
        for (int i = 1; i < NR; ++i) v ^= r[s % DEPTH][i];
        acc[s] = v;
        __syncthreads();
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) st16<false>(a.out[0] + ((((long)bh * 8 + jg) * 8 + s) * T + t) * 16, acc[s]);
}
template <int NR, int DEPTH, bool NT>
static void run_pair(const char* name, std::vector<char*>& bufs, size_t bytes_per_tensor) {
    const int wgs = (int)(bytes_per_tensor / (512 * 16) / 8) / 2;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nsets = (int)bufs.size() / 5, iters = 24;
    auto launch = [&](int it) {
        Args a;
        char** b = &bufs[(it % nsets) * 5];
        a.in[0] = b[0]; a.in[1] = b[1]; a.in[2] = b[2]; a.out[0] = b[3]; a.out[1] = b[4];
        a.tiles = 0; a.tiles_per_wg = 8;
        hipLaunchKernelGGL((k_stream_pair<NR, DEPTH, NT>), dim3(wgs), dim3(1024), 0, 0, a);
    };
    for (int i = 0; i < 4; ++i) launch(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch(i);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, gb = (double)(NR + 1) * bytes_per_tensor / 1e9;
    printf("%-52s wgs=%5d depth=%d nt=%d 1024 threads R%d:W1  %7.1f us  %6.2f TB/s\n", name, wgs, DEPTH, (int)NT, NR, us, gb / us * 1e3);
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
    run_layout<3, 3, true, 0>("3r1w (b,h)-contiguous tiles, deferred store", bufs, bytes);
    run_layout<3, 3, true, 1>("3r1w (B,N,H,D) head-strided rows, deferred store", bufs, bytes);
    run_layout<3, 3, false, 1>("3r1w (B,N,H,D) head-strided rows, no nt", bufs, bytes);
    run_layout<3, 3, true, 2>("3r1w (B,N,H,D) rows of 2 heads (256 B)", bufs, bytes);
    run_layout<3, 3, true, 4>("3r1w (B,N,H,D) rows of 4 heads (512 B)", bufs, bytes);
    run_layout<3, 3, true, 8>("3r1w (B,N,H,D) rows of 8 heads (1 KB)", bufs, bytes);
    run_pair<3, 3, true>("3r1w (B,N,H,D) head pair per 1024-thread workgroup", bufs, bytes);
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
