// How many bytes must a CU keep in flight to stream HBM at full rate, and what does the "commit to LDS, barrier, refill" round
// structure of the causal token kernels cost on top?    Every workgroup (512 threads) walks its own contiguous
// range of 8 KB tiles with a rolling ring of DEPTH tiles in registers (one uint4 per thread and tile):
//   mode 0: consume the oldest tile (xor into a register), refill its slot                -- pure streaming
//   mode 1: write the oldest tile to LDS, barrier, refill, read it back, barrier          -- the token kernels' round
//   hipcc --offload-arch=gfx950 -O3 -o inflight inflight.hip && ./inflight
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define GAS __attribute__((address_space(1)))

template <int DEPTH, int MODE>
__global__ __launch_bounds__(512) void k_ring(const char* buf, long tiles_per_wg, unsigned* sink) {
    __shared__ u32x4 lds[2][512];
    const int t = threadIdx.x;
    const GAS u32x4* g = (const GAS u32x4*)(buf + (long)blockIdx.x * tiles_per_wg * 8192) + t;
    u32x4 ring[DEPTH], acc = {0, 0, 0, 0};
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) ring[d] = g[(long)d * 512];
    for (long s = 0; s < tiles_per_wg; s += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            long nxt = s + d + DEPTH; if (nxt >= tiles_per_wg) nxt = s + d;   // (the last ring re-reads its own tiles)
            if (MODE == 0) {
                acc ^= ring[d];
                ring[d] = g[nxt * 512];
            } else {
                lds[d & 1][t] = ring[d];
                __syncthreads();
                ring[d] = g[nxt * 512];
                acc ^= lds[d & 1][t ^ 64];
                if (MODE == 2) __syncthreads();
            }
        }
    }
    if (acc.x == 0x12345u) sink[t] = acc.y;
}

// mode 3: the token kernel's address pattern -- a workgroup reads the two 256 KB summaries (P, dS: [4][8] tiles of 8 KB) of one
// chunk in the order (v slice, k slice), i.e. tiles 0, 8, 16, 24, 1, 9, ...; ring of 4 rounds x 2 tiles; then the next chunk.
// PERM 0 walks the tiles in memory order instead (what a [V / 64][K / 64] tile order would give).
template <int PERM>
__global__ __launch_bounds__(512) void k_tok(const char* bufP, const char* bufS, long chunks, unsigned* sink) {
    __shared__ u32x4 lds[2][512];
    const int t = threadIdx.x;
    u32x4 rp[4], rs[4], acc = {0, 0, 0, 0};
    for (long c = blockIdx.x; c < chunks; c += gridDim.x) {
        const GAS u32x4* gp = (const GAS u32x4*)(bufP + c * 262144) + t;
        const GAS u32x4* gs = (const GAS u32x4*)(bufS + c * 262144) + t;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { rp[kk] = gp[(PERM ? kk * 8 : kk) * 512]; rs[kk] = gs[(PERM ? kk * 8 : kk) * 512]; }
        for (int vs = 0; vs < 8; ++vs) {
            const int vn = vs < 7 ? vs + 1 : vs;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                lds[0][t] = rp[kk]; lds[1][t] = rs[kk];
                __syncthreads();
                const int tile = PERM ? kk * 8 + vn : vn * 4 + kk;
                rp[kk] = gp[tile * 512]; rs[kk] = gs[tile * 512];
                acc ^= lds[0][t ^ 64] ^ lds[1][t ^ 128];
                __syncthreads();
            }
        }
    }
    if (acc.x == 0x12345u) sink[t] = acc.y;
}
template <int PERM>
static void run_tok(const char* buf, unsigned* sink, int wgs, hipEvent_t e0, hipEvent_t e1) {
    const long chunks = 4096;   // 2 x 1 GB
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_tok<PERM>), dim3(wgs), dim3(512), 0, 0, buf, buf + ((size_t)1 << 30), chunks, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    printf("token-kernel pattern, %s tile order, %4d workgroups: %6.2f TB/s\n", PERM ? "(v, k)" : "memory", wgs, (double)chunks * 2 * 262144 / best * 1e-9);
}

template <int DEPTH, int MODE>
static void run(const char* buf, unsigned* sink, int wg_per_cu, hipEvent_t e0, hipEvent_t e1) {
    const int wgs = 256 * wg_per_cu;
    const size_t total = (size_t)3 << 30;
    long tpw = (long)(total / 8192 / wgs);
    tpw -= tpw % DEPTH;
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_ring<DEPTH, MODE>), dim3(wgs), dim3(512), 0, 0, buf, tpw, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    printf("mode %d  wg/CU %d  depth %2d (%4d KB in flight per CU): %6.2f TB/s\n", MODE, wg_per_cu, DEPTH, wg_per_cu * DEPTH * 8,
           (double)wgs * tpw * 8192 / best * 1e-9);
}

int main() {
    char* buf; unsigned* sink;
    CK(hipMalloc(&buf, (size_t)3 << 30)); CK(hipMalloc(&sink, 4096));
    CK(hipMemset(buf, 1, (size_t)3 << 30));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w : {256, 512, 4096}) { run_tok<1>(buf, sink, w, e0, e1); run_tok<0>(buf, sink, w, e0, e1); }
    if (getenv("INFLIGHT_TOK_ONLY")) return 0;
    for (int w : {1, 2, 4}) {
        run<2, 0>(buf, sink, w, e0, e1); run<4, 0>(buf, sink, w, e0, e1); run<8, 0>(buf, sink, w, e0, e1); run<16, 0>(buf, sink, w, e0, e1);
    }
    for (int w : {1, 2, 4}) {
        run<2, 1>(buf, sink, w, e0, e1); run<4, 1>(buf, sink, w, e0, e1); run<8, 1>(buf, sink, w, e0, e1); run<16, 1>(buf, sink, w, e0, e1);
    }
    for (int w : {1, 2}) {
        run<4, 2>(buf, sink, w, e0, e1); run<8, 2>(buf, sink, w, e0, e1); run<16, 2>(buf, sink, w, e0, e1);
    }
    return 0;
}
