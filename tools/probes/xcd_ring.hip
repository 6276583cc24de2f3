// XCD-resident summary ring probe (verdict r4 item 3).  Question: can the block summaries of the C2 step stay
// in an XCD's 4 MB L2 between the workgroups that produce and consume them inside ONE launch -- i.e. do ring writes stay out of HBM,
// what does an XCD-local barrier cost, and at what rate does a workgroup read what other CUs of its XCD have just written?
//
//   hipcc --offload-arch=gfx950 -O3 -o xcd_ring xcd_ring.hip && ./xcd_ring            (timings)
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE FETCH_SIZE ... -- ./xcd_ring pmc         (one pass of each kernel: counters)
//
// Kernel `k_ring`: 256 persistent workgroups (one per CU; 100 KB of LDS forces that), grouped by the XCC they actually run on
// (s_getreg XCC_ID + a per-XCC arrival counter -> rank inside the XCD).  Per iteration ("one (b,h) of C2 per XCD"):
//   phase A  every workgroup streams TOK bytes of "tokens" from HBM (nontemporal) and writes its 1/32 of the XCD's ring slot
//            (RING bytes per XCD: 1 MB = the fp32 summaries of one (b,h)) with plain stores;
//   barrier  XCD-local (monotonic counter per XCC in global memory, relaxed agent atomics, s_sleep poll);
//   phase B  every workgroup reads a 1/32 COLUMN slice of the whole ring (pieces written by all 32 workgroups: the e-sliced
//            mixing's access pattern) with sc1 (L1-bypassing) loads and writes it back in place;
//   barrier
//   phase C  every workgroup reads its own 1/32 of the ring again (sc1) and writes OUTB bytes of "output" (nontemporal).
// Modes: 0 = all phases; 1 = no ring traffic (tokens + barriers only); 2 = ring only (no token streaming); 3 = no barriers (wrong
// data, timing only).  The data is checked in mode 0 (every word carries (iteration, writer) and is verified by its readers).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define GAS __attribute__((address_space(1)))

constexpr int NT = 512;
constexpr int WPX = 32;                 // workgroups per XCD
constexpr long RING = 1 << 20;          // bytes per XCD ring slot
constexpr long PIECE = RING / WPX;      // 32 KB: what one workgroup writes in phase A
constexpr long TOK = 48 * 1024;         // token bytes a workgroup streams per iteration (2 blocks x Q, K, V)
constexpr long OUTB = 16 * 1024;

__device__ __forceinline__ u32x4 ld_sc1(const void* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

struct Args {
    const char* tokens;   // >= 256 * iters * TOK bytes
    char* out;
    char* ring;           // 8 * RING bytes
    int* arrive;          // [8] registration counters
    int* bar;             // [8] barrier counters (monotonic)
    int* err;
    unsigned long long* ticks;   // [256][4] accumulated s_memtime ticks per phase (A, barrier, B, C)
    int iters, mode;
};

__device__ __forceinline__ void xcd_barrier(int* ctr, int target, int tid) {
    wait_vm0();                      // this workgroup's stores have reached the L2
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
}

__global__ __launch_bounds__(NT, 2) void k_ring(const Args a) {
    extern __shared__ unsigned char smem[];   // 100 KB: one workgroup per CU
    __shared__ int s_rank, s_xcc;
    const int tid = threadIdx.x;
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7;
        s_xcc = (int)xcc;
        s_rank = __hip_atomic_fetch_add(a.arrive + xcc, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int xcc = s_xcc, rank = s_rank;
    if (rank >= WPX) { if (tid == 0) atomicAdd(a.err, 1000); return; }   // placement other than 32 per XCD: the probe gives up
    char* ring = a.ring + (long)xcc * RING;
    int* bar = a.bar + xcc;
    const char* tok = a.tokens + ((long)(xcc * WPX + rank) * a.iters) * TOK;
    unsigned long long tA = 0, tBar = 0, tB = 0, tC = 0;
    u32x4 sink = {0, 0, 0, 0};
    int bad = 0;
    for (int it = 0; it < a.iters; ++it) {
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        // ---- phase A
        if (a.mode != 2) {
#pragma unroll
            for (int i = 0; i < TOK / (NT * 16); ++i) sink ^= __builtin_nontemporal_load((const GAS u32x4*)(tok + (long)it * TOK + ((long)i * NT + tid) * 16));
        }
        if (a.mode != 1) {
#pragma unroll
            for (int i = 0; i < PIECE / (NT * 16); ++i) {
                const u32x4 v = {(unsigned)it, (unsigned)rank, (unsigned)i, sink.x & 1u};
                *(GAS u32x4*)(ring + rank * PIECE + ((long)i * NT + tid) * 16) = v;
            }
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (a.mode != 3) xcd_barrier(bar, (2 * it + 1) * WPX, tid);
        unsigned long long t2 = __builtin_amdgcn_s_memtime();
        // ---- phase B: column slice `rank` of every workgroup's piece: piece p, bytes [rank * 1 KB, + 1 KB)
        if (a.mode != 1) {
            u32x4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {   // 32 pieces x 1 KB = 2048 x 16 B = 4 loads per thread
                const int e = i * NT + tid, p = e >> 6, o = e & 63;
                v[i] = ld_sc1(ring + p * PIECE + rank * 1024 + o * 16);
            }
            wait_vm0();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = i * NT + tid, p = e >> 6;
                if (a.mode == 0 && (v[i].x != (unsigned)it || v[i].y != (unsigned)p)) ++bad;
                v[i].z ^= 0x5a5a5a5au;   // "mixed"
                *(GAS u32x4*)(ring + p * PIECE + rank * 1024 + (e & 63) * 16) = v[i];
            }
        }
        unsigned long long t3 = __builtin_amdgcn_s_memtime();
        if (a.mode != 3) xcd_barrier(bar, (2 * it + 2) * WPX, tid);
        unsigned long long t4 = __builtin_amdgcn_s_memtime();
        // ---- phase C: own piece again (now holding the other workgroups' phase-B writes), then the output
        if (a.mode != 1) {
            u32x4 v[PIECE / (NT * 16)];
#pragma unroll
            for (int i = 0; i < PIECE / (NT * 16); ++i) v[i] = ld_sc1(ring + rank * PIECE + ((long)i * NT + tid) * 16);
            wait_vm0();   // (the asm loads are invisible to the compiler's wait bookkeeping: nothing may touch v[] before this)
#pragma unroll
            for (int i = 0; i < PIECE / (NT * 16); ++i) {
                if (a.mode == 0 && (v[i].x != (unsigned)it || v[i].y != (unsigned)rank)) ++bad;
                sink ^= v[i];
            }
        }
        if (a.mode != 2) {
#pragma unroll
            for (int i = 0; i < OUTB / (NT * 16); ++i)
                __builtin_nontemporal_store(sink, (GAS u32x4*)(a.out + ((long)(xcc * WPX + rank) * a.iters + it) * OUTB + ((long)i * NT + tid) * 16));
        }
        unsigned long long t5 = __builtin_amdgcn_s_memtime();
        tA += t1 - t0; tBar += (t2 - t1) + (t4 - t3); tB += t3 - t2; tC += t5 - t4;
    }
    if (bad) atomicAdd(a.err, bad);
    if (tid == 0) {
        unsigned long long* t = a.ticks + (long)(xcc * WPX + rank) * 4;
        t[0] = tA; t[1] = tBar; t[2] = tB; t[3] = tC;
    }
    if (sink.x == 0x1234567u) a.out[0] = 1;
}

int main(int argc, char** argv) {
    const bool pmc = argc > 1 && !strcmp(argv[1], "pmc");
    const int iters = 16;   // (b,h) pairs per XCD at C2
    Args a{};
    char* tok; CK(hipMalloc(&tok, 256L * iters * TOK)); CK(hipMemset(tok, 1, 256L * iters * TOK));
    CK(hipMalloc(&a.out, 256L * iters * OUTB));
    CK(hipMalloc(&a.ring, 8 * RING));
    CK(hipMalloc(&a.arrive, 64)); CK(hipMalloc(&a.bar, 64)); CK(hipMalloc(&a.err, 4));
    CK(hipMalloc(&a.ticks, 256 * 4 * 8));
    a.tokens = tok; a.iters = iters;
    CK(hipFuncSetAttribute((const void*)k_ring, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[] = {"all phases", "tokens + barriers, no ring", "ring + barriers, no tokens", "all phases, NO barriers (timing only)"};
    printf("per launch: 256 workgroups x %d iterations; tokens %.1f MB read + %.1f MB written, ring writes %.1f MB (A) + %.1f MB (B), ring reads 2 x %.1f MB\n",
           iters, 256.0 * iters * TOK / 1e6, 256.0 * iters * OUTB / 1e6, 256.0 * iters * PIECE / 1e6, 256.0 * iters * PIECE / 1e6, 256.0 * iters * PIECE / 1e6);
    for (int mode = 0; mode < 4; ++mode) {
        a.mode = mode;
        float best = 1e9f;
        int err = 0;
        unsigned long long ticks[256 * 4];
        for (int rep = 0; rep < (pmc ? 1 : 5); ++rep) {
            CK(hipMemset(a.arrive, 0, 64)); CK(hipMemset(a.bar, 0, 64)); CK(hipMemset(a.err, 0, 4));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_ring, dim3(256), dim3(NT), 100 * 1024, 0, a);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            int e; CK(hipMemcpy(&e, a.err, 4, hipMemcpyDeviceToHost)); err |= e;
        }
        CK(hipMemcpy(ticks, a.ticks, sizeof(ticks), hipMemcpyDeviceToHost));
        double s[4] = {0, 0, 0, 0};
        for (int w = 0; w < 256; ++w) for (int p = 0; p < 4; ++p) s[p] += (double)ticks[w * 4 + p] / 256 / iters;
        printf("mode %d (%s): %.1f us per launch = %.2f us per iteration; per-iteration ticks (100 MHz s_memtime: x10 ns) A %.0f  barriers %.0f  B %.0f  C %.0f   errors %d\n",
               mode, names[mode], best * 1e3, best * 1e3 / iters, s[0], s[1], s[2], s[3], err);
    }
    return 0;
}
