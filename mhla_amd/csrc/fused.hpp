// Fast path of the block-mixing MHLA operator for bf16 tensors, head dim 64, M <= 64 blocks.
//
// Design.  The per-(b,h) block summaries KV_j (64x64) are the only state that
// links blocks.  They are kept in bf16 in an "8-block interleaved, transposed" layout
//      state[bh][jg][e'][jj]   e' = d2*64 + d1,  j = 8*jg + jj,   (KV_j^T = V_j^T K_j)
// so that (a) the producer stores 16-byte pieces (8 blocks of one element) in 256-byte runs, and (b) the
// mixing GEMM  G_i = sum_j W_ij KV_j  reads its MFMA A-operand (16 e' x 32 j) straight from global/L2 as
// one 16-byte load per lane -- no LDS staging, no transposition.  The mixed summaries G_i of an 8-block
// tile never leave the CU: they are written to LDS (bf16, [i][d2][d1]) and consumed at once by
// O_i = Q_i G_i (forward) or dQ/dK/dV (backward).  W is split into bf16 hi + lo parts (2 MFMAs) so the
// mixing weights keep ~16 mantissa bits.  All contractions: v_mfma_f32_16x16x32_bf16, fp32 accumulate.
//
//   forward : k_fs_state1c<0> (KV^T, ksum, z) -> k_fs_wz<0> (1/n) -> k_tile_out (mix + O)                    [fused_tile16.hpp]
//   backward: k_fs_state1c<1> (dG^T, dn) -> k_fs_dw (dW partials; extra workgroups: dz = W^T dn) -> k_tile_bwd: dQ tiles (mix G;
//             dQ, dksum), dK/dV tiles (mix dKV; dK, dV), dW reduction -- one launch, three roles
//   (k_fs_state / k_fs_state_fwd: the summaries of blocks of more than 64 tokens, in synchronous 64-row chunks)
// The streaming kernels run 8-wave workgroups at <= 128 VGPRs (two workgroups = 16 waves per CU); the tile kernels one 8-wave
// workgroup per CU (160 KB of LDS).  What bounds each of them is measured in DESIGN.md section 3b.
#pragma once
#include <type_traits>

#include "common.hpp"

namespace mhla {
namespace fast {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int FD = 64;         // head dim
constexpr int FE = FD * FD;    // elements of one block summary
constexpr int TLD = 80;        // LDS row stride (bf16) of [rows][64] token tiles read with ds_read_b64_tr_b16
constexpr int GLD = 64;        // LDS row stride (bf16) of the mixed summaries Gt[i][d2][d1]: unpadded 128-byte rows, 16-byte pieces XOR-swizzled (gt_off)
// Layout of a mixed-summary slot / staging tile [64 rows][64 columns] bf16 (round 4).  gfx950's LDS has 64 banks and services a
// wave in instruction-specific lane groups (MI355X_MICROARCH.md, LDS): under that model the round-1 layout (rows of 72 elements,
// derived for 32 banks) was 2-way conflicted in EVERY access pattern of the tile kernels -- the 16-byte operand reads, the
// transpose reads, the 8-byte mixing / staging writes and the 16-byte store reads (tools/lds_conflicts.py reproduces the
// counters' 0.45-0.49).  Rows are now unpadded and the 16-byte piece p of row r is stored at piece p ^ ((r ^ (r >> 1)) & 7):
// all four read patterns and the (now 16-byte) write patterns are conflict-free in the same model.
__device__ __forceinline__ int gt_off(int row, int col) {
    return row * GLD + ((((col >> 3) ^ ((row ^ (row >> 1)) & 7)) << 3) | (col & 7));
}
// The swizzled addresses are functions of the lane that hipcc would otherwise compute at kernel entry and carry (in the 256-VGPR
// tile kernels: spill) across the mixing phase: the helpers that use them launder the lane number first, so the address
// arithmetic (a handful of VALU operations) is redone where it is used.
__device__ __forceinline__ int opaque_lane(int lane) {
    asm volatile("" : "+v"(lane));
    return lane;
}
// transpose read of tr_read8 from a swizzled slot: lane (c = lane & 15, g = lane >> 4) receives T[k0 + 8 g + 0..7][c0 + c]
__device__ __forceinline__ bf16x8 tr_read8_gt(const u16* tile, int k0, int c0, int lane);
constexpr int IT = 8;          // blocks per workgroup tile (= interleave factor of the state layout)
constexpr int FT = 256;        // threads per workgroup of the small kernels
constexpr int FT8 = 512;       // 8-wave workgroups of the streaming kernels: <= 128 VGPRs -> 16 waves per CU in flight

#define LDS_S16X4(p) ((__attribute__((address_space(3))) s16x4*)(p))

__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// ... on fp16 operands (the h16 summary payloads, common.hpp): same shapes and lane layout
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma_f16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f16x8 as_f16x8(const bf16x8& v) { return __builtin_bit_cast(f16x8, v); }

// MFMA operand from a row-major LDS tile T[k][c] (k = reduction index): lane (c = lane & 15, g = lane >> 4)
// receives T[k0 + 8 g + 0..7][c0 + c].  Two hardware transpose reads of a 4 x 16 block each.
__device__ __forceinline__ bf16x8 tr_read8(const u16* tile, int ld, int k0, int c0, int lane) {
    const int g = lane >> 4, li = lane & 15;
    const u16* p = tile + (k0 + g * 8 + (li >> 2)) * ld + c0 + (li & 3) * 4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(p + 4 * ld));
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(bf16x8, r);
}

__device__ __forceinline__ bf16x8 tr_read8_gt(const u16* tile, int k0, int c0, int lane) {
    const int g = lane >> 4, li = lane & 15, row = k0 + g * 8 + (li >> 2), col = c0 + (li & 3) * 4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(tile + gt_off(row, col)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(tile + gt_off(row + 4, col)));
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(bf16x8, r);
}
// Two packed 8-byte results (4 columns each) of neighbouring column tiles a, b -> one 16-byte piece per lane: lane pairs
// (kg, kg ^ 1) swap halves (v_permlane16_swap: odd rows of the first operand <-> even rows of the second), after which lane kg
// holds columns 8 (kg >> 1) .. + 7 of tile (kg & 1 ? b : a).  8-byte LDS stores of 16 lanes with one column are 2-way
// conflicted whatever the layout; the 16-byte ones are conflict-free.
__device__ __forceinline__ uint4 pair_pieces(uint2 a, uint2 b) {
    const auto x = __builtin_amdgcn_permlane16_swap(a.x, b.x, false, false);
    const auto y = __builtin_amdgcn_permlane16_swap(a.y, b.y, false, false);
    return make_uint4(x[0], y[0], x[1], y[1]);
}

__device__ __forceinline__ uint4 relu_eps8(uint4 v, float eps) {
    unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float lo = fmaxf(__uint_as_float(w[i] << 16), 0.f) + eps;
        const float hi = fmaxf(__uint_as_float(w[i] & 0xffff0000u), 0.f) + eps;
        w[i] = pack_bf16x2(lo, hi);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ float bf(u16 h) { return __uint_as_float(((unsigned)h) << 16); }

// -------------------------------------------------------------------------------------------------
// k_fs_wz: the two small [M x M] x [M x S] products per (b,h), done once instead of per consumer wave:
//   MODE 0: ninv[i][s] = 1 / (eps + sum_j W[i][j] z[j][s])         (normaliser, mhla.py:266)
//   MODE 1: dz[j][s]   = sum_i W[i][j] dn[i][s]
// grid (ceil(S / 64), bh), M <= 64.
// -------------------------------------------------------------------------------------------------
constexpr int WZ_C = 16;   // columns (intra-block positions s) per workgroup: ceil(S / 16) x bh workgroups
constexpr int WZ_SMEM = (64 * 65 + 64 * WZ_C) * 4;
// body shared by the stand-alone launch (forward: 1 / n) and the extra workgroups of the dW launch (backward: dz); NT threads
template <int MODE, int NT>
__device__ __forceinline__ void wz_body(float* __restrict__ smem, const float* __restrict__ W, int ldw, const float* __restrict__ x,
                                        float* __restrict__ out, int M, int S, float eps, int cblk, int bh, int tid) {
    float* Ws = smem;                 // [64][65]
    float* xs = smem + 64 * 65;       // [64][WZ_C], 16-byte aligned (64 * 65 * 4 = 16640)
    const int c0 = cblk * WZ_C, rv = min(WZ_C, S - c0);
    constexpr int NWL = 4096 / NT, NXL = 64 * WZ_C / NT;
    // all global loads first (clamped indices: no branches), LDS writes after
    float wreg[NWL], xreg[NXL];
#pragma unroll
    for (int t = 0; t < NWL; ++t) {
        const int v = tid + t * NT, r = min(v >> 6, M - 1), c = min(v & 63, M - 1);
        wreg[t] = gld<float>(MODE ? W + (long)c * ldw + r : W + (long)r * ldw + c);
    }
#pragma unroll
    for (int t = 0; t < NXL; ++t) {
        const int v = tid + t * NT, r = min(v / WZ_C, M - 1), c = min(v % WZ_C, rv - 1);
        xreg[t] = gld<float>(x + ((long)bh * M + r) * S + c0 + c);
    }
#pragma unroll
    for (int t = 0; t < NWL; ++t) {
        const int v = tid + t * NT, r = v >> 6, c = v & 63;
        Ws[r * 65 + c] = (r < M && c < M) ? wreg[t] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < NXL; ++t) {
        const int v = tid + t * NT, r = v / WZ_C, c = v % WZ_C;
        xs[r * WZ_C + c] = (r < M && c < rv) ? xreg[t] : 0.f;
    }
    __syncthreads();
    // thread -> row r = tid >> 2, 4 columns 4 (tid & 3) ..: x read as float4, one W element per step (first 256 threads)
    if (tid < 256) {
        const int r = tid >> 2, sq = tid & 3;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        // With packed fp32 math enabled the straightforward loop (WZ_VARIANT=0: `acc += Ws[..] * x4` unrolled by 8) compiles to
        // v_pk_fma_f32 fed by ds_read2_b32 / ds_read_b128 behind counted s_waitcnt lgkmcnt(N) waits, and the LOW halves of its
        // results in lanes 16-31 / 48-63 differ from run to run whenever the dW workgroups of this launch share the CU
        // (tools/probes/pk_fma_repro.hip reproduces it without the library).  The same loop with scalar
        // v_fmac_f32 (the shipped build has no packed fp32 ops at all) is bit-stable.  Shipped: WZ_VARIANT=2, the W elements read
        // one by one (ds_read_b32, no ds_read2_b32 pairs) -- bit-stable with packed math on as well, and as fast as variant 0
        // (k_fs_dw 33.4 vs 32.8 us at C2).  Variant 1 (wait for lgkmcnt(0) before each group of eight multiply-adds) is also
        // stable but doubled k_fs_dw (64 us): the dz workgroups became the launch's critical path.
#ifndef WZ_VARIANT
#define WZ_VARIANT 2
#endif
#if WZ_VARIANT == 0   // the reproducer's variant
#pragma unroll 8
        for (int c = 0; c < 64; ++c) acc += Ws[r * 65 + c] * *reinterpret_cast<const f32x4*>(xs + c * WZ_C + sq * 4);
#elif WZ_VARIANT == 1
        for (int c0 = 0; c0 < 64; c0 += 8) {
            float w[8];
            f32x4 xv[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                w[c] = Ws[r * 65 + c0 + c];
                xv[c] = *reinterpret_cast<const f32x4*>(xs + (c0 + c) * WZ_C + sq * 4);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int c = 0; c < 8; ++c) acc += w[c] * xv[c];
        }
#elif WZ_VARIANT == 2
#pragma unroll 8
        for (int c = 0; c < 64; ++c) {
            float w = Ws[r * 65 + c];
            asm volatile("" : "+v"(w));
            acc += w * *reinterpret_cast<const f32x4*>(xs + c * WZ_C + sq * 4);
        }
#endif
        if (r < M) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int sc = sq * 4 + t;
                if (sc < rv) out[((long)bh * M + r) * S + c0 + sc] = MODE ? acc[t] : 1.f / (eps + acc[t]);
            }
        }
    }
}
template <int MODE>
__global__ __launch_bounds__(FT) void k_fs_wz(const float* __restrict__ W, int ldw, const float* __restrict__ x,
                                              float* __restrict__ out, int M, int S, float eps) {
    __shared__ __attribute__((aligned(16))) float smem[WZ_SMEM / 4];
    wz_body<MODE, FT>(smem, W, ldw, x, out, M, S, eps, blockIdx.x, blockIdx.y, threadIdx.x);
}

// -------------------------------------------------------------------------------------------------
// k_fs_state / k_fs_state_fwd: the summaries of blocks of MORE than 64 tokens (S > 64: synchronous 64-row chunks; the common case
// S <= 64 is k_fs_state1c below).  Per (block group jg, bh): 8 block summaries in the interleaved transposed layout.
//   MODE 0 (forward) : state = V_j^T K_j ; ksum_j ; z_j[s] = Q_j[s] . ksum_j
//   MODE 1 (backward): state = dP_i^T Q_i with dP = dO / n ; dn_i[s] = -(dO_i[s] . O_i[s]) / n_i[s]
// 8 waves: wave w owns rows d2 = 16 (w & 3) .. and columns d1 = 32 (w >> 2) .. of every summary.
// Token tiles are fetched two blocks ahead into registers (S <= 64) and committed to LDS when needed.
// -------------------------------------------------------------------------------------------------
struct FsStateArgs {
    View x;   // MODE 0: k     MODE 1: q     (B operand: columns d1)
    View y;   // MODE 0: v     MODE 1: dout  (A operand: rows d2)
    View t;   // MODE 0: q (for z)   MODE 1: out (for the row dot)
    const int* idx;
    const float* W;
    int ldw;
    const float* ninv;   // MODE 1: [bh][M][S]  1 / n  (k_fs_wz<0>)
    u16* state;          // [bh][njg][4096][8]
    float* ksum;         // MODE 0: [bh][M][64]
    float* z_out;        // MODE 0: [bh][M][S]
    float* dn;           // MODE 1: [bh][M][S]
    int H, M, S;
    float eps;
    int relu, normalize;
};
constexpr int FS_STATE_SMEM = 2 * 64 * TLD * 2 + 8 * 64 * 4;

struct TileRegs { uint4 x, y, t; float ninv; };

// dot of 8 bf16 pairs held as two uint4
__device__ __forceinline__ float dot8(uint4 a, uint4 b) {
    const unsigned aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        d += __uint_as_float(aw[i] << 16) * __uint_as_float(bw[i] << 16) +
             __uint_as_float(aw[i] & 0xffff0000u) * __uint_as_float(bw[i] & 0xffff0000u);
    return d;
}

template <int MODE>
__global__ __launch_bounds__(FT8, 3) void k_fs_state(const FsStateArgs a) {   // (3: the second register set of the pair loop does not fit 128 VGPRs -- 9 spilled)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Xs = reinterpret_cast<u16*>(smem_raw);
    u16* Ys = Xs + 64 * TLD;
    float* part = reinterpret_cast<float*>(Ys + 64 * TLD);   // [8][64] column-sum partials of K (MODE 0)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int dt = wave & 3, th = wave >> 2;   // wave-uniform (scalar) so branches around MFMAs are scalar branches
    const int jg = blockIdx.x, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int S = a.S, M = a.M, njg = gridDim.x;
    const u16* xb = (const u16*)a.x.ptr + b * a.x.sb + h * a.x.sh;
    const u16* yb = (const u16*)a.y.ptr + b * a.y.sb + h * a.y.sh;
    const u16* tb = a.normalize ? (const u16*)a.t.ptr + b * a.t.sb + h * a.t.sh : nullptr;
    const int srow = tid >> 3, scol = (tid & 7) * 8;   // staging: thread -> (row, 8 columns = 16 bytes)

    // global -> registers for one 64-row chunk of block j (rows >= rv give zeros)
    // (every load unconditional: rows past the chunk's end are clamped onto its last row and zeroed in `commit` -- a load behind a
    // divergent branch makes the compiler wait for everything at the join, which would drain the one-chunk-ahead prefetch)
    auto issue = [&](int j, int c0, int rv, TileRegs& R) {
        const int sr = min(srow, rv - 1);
        const long tr = tok_row(a.idx, (long)j * S + c0 + sr);
        // MODE 1: Q and dO are read again by the next kernel (dQ): regular loads; O is not: streaming.  MODE 0: K, V stream.
        R.x = MODE == 1 ? gld<uint4>(xb + tr * a.x.sn + scol) : gld_stream16(xb + tr * a.x.sn + scol);
        R.y = MODE == 1 ? gld<uint4>(yb + tr * a.y.sn + scol) : gld_stream16(yb + tr * a.y.sn + scol);
        R.t = make_uint4(0, 0, 0, 0);
        R.ninv = 0.f;
        if (a.normalize) {   // (uniform)
            R.t = MODE == 1 ? gld_stream16(tb + tr * a.t.sn + scol) : gld<uint4>(tb + tr * a.t.sn + scol);
            if (MODE == 1) R.ninv = gld<float>(a.ninv + ((long)bh * M + j) * S + c0 + sr);
        }
    };
    // registers -> LDS.  MODE 1 folds dn = -(dO . O) / n and dP = dO / n into this step (no LDS pass).
    auto commit = [&](TileRegs& R, int j, int c0, int rv, int rfill) {
        if (srow >= rv) {
            R.x = R.y = R.t = make_uint4(0, 0, 0, 0);
            R.ninv = 0.f;
        }
        uint4 x = R.x, y = R.y;
        if (a.relu && srow < rv) {
            x = relu_eps8(x, a.eps);
            if (MODE == 0 && a.normalize) R.t = relu_eps8(R.t, a.eps);
        }
        if (MODE == 1 && a.normalize) {
            float d = dot8(R.y, R.t);
            d += __shfl_xor(d, 1, 64);
            d += __shfl_xor(d, 2, 64);
            d += __shfl_xor(d, 4, 64);
            if (srow < rv && (tid & 7) == 0) a.dn[((long)bh * M + j) * S + c0 + srow] = -d * R.ninv;
            unsigned yw[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                yw[i] = pack_bf16x2(__uint_as_float(yw[i] << 16) * R.ninv, __uint_as_float(yw[i] & 0xffff0000u) * R.ninv);
            y = make_uint4(yw[0], yw[1], yw[2], yw[3]);
        }
        if (srow < rfill) {
            *reinterpret_cast<uint4*>(Xs + srow * TLD + scol) = x;
            *reinterpret_cast<uint4*>(Ys + srow * TLD + scol) = y;
        }
    };

    f32x4 acc[IT][2];
#pragma unroll
    for (int jj = 0; jj < IT; ++jj) acc[jj][0] = acc[jj][1] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one staged chunk (tiles in LDS, barrier done): (MODE 0) column-sum partials of K, then the summaries
    auto chunk = [&](auto jjc, int rv, int rfill, float& ks) {
        constexpr int jj = decltype(jjc)::value;
        if (MODE == 0 && a.normalize) {   // thread -> column tid & 63, rows 8 (tid >> 6) ..
            const int col = tid & 63, pr = tid >> 6;
            for (int r = pr * 8; r < min(rv, pr * 8 + 8); ++r) ks += bf(Xs[r * TLD + col]);
        }
        for (int k0 = 0; k0 < rfill; k0 += 32) {
            const bf16x8 av = tr_read8(Ys, TLD, k0, dt * 16, lane);
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[jj][t] = mfma_bf16(av, tr_read8(Xs, TLD, k0, (2 * th + t) * 16, lane), acc[jj][t]);
        }
    };
    // after the partials are in `part` and a barrier: every thread sums the 8 partials of its 8 staging columns
    auto ksum8 = [&](float (&kv)[8]) {
#pragma unroll
        for (int t = 0; t < 8; ++t) kv[t] = 0.f;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(part + p * 64 + scol);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(part + p * 64 + scol + 4);
#pragma unroll
            for (int t = 0; t < 4; ++t) { kv[t] += lo[t]; kv[4 + t] += hi[t]; }
        }
    };
    auto write_ksum_z = [&](int j, int c0, int rv, const float (&kv)[8], uint4 tq, bool write_ksum) {
        if (write_ksum && tid < 8) {
#pragma unroll
            for (int t = 0; t < 8; ++t) a.ksum[((long)bh * M + j) * 64 + scol + t] = kv[t];
        }
        const unsigned qw[4] = {tq.x, tq.y, tq.z, tq.w};
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            d += __uint_as_float(qw[i] << 16) * kv[2 * i] + __uint_as_float(qw[i] & 0xffff0000u) * kv[2 * i + 1];
        d += __shfl_xor(d, 1, 64);
        d += __shfl_xor(d, 2, 64);
        d += __shfl_xor(d, 4, 64);
        if (srow < rv && (tid & 7) == 0) a.z_out[((long)bh * M + j) * S + c0 + srow] = d;
    };

    // multi-chunk blocks (S > 64): the next chunk (of this block or the first of the next one) is requested while the current one is
    // multiplied; z needs the complete ksum -> second pass over Q
    // Blocks of a whole number of 128-token pairs keep TWO chunks in flight (register sets RA / RB alternate inside one loop body, so
    // their indices stay static); other lengths one.
    TileRegs RA, RB;
    const int jend = min(M, (jg + 1) * IT);
    const bool pairs = (S & 127) == 0;
    if (jg * IT < M) {
        issue(jg * IT, 0, min(64, S), RA);
        if (pairs) issue(jg * IT, 64, 64, RB);
    }
    auto blockloop = [&](auto jjc) {
        constexpr int jj = decltype(jjc)::value;
        const int j = jg * IT + jj;
        if (j >= M) return;
        float ks = 0.f;
        // one chunk: stage the registers, request the chunk `ahead` steps on into them, multiply
        auto step = [&](int c0, TileRegs& R, int ahead) {
            const int rv = min(64, S - c0), rfill = (rv + 31) & ~31;
            commit(R, j, c0, rv, rfill);
            __syncthreads();
            {
                int c2 = c0 + 64 * ahead, j2 = j;
                if (c2 >= S) { c2 = ahead == 2 ? c2 - S : 0; ++j2; }   // (pairs: S is a multiple of 128; single steps: S may be ragged)
                if (j2 < jend) issue(j2, c2, min(64, S - c2), R);
            }
            chunk(jjc, rv, rfill, ks);
            __syncthreads();
        };
        if (pairs) {
            for (int c0 = 0; c0 < S; c0 += 128) {
                step(c0, RA, 2);
                step(c0 + 64, RB, 2);
            }
        } else {
            for (int c0 = 0; c0 < S; c0 += 64) step(c0, RA, 1);
        }
        if (MODE == 0 && a.normalize) {
            part[tid] = ks;
            __syncthreads();
            float kv[8];
            ksum8(kv);
            for (int c0 = 0; c0 < S; c0 += 64) {
                const int rv = min(64, S - c0);
                uint4 tq = make_uint4(0, 0, 0, 0);
                if (srow < rv) {
                    tq = *reinterpret_cast<const uint4*>(tb + tok_row(a.idx, (long)j * S + c0 + srow) * a.t.sn + scol);
                    if (a.relu) tq = relu_eps8(tq, a.eps);
                }
                write_ksum_z(j, c0, rv, kv, tq, c0 == 0);
            }
            __syncthreads();                       // `part` is rewritten by the next block
        }
    };
    blockloop(std::integral_constant<int, 0>{});
    blockloop(std::integral_constant<int, 1>{});
    blockloop(std::integral_constant<int, 2>{});
    blockloop(std::integral_constant<int, 3>{});
    blockloop(std::integral_constant<int, 4>{});
    blockloop(std::integral_constant<int, 5>{});
    blockloop(std::integral_constant<int, 6>{});
    blockloop(std::integral_constant<int, 7>{});

    // 16-byte interleaved store: lane -> (d2 = 16 dt + 4 (lane >> 4) + r, d1 = 16 (2 th + t) + (lane & 15))
    u16* sb = a.state + ((long)bh * njg + jg) * FE * IT;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int d2 = dt * 16 + (lane >> 4) * 4 + r, d1 = (2 * th + t) * 16 + (lane & 15);
            unsigned w[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) w[p] = pack_bf16x2(acc[2 * p][t][r], acc[2 * p + 1][t][r]);
            *reinterpret_cast<uint4*>(sb + ((long)d2 * FD + d1) * IT) = make_uint4(w[0], w[1], w[2], w[3]);
        }
}

constexpr int FS_STATE_FWD_SMEM = 3 * 64 * TLD * 2 + (8 * 64 + 64) * 4;

// Forward summaries (MODE 0 only): Q tile staged in LDS with K and V, two blocks ahead in registers.
struct TileRegs3 { uint4 x, y, t; };

template <int MODE>
__global__ __launch_bounds__(FT8, 4) void k_fs_state_fwd(const FsStateArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Xs = reinterpret_cast<u16*>(smem_raw);
    u16* Ys = Xs + 64 * TLD;
    u16* Ts = Ys + 64 * TLD;
    float* part = reinterpret_cast<float*>(Ts + 64 * TLD);   // [8][64]
    float* ksum_s = part + 512;                              // [64]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int dt = wave & 3, th = wave >> 2;
    const int jg = blockIdx.x, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int S = a.S, M = a.M, njg = gridDim.x;
    const u16* xb = (const u16*)a.x.ptr + b * a.x.sb + h * a.x.sh;
    const u16* yb = (const u16*)a.y.ptr + b * a.y.sb + h * a.y.sh;
    const u16* tb = a.normalize ? (const u16*)a.t.ptr + b * a.t.sb + h * a.t.sh : nullptr;
    const int srow = tid >> 3, scol = (tid & 7) * 8;         // staging: thread -> (row, 8 columns = 16 bytes)
    const bool tile_t = a.normalize && MODE == 1;   // third tile travels with the chunk

    // (every load unconditional: rows past the chunk's end are clamped onto its last row and zeroed in `commit`)
    auto issue = [&](long p, int rv, TileRegs3& R) {
        const long tr = tok_row(a.idx, p + min(srow, rv - 1));
        R.x = gld_stream16(xb + tr * a.x.sn + scol);                     // K, V: not read again in the forward
        R.y = gld_stream16(yb + tr * a.y.sn + scol);
        R.t = make_uint4(0, 0, 0, 0);
        if (tile_t) R.t = gld<uint4>(tb + tr * a.t.sn + scol);          // Q: the output kernel reads it next
    };
    auto commit = [&](TileRegs3& R, int rv, int rfill) {
        if (srow >= rv) R.x = R.y = R.t = make_uint4(0, 0, 0, 0);
        if (srow < rfill) {
            uint4 x = R.x, t = R.t;
            if (a.relu && srow < rv) {
                x = relu_eps8(x, a.eps);
                if (MODE == 0 && tile_t) t = relu_eps8(t, a.eps);
            }
            *reinterpret_cast<uint4*>(Xs + srow * TLD + scol) = x;
            *reinterpret_cast<uint4*>(Ys + srow * TLD + scol) = R.y;
            if (tile_t) *reinterpret_cast<uint4*>(Ts + srow * TLD + scol) = t;
        }
    };

    f32x4 acc[IT][2];
#pragma unroll
    for (int jj = 0; jj < IT; ++jj) acc[jj][0] = acc[jj][1] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one staged 64-row chunk of block jj (tiles in LDS, barrier done): side products + MFMAs
    auto chunk = [&](auto jjc, int j, int c0, int rv, int rfill, float& ks) {
        constexpr int jj = decltype(jjc)::value;
        if (MODE == 0 && a.normalize) {   // column sums of K: thread -> column tid & 63, rows 8 (tid >> 6) ..
            const int col = tid & 63, pr = tid >> 6;
            for (int r = pr * 8; r < min(rv, pr * 8 + 8); ++r) ks += bf(Xs[r * TLD + col]);
        }
        if (MODE == 1 && a.normalize) {   // dn and dP = dO / n (rounded to bf16); 8 threads per row, 16 bytes each
            float d = 0.f;
            uint4 yv = make_uint4(0, 0, 0, 0);
            if (srow < rv) {
                yv = *reinterpret_cast<const uint4*>(Ys + srow * TLD + scol);
                const uint4 ov = *reinterpret_cast<const uint4*>(Ts + srow * TLD + scol);
                const unsigned yw[4] = {yv.x, yv.y, yv.z, yv.w}, ow[4] = {ov.x, ov.y, ov.z, ov.w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    d += __uint_as_float(yw[i] << 16) * __uint_as_float(ow[i] << 16) +
                         __uint_as_float(yw[i] & 0xffff0000u) * __uint_as_float(ow[i] & 0xffff0000u);
            }
            d += __shfl_xor(d, 1, 64);
            d += __shfl_xor(d, 2, 64);
            d += __shfl_xor(d, 4, 64);
            if (srow < rv) {
                const float ni = a.ninv[((long)bh * M + j) * S + c0 + srow];
                if ((tid & 7) == 0) a.dn[((long)bh * M + j) * S + c0 + srow] = -d * ni;
                unsigned yw[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    yw[i] = pack_bf16x2(__uint_as_float(yw[i] << 16) * ni, __uint_as_float(yw[i] & 0xffff0000u) * ni);
                *reinterpret_cast<uint4*>(Ys + srow * TLD + scol) = make_uint4(yw[0], yw[1], yw[2], yw[3]);
            }
            __syncthreads();
        }
        for (int k0 = 0; k0 < rfill; k0 += 32) {
            const bf16x8 av = tr_read8(Ys, TLD, k0, dt * 16, lane);
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[jj][t] = mfma_bf16(av, tr_read8(Xs, TLD, k0, (2 * th + t) * 16, lane), acc[jj][t]);
        }
    };
    // ksum_j and z_j once all chunks of block j went through (MODE 0)
    auto finish_block = [&](int j, float ks) {
        const long p0 = (long)j * S;
        part[(tid >> 6) * 64 + (tid & 63)] = ks;
        __syncthreads();
        if (tid < 64) {
            float sacc = 0.f;
#pragma unroll
            for (int p = 0; p < 8; ++p) sacc += part[p * 64 + tid];
            ksum_s[tid] = sacc;
            a.ksum[((long)bh * M + j) * 64 + tid] = sacc;
        }
        __syncthreads();
        // second pass over Q: every thread dots its own 16-byte piece of a row with ksum (no staging: the piece is used by the thread
        // that loaded it), two chunks' loads in flight, 8 lanes per row reduced by shuffles
        float kv8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) kv8[i] = ksum_s[scol + i];
        for (int c0 = 0; c0 < S; c0 += 128) {
            uint4 tq[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int cu = min(c0 + 64 * u, S - 1), rv = min(64, S - cu);
                tq[u] = gld<uint4>(tb + tok_row(a.idx, p0 + cu + min(srow, rv - 1)) * a.t.sn + scol);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int cu = c0 + 64 * u, rv = min(64, S - cu);
                uint4 t = tq[u];
                if (a.relu) t = relu_eps8(t, a.eps);
                const unsigned qw[4] = {t.x, t.y, t.z, t.w};
                float d = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    d += __uint_as_float(qw[i] << 16) * kv8[2 * i] + __uint_as_float(qw[i] & 0xffff0000u) * kv8[2 * i + 1];
                d += __shfl_xor(d, 1, 64);
                d += __shfl_xor(d, 2, 64);
                d += __shfl_xor(d, 4, 64);
                if (cu < S && srow < rv && (tid & 7) == 0) a.z_out[((long)bh * M + j) * S + cu + srow] = d;
            }
        }
        __syncthreads();   // ksum_s / part are rewritten by the next block
    };

    // the next chunk (of this block or the first of the next one) is requested while the current one is multiplied
    // (blocks of a whole number of 128-token pairs: two chunks in flight, register sets RA / RB alternating inside one loop body)
    TileRegs3 RA, RB;
    const int jend = min(M, (jg + 1) * IT);
    const bool pairs = (S & 127) == 0;
    if (jg * IT < M) {
        issue((long)(jg * IT) * S, min(64, S), RA);
        if (pairs) issue((long)(jg * IT) * S + 64, 64, RB);
    }
    auto blockloop = [&](auto jjc) {
        constexpr int jj = decltype(jjc)::value;
        const int j = jg * IT + jj;
        if (j >= M) return;
        float ks = 0.f;
        auto step = [&](int c0, TileRegs3& R, int ahead) {
            const int rv = min(64, S - c0), rfill = (rv + 31) & ~31;
            commit(R, rv, rfill);
            __syncthreads();
            {
                int c2 = c0 + 64 * ahead, j2 = j;
                if (c2 >= S) { c2 = ahead == 2 ? c2 - S : 0; ++j2; }   // (pairs: S is a multiple of 128; single steps: S may be ragged)
                if (j2 < jend) issue((long)j2 * S + c2, min(64, S - c2), R);
            }
            chunk(jjc, j, c0, rv, rfill, ks);
            __syncthreads();
        };
        if (pairs) {
            for (int c0 = 0; c0 < S; c0 += 128) {
                step(c0, RA, 2);
                step(c0 + 64, RB, 2);
            }
        } else {
            for (int c0 = 0; c0 < S; c0 += 64) step(c0, RA, 1);
        }
        if (MODE == 0 && a.normalize) finish_block(j, ks);
    };
    blockloop(std::integral_constant<int, 0>{});
    blockloop(std::integral_constant<int, 1>{});
    blockloop(std::integral_constant<int, 2>{});
    blockloop(std::integral_constant<int, 3>{});
    blockloop(std::integral_constant<int, 4>{});
    blockloop(std::integral_constant<int, 5>{});
    blockloop(std::integral_constant<int, 6>{});
    blockloop(std::integral_constant<int, 7>{});

    // 16-byte interleaved store: lane -> (d2 = 16 dt + 4 (lane >> 4) + r, d1 = 16 (2 th + t) + (lane & 15))
    u16* sb = a.state + ((long)bh * njg + jg) * FE * IT;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int d2 = dt * 16 + (lane >> 4) * 4 + r, d1 = (2 * th + t) * 16 + (lane & 15);
            unsigned w[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) w[p] = pack_bf16x2(acc[2 * p][t][r], acc[2 * p + 1][t][r]);
            *reinterpret_cast<uint4*>(sb + ((long)d2 * FD + d1) * IT) = make_uint4(w[0], w[1], w[2], w[3]);
        }
}

// -------------------------------------------------------------------------------------------------
// k_fs_state1c<MODE, IDX, NORM>: the summaries of 8 blocks of <= 64 tokens each (the common case S <= 64), both directions, written
// as straight-line code: every global load of the 8-step pipeline is unconditional (rows / blocks past the end are clamped to valid
// addresses and zeroed after they arrive), and the small side outputs (ksum, z; dn) are parked in LDS and stored once, with the
// summaries, after the last block.  The earlier kernels guarded loads and 4-byte side stores with divergent branches inside
// the pipeline: the compiler then has to assume the worst at every join (s_waitcnt vmcnt(0) right after the prefetch was
// issued), which serialised the loads -- 63 -> 41 us at C2 for the forward summaries once the side stores were out of the loop.
//   MODE 0: x = K (relu + eps), y = V, t = Q (relu + eps):  state = V_j^T K_j, ksum_j, z_j[s] = Q_j[s] . ksum_j
//   MODE 1: x = Q (relu + eps), y = dO, t = O:              state = dP_i^T Q_i, dP = dO / n, dn_i[s] = -(dO_i[s] . O_i[s]) / n_i[s]
// -------------------------------------------------------------------------------------------------
constexpr int FS_STATE1C_SMEM = 2 * 64 * TLD * 2 + (64 + 2 * IT * 64) * 4;

template <int MODE, bool IDX, bool NORM>
__global__ __launch_bounds__(FT8, 4) void k_fs_state1c(const FsStateArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Xs = reinterpret_cast<u16*>(smem_raw);
    u16* Ys = Xs + 64 * TLD;
    float* ksum_s = reinterpret_cast<float*>(Ys + 64 * TLD);   // [64]      column sums of the block in flight (MODE 0)
    float* side = ksum_s + 64;                                  // [8][64]   z (MODE 0) / dn (MODE 1) of the 8 blocks
    float* ksum_all = side + IT * 64;                           // [8][64]   (MODE 0)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int dt = wave & 3, th = wave >> 2;
    const int jg = blockIdx.x, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int S = a.S, M = a.M, njg = gridDim.x;
    const u16* xb = (const u16*)a.x.ptr + b * a.x.sb + h * a.x.sh;
    const u16* yb = (const u16*)a.y.ptr + b * a.y.sb + h * a.y.sh;
    const u16* tb = NORM ? (const u16*)a.t.ptr + b * a.t.sb + h * a.t.sh : nullptr;
    const int srow = tid >> 3, scol = (tid & 7) * 8;            // staging: thread -> (row, 8 columns = 16 bytes)
    const int lrow = min(srow, S - 1);                          // rows past the block: a valid address, zeroed on arrival
    const bool valid = srow < S;
    const int nblk = min(IT, M - jg * IT);                      // blocks of this group (uniform)
    const int rfill = (S + 31) & ~31;

    // gather map: the 8 blocks' row indices are fetched up front, so the pipeline's loads depend on nothing in flight
    int tix[IDX ? IT : 1];
    if (IDX) {
#pragma unroll
        for (int jj = 0; jj < IT; ++jj) tix[jj] = gld<int>(a.idx + (long)(jg * IT + min(jj, nblk - 1)) * S + lrow);
    }
    auto issue = [&](auto jjc, TileRegs& R) {
        constexpr int jj = decltype(jjc)::value;
        const int jb = jg * IT + min(jj, nblk - 1);              // (uniform) blocks past the end: the last one again, never used
        const long tr = IDX ? (long)tix[IDX ? jj : 0] : (long)jb * S + lrow;
        // MODE 0: K, V are not read again in the forward (streaming loads); Q is (k_t16_out).  MODE 1: Q, dO are read by the
        // dQ kernel next; O is not.
        R.x = MODE == 1 ? gld<uint4>(xb + tr * a.x.sn + scol) : gld_stream16(xb + tr * a.x.sn + scol);
        R.y = MODE == 1 ? gld<uint4>(yb + tr * a.y.sn + scol) : gld_stream16(yb + tr * a.y.sn + scol);
        if (NORM) {
            R.t = MODE == 1 ? gld_stream16(tb + tr * a.t.sn + scol) : gld<uint4>(tb + tr * a.t.sn + scol);
            if (MODE == 1) R.ninv = gld<float>(a.ninv + ((long)bh * M + jb) * S + lrow);
        }
    };

    // accumulators of the block pair in flight; finished pairs are kept packed (bf16 x 2: even block low, odd block high)
    f32x4 acc[2][2];
    unsigned pk[IT / 2][2][4];
    s16x8 ones_;
#pragma unroll
    for (int t = 0; t < 8; ++t) ones_[t] = (short)0x3F80;       // bf16 1.0
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_);
    const uint4 zero4 = make_uint4(0, 0, 0, 0);

    TileRegs R0, R1;
    issue(std::integral_constant<int, 0>{}, R0);
    issue(std::integral_constant<int, 1>{}, R1);
    auto step = [&](auto jjc, auto fullc, TileRegs& R) {
        constexpr int jj = decltype(jjc)::value;
        constexpr bool FULL = decltype(fullc)::value;
        acc[jj & 1][0] = acc[jj & 1][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (FULL || jj < nblk) {                                // (uniform) short last group
            uint4 x = R.x, y = R.y, tq = zero4;
            if (NORM) tq = R.t;
            if (a.relu) {
                x = relu_eps8(x, a.eps);
                if (MODE == 0 && NORM) tq = relu_eps8(tq, a.eps);
            }
            if (MODE == 1 && NORM) {                            // dn = -(dO . O) / n and dP = dO / n (rounded to bf16)
                float d = dot8(y, tq);
                d += __shfl_xor(d, 1, 64);
                d += __shfl_xor(d, 2, 64);
                d += __shfl_xor(d, 4, 64);
                if ((tid & 7) == 0) side[jj * 64 + srow] = -d * R.ninv;
                unsigned yw[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    yw[i] = pack_bf16x2(__uint_as_float(yw[i] << 16) * R.ninv, __uint_as_float(yw[i] & 0xffff0000u) * R.ninv);
                y = make_uint4(yw[0], yw[1], yw[2], yw[3]);
            }
            // (component-wise selects: `valid ? x : zero` on the struct type becomes a select of two stack addresses)
            *reinterpret_cast<uint4*>(Xs + srow * TLD + scol) = make_uint4(valid ? x.x : 0u, valid ? x.y : 0u, valid ? x.z : 0u, valid ? x.w : 0u);
            *reinterpret_cast<uint4*>(Ys + srow * TLD + scol) = make_uint4(valid ? y.x : 0u, valid ? y.y : 0u, valid ? y.z : 0u, valid ? y.w : 0u);
            if (jj + 2 < IT) issue(std::integral_constant<int, (jj + 2 < IT ? jj + 2 : 0)>{}, R);
            __syncthreads();
            f32x4 ks[2];
            ks[0] = ks[1] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int k0 = 0; k0 < rfill; k0 += 32) {
                const bf16x8 av = tr_read8(Ys, TLD, k0, dt * 16, lane);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const bf16x8 bv = tr_read8(Xs, TLD, k0, (2 * th + t) * 16, lane);
                    acc[jj & 1][t] = mfma_bf16(av, bv, acc[jj & 1][t]);
                    if (MODE == 0 && NORM) ks[t] = mfma_bf16(ones, bv, ks[t]);   // column sums of K ride on the matrix pipe
                }
            }
            if (MODE == 0 && NORM && dt == 0 && lane < 16) {    // every row of the ones-product is the column sum
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int col = (2 * th + t) * 16 + lane;
                    ksum_s[col] = ks[t][0];
                    ksum_all[jj * 64 + col] = ks[t][0];
                }
            }
            __syncthreads();                                    // tiles consumed; column sums visible
            if (MODE == 0 && NORM) {                            // z_j[s] = Q_j[s] . ksum_j from the thread's own piece of Q
                const f32x4 lo = *reinterpret_cast<const f32x4*>(ksum_s + scol), hi = *reinterpret_cast<const f32x4*>(ksum_s + scol + 4);
                const unsigned qw[4] = {tq.x, tq.y, tq.z, tq.w};
                float d = __uint_as_float(qw[0] << 16) * lo[0] + __uint_as_float(qw[0] & 0xffff0000u) * lo[1] +
                          __uint_as_float(qw[1] << 16) * lo[2] + __uint_as_float(qw[1] & 0xffff0000u) * lo[3] +
                          __uint_as_float(qw[2] << 16) * hi[0] + __uint_as_float(qw[2] & 0xffff0000u) * hi[1] +
                          __uint_as_float(qw[3] << 16) * hi[2] + __uint_as_float(qw[3] & 0xffff0000u) * hi[3];
                d += __shfl_xor(d, 1, 64);
                d += __shfl_xor(d, 2, 64);
                d += __shfl_xor(d, 4, 64);
                if ((tid & 7) == 0) side[jj * 64 + srow] = d;
            }
        }
        if (jj & 1) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) pk[jj >> 1][t][r] = pack_bf16x2(acc[0][t][r], acc[1][t][r]);
        }
    };
    auto run = [&](auto fullc) {
        step(std::integral_constant<int, 0>{}, fullc, R0);
        step(std::integral_constant<int, 1>{}, fullc, R1);
        step(std::integral_constant<int, 2>{}, fullc, R0);
        step(std::integral_constant<int, 3>{}, fullc, R1);
        step(std::integral_constant<int, 4>{}, fullc, R0);
        step(std::integral_constant<int, 5>{}, fullc, R1);
        step(std::integral_constant<int, 6>{}, fullc, R0);
        step(std::integral_constant<int, 7>{}, fullc, R1);
    };
    if (nblk == IT) run(std::true_type{});
    else run(std::false_type{});

    // 16-byte interleaved store: lane -> (d2 = 16 dt + 4 (lane >> 4) + r, d1 = 16 (2 th + t) + (lane & 15))
    u16* sb = a.state + ((long)bh * njg + jg) * FE * IT;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int d2 = dt * 16 + (lane >> 4) * 4 + r, d1 = (2 * th + t) * 16 + (lane & 15);
            gst<uint4>(sb + ((long)d2 * FD + d1) * IT, make_uint4(pk[0][t][r], pk[1][t][r], pk[2][t][r], pk[3][t][r]));
        }
    if (NORM) {                                                 // side outputs: thread -> (block tid >> 6, position / column tid & 63)
        __syncthreads();
        const int jj = tid >> 6, c = tid & 63, j = jg * IT + jj;
        if (jj < nblk) {
            if (c < S) gst<float>((MODE == 0 ? a.z_out : a.dn) + ((long)bh * M + j) * S + c, side[jj * 64 + c]);
            if (MODE == 0) gst<float>(a.ksum + ((long)bh * M + j) * 64 + c, ksum_all[jj * 64 + c]);
        }
    }
}

// XCD-aware logical work index (bijective): workgroups that share a (b,h)'s summaries land on one XCD's L2.
__device__ __forceinline__ int xcd_swizzle(int wg, int nwg) {
    const int xcd = wg & 7, slot = wg >> 3, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// zero the bf16 lanes of v where the corresponding element of m is <= 0 (relu gradient mask)
__device__ __forceinline__ uint4 mask_pos8(uint4 v, uint4 m) {
    unsigned vv[4] = {v.x, v.y, v.z, v.w}, mm[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned lo = mm[i] & 0xffffu, hi = mm[i] >> 16;
        unsigned keep = 0;
        if ((lo & 0x7fffu) != 0 && !(lo & 0x8000u)) keep |= 0x0000ffffu;
        if ((hi & 0x7fffu) != 0 && !(hi & 0x8000u)) keep |= 0xffff0000u;
        vv[i] &= keep;
    }
    return make_uint4(vv[0], vv[1], vv[2], vv[3]);
}

// One wave stores a staged 64 x 64 bf16 tile: 8 passes of 8 full 128-byte rows.
// STREAM: the rows are not read again soon (gradients): nontemporal stores; the forward's output, which the backward's first
// kernel reads back, goes through the regular path
template <bool MASK, bool STREAM = true>
__device__ __forceinline__ void store64(u16* __restrict__ base, long sn, const int* __restrict__ idx, long p0, int rv,
                                        const u16* __restrict__ Os, const u16* __restrict__ mbase, long msn, int lane) {
#ifdef T16_NOSTORE
    if (rv != 12345) return;
#endif
    const int lane_o = opaque_lane(lane);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int row = p * 8 + (lane >> 3), c = (lane & 7) * 8;
        if (row < rv) {
            const long tr = tok_row(idx, p0 + row);
            uint4 v = *reinterpret_cast<const uint4*>(Os + gt_off(p * 8 + (lane_o >> 3), (lane_o & 7) * 8));
            if (MASK) v = mask_pos8(v, *reinterpret_cast<const uint4*>(mbase + tr * msn + c));
            if (STREAM) gst_stream16(base + tr * sn + c, v);
            else        gst<uint4>(base + tr * sn + c, v);
        }
    }
}
// -------------------------------------------------------------------------------------------------
// Arguments of the output kernel (k_t16_out, fused_tile16.hpp): mix the summaries of a tile into LDS, then
// O_i = (Q_i G_i) / n_i per block, staged in the block's own (dead) LDS slot -- no block barriers.
// -------------------------------------------------------------------------------------------------
struct FsOutArgs {
    View q;
    MView o;
    const int* idx;
    const float* W;
    int ldw;
    const u16* state;   // [bh][njg][4096][8]
    const float* ninv;  // [bh][M][S]  1 / n  (k_fs_wz<0>)
    int H, M, S, njg;
    float eps;
    int relu, normalize;
    unsigned long long* trace;   // debugging aid (mhla_debug_set_trace): per-workgroup phase timestamps, or null
    int cs;             // blocks of several 64-token chunks (S > 64): `cs` workgroups per tile, part p takes chunks p, p + cs, ..; 1 otherwise
};
// phase timestamp k of this workgroup (s_memtime ticks), first lane only; slot 15 of each record holds the XCC id
constexpr int TRACE_SLOTS = 16;
__device__ __forceinline__ void trace_mark(unsigned long long* trace, int k) {
    if (trace && threadIdx.x == 0) {
        trace[(long)blockIdx.x * TRACE_SLOTS + k] = __builtin_amdgcn_s_memtime();
        if (k == 0) {
            unsigned xcc, hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            trace[(long)blockIdx.x * TRACE_SLOTS + 15] = ((unsigned long long)xcc << 32) | hwid;
        }
    }
}
constexpr int FS_GT_BYTES = IT * FD * GLD * 2;
constexpr int FS_OUT_SMEM = FS_GT_BYTES;

// -------------------------------------------------------------------------------------------------
// k_fs_dw: dWp[bh][q][i][j] = sum_{e' in quarter q} dG[i][e'] KV[j][e']   (both in the interleaved layout)
// LDS images [e'][64 blocks] built from 16-byte pieces; both MFMA operands via transpose reads.
// grid (DW_SPLIT, bh).  The <dn_i, z_j> term (fp32 rows, the reduction index s contiguous) rides in split 0: both operands come
// straight from global memory in MFMA layout, split into bf16 hi + lo parts (three products: ~16 mantissa bits).
// -------------------------------------------------------------------------------------------------
struct FsDwArgs {
    const u16* dg;
    const u16* kv;
    const float* dn;   // [bh][M][S] or null
    const float* z;
    float* dwp;        // [bh][DW_SPLIT][64][64]
    int M, S, njg;
    // extra workgroups blockIdx.x >= DW_SPLIT of the same launch: dz = W^T dn (k_fs_wz<1>'s work; independent of dW, needed
    // by the next kernel) -- a latency-bound step that disappears beside the bandwidth-bound one
    const float* W;
    int ldw;
    float* dz;
    // per-tile "dQ done" flags of the token-gradient launch that follows (k_tile_bwd): cleared here, one workgroup per (b,h)
    int* done;
    int* err;    // the launch's error word, cleared here as well
    int ntt;
};
constexpr int DW_EC = 128;                       // e' rows per LDS image
constexpr int DW_SPLIT = 8;                      // e' splits per (b,h)
constexpr int DW_LDI = 72;
constexpr int FS_DW_SMEM = 2 * DW_EC * DW_LDI * 2;   // 36864 >= WZ_SMEM

template <int UNIT = 0>   // (a template only so that fused.hpp can be part of several translation units)
__global__ __launch_bounds__(FT8) void k_fs_dw(const FsDwArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Ai = reinterpret_cast<u16*>(smem_raw);   // dG image [256 e'][72]
    u16* Bi = Ai + DW_EC * DW_LDI;                // KV image
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, kg = lane >> 4;
    const int wi = wave & 3, wj = wave >> 2;   // wave -> rows i = 16 wi .., columns j = 32 wj ..
    const int qtr = blockIdx.x, bh = blockIdx.y, njg = a.njg;
    if (qtr == 0 && a.done && tid < a.ntt) a.done[bh * a.ntt + tid] = 0;
    if (qtr == 0 && bh == 0 && a.err && tid == 0) *a.err = 0;   // the launch's error word (tail of the workspace: bwd_err_word)
    if (qtr >= DW_SPLIT) {   // workgroup-uniform role switch
        wz_body<1, FT8>(reinterpret_cast<float*>(smem_raw), a.W, a.ldw, a.dn, a.dz, a.M, a.S, 0.f, qtr - DW_SPLIT, bh, tid);
        return;
    }
#ifdef FS_DW_PROBE_NO_DW_ROLE   // tools/probes/pk_fma_repro.hip: only the dz workgroups do anything (1: dW code compiled out;
#if FS_DW_PROBE_NO_DW_ROLE == 1 //  2: dW code kept -- same register allocation -- but skipped at run time when dwp is null)
    return;
#else
    if (a.dwp == nullptr) return;
#endif
#endif
    const u16* dg = a.dg + (long)bh * njg * FE * IT;
    const u16* kv = a.kv + (long)bh * njg * FE * IT;
    f32x4 acc[2];
    acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int NP = 2 * DW_EC * 8 / FT8;  // 16-byte pieces per thread per chunk
    uint4 pr[NP];
    auto issue = [&](long e0) {
#pragma unroll
        for (int t = 0; t < NP; ++t) {
            const int v = tid + t * FT8;
            const int which = v / (DW_EC * 8), rem = v - which * DW_EC * 8, g = rem / DW_EC, r = rem - g * DW_EC;
            const uint4 ld = gld<uint4>((which ? kv : dg) + ((long)min(g, njg - 1) * FE + e0 + r) * IT);
            pr[t] = (g < njg) ? ld : make_uint4(0, 0, 0, 0);
        }
    };
    // The <dn_i, z_j> term: 32-token steps of the reduction over s are dealt to the splits (step t -> split t % DW_SPLIT).  The
    // rows are fp32 with s contiguous, i.e. already in MFMA operand order: lane (n, kg) holds s = 32 step + 8 kg .. + 7 of row
    // i = 16 wi + n (A) / j = 16 (2 wj + t) + n (B).  Loaded first, multiplied last (bf16 hi + lo parts, three products).
    const int nsteps = (a.S + 31) / 32;
    float dnr[8], zr[2][8];
    auto load_row8 = [&](float (&dst)[8], const float* base, int row, int s0) {
        const bool ok = row < a.M;
        const float* p = base + ((long)bh * a.M + min(row, a.M - 1)) * a.S;
        if ((a.S & 7) == 0) {   // whole groups of 8: two 16-byte loads (rows are 32-byte aligned)
            const int sc = min(s0, a.S - 8);
            const f32x4 lo = gld<f32x4>(p + sc), hi = gld<f32x4>(p + sc + 4);
#pragma unroll
            for (int t = 0; t < 4; ++t) { dst[t] = (ok && s0 < a.S) ? lo[t] : 0.f; dst[4 + t] = (ok && s0 < a.S) ? hi[t] : 0.f; }
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float v = gld<float>(p + min(s0 + t, a.S - 1));
                dst[t] = (ok && s0 + t < a.S) ? v : 0.f;
            }
        }
    };
    auto load_dnz = [&](int step) {
        const int s0 = step * 32 + kg * 8;
        load_row8(dnr, a.dn, wi * 16 + n, step < nsteps ? s0 : a.S);
        load_row8(zr[0], a.z, (2 * wj) * 16 + n, step < nsteps ? s0 : a.S);
        load_row8(zr[1], a.z, (2 * wj + 1) * 16 + n, step < nsteps ? s0 : a.S);
    };
    auto mma_dnz = [&](const float (&x)[8], const float (&y)[8], f32x4& c) {
        s16x8 xh, xl, yh, yl;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const u16 a_ = cvt_bf16(x[t]), b_ = cvt_bf16(y[t]);
            xh[t] = (short)a_; xl[t] = (short)cvt_bf16(x[t] - bf(a_));
            yh[t] = (short)b_; yl[t] = (short)cvt_bf16(y[t] - bf(b_));
        }
        c = mfma_bf16(__builtin_bit_cast(bf16x8, xh), __builtin_bit_cast(bf16x8, yh), c);
        c = mfma_bf16(__builtin_bit_cast(bf16x8, xh), __builtin_bit_cast(bf16x8, yl), c);
        c = mfma_bf16(__builtin_bit_cast(bf16x8, xl), __builtin_bit_cast(bf16x8, yh), c);
    };
    if (a.dn) load_dnz(qtr);   // steps beyond the last one load clamped addresses and contribute zeros
    const long ebase = (long)qtr * (FE / DW_SPLIT);
    issue(ebase);
    for (int ec = 0; ec < FE / DW_SPLIT; ec += DW_EC) {
#pragma unroll
        for (int t = 0; t < NP; ++t) {
            const int v = tid + t * FT8;
            const int which = v / (DW_EC * 8), rem = v - which * DW_EC * 8, g = rem / DW_EC, r = rem - g * DW_EC;
            *reinterpret_cast<uint4*>((which ? Bi : Ai) + r * DW_LDI + g * 8) = pr[t];
        }
        if (ec + DW_EC < FE / DW_SPLIT) issue(ebase + ec + DW_EC);   // next chunk in flight during the MFMAs
        __syncthreads();
        for (int k0 = 0; k0 < DW_EC; k0 += 32) {
            const bf16x8 av = tr_read8(Ai, DW_LDI, k0, wi * 16, lane);
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = mfma_bf16(av, tr_read8(Bi, DW_LDI, k0, (2 * wj + t) * 16, lane), acc[t]);
        }
        __syncthreads();
    }
    if (a.dn) {
        mma_dnz(dnr, zr[0], acc[0]);
        mma_dnz(dnr, zr[1], acc[1]);
        for (int step = qtr + DW_SPLIT; step < nsteps; step += DW_SPLIT) {   // blocks longer than 256 tokens (rare)
            load_dnz(step);
            mma_dnz(dnr, zr[0], acc[0]);
            mma_dnz(dnr, zr[1], acc[1]);
        }
    }
    float* out = a.dwp + ((long)bh * DW_SPLIT + qtr) * 64 * 64;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(wi * 16 + kg * 4 + r) * 64 + (2 * wj + t) * 16 + n] = acc[t][r];
}

// Deterministic reduction of the per-(b,h, split) dW partials, run by the LAST workgroups of the k_tile_bwd launch (they
// fill the tail of that launch instead of costing a latency-bound launch of their own): workgroup r -> 32 consecutive elements
// of the padded [64][64] matrix; 64 part-lanes of 8 threads (16-byte loads), each summing every 64th partial in four
// independent chains, then a fixed-order sum over the part-lanes through LDS.  512 threads, DWR_WGS workgroups.
constexpr int DWR_WGS = 128;
__device__ __forceinline__ void dw_reduce_body(float* __restrict__ red /* [64][32] */, const float* __restrict__ dwp,
                                               float* __restrict__ dW, int M, int nparts, int r, int tid) {
    const int pl = tid >> 3, q = tid & 7, e0 = r * 32 + q * 4;
    f32x4 s[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) s[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int p0 = pl; p0 < nparts; p0 += 256) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int p = p0 + 64 * c;
            const f32x4 v = gld<f32x4>(dwp + (long)min(p, nparts - 1) * 4096 + e0);
            if (p < nparts) s[c] += v;
        }
    }
    *reinterpret_cast<f32x4*>(red + pl * 32 + q * 4) = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (tid < 32) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 64; ++k) v += red[k * 32 + tid];
        const int e = r * 32 + tid, i = e >> 6, j = e & 63;
        if (i < M && j < M) dW[(long)i * M + j] = v;
    }
}

// -------------------------------------------------------------------------------------------------
// Arguments of the token-gradient launch (k_tile_bwd: dQ role, dK/dV role, dW reduction; fused_tile16.hpp):
//   dq : Gt = mix(W, KV)    -> dQ_j = (dO_j G_j^T) / n_j + dz_j (x) ksum_j  (relu mask) ; dksum_j
//   dkv: Gt = mix(W^T, dG)  -> dK_j = V_j dKV_j^T + 1 dksum_j^T (relu mask) ; dV_j = K_j dKV_j
// -------------------------------------------------------------------------------------------------
struct FsTokArgs {
    View q, k, v, dout;
    MView dq, dk, dv;
    const int* idx;
    const float* W;
    int ldw;
    const u16* state;   // KV^T
    const u16* dstate;  // dG^T
    const float* ninv;  // [bh][M][S]
    const float* dz;    // [bh][M][S]  (the dz workgroups of k_fs_dw)
    const float* ksum;
    float* dksum;       // [bh][M][64]: written by the dQ role, read by the dK/dV role
    int H, M, S, njg;
    float eps;
    int relu, normalize;
    // the launch's last DWR_WGS workgroups reduce the dW partials (dw_reduce_body)
    const float* dwp;
    float* dW;
    int nparts, ntiles;
    int* done;          // [ntiles]: set by a tile's dQ workgroup once its dksum rows are written, awaited by its dK/dV workgroup
    int* err;           // error word of the launch: raised by a dK/dV workgroup whose flag did not arrive (tile_wait)
    int x0;             // role offset added to blockIdx.x (0: one launch for all roles; ntiles: the second of two launches)
    int drop_signal;    // testing aid (MHLA_DEBUG_DROP_SIGNAL=1): the dQ role does not raise its flag
    unsigned long long* trace;
    // blocks of several 64-token chunks (S > 64): `cs` workgroups per tile and role, part p takes chunks p, p + cs, ..; the dQ
    // parts write their share of dksum to dksum + p * dks_part and raise flag `cs * tile + p`; a dK/dV part waits for all `cs`
    // flags of its tile and adds the shares in part order (deterministic).  cs = 1 (and ntiles = tiles) when S <= 64.
    int cs;
    long dks_part;
};
// Hand-over of a tile's dksum rows between two workgroups of one launch (the waiting one has the higher blockIdx: it is dispatched
// after the signalling one, which never waits itself).  No fences: an agent-scope release / acquire pair writes back and
// invalidates the XCD's whole L2, which every other workgroup on it pays for (measured: 3.5x the kernel time).  Instead the few
// values that cross are written and read with agent-coherent accesses (sc1: through to / from the memory side), the writer waits
// for its stores to be acknowledged (vmcnt(0)) before the flag goes out, and the flag is polled with the same kind of load.
__device__ __forceinline__ void coherent_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float coherent_load(const float* p) {
    return __hip_atomic_load(const_cast<float*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void tile_signal(int* flag, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The wait is bounded: a flag that never arrives (a dispatch order other than the one this protocol relies on, a wrong flag index)
// must not hang the GPU.  After TILE_WAIT_POLLS polls (~1 us each: a second or more, against tile lifetimes of ~30 us) the waiter
// raises the launch's error word and reports the expiry to its whole workgroup through `lds_word` (an LDS word of the caller,
// returned to every thread) -- the caller then poisons its dksum rows with NaN, so that the tile's dk comes out as NaN: loud in
// the next loss / gradient-norm check without any host synchronisation.  mhla_blockmix_bwd_status is the synchronous form of
// the same report (MHLA_ELAUNCH).
constexpr int TILE_WAIT_POLLS = 1 << 20;
__device__ __forceinline__ bool tile_wait(int* flag, int* err, int* lds_word, int tid) {
    if (tid == 0) {
        int polls = 0, expired = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            if (++polls > TILE_WAIT_POLLS) {
                __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                expired = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(16);
        }
        *lds_word = expired;
    }
    __syncthreads();
    return *lds_word != 0;
}
// the same wait for `n` consecutive flags (the dQ parts of a tile with multi-chunk blocks)
__device__ __forceinline__ bool tile_wait_n(int* flags, int n, int* err, int* lds_word, int tid) {
    if (tid == 0) {
        int polls = 0, expired = 0;
        for (int i = 0; i < n && !expired; ++i)
            while (__hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                if (++polls > TILE_WAIT_POLLS) {
                    __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    expired = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(16);
            }
        *lds_word = expired;
    }
    __syncthreads();
    return *lds_word != 0;
}
// 16-block (TTP) tiles per (b,h) for njg groups of 8 blocks: the one place the flag array, the flag clearing and the tile
// launches take their count from
__host__ __device__ constexpr int tiles_per_bh(int njg, int ttp) { return (njg * IT + ttp - 1) / ttp; }
constexpr int FS_TOK_SMEM = FS_GT_BYTES;

}  // namespace fast
}  // namespace mhla
