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
//   forward : k_fs_state<0> (KV^T, ksum, z)  ->  k_fs_out (mix + n + O)
//   backward: k_fs_state<1> (dG^T, dn)  ->  k_fs_dw (dW partials)  ->  k_fs_bwd_tok (mix G, dQ; mix dKV, dK, dV)
#pragma once
#include "common.cuh"

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
constexpr int GLD = 72;        // LDS row stride (bf16) of the mixed summaries Gt[i][d2][d1]
constexpr int IT = 8;          // blocks per workgroup tile (= interleave factor of the state layout)
constexpr int FT = 256;        // threads per workgroup

#define LDS_S16X4(p) ((__attribute__((address_space(3))) s16x4*)(p))

__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

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

__device__ __forceinline__ uint4 relu_eps8(uint4 v, float eps) {
    unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float lo = fmaxf(__uint_as_float(w[i] << 16), 0.f) + eps;
        const float hi = fmaxf(__uint_as_float(w[i] & 0xffff0000u), 0.f) + eps;
        w[i] = (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// Stage rows [p0, p0 + rv) x 64 bf16 of a token view into an LDS tile [rfill][TLD] (rows >= rv zero).
// 256 threads: thread t -> row t >> 2, 16 columns starting at (t & 3) * 16.
template <bool RELU>
__device__ __forceinline__ void stage_rows(u16* __restrict__ dst, const u16* __restrict__ base, long sn,
                                           const int* __restrict__ idx, long p0, int rv, int rfill, float eps, int tid) {
    for (int r = tid >> 2; r < rfill; r += FT / 4) {
        const int c = (tid & 3) * 16;
        uint4 a = make_uint4(0, 0, 0, 0), b = a;
        if (r < rv) {
            const u16* src = base + tok_row(idx, p0 + r) * sn + c;
            a = *reinterpret_cast<const uint4*>(src);
            b = *reinterpret_cast<const uint4*>(src + 8);
            if (RELU) { a = relu_eps8(a, eps); b = relu_eps8(b, eps); }
        }
        *reinterpret_cast<uint4*>(dst + r * TLD + c) = a;
        *reinterpret_cast<uint4*>(dst + r * TLD + c + 8) = b;
    }
}

__device__ __forceinline__ float bf(u16 h) { return __uint_as_float(((unsigned)h) << 16); }

// Partial sums of a [M] x [M, S] product for one 64-token chunk: thread (s = tid & 63, q = tid >> 6) adds the
// terms j = q, q + 4, ...  of  sum_j w[j * wstride] * x[j * S + s]; the four partials of a column are summed
// by the reader after a barrier (wz_sum).  Used for n_i = W[i,:] z + eps and dz_j = W[:,j]^T dn.
__device__ __forceinline__ float wz_partial(const float* __restrict__ w, long wstride, const float* __restrict__ x,
                                            int M, int S, int s, bool valid, int q) {
    float acc = 0.f;
    if (valid)
        for (int j = q; j < M; j += 4) acc += w[(long)j * wstride] * x[(long)j * S + s];
    return acc;
}
__device__ __forceinline__ float wz_sum(const float* __restrict__ part, int row) {
    return part[row] + part[64 + row] + part[128 + row] + part[192 + row];
}

// -------------------------------------------------------------------------------------------------
// k_fs_state: per (block group jg, bh): 8 block summaries in the interleaved transposed layout.
//   MODE 0 (forward) : state = V_j^T K_j ; ksum_j ; z_j[s] = Q_j[s] . ksum_j
//   MODE 1 (backward): state = dP_i^T Q_i with dP = dO / n ; dn_i[s] = -(dO_i[s] . O_i[s]) / n_i[s]
// -------------------------------------------------------------------------------------------------
struct FsStateArgs {
    View x;   // MODE 0: k     MODE 1: q     (B operand: columns d1)
    View y;   // MODE 0: v     MODE 1: dout  (A operand: rows d2)
    View t;   // MODE 0: q (for z)   MODE 1: out (for the row dot)
    const int* idx;
    const float* W;
    int ldw;
    const float* z_in;   // MODE 1: [bh][M][S]
    u16* state;          // [bh][njg][4096][8]
    float* ksum;         // MODE 0: [bh][M][64]
    float* z_out;        // MODE 0: [bh][M][S]
    float* dn;           // MODE 1: [bh][M][S]
    int H, M, S;
    float eps;
    int relu, normalize;
};
constexpr int FS_STATE_SMEM = 3 * 64 * TLD * 2 + (4 * 64 + 64) * 4;

template <int MODE>
__global__ __launch_bounds__(FT) void k_fs_state(const FsStateArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Xs = reinterpret_cast<u16*>(smem_raw);
    u16* Ys = Xs + 64 * TLD;
    u16* Ts = Ys + 64 * TLD;
    float* part = reinterpret_cast<float*>(Ts + 64 * TLD);   // [4][64]
    float* ksum_s = part + 256;                              // [64]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int jg = blockIdx.x, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int S = a.S, M = a.M, njg = gridDim.x;
    const u16* xb = (const u16*)a.x.ptr + b * a.x.sb + h * a.x.sh;
    const u16* yb = (const u16*)a.y.ptr + b * a.y.sb + h * a.y.sh;
    const u16* tb = a.normalize ? (const u16*)a.t.ptr + b * a.t.sb + h * a.t.sh : nullptr;
    const bool single = S <= 64;

    f32x4 acc[IT][4];
#pragma unroll
    for (int jj = 0; jj < IT; ++jj)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[jj][tn] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int jj = 0; jj < IT; ++jj) {
        const int j = jg * IT + jj;
        if (j >= M) continue;
        const long p0 = (long)j * S;
        float ks = 0.f;
        for (int c0 = 0; c0 < S; c0 += 64) {
            const int rv = min(64, S - c0), rfill = (rv + 31) & ~31;
            if (a.relu) stage_rows<true>(Xs, xb, a.x.sn, a.idx, p0 + c0, rv, rfill, a.eps, tid);
            else        stage_rows<false>(Xs, xb, a.x.sn, a.idx, p0 + c0, rv, rfill, a.eps, tid);
            stage_rows<false>(Ys, yb, a.y.sn, a.idx, p0 + c0, rv, rfill, 0.f, tid);
            if (a.normalize && (MODE == 1 || single)) {
                if (MODE == 0 && a.relu) stage_rows<true>(Ts, tb, a.t.sn, a.idx, p0 + c0, rv, rfill, a.eps, tid);
                else                     stage_rows<false>(Ts, tb, a.t.sn, a.idx, p0 + c0, rv, rfill, 0.f, tid);
            }
            if (MODE == 1 && a.normalize)
                part[tid] = wz_partial(a.W + (long)j * a.ldw, 1, a.z_in + (long)bh * M * S, M, S, c0 + (tid & 63),
                                       (tid & 63) < rv, tid >> 6);
            __syncthreads();
            if (MODE == 0 && a.normalize) {   // column sums of K
                const int col = tid & 63, pr = tid >> 6;
                for (int r = pr * 16; r < min(rv, pr * 16 + 16); ++r) ks += bf(Xs[r * TLD + col]);
            }
            if (MODE == 1 && a.normalize) {   // dn and dP = dO / n (rounded to bf16)
                const int r = tid >> 2, cq = (tid & 3) * 16;
                float d = 0.f;
                if (r < rv) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) d += bf(Ys[r * TLD + cq + c]) * bf(Ts[r * TLD + cq + c]);
                }
                d += __shfl_xor(d, 1, 64);
                d += __shfl_xor(d, 2, 64);
                if (r < rv) {
                    const float ni = 1.f / (a.eps + wz_sum(part, r));
                    if ((tid & 3) == 0) a.dn[((long)bh * M + j) * S + c0 + r] = -d * ni;
#pragma unroll
                    for (int c = 0; c < 16; ++c) Ys[r * TLD + cq + c] = f32_to_bf16(bf(Ys[r * TLD + cq + c]) * ni);
                }
                __syncthreads();
            }
            for (int k0 = 0; k0 < rfill; k0 += 32) {
                const bf16x8 av = tr_read8(Ys, TLD, k0, wave * 16, lane);
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) acc[jj][tn] = mfma_bf16(av, tr_read8(Xs, TLD, k0, tn * 16, lane), acc[jj][tn]);
            }
            __syncthreads();
        }
        if (MODE == 0 && a.normalize) {
            part[(tid >> 6) * 64 + (tid & 63)] = ks;
            __syncthreads();
            if (tid < 64) {
                const float s = part[tid] + part[64 + tid] + part[128 + tid] + part[192 + tid];
                ksum_s[tid] = s;
                a.ksum[((long)bh * M + j) * 64 + tid] = s;
            }
            __syncthreads();
            for (int c0 = 0; c0 < S; c0 += 64) {
                const int rv = min(64, S - c0);
                if (!single) {
                    if (a.relu) stage_rows<true>(Ts, tb, a.t.sn, a.idx, p0 + c0, rv, rv, a.eps, tid);
                    else        stage_rows<false>(Ts, tb, a.t.sn, a.idx, p0 + c0, rv, rv, 0.f, tid);
                    __syncthreads();
                }
                const int r = tid >> 2, cq = (tid & 3) * 16;
                float d = 0.f;
                if (r < rv) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) d += bf(Ts[r * TLD + cq + c]) * ksum_s[cq + c];
                }
                d += __shfl_xor(d, 1, 64);
                d += __shfl_xor(d, 2, 64);
                if (r < rv && (tid & 3) == 0) a.z_out[((long)bh * M + j) * S + c0 + r] = d;
                __syncthreads();
            }
        }
    }

    // 16-byte interleaved store: lane -> (d2 = 16 wave + 4 (lane >> 4) + r, d1 = 16 tn + (lane & 15))
    u16* sb = a.state + ((long)bh * njg + jg) * FE * IT;
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int d2 = wave * 16 + (lane >> 4) * 4 + r, d1 = tn * 16 + (lane & 15);
            unsigned w[4];
#pragma unroll
            for (int p = 0; p < 4; ++p)
                w[p] = (unsigned)f32_to_bf16(acc[2 * p][tn][r]) | ((unsigned)f32_to_bf16(acc[2 * p + 1][tn][r]) << 16);
            *reinterpret_cast<uint4*>(sb + ((long)d2 * FD + d1) * IT) = make_uint4(w[0], w[1], w[2], w[3]);
        }
}

// -------------------------------------------------------------------------------------------------
// Mixing of one 8-block tile into LDS:  Gt[ii][d2][d1] = sum_j Wm(i0 + ii, j) state[j][d2][d1]
//   TRANSW 0: Wm(i, j) = W[i][j]      TRANSW 1: Wm(i, j) = W[j][i]
// MFMA: rows = 16 consecutive e' (A operand: one 16-byte global load per lane), cols = the 8 blocks of
// the tile (B operand: W hi / lo bf16 parts), reduction over j in steps of 32 blocks (M <= 64: 2 steps).
// -------------------------------------------------------------------------------------------------
template <int TRANSW>
__device__ __forceinline__ void mix_tile_to_lds(u16* __restrict__ Gt, const u16* __restrict__ state_bh, int njg,
                                                const float* __restrict__ W, int ldw, int M, int i0, int tid) {
    const int wave = tid >> 6, lane = tid & 63, n = lane & 15, kg = lane >> 4;
    bf16x8 bhi[2], blo[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        s16x8 hi, lo;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int j = ks * 32 + kg * 8 + t, i = i0 + n;
            float w = 0.f;
            if (n < IT && i < M && j < M) w = TRANSW ? W[(long)j * ldw + i] : W[(long)i * ldw + j];
            const u16 h = f32_to_bf16(w);
            hi[t] = (short)h;
            lo[t] = (short)f32_to_bf16(w - bf(h));
        }
        bhi[ks] = __builtin_bit_cast(bf16x8, hi);
        blo[ks] = __builtin_bit_cast(bf16x8, lo);
    }
    const bool two = njg > 4;
    constexpr int UN = 8;
    for (int et0 = wave * UN; et0 < FE / 16; et0 += 4 * UN) {
        uint4 av[UN][2];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long e = (long)(et0 + u) * 16 + n;
            av[u][0] = (kg < njg) ? *reinterpret_cast<const uint4*>(state_bh + ((long)kg * FE + e) * IT) : make_uint4(0, 0, 0, 0);
            av[u][1] = (two && 4 + kg < njg) ? *reinterpret_cast<const uint4*>(state_bh + ((long)(4 + kg) * FE + e) * IT)
                                             : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, av[u][0]);
            c = mfma_bf16(a0, bhi[0], c);
            c = mfma_bf16(a0, blo[0], c);
            if (two) {
                const bf16x8 a1 = __builtin_bit_cast(bf16x8, av[u][1]);
                c = mfma_bf16(a1, bhi[1], c);
                c = mfma_bf16(a1, blo[1], c);
            }
            if (n < IT) {
                const int et = et0 + u, d2 = et >> 2, d1 = (et & 3) * 16 + kg * 4;
                uint2 pk;
                pk.x = (unsigned)f32_to_bf16(c[0]) | ((unsigned)f32_to_bf16(c[1]) << 16);
                pk.y = (unsigned)f32_to_bf16(c[2]) | ((unsigned)f32_to_bf16(c[3]) << 16);
                *reinterpret_cast<uint2*>(Gt + ((long)(n * FD + d2)) * GLD + d1) = pk;
            }
        }
    }
}

// A operand (16 rows x 32 k) straight from a token view: lane (m = lane & 15, kg = lane >> 4) loads
// row (p0 + m), columns k0 + 8 kg .. + 7.  Rows >= rv give zeros.
template <bool RELU>
__device__ __forceinline__ bf16x8 load_a_rows(const u16* __restrict__ base, long sn, const int* __restrict__ idx,
                                              long p0, int rv, int k0, float eps, int lane) {
    const int m = lane & 15, kg = lane >> 4;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (m < rv) {
        v = *reinterpret_cast<const uint4*>(base + tok_row(idx, p0 + m) * sn + k0 + kg * 8);
        if (RELU) v = relu_eps8(v, eps);
    }
    return __builtin_bit_cast(bf16x8, v);
}

// -------------------------------------------------------------------------------------------------
// k_fs_out: per (tile it, bh): mix 8 summaries into LDS, then O_i = (Q_i G_i) / n_i for the 8 blocks.
// -------------------------------------------------------------------------------------------------
struct FsOutArgs {
    View q;
    MView o;
    const int* idx;
    const float* W;
    int ldw;
    const u16* state;   // [bh][njg][4096][8]
    const float* z;     // [bh][M][S]
    int H, M, S, njg;
    float eps;
    int relu, normalize;
};
constexpr int FS_GT_BYTES = IT * FD * GLD * 2;
constexpr int FS_OUT_SMEM = FS_GT_BYTES + 256 * 4;

// one 64-row chunk of one block: acc[tn] = A(rows 16 wave ..) x Gt-block (B, k = d1 contiguous)
__device__ __forceinline__ void rows_times_gt(f32x4 (&acc)[4], const bf16x8 (&a)[2], const u16* __restrict__ Gb, int lane) {
    const int n = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 bv = *reinterpret_cast<const bf16x8*>(Gb + (tn * 16 + n) * GLD + ks * 32 + kg * 8);
            acc[tn] = mfma_bf16(a[ks], bv, acc[tn]);
        }
    }
}
// same with the B operand transposed: B[k = row of Gt][n = column of Gt] (hardware transpose read)
__device__ __forceinline__ void rows_times_gt_t(f32x4 (&acc)[4], const bf16x8 (&a)[2], const u16* __restrict__ Gb, int lane) {
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) acc[tn] = mfma_bf16(a[ks], tr_read8(Gb, GLD, ks * 32, tn * 16, lane), acc[tn]);
    }
}

// stage a wave's 16 x 64 fp32 result tile (C layout) as bf16 into Os[row][GLD] and store 64 rows coalesced
__device__ __forceinline__ void stage_c_tile(u16* __restrict__ Os, const f32x4 (&acc)[4], int wave, int lane) {
    const int n = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) Os[(wave * 16 + kg * 4 + r) * GLD + tn * 16 + n] = f32_to_bf16(acc[tn][r]);
}
__device__ __forceinline__ void store_rows(u16* __restrict__ base, long sn, const int* __restrict__ idx, long p0, int rv,
                                           const u16* __restrict__ Os, int tid) {
    const int r = tid >> 2, c = (tid & 3) * 16;
    if (r < rv) {
        u16* dst = base + tok_row(idx, p0 + r) * sn + c;
        *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(Os + r * GLD + c);
        *reinterpret_cast<uint4*>(dst + 8) = *reinterpret_cast<const uint4*>(Os + r * GLD + c + 8);
    }
}

__global__ __launch_bounds__(FT) void k_fs_out(const FsOutArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Gt = reinterpret_cast<u16*>(smem_raw);                  // [8][64 d2][72]
    float* part = reinterpret_cast<float*>(smem_raw + FS_GT_BYTES);   // [4][64] partials of n
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int it = blockIdx.x, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int S = a.S, M = a.M;
    const u16* qb = (const u16*)a.q.ptr + b * a.q.sb + h * a.q.sh;
    u16* ob = (u16*)a.o.ptr + b * a.o.sb + h * a.o.sh;
    const u16* state_bh = a.state + (long)bh * a.njg * FE * IT;

    mix_tile_to_lds<0>(Gt, state_bh, a.njg, a.W, a.ldw, M, it * IT, tid);
    __syncthreads();

    for (int ii = 0; ii < IT; ++ii) {
        const int i = it * IT + ii;
        if (i >= M) break;
        u16* Gb = Gt + ii * FD * GLD;
        for (int c0 = 0; c0 < S; c0 += 64) {
            const long p0 = (long)i * S + c0;
            const int rv = min(64, S - c0);
            if (a.normalize)
                part[tid] = wz_partial(a.W + (long)i * a.ldw, 1, a.z + (long)bh * M * S, M, S, c0 + (tid & 63), (tid & 63) < rv, tid >> 6);
            bf16x8 av[2];
            const int rvw = rv - wave * 16;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                av[ks] = a.relu ? load_a_rows<true>(qb, a.q.sn, a.idx, p0 + wave * 16, rvw, ks * 32, a.eps, lane)
                                : load_a_rows<false>(qb, a.q.sn, a.idx, p0 + wave * 16, rvw, ks * 32, a.eps, lane);
            f32x4 acc[4];
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) acc[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
            rows_times_gt(acc, av, Gb, lane);
            __syncthreads();   // partials of n ready; on the last chunk every wave is done reading Gb
            const bool last = c0 + 64 >= S;
            // the last chunk stages into this block's (now dead) Gt slot; earlier chunks into the previous block's slot
            u16* Os = last ? Gb : (ii > 0 ? Gb - FD * GLD : nullptr);
            const int kg = lane >> 4;
            if (a.normalize) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float ni = 1.f / (a.eps + wz_sum(part, wave * 16 + kg * 4 + r));
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) acc[tn][r] *= ni;
                }
            }
            if (Os) {
                stage_c_tile(Os, acc, wave, lane);
                __syncthreads();
                store_rows(ob, a.o.sn, a.idx, p0, rv, Os, tid);
            } else {   // multi-chunk first block: no free slot yet -> direct (narrow) stores
                const int n = lane & 15;
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = wave * 16 + kg * 4 + r;
                        if (row < rv) ob[tok_row(a.idx, p0 + row) * a.o.sn + tn * 16 + n] = f32_to_bf16(acc[tn][r]);
                    }
            }
            __syncthreads();
        }
    }
}

// -------------------------------------------------------------------------------------------------
// k_fs_dw: dWp[bh][q][i][j] = sum_{e' in quarter q} dG[i][e'] KV[j][e']   (both in the interleaved layout)
// LDS images [e'][64 blocks] built from 16-byte pieces; both MFMA operands via transpose reads.
// grid (4 quarters, bh).  The <dn_i, z_j> term is added by quarter 0 with VALU.
// -------------------------------------------------------------------------------------------------
struct FsDwArgs {
    const u16* dg;
    const u16* kv;
    const float* dn;   // [bh][M][S] or null
    const float* z;
    float* dwp;        // [bh][4][64][64]
    int M, S, njg;
};
constexpr int DW_EC = 256;                       // e' rows per LDS image
constexpr int DW_LDI = 72;
constexpr int FS_DW_SMEM = 2 * DW_EC * DW_LDI * 2;

__global__ __launch_bounds__(FT) void k_fs_dw(const FsDwArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Ai = reinterpret_cast<u16*>(smem_raw);   // dG image [256 e'][72]
    u16* Bi = Ai + DW_EC * DW_LDI;                // KV image
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, kg = lane >> 4;
    const int qtr = blockIdx.x, bh = blockIdx.y, M = a.M, njg = a.njg;
    const u16* dg = a.dg + (long)bh * njg * FE * IT;
    const u16* kv = a.kv + (long)bh * njg * FE * IT;
    f32x4 acc[4];
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) acc[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ec = 0; ec < FE / 4; ec += DW_EC) {
        const long e0 = (long)qtr * (FE / 4) + ec;
        // 2 images x 256 rows x 8 groups of 16 bytes
        for (int v = tid; v < 2 * DW_EC * 8; v += FT) {
            const int which = v / (DW_EC * 8), rem = v - which * DW_EC * 8, g = rem / DW_EC, r = rem - g * DW_EC;
            uint4 x = make_uint4(0, 0, 0, 0);
            if (g < njg) x = *reinterpret_cast<const uint4*>((which ? kv : dg) + ((long)g * FE + e0 + r) * IT);
            *reinterpret_cast<uint4*>((which ? Bi : Ai) + r * DW_LDI + g * 8) = x;
        }
        __syncthreads();
        for (int k0 = 0; k0 < DW_EC; k0 += 32) {
            const bf16x8 av = tr_read8(Ai, DW_LDI, k0, wave * 16, lane);
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) acc[tn] = mfma_bf16(av, tr_read8(Bi, DW_LDI, k0, tn * 16, lane), acc[tn]);
        }
        __syncthreads();
    }
    float* out = a.dwp + ((long)bh * 4 + qtr) * 64 * 64;
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = wave * 16 + kg * 4 + r, j = tn * 16 + n;
            float v = acc[tn][r];
            if (qtr == 0 && a.dn && i < M && j < M) {
                const float* dr = a.dn + ((long)bh * M + i) * a.S;
                const float* zr = a.z + ((long)bh * M + j) * a.S;
                for (int s = 0; s < a.S; ++s) v += dr[s] * zr[s];
            }
            out[i * 64 + j] = v;
        }
}

// dW[i][j] = sum over (bh, quarter) of dWp  (fixed order: deterministic)
__global__ void k_fs_dw_reduce(const float* __restrict__ dwp, float* __restrict__ dW, int M, int nparts) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= M * M) return;
    const int i = e / M, j = e - i * M;
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += dwp[(long)p * 4096 + i * 64 + j];
    dW[(long)i * M + j] = s;
}

// -------------------------------------------------------------------------------------------------
// k_fs_bwd_tok: per (tile jg, bh):
//   phase 1: Gt = mix(W, KV)        -> dQ_j = (dO_j G_j^T) / n_j + dz_j (x) ksum_j     (relu mask)
//   phase 2: Gt = mix(W^T, dG)      -> dK_j = V_j dKV_j^T + 1 dksum_j^T (relu mask) ; dV_j = K_j dKV_j
// -------------------------------------------------------------------------------------------------
struct FsTokArgs {
    View q, k, v, dout;
    MView dq, dk, dv;
    const int* idx;
    const float* W;
    int ldw;
    const u16* state;   // KV^T
    const u16* dstate;  // dG^T
    const float* z;
    const float* dn;
    const float* ksum;
    int H, M, S, njg;
    float eps;
    int relu, normalize;
};
constexpr int FS_TOK_SMEM = FS_GT_BYTES + (256 + 256 + 64 + 64 + 256) * 4;

__global__ __launch_bounds__(FT) void k_fs_bwd_tok(const FsTokArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Gt = reinterpret_cast<u16*>(smem_raw);
    float* pn = reinterpret_cast<float*>(smem_raw + FS_GT_BYTES);     // [4][64] partials of n
    float* pz = pn + 256;                                             // [4][64] partials of dz
    float* ksum_s = pz + 256;                                         // [64]
    float* dks = ksum_s + 64;                                         // [64]
    float* part = dks + 64;                                           // [4 waves][64] dksum partials
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, kg = lane >> 4;
    part[tid] = 0.f;
    const int jgx = blockIdx.x, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int S = a.S, M = a.M;
    auto base = [&](const View& w) { return (const u16*)w.ptr + b * w.sb + h * w.sh; };
    auto mbase = [&](const MView& w) { return (u16*)w.ptr + b * w.sb + h * w.sh; };
    const u16 *qb = base(a.q), *kb = base(a.k), *vb = base(a.v), *gb = base(a.dout);
    u16 *dqb = mbase(a.dq), *dkb = mbase(a.dk), *dvb = mbase(a.dv);
    const long sofs = (long)bh * a.njg * FE * IT;
    const float* z_bh = a.z + (long)bh * M * S;
    const float* dn_bh = a.dn + (long)bh * M * S;

    // ---------------- phase 1: dQ ----------------
    mix_tile_to_lds<0>(Gt, a.state + sofs, a.njg, a.W, a.ldw, M, jgx * IT, tid);
    __syncthreads();
    // dksum per block is accumulated here and kept for phase 2 in registers of threads < 64 (8 blocks)
    float dks_keep[IT];
#pragma unroll
    for (int jj = 0; jj < IT; ++jj) dks_keep[jj] = 0.f;
#pragma unroll
    for (int jj = 0; jj < IT; ++jj) {
        const int j = jgx * IT + jj;
        if (j >= M) continue;
        u16* Gb = Gt + jj * FD * GLD;
        if (tid < 64) ksum_s[tid] = a.normalize ? a.ksum[((long)bh * M + j) * 64 + tid] : 0.f;
        for (int c0 = 0; c0 < S; c0 += 64) {
            const long p0 = (long)j * S + c0;
            const int rv = min(64, S - c0), rvw = rv - wave * 16;
            if (a.normalize) {
                const bool ok = (tid & 63) < rv;
                pn[tid] = wz_partial(a.W + (long)j * a.ldw, 1, z_bh, M, S, c0 + (tid & 63), ok, tid >> 6);
                pz[tid] = wz_partial(a.W + j, a.ldw, dn_bh, M, S, c0 + (tid & 63), ok, tid >> 6);
            }
            bf16x8 av[2], qv[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                av[ks] = load_a_rows<false>(gb, a.dout.sn, a.idx, p0 + wave * 16, rvw, ks * 32, 0.f, lane);
                qv[ks] = a.relu ? load_a_rows<true>(qb, a.q.sn, a.idx, p0 + wave * 16, rvw, ks * 32, a.eps, lane)
                                : load_a_rows<false>(qb, a.q.sn, a.idx, p0 + wave * 16, rvw, ks * 32, a.eps, lane);
            }
            f32x4 acc[4];
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) acc[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
            rows_times_gt_t(acc, av, Gb, lane);      // (dO G^T)[s][d1] : B[k = d2][n = d1] = Gt[d2][d1]
            __syncthreads();                         // partials ready; all waves done with Gb on the last chunk
            // dksum[d] += sum_s dz[s] q[s][d] : lane holds q[row = 16 wave + n][cols ks*32 + 8 kg ..]
            if (a.normalize) {
                const float dzr = wz_sum(pz, wave * 16 + n);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const s16x8 qs = __builtin_bit_cast(s16x8, qv[ks]);
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        float v = dzr * bf((u16)qs[t]);
                        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64);
                        v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                        if (n == 0) part[wave * 64 + ks * 32 + kg * 8 + t] += v;
                    }
                }
            }
            const bool last = c0 + 64 >= S;
            u16* Os = last ? Gb : (jj > 0 ? Gb - FD * GLD : nullptr);
            // epilogue in C layout: row s = 16 wave + 4 kg + r, col d1 = 16 tn + n
            if (a.normalize) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wave * 16 + kg * 4 + r;
                    const float ni = 1.f / (a.eps + wz_sum(pn, row)), dzr = wz_sum(pz, row);
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) acc[tn][r] = acc[tn][r] * ni + dzr * ksum_s[tn * 16 + n];
                }
            }
            if (a.relu) {   // mask by q > 0: re-read q in C layout from global (L1/L2 hot)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = wave * 16 + kg * 4 + r;
                        if (row < rv && !(bf(qb[tok_row(a.idx, p0 + row) * a.q.sn + tn * 16 + n]) > 0.f)) acc[tn][r] = 0.f;
                    }
            }
            if (Os) {
                stage_c_tile(Os, acc, wave, lane);
                __syncthreads();
                store_rows(dqb, a.dq.sn, a.idx, p0, rv, Os, tid);
            } else {
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = wave * 16 + kg * 4 + r;
                        if (row < rv) dqb[tok_row(a.idx, p0 + row) * a.dq.sn + tn * 16 + n] = f32_to_bf16(acc[tn][r]);
                    }
            }
            __syncthreads();
        }
        if (a.normalize) {
            if (tid < 64) {
                dks_keep[jj] = part[tid] + part[64 + tid] + part[128 + tid] + part[192 + tid];
            }
            __syncthreads();
        }
        if (tid < 256) part[tid] = 0.f;
        __syncthreads();
    }

    // ---------------- phase 2: dK, dV ----------------
    __syncthreads();
    mix_tile_to_lds<1>(Gt, a.dstate + sofs, a.njg, a.W, a.ldw, M, jgx * IT, tid);
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < IT; ++jj) {
        const int j = jgx * IT + jj;
        if (j >= M) continue;
        u16* Gb = Gt + jj * FD * GLD;
        if (tid < 64) dks[tid] = dks_keep[jj];
        for (int c0 = 0; c0 < S; c0 += 64) {
            const long p0 = (long)j * S + c0;
            const int rv = min(64, S - c0), rvw = rv - wave * 16;
            bf16x8 vv[2], kv[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                vv[ks] = load_a_rows<false>(vb, a.v.sn, a.idx, p0 + wave * 16, rvw, ks * 32, 0.f, lane);
                kv[ks] = a.relu ? load_a_rows<true>(kb, a.k.sn, a.idx, p0 + wave * 16, rvw, ks * 32, a.eps, lane)
                                : load_a_rows<false>(kb, a.k.sn, a.idx, p0 + wave * 16, rvw, ks * 32, a.eps, lane);
            }
            f32x4 accK[4], accV[4];
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) accK[tn] = accV[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
            rows_times_gt_t(accK, vv, Gb, lane);   // dK[s][d1] = sum_d2 V[s][d2] dKVt[d2][d1]
            rows_times_gt(accV, kv, Gb, lane);     // dV[s][d2] = sum_d1 K[s][d1] dKVt[d2][d1]
            __syncthreads();
            const bool last = c0 + 64 >= S;
            u16* Os = last ? Gb : (jj > 0 ? Gb - FD * GLD : nullptr);
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wave * 16 + kg * 4 + r;
                    accK[tn][r] += dks[tn * 16 + n];
                    if (a.relu && row < rv && !(bf(kb[tok_row(a.idx, p0 + row) * a.k.sn + tn * 16 + n]) > 0.f)) accK[tn][r] = 0.f;
                }
            if (Os) {
                stage_c_tile(Os, accK, wave, lane);
                __syncthreads();
                store_rows(dkb, a.dk.sn, a.idx, p0, rv, Os, tid);
                __syncthreads();
                stage_c_tile(Os, accV, wave, lane);
                __syncthreads();
                store_rows(dvb, a.dv.sn, a.idx, p0, rv, Os, tid);
            } else {
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = wave * 16 + kg * 4 + r;
                        if (row < rv) {
                            dkb[tok_row(a.idx, p0 + row) * a.dk.sn + tn * 16 + n] = f32_to_bf16(accK[tn][r]);
                            dvb[tok_row(a.idx, p0 + row) * a.dv.sn + tn * 16 + n] = f32_to_bf16(accV[tn][r]);
                        }
                    }
            }
            __syncthreads();
        }
    }
}

}  // namespace fast
}  // namespace mhla
