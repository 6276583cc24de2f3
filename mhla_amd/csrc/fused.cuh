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
//   backward: k_fs_state<1> (dG^T, dn) -> k_fs_dw (dW partials) -> k_fs_bwd_dq (mix G; dQ, dksum) -> k_fs_bwd_dkv (mix dKV; dK, dV)
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
        w[i] = pack_bf16x2(lo, hi);
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
// k_fs_wz: the two small [M x M] x [M x S] products per (b,h), done once instead of per consumer wave:
//   MODE 0: ninv[i][s] = 1 / (eps + sum_j W[i][j] z[j][s])         (normaliser, mhla.py:266)
//   MODE 1: dz[j][s]   = sum_i W[i][j] dn[i][s]
// grid (ceil(S / 64), bh), M <= 64.
// -------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(FT) void k_fs_wz(const float* __restrict__ W, int ldw, const float* __restrict__ x,
                                              float* __restrict__ out, int M, int S, float eps) {
    __shared__ float Ws[64 * 65];
    __shared__ __attribute__((aligned(16))) float xs[64 * 64];
    const int tid = threadIdx.x, c0 = blockIdx.x * 64, bh = blockIdx.y, rv = min(64, S - c0);
    // all global loads first (16 + 16 per thread in flight), LDS writes after
    float wreg[16], xreg[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int v = tid + t * FT, r = v >> 6, c = v & 63;
        wreg[t] = (r < M && c < M) ? (MODE ? W[(long)c * ldw + r] : W[(long)r * ldw + c]) : 0.f;
        xreg[t] = (r < M && c < rv) ? x[((long)bh * M + r) * S + c0 + c] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int v = tid + t * FT, r = v >> 6, c = v & 63;
        Ws[r * 65 + c] = wreg[t];
        xs[r * 64 + c] = xreg[t];
    }
    __syncthreads();
    // thread -> 4 rows (r0, r0 + 16, ...) x 4 columns (4 sq ..): W element reused for 4 columns, x read as float4
    const int sq = tid & 15, r0 = tid >> 4;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int c = 0; c < 64; ++c) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + c * 64 + sq * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += Ws[(r0 + 16 * i) * 65 + c] * xv;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + 16 * i;
        if (r < M) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int sc = sq * 4 + t;
                if (sc < rv) out[((long)bh * M + r) * S + c0 + sc] = MODE ? acc[i][t] : 1.f / (eps + acc[i][t]);
            }
        }
    }
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
    const float* ninv;   // MODE 1: [bh][M][S]  1 / n  (k_fs_wz<0>)
    u16* state;          // [bh][njg][4096][8]
    float* ksum;         // MODE 0: [bh][M][64]
    float* z_out;        // MODE 0: [bh][M][S]
    float* dn;           // MODE 1: [bh][M][S]
    int H, M, S;
    float eps;
    int relu, normalize;
};
constexpr int FS_STATE_SMEM = 3 * 64 * TLD * 2 + (4 * 64 + 64) * 4;

struct TileRegs { uint4 x0, x1, y0, y1, t0, t1; };

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
    const int srow = tid >> 2, scol = (tid & 3) * 16;        // staging: thread -> (row, 16 columns)

    // global -> registers for one 64-row chunk (rows >= rv give zeros); registers -> LDS separately so that
    // the next chunk's loads are in flight while the current one is being multiplied
    auto issue = [&](long p, int rv, bool with_t, TileRegs& R) {
        const uint4 zero = make_uint4(0, 0, 0, 0);
        R.x0 = R.x1 = R.y0 = R.y1 = R.t0 = R.t1 = zero;
        if (srow < rv) {
            const long tr = tok_row(a.idx, p + srow);
            const u16* px = xb + tr * a.x.sn + scol;
            const u16* py = yb + tr * a.y.sn + scol;
            R.x0 = *reinterpret_cast<const uint4*>(px);
            R.x1 = *reinterpret_cast<const uint4*>(px + 8);
            R.y0 = *reinterpret_cast<const uint4*>(py);
            R.y1 = *reinterpret_cast<const uint4*>(py + 8);
            if (with_t) {
                const u16* pt = tb + tr * a.t.sn + scol;
                R.t0 = *reinterpret_cast<const uint4*>(pt);
                R.t1 = *reinterpret_cast<const uint4*>(pt + 8);
            }
        }
    };
    auto commit = [&](const TileRegs& R, int rv, int rfill, bool with_t) {
        if (srow < rfill) {
            uint4 x0 = R.x0, x1 = R.x1, t0 = R.t0, t1 = R.t1;
            if (a.relu && srow < rv) {
                x0 = relu_eps8(x0, a.eps); x1 = relu_eps8(x1, a.eps);
                if (MODE == 0 && with_t) { t0 = relu_eps8(t0, a.eps); t1 = relu_eps8(t1, a.eps); }
            }
            *reinterpret_cast<uint4*>(Xs + srow * TLD + scol) = x0;
            *reinterpret_cast<uint4*>(Xs + srow * TLD + scol + 8) = x1;
            *reinterpret_cast<uint4*>(Ys + srow * TLD + scol) = R.y0;
            *reinterpret_cast<uint4*>(Ys + srow * TLD + scol + 8) = R.y1;
            if (with_t) {
                *reinterpret_cast<uint4*>(Ts + srow * TLD + scol) = t0;
                *reinterpret_cast<uint4*>(Ts + srow * TLD + scol + 8) = t1;
            }
        }
    };

    f32x4 acc[IT][4];
#pragma unroll
    for (int jj = 0; jj < IT; ++jj)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[jj][tn] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool tile_t = a.normalize && (MODE == 1 || single);   // third tile staged with the chunk
    TileRegs R;
    if (jg * IT < M) issue((long)jg * IT * S, min(64, S), tile_t, R);

#pragma unroll
    for (int jj = 0; jj < IT; ++jj) {
        const int j = jg * IT + jj;
        if (j >= M) continue;
        const long p0 = (long)j * S;
        float ks = 0.f;
        for (int c0 = 0; c0 < S; c0 += 64) {
            const int rv = min(64, S - c0), rfill = (rv + 31) & ~31;
            commit(R, rv, rfill, tile_t);
            {   // prefetch the next chunk (same block, or the next block of the group)
                long pn = -1;
                int rvn = 0;
                if (c0 + 64 < S) { pn = p0 + c0 + 64; rvn = min(64, S - c0 - 64); }
                else if (jj + 1 < IT && j + 1 < M) { pn = (long)(j + 1) * S; rvn = min(64, S); }
                if (pn >= 0) issue(pn, rvn, tile_t, R);
            }
            __syncthreads();
            if (MODE == 0 && a.normalize) {   // column sums of K
                const int col = tid & 63, pr = tid >> 6;
                for (int r = pr * 16; r < min(rv, pr * 16 + 16); ++r) ks += bf(Xs[r * TLD + col]);
            }
            if (MODE == 1 && a.normalize) {   // dn and dP = dO / n (rounded to bf16)
                const int r = srow, cq = scol;
                float d = 0.f;
                if (r < rv) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) d += bf(Ys[r * TLD + cq + c]) * bf(Ts[r * TLD + cq + c]);
                }
                d += __shfl_xor(d, 1, 64);
                d += __shfl_xor(d, 2, 64);
                if (r < rv) {
                    const float ni = a.ninv[((long)bh * M + j) * S + c0 + r];
                    if ((tid & 3) == 0) a.dn[((long)bh * M + j) * S + c0 + r] = -d * ni;
#pragma unroll
                    for (int c = 0; c < 16; ++c) Ys[r * TLD + cq + c] = cvt_bf16(bf(Ys[r * TLD + cq + c]) * ni);
                }
                __syncthreads();
            }
            for (int k0 = 0; k0 < rfill; k0 += 32) {
                const bf16x8 av = tr_read8(Ys, TLD, k0, wave * 16, lane);
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) acc[jj][tn] = mfma_bf16(av, tr_read8(Xs, TLD, k0, tn * 16, lane), acc[jj][tn]);
            }
            if (!(MODE == 0 && a.normalize && single)) __syncthreads();   // (single: the z pass below syncs)
        }
        if (MODE == 0 && a.normalize) {
            part[(tid >> 6) * 64 + (tid & 63)] = ks;
            __syncthreads();
            if (tid < 64) {
                const float sacc = part[tid] + part[64 + tid] + part[128 + tid] + part[192 + tid];
                ksum_s[tid] = sacc;
                a.ksum[((long)bh * M + j) * 64 + tid] = sacc;
            }
            __syncthreads();
            for (int c0 = 0; c0 < S; c0 += 64) {
                const int rv = min(64, S - c0);
                if (!single) {   // (rare) multi-chunk blocks: second pass over Q, synchronous
                    if (a.relu) stage_rows<true>(Ts, tb, a.t.sn, a.idx, p0 + c0, rv, rv, a.eps, tid);
                    else        stage_rows<false>(Ts, tb, a.t.sn, a.idx, p0 + c0, rv, rv, 0.f, tid);
                    __syncthreads();
                }
                float d = 0.f;
                if (srow < rv) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) d += bf(Ts[srow * TLD + scol + c]) * ksum_s[scol + c];
                }
                d += __shfl_xor(d, 1, 64);
                d += __shfl_xor(d, 2, 64);
                if (srow < rv && (tid & 3) == 0) a.z_out[((long)bh * M + j) * S + c0 + srow] = d;
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
                w[p] = pack_bf16x2(acc[2 * p][tn][r], acc[2 * p + 1][tn][r]);
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
            const u16 h = cvt_bf16(w);
            hi[t] = (short)h;
            lo[t] = (short)cvt_bf16(w - bf(h));
        }
        bhi[ks] = __builtin_bit_cast(bf16x8, hi);
        blo[ks] = __builtin_bit_cast(bf16x8, lo);
    }
    const bool two = njg > 4;
    constexpr int UN = 4;   // tiles per batch; two batches in flight (software double buffer)
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    auto load_batch = [&](uint4 (&av)[UN][2], int et0) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long e = (long)(et0 + u) * 16 + n;
            av[u][0] = (kg < njg) ? *reinterpret_cast<const uint4*>(state_bh + ((long)kg * FE + e) * IT) : zero4;
            av[u][1] = (two && 4 + kg < njg) ? *reinterpret_cast<const uint4*>(state_bh + ((long)(4 + kg) * FE + e) * IT) : zero4;
        }
    };
    auto do_batch = [&](const uint4 (&av)[UN][2], int et0) {
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
                pk.x = pack_bf16x2(c[0], c[1]);
                pk.y = pack_bf16x2(c[2], c[3]);
                *reinterpret_cast<uint2*>(Gt + ((long)(n * FD + d2)) * GLD + d1) = pk;
            }
        }
    };
    // wave w owns tiles et = 4 UN (w + 4 b) .. : 64 tiles per wave in 16 batches of UN
    constexpr int NB = FE / 16 / 4 / UN;
    uint4 bufA[UN][2], bufB[UN][2];
    load_batch(bufA, wave * UN);
#pragma unroll 1
    for (int bt = 0; bt < NB; bt += 2) {
        load_batch(bufB, (wave + 4 * (bt + 1)) * UN);
        do_batch(bufA, (wave + 4 * bt) * UN);
        if (bt + 2 < NB) load_batch(bufA, (wave + 4 * (bt + 2)) * UN);
        do_batch(bufB, (wave + 4 * (bt + 1)) * UN);
    }
}

// XCD-aware logical work index (bijective): workgroups that share a (b,h)'s summaries land on one XCD's L2.
__device__ __forceinline__ int xcd_swizzle(int wg, int nwg) {
    const int xcd = wg & 7, slot = wg >> 3, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// A operands of a 64-row chunk straight from a token view: a[st][ks] = rows 16 st + (lane & 15),
// columns 32 ks + 8 (lane >> 4) .. + 7.  Rows >= rv give zeros.
template <bool RELU>
__device__ __forceinline__ void load_a64(bf16x8 (&a)[4][2], const u16* __restrict__ base, long sn,
                                         const int* __restrict__ idx, long p0, int rv, float eps, int lane) {
    const int m = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const int row = st * 16 + m;
        const u16* src = base + (row < rv ? tok_row(idx, p0 + row) : 0) * sn + kg * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < rv) {
                v = *reinterpret_cast<const uint4*>(src + ks * 32);
                if (RELU) v = relu_eps8(v, eps);
            }
            a[st][ks] = __builtin_bit_cast(bf16x8, v);
        }
    }
}

// acc[st][tn] += A[st] x B  with B from one mixed summary Gb[d2][d1] (GLD stride):
//   TRB false: B[k = d1][n = d2] = Gb[n][k]  (k contiguous: plain 16-byte LDS reads)
//   TRB true : B[k = d2][n = d1] = Gb[k][n]  (hardware transpose reads)
template <bool TRB>
__device__ __forceinline__ void chunk_times_gt(f32x4 (&acc)[4][4], const bf16x8 (&a)[4][2], const u16* __restrict__ Gb, int lane) {
    const int n = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 bv = TRB ? tr_read8(Gb, GLD, ks * 32, tn * 16, lane)
                                  : *reinterpret_cast<const bf16x8*>(Gb + (tn * 16 + n) * GLD + ks * 32 + kg * 8);
#pragma unroll
            for (int st = 0; st < 4; ++st) acc[st][tn] = mfma_bf16(a[st][ks], bv, acc[st][tn]);
        }
    }
}

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// Wave-private staging of a 64 x 64 fp32 result (C layout: row = 16 st + 4 (lane >> 4) + r, col = 16 tn + (lane & 15))
__device__ __forceinline__ void stage64(u16* __restrict__ Os, const f32x4 (&acc)[4][4], int lane) {
    const int n = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r) Os[(st * 16 + kg * 4 + r) * GLD + tn * 16 + n] = cvt_bf16(acc[st][tn][r]);
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
template <bool MASK>
__device__ __forceinline__ void store64(u16* __restrict__ base, long sn, const int* __restrict__ idx, long p0, int rv,
                                        const u16* __restrict__ Os, const u16* __restrict__ mbase, long msn, int lane) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int row = p * 8 + (lane >> 3), c = (lane & 7) * 8;
        if (row < rv) {
            const long tr = tok_row(idx, p0 + row);
            uint4 v = *reinterpret_cast<const uint4*>(Os + row * GLD + c);
            if (MASK) v = mask_pos8(v, *reinterpret_cast<const uint4*>(mbase + tr * msn + c));
            *reinterpret_cast<uint4*>(base + tr * sn + c) = v;
        }
    }
}
// narrow fallback (no free staging slot): direct stores from the C layout
template <bool MASK>
__device__ __forceinline__ void store64_direct(u16* __restrict__ base, long sn, const int* __restrict__ idx, long p0, int rv,
                                               const f32x4 (&acc)[4][4], const u16* __restrict__ mbase, long msn, int lane) {
    const int n = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = st * 16 + kg * 4 + r;
            if (row < rv) {
                const long tr = tok_row(idx, p0 + row);
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) {
                    float v = acc[st][tn][r];
                    if (MASK && !(bf(mbase[tr * msn + tn * 16 + n]) > 0.f)) v = 0.f;
                    base[tr * sn + tn * 16 + n] = cvt_bf16(v);
                }
            }
        }
}

// lane = row of the chunk: sum_j w[j * wstride] * x[j * S + s]
__device__ __forceinline__ float col_dot(const float* __restrict__ w, long wstride, const float* __restrict__ x, int M, int S,
                                         int s, bool valid) {
    float acc = 0.f;
    if (valid) {
#pragma unroll 8
        for (int j = 0; j < M; ++j) acc += w[(long)j * wstride] * x[(long)j * S + s];
    }
    return acc;
}

// -------------------------------------------------------------------------------------------------
// k_fs_out: per (tile it, bh): all 4 waves mix the 8 summaries of the tile into LDS; then each wave owns
// 2 blocks: O_i = (Q_i G_i) / n_i, staged in the block's own (dead) Gt slot -- no block-level barriers.
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
};
constexpr int FS_GT_BYTES = IT * FD * GLD * 2;
constexpr int FS_OUT_SMEM = FS_GT_BYTES;

__global__ __launch_bounds__(FT, 2) void k_fs_out(const FsOutArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Gt = reinterpret_cast<u16*>(smem_raw);   // [8][64 d2][72]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kg = lane >> 4;
    const int L = xcd_swizzle(blockIdx.x, gridDim.x);
    const int bh = L / a.njg, it = L - bh * a.njg, b = bh / a.H, h = bh - b * a.H;
    const int S = a.S, M = a.M;
    const u16* qb = (const u16*)a.q.ptr + b * a.q.sb + h * a.q.sh;
    u16* ob = (u16*)a.o.ptr + b * a.o.sb + h * a.o.sh;
    const float* ninv_bh = a.ninv + (long)bh * M * S;

    const u16* state_bh = a.state + (long)bh * a.njg * FE * IT;
    auto load_blk = [&](bf16x8 (&av)[4][2], float& ninv, int i, int c0, int rv) {
        const long p0 = (long)i * S + c0;
        if (a.relu) load_a64<true>(av, qb, a.q.sn, a.idx, p0, rv, a.eps, lane);
        else        load_a64<false>(av, qb, a.q.sn, a.idx, p0, rv, a.eps, lane);
        ninv = (a.normalize && lane < rv) ? ninv_bh[(long)i * S + c0 + lane] : 1.f;
    };
    auto compute_store = [&](const bf16x8 (&av)[4][2], float ninv, int bi, int i, int c0, int rv) {
        const long p0 = (long)i * S + c0;
        u16* Gb = Gt + bi * FD * GLD;
        f32x4 acc[4][4];
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) acc[st][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
        chunk_times_gt<false>(acc, av, Gb, lane);
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ni = __shfl(ninv, st * 16 + kg * 4 + r, 64);
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) acc[st][tn][r] *= ni;
            }
        if (c0 + 64 >= S) {   // last chunk of the block: its Gt slot is dead for this wave -> staging buffer
            wave_lds_fence();
            stage64(Gb, acc, lane);
            wave_lds_fence();
            store64<false>(ob, a.o.sn, a.idx, p0, rv, Gb, nullptr, 0, lane);
        } else {
            store64_direct<false>(ob, a.o.sn, a.idx, p0, rv, acc, nullptr, 0, lane);
        }
    };

    if (S <= 64) {
        // each wave owns blocks (wave, wave + 4): the first block's operands are fetched before the mixing,
        // the second block's while the first is being multiplied
        const int iA = it * IT + wave, iB = iA + 4;
        bf16x8 avA[4][2], avB[4][2];
        float ninvA = 1.f, ninvB = 1.f;
        if (iA < M) load_blk(avA, ninvA, iA, 0, S);
        mix_tile_to_lds<0>(Gt, state_bh, a.njg, a.W, a.ldw, M, it * IT, tid);
        __syncthreads();
        if (iB < M) load_blk(avB, ninvB, iB, 0, S);
        if (iA < M) compute_store(avA, ninvA, wave, iA, 0, S);
        if (iB < M) compute_store(avB, ninvB, wave + 4, iB, 0, S);
        return;
    }

    mix_tile_to_lds<0>(Gt, state_bh, a.njg, a.W, a.ldw, M, it * IT, tid);
    __syncthreads();
    for (int bi = wave; bi < IT; bi += 4) {
        const int i = it * IT + bi;
        if (i >= M) continue;
        for (int c0 = 0; c0 < S; c0 += 64) {
            const int rv = min(64, S - c0);
            bf16x8 av[4][2];
            float ninv;
            load_blk(av, ninv, i, c0, rv);
            compute_store(av, ninv, bi, i, c0, rv);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// k_fs_dw: dWp[bh][q][i][j] = sum_{e' in quarter q} dG[i][e'] KV[j][e']   (both in the interleaved layout)
// LDS images [e'][64 blocks] built from 16-byte pieces; both MFMA operands via transpose reads.
// grid (DW_SPLIT, bh).  The <dn_i, z_j> term is added by split 0 from LDS tiles.
// -------------------------------------------------------------------------------------------------
struct FsDwArgs {
    const u16* dg;
    const u16* kv;
    const float* dn;   // [bh][M][S] or null
    const float* z;
    float* dwp;        // [bh][DW_SPLIT][64][64]
    int M, S, njg;
};
constexpr int DW_EC = 128;                       // e' rows per LDS image
constexpr int DW_SPLIT = 8;                      // e' splits per (b,h)
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
    constexpr int NP = 2 * DW_EC * 8 / FT;   // 16-byte pieces per thread per chunk
    uint4 pr[NP];
    auto issue = [&](long e0) {
#pragma unroll
        for (int t = 0; t < NP; ++t) {
            const int v = tid + t * FT;
            const int which = v / (DW_EC * 8), rem = v - which * DW_EC * 8, g = rem / DW_EC, r = rem - g * DW_EC;
            pr[t] = (g < njg) ? *reinterpret_cast<const uint4*>((which ? kv : dg) + ((long)g * FE + e0 + r) * IT) : make_uint4(0, 0, 0, 0);
        }
    };
    const long ebase = (long)qtr * (FE / DW_SPLIT);
    issue(ebase);
    for (int ec = 0; ec < FE / DW_SPLIT; ec += DW_EC) {
#pragma unroll
        for (int t = 0; t < NP; ++t) {
            const int v = tid + t * FT;
            const int which = v / (DW_EC * 8), rem = v - which * DW_EC * 8, g = rem / DW_EC, r = rem - g * DW_EC;
            *reinterpret_cast<uint4*>((which ? Bi : Ai) + r * DW_LDI + g * 8) = pr[t];
        }
        if (ec + DW_EC < FE / DW_SPLIT) issue(ebase + ec + DW_EC);   // next chunk in flight during the MFMAs
        __syncthreads();
        for (int k0 = 0; k0 < DW_EC; k0 += 32) {
            const bf16x8 av = tr_read8(Ai, DW_LDI, k0, wave * 16, lane);
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) acc[tn] = mfma_bf16(av, tr_read8(Bi, DW_LDI, k0, tn * 16, lane), acc[tn]);
        }
        __syncthreads();
    }
    // <dn_i, z_j> term (quarter 0): stage dn[64][S<=64 chunk] and z[64][chunk] in LDS as fp32
    if (qtr == 0 && a.dn) {
        float* dns = reinterpret_cast<float*>(smem_raw);     // [64][65]
        float* zs = dns + 64 * 65;                           // [64][65]   (2 x 16.6 KB <= the image buffers)
        for (int c0 = 0; c0 < a.S; c0 += 64) {
            const int rv = min(64, a.S - c0);
            __syncthreads();
            float zr[32];
#pragma unroll
            for (int t = 0; t < 32; ++t) {
                const int v = tid + t * FT, which = v >> 12, row = (v >> 6) & 63, col = v & 63;
                zr[t] = (row < M && col < rv) ? (which ? a.z : a.dn)[((long)bh * M + row) * a.S + c0 + col] : 0.f;
            }
#pragma unroll
            for (int t = 0; t < 32; ++t) {
                const int v = tid + t * FT, which = v >> 12, row = (v >> 6) & 63, col = v & 63;
                (which ? zs : dns)[row * 65 + col] = zr[t];
            }
            __syncthreads();
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = wave * 16 + kg * 4 + r, j = tn * 16 + n;
                    float v = 0.f;
#pragma unroll 4
                    for (int c = 0; c < 64; ++c) v += dns[i * 65 + c] * zs[j * 65 + c];
                    acc[tn][r] += v;
                }
        }
    }
    float* out = a.dwp + ((long)bh * DW_SPLIT + qtr) * 64 * 64;
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(wave * 16 + kg * 4 + r) * 64 + tn * 16 + n] = acc[tn][r];
}

// Deterministic two-stage reduction of the partials: stage 1 sums groups of DWR_G partials (grid (16, ngroups)),
// stage 2 (MODE 1) sums the group results into dW[M][M].
constexpr int DWR_G = 64;
__global__ void k_fs_dw_reduce1(const float* __restrict__ dwp, float* __restrict__ tmp, int nparts) {
    const int e = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
    float s = 0.f;
    const int p1 = min(nparts, (g + 1) * DWR_G);
#pragma unroll 16
    for (int p = g * DWR_G; p < p1; ++p) s += dwp[(long)p * 4096 + e];
    tmp[(long)g * 4096 + e] = s;
}
__global__ void k_fs_dw_reduce2(const float* __restrict__ tmp, float* __restrict__ dW, int M, int ngroups) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= M * M) return;
    const int i = e / M, j = e - i * M;
    float s = 0.f;
    for (int g = 0; g < ngroups; ++g) s += tmp[(long)g * 4096 + i * 64 + j];
    dW[(long)i * M + j] = s;
}

// -------------------------------------------------------------------------------------------------
// k_fs_bwd_tok: per (tile jg, bh); mixing by all 4 waves, then each wave owns 2 blocks (no block barriers):
//   phase 1: Gt = mix(W, KV)    -> dQ_j = (dO_j G_j^T) / n_j + dz_j (x) ksum_j  (relu mask) ; dksum_j
//   phase 2: Gt = mix(W^T, dG)  -> dK_j = V_j dKV_j^T + 1 dksum_j^T (relu mask) ; dV_j = K_j dKV_j
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
    const float* dz;    // [bh][M][S]  (k_fs_wz<1>)
    const float* ksum;
    float* dksum;       // [bh][M][64]: written by k_fs_bwd_dq, read by k_fs_bwd_dkv
    int H, M, S, njg;
    float eps;
    int relu, normalize;
};
constexpr int FS_TOK_SMEM = FS_GT_BYTES;

__global__ __launch_bounds__(FT, 2) void k_fs_bwd_dq(const FsTokArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Gt = reinterpret_cast<u16*>(smem_raw);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, kg = lane >> 4;
    const int L = xcd_swizzle(blockIdx.x, gridDim.x);
    const int bh = L / a.njg, jgx = L - bh * a.njg, b = bh / a.H, h = bh - b * a.H;
    const int S = a.S, M = a.M;
    auto base = [&](const View& w) { return (const u16*)w.ptr + b * w.sb + h * w.sh; };
    auto mbase = [&](const MView& w) { return (u16*)w.ptr + b * w.sb + h * w.sh; };
    const u16 *qb = base(a.q), *gb = base(a.dout);
    u16* dqb = mbase(a.dq);
    const long sofs = (long)bh * a.njg * FE * IT;

    struct Side { float ninv, dz, ksum; };
    auto load_blk = [&](bf16x8 (&gv)[4][2], Side& sd, int j, int c0, int rv) {
        load_a64<false>(gv, gb, a.dout.sn, a.idx, (long)j * S + c0, rv, 0.f, lane);
        sd.ninv = 1.f; sd.dz = 0.f;
        sd.ksum = a.normalize ? a.ksum[((long)bh * M + j) * 64 + lane] : 0.f;   // lane = column d1
        if (a.normalize && lane < rv) {
            sd.ninv = a.ninv[((long)bh * M + j) * S + c0 + lane];
            sd.dz = a.dz[((long)bh * M + j) * S + c0 + lane];
        }
    };
    // one 64-row chunk: dQ rows, and the chunk's contribution to dksum (per-lane partials in the A layout)
    auto compute_store = [&](const bf16x8 (&gv)[4][2], const Side& sd, float (&dks_acc)[2][8], int bi, int j, int c0, int rv) {
        const long p0 = (long)j * S + c0;
        u16* Gb = Gt + bi * FD * GLD;
        f32x4 acc[4][4];
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) acc[st][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
        chunk_times_gt<true>(acc, gv, Gb, lane);   // (dO G^T)[s][d1] : B[k = d2][n = d1] = Gt[d2][d1]
        __builtin_amdgcn_sched_barrier(0);         // keep the q loads below the MFMAs (register pressure)
        if (a.normalize) {
            bf16x8 qv[4][2];
            if (a.relu) load_a64<true>(qv, qb, a.q.sn, a.idx, p0, rv, a.eps, lane);
            else        load_a64<false>(qv, qb, a.q.sn, a.idx, p0, rv, a.eps, lane);
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const float dzr = __shfl(sd.dz, st * 16 + n, 64);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const s16x8 qs = __builtin_bit_cast(s16x8, qv[st][ks]);
#pragma unroll
                    for (int t = 0; t < 8; ++t) dks_acc[ks][t] += dzr * bf((u16)qs[t]);
                }
            }
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = st * 16 + kg * 4 + r;
                    const float ni = __shfl(sd.ninv, row, 64), dzr = __shfl(sd.dz, row, 64);
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) acc[st][tn][r] = acc[st][tn][r] * ni + dzr * __shfl(sd.ksum, tn * 16 + n, 64);
                }
        }
        if (c0 + 64 >= S) {
            wave_lds_fence();
            stage64(Gb, acc, lane);
            wave_lds_fence();
            if (a.relu) store64<true>(dqb, a.dq.sn, a.idx, p0, rv, Gb, qb, a.q.sn, lane);
            else        store64<false>(dqb, a.dq.sn, a.idx, p0, rv, Gb, nullptr, 0, lane);
        } else {
            if (a.relu) store64_direct<true>(dqb, a.dq.sn, a.idx, p0, rv, acc, qb, a.q.sn, lane);
            else        store64_direct<false>(dqb, a.dq.sn, a.idx, p0, rv, acc, nullptr, 0, lane);
        }
    };
    // dksum[col]: reduce the per-lane partials over the 16 row-lanes (n); columns = 32 ks + 8 kg + t
    auto finish_dks = [&](const float (&dks_acc)[2][8], int j) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                float v = dks_acc[ks][t];
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64);
                v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                if (n == 0) a.dksum[((long)bh * M + j) * 64 + ks * 32 + kg * 8 + t] = v;
            }
    };
    auto zero_dks = [](float (&d)[2][8]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < 8; ++t) d[ks][t] = 0.f;
    };

    if (S <= 64) {   // operands of the wave's first block are fetched before the mixing
        const int jA = jgx * IT + wave, jB = jA + 4;
        bf16x8 gvA[4][2], gvB[4][2];
        Side sA, sB;
        if (jA < M) load_blk(gvA, sA, jA, 0, S);
        mix_tile_to_lds<0>(Gt, a.state + sofs, a.njg, a.W, a.ldw, M, jgx * IT, tid);
        __syncthreads();
        if (jB < M) load_blk(gvB, sB, jB, 0, S);
        float dks_acc[2][8];
        if (jA < M) { zero_dks(dks_acc); compute_store(gvA, sA, dks_acc, wave, jA, 0, S); if (a.normalize) finish_dks(dks_acc, jA); }
        if (jB < M) { zero_dks(dks_acc); compute_store(gvB, sB, dks_acc, wave + 4, jB, 0, S); if (a.normalize) finish_dks(dks_acc, jB); }
        return;
    }
    mix_tile_to_lds<0>(Gt, a.state + sofs, a.njg, a.W, a.ldw, M, jgx * IT, tid);
    __syncthreads();
    for (int bi = wave; bi < IT; bi += 4) {
        const int j = jgx * IT + bi;
        if (j >= M) continue;
        float dks_acc[2][8];
        zero_dks(dks_acc);
        for (int c0 = 0; c0 < S; c0 += 64) {
            const int rv = min(64, S - c0);
            bf16x8 gv[4][2];
            Side sd;
            load_blk(gv, sd, j, c0, rv);
            compute_store(gv, sd, dks_acc, bi, j, c0, rv);
        }
        if (a.normalize) finish_dks(dks_acc, j);
    }
}

__global__ __launch_bounds__(FT, 2) void k_fs_bwd_dkv(const FsTokArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Gt = reinterpret_cast<u16*>(smem_raw);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, kg = lane >> 4;
    const int L = xcd_swizzle(blockIdx.x, gridDim.x);
    const int bh = L / a.njg, jgx = L - bh * a.njg, b = bh / a.H, h = bh - b * a.H;
    const int S = a.S, M = a.M;
    auto base = [&](const View& w) { return (const u16*)w.ptr + b * w.sb + h * w.sh; };
    auto mbase = [&](const MView& w) { return (u16*)w.ptr + b * w.sb + h * w.sh; };
    const u16 *kb = base(a.k), *vb = base(a.v);
    u16 *dkb = mbase(a.dk), *dvb = mbase(a.dv);
    const long sofs = (long)bh * a.njg * FE * IT;
    auto load_k = [&](bf16x8 (&kv)[4][2], int j, int c0, int rv) {
        if (a.relu) load_a64<true>(kv, kb, a.k.sn, a.idx, (long)j * S + c0, rv, a.eps, lane);
        else        load_a64<false>(kv, kb, a.k.sn, a.idx, (long)j * S + c0, rv, a.eps, lane);
    };
    auto compute_store = [&](const bf16x8 (&kv)[4][2], int bi, int j, int c0, int rv) {
        const long p0 = (long)j * S + c0;
        u16* Gb = Gt + bi * FD * GLD;
        const bool last = c0 + 64 >= S;
        bf16x8 vv[4][2];
        load_a64<false>(vv, vb, a.v.sn, a.idx, p0, rv, 0.f, lane);
        // dV first, kept packed as bf16 pairs while dK is computed (both need the intact Gb)
        unsigned pv[4][4][2];
        {
            f32x4 accV[4][4];
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) accV[st][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
            chunk_times_gt<false>(accV, kv, Gb, lane);   // dV[s][d2] = sum_d1 K[s][d1] dKVt[d2][d1]
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) {
                    pv[st][tn][0] = pack_bf16x2(accV[st][tn][0], accV[st][tn][1]);
                    pv[st][tn][1] = pack_bf16x2(accV[st][tn][2], accV[st][tn][3]);
                }
        }
        f32x4 accK[4][4];
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) accK[st][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
        chunk_times_gt<true>(accK, vv, Gb, lane);        // dK[s][d1] = sum_d2 V[s][d2] dKVt[d2][d1]
        if (a.normalize) {
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
                const float dk = a.dksum[((long)bh * M + j) * 64 + tn * 16 + n];
#pragma unroll
                for (int st = 0; st < 4; ++st)
#pragma unroll
                    for (int r = 0; r < 4; ++r) accK[st][tn][r] += dk;
            }
        }
        if (last) {
            wave_lds_fence();
            stage64(Gb, accK, lane);
            wave_lds_fence();
            if (a.relu) store64<true>(dkb, a.dk.sn, a.idx, p0, rv, Gb, kb, a.k.sn, lane);
            else        store64<false>(dkb, a.dk.sn, a.idx, p0, rv, Gb, nullptr, 0, lane);
            wave_lds_fence();
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        Gb[(st * 16 + kg * 4 + r) * GLD + tn * 16 + n] = (u16)(pv[st][tn][r >> 1] >> ((r & 1) * 16));
            wave_lds_fence();
            store64<false>(dvb, a.dv.sn, a.idx, p0, rv, Gb, nullptr, 0, lane);
        } else {
            if (a.relu) store64_direct<true>(dkb, a.dk.sn, a.idx, p0, rv, accK, kb, a.k.sn, lane);
            else        store64_direct<false>(dkb, a.dk.sn, a.idx, p0, rv, accK, nullptr, 0, lane);
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = st * 16 + kg * 4 + r;
                    if (row < rv) {
                        const long tr = tok_row(a.idx, p0 + row);
#pragma unroll
                        for (int tn = 0; tn < 4; ++tn)
                            dvb[tr * a.dv.sn + tn * 16 + n] = (u16)(pv[st][tn][r >> 1] >> ((r & 1) * 16));
                    }
                }
        }
    };

    if (S <= 64) {
        const int jA = jgx * IT + wave, jB = jA + 4;
        bf16x8 kvA[4][2], kvB[4][2];
        if (jA < M) load_k(kvA, jA, 0, S);
        mix_tile_to_lds<1>(Gt, a.dstate + sofs, a.njg, a.W, a.ldw, M, jgx * IT, tid);
        __syncthreads();
        if (jB < M) load_k(kvB, jB, 0, S);
        if (jA < M) compute_store(kvA, wave, jA, 0, S);
        if (jB < M) compute_store(kvB, wave + 4, jB, 0, S);
        return;
    }
    mix_tile_to_lds<1>(Gt, a.dstate + sofs, a.njg, a.W, a.ldw, M, jgx * IT, tid);
    __syncthreads();
    for (int bi = wave; bi < IT; bi += 4) {
        const int j = jgx * IT + bi;
        if (j >= M) continue;
        for (int c0 = 0; c0 < S; c0 += 64) {
            const int rv = min(64, S - c0);
            bf16x8 kv[4][2];
            load_k(kv, j, c0, rv);
            compute_store(kv, bi, j, c0, rv);
        }
    }
}

}  // namespace fast
}  // namespace mhla
