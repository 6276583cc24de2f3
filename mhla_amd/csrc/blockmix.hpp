// Block-mixing MHLA kernels, generic fp32-compute path (any dtype in/out, D <= 128, any M, S).
//
// Forward  (reference: mhla_dit/mhla/mhla.py:262-268, wan/mhla_utils.py:331-341)
//   k_bm_state<MODE 0>   per (block j, bh): KV_j = K_j^T V_j, ksum_j, z_j          -> ws
//   k_mix                per bh:            G = W . KV      ([M x M] . [M x D^2])   -> ws
//   k_bm_out             per (block i, bh): n_i = W z + eps ; O_i = Q_i G_i / n_i
// Backward
//   state, mix (recompute KV, G) ; k_bm_state<MODE 1>: dG_i = Q_i^T (dO_i/n_i), dn_i
//   k_mix<TRANS>: dKV = W^T dG ; k_dw + k_dw_reduce: dW ; k_bm_bwd_tok: dQ, dK, dV.
//
// All contractions run on the fp32-input MFMA (v_mfma_f32_16x16x4_f32): exact fp32, so the
// same kernels serve fp32, bf16 and fp16 tensors (converted while staging tiles into LDS).
#pragma once
#include "common.hpp"

namespace mhla {

constexpr int NTHREADS = 256;

__host__ __device__ constexpr int bm_chunk(int DT) { return DT <= 5 ? 64 : 32; }

// out[bh][r][s] = f(sum_c Wm(r, c) x[bh][c][s])  -- the two small [M x M] x [M x S] products of the normaliser path:
//   MODE 0: Wm = W,   f(v) = 1 / (eps + v)   -> 1/n      MODE 1: Wm = W^T, f(v) = v -> dz = W^T dn
// grid (ceil(S/64), ceil(M/64), bh); LDS tiles of 64 x 64, all global loads of a tile issued before the LDS stores.
template <int MODE>
__global__ __launch_bounds__(NTHREADS) void k_wz(const float* __restrict__ W, int ldw, const float* __restrict__ x,
                                                 float* __restrict__ out, int M, int S, float eps) {
    __shared__ float Ws[64 * 65];
    __shared__ __attribute__((aligned(16))) float xs[64 * 64];
    const int tid = threadIdx.x, c0 = blockIdx.x * 64, r0 = blockIdx.y * 64, bh = blockIdx.z, rv = min(64, S - c0);
    const int sq = tid & 15, rq = tid >> 4;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < M; k0 += 64) {
        float wreg[16], xreg[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int v = tid + t * NTHREADS, r = v >> 6, c = v & 63;
            const int gr = r0 + r, gk = k0 + c;       // Ws[r][c] = Wm(r0 + r, k0 + c)
            wreg[t] = (gr < M && gk < M) ? (MODE ? W[(long)gk * ldw + gr] : W[(long)gr * ldw + gk]) : 0.f;
            const int xr = k0 + r;                     // xs[r][c] = x[k0 + r][c0 + c]
            xreg[t] = (xr < M && c < rv) ? x[((long)bh * M + xr) * S + c0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int v = tid + t * NTHREADS, r = v >> 6, c = v & 63;
            Ws[r * 65 + c] = wreg[t];
            xs[r * 64 + c] = xreg[t];
        }
        __syncthreads();
#pragma unroll 4
        for (int c = 0; c < 64; ++c) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + c * 64 + sq * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] += Ws[(rq + 16 * i) * 65 + c] * xv;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + rq + 16 * i;
        if (r < M) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int sc = sq * 4 + t;
                if (sc < rv) out[((long)bh * M + r) * S + c0 + sc] = MODE ? acc[i][t] : 1.f / (eps + acc[i][t]);
            }
        }
    }
}

struct StateArgs {
    View x, y;        // MODE 0: k_num, v        MODE 1: q_num, dout
    View kd, qd;      // MODE 0: k_den, q_den    (used when normalize)
    View o;           // MODE 1: forward output (row dot with dout)
    const float* g;   // MODE 1, split.hpp's row-dots-from-G variant: the mixed summaries G (24-bit floats) instead of o
    const int* idx;
    const float* W;   // MODE 1
    int ldw;
    const float* ninv;   // MODE 1 in  [bh][M][S]  1 / n  (k_wz<0>)
    float* out;       // [bh][M][D][D]
    float* ksum;      // MODE 0 out [bh][M][D]
    float* zo;        // MODE 0 out [bh][M][S]
    float* dn;        // MODE 1 out [bh][M][S]
    int H, M, S, D;   // D: head dim of the block-mix modes (DX = DY = D)
    float eps;
    int relu, normalize, split;
    // MODE 2 (plain X^T Y, causal path): X is [.., DX], Y is [.., DY]; blockIdx.z enumerates
    // (x-strip, y-strip) of 16*DT columns each; rows past `T` tokens are zero; output scaled by alpha.
    int DX, DY;
    long T;
    float alpha;
    // fused rotary prologue (split.hpp, MODE 0): x feeds KV rotated by the token's angles, cos/sin [rows][D/2] fp32
    const float* rcos;
    const float* rsin;
    long ldr;
    long es;   // elements from one block's summary to the next in `out` (split.hpp / split16.hpp; the kernels of this file use D D)
    // MODE 1, 16-bit tensors at the default arithmetic: what the forward's store of O rounded away, bf16 [bh][M S][D] in block-major
    // token order (OutArgs::olo); added to O before the row dot.  Null: the row dot uses O as stored.
    const unsigned short* olo;
    // split.hpp PRO (Wan inference, mhla_blockmix_wan_pro_fwd): x (keys) and qd (queries) are the 16-bit PROJECTION outputs; the q / k
    // prologue of wan/mhla_utils.py:268-272 -- relu(x * rstd[token] * w[channel]) + eps, fp32 -- is applied while they are loaded.
    // rstd [B][pro_n] = 1 / sqrt(mean_C(x^2) + norm_eps) per token (null: 1), w [H D] the RMSNorm weights (null: 1)
    const float *pro_rk, *pro_wk, *pro_rq, *pro_wq;
    long pro_n;
};

template <int DT>
__host__ __device__ constexpr int state_smem_floats() {
    return 3 * bm_chunk(DT) * ld_kmajor(DT * 16) + DT * 16 + bm_chunk(DT);
}

// acc[i] += X^T Y over the staged chunk; output tile t = wave + 4 i -> (tm, tn) = (t / DT, t % DT)
template <int DT, int NT>
__device__ __forceinline__ void xty_accum(f32x4 (&acc)[NT], const float* __restrict__ Xs, const float* __restrict__ Ys,
                                          int ld, int kend, int wave, int lane) {
    const int r16 = lane & 15, kq = lane >> 4;
    for (int k0 = 0; k0 < kend; k0 += 4) {
        const float* xr = Xs + (k0 + kq) * ld + r16;
        const float* yr = Ys + (k0 + kq) * ld + r16;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int t = wave + 4 * i;
            if (t < DT * DT) {
                const int tm = t / DT, tn = t - tm * DT;
                acc[i] = mfma4(xr[tm * 16], yr[tn * 16], acc[i]);
            }
        }
    }
}

template <typename T, int DT, int MODE>
__global__ __launch_bounds__(NTHREADS) void k_bm_state(const StateArgs a) {
    constexpr int DP = DT * 16, CH = bm_chunk(DT), LD = ld_kmajor(DP), NT = (DT * DT + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;
    float* Ys = Xs + CH * LD;
    float* Es = Ys + CH * LD;
    float* vecd = Es + CH * LD;   // [DP]
    float* vecr = vecd + DP;      // [CH]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int blk = blockIdx.x, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int D = a.D;
    const long p0 = (long)blk * a.S;
    const T* xb = (const T*)a.x.ptr + b * a.x.sb + h * a.x.sh;
    const T* yb = (const T*)a.y.ptr + b * a.y.sb + h * a.y.sh;
    // MODE 2: strips and token tail
    int x0 = 0, y0 = 0, S = a.S;
    if (MODE == 2) {
        const int nsy = (a.DY + DP - 1) / DP;
        x0 = (blockIdx.z / nsy) * DP;
        y0 = (blockIdx.z % nsy) * DP;
        S = (int)max(0L, min((long)a.S, a.T - p0));
    }

    f32x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float ks = 0.f;

    for (int c0 = 0; c0 < S; c0 += CH) {
        const int rv = min(CH, S - c0), kend = (rv + 3) & ~3;
        if (MODE == 2) {
            load_tile<T, DP, false>(Xs, LD, xb + x0, a.x.sn, a.idx, p0 + c0, rv, kend, a.DX - x0, 0.f, tid, NTHREADS);
            load_tile<T, DP, false>(Ys, LD, yb + y0, a.y.sn, a.idx, p0 + c0, rv, kend, a.DY - y0, 0.f, tid, NTHREADS);
            __syncthreads();
        } else if (MODE == 0) {
            if (a.relu) load_tile<T, DP, true>(Xs, LD, xb, a.x.sn, a.idx, p0 + c0, rv, kend, D, a.eps, tid, NTHREADS);
            else        load_tile<T, DP, false>(Xs, LD, xb, a.x.sn, a.idx, p0 + c0, rv, kend, D, a.eps, tid, NTHREADS);
            load_tile<T, DP, false>(Ys, LD, yb, a.y.sn, a.idx, p0 + c0, rv, kend, D, 0.f, tid, NTHREADS);
            if (a.normalize && a.split) {
                const T* kb = (const T*)a.kd.ptr + b * a.kd.sb + h * a.kd.sh;
                load_tile<T, DP, false>(Es, LD, kb, a.kd.sn, a.idx, p0 + c0, rv, kend, D, 0.f, tid, NTHREADS);
            }
            __syncthreads();
            if (a.normalize && tid < DP) {
                const float* src = a.split ? Es : Xs;
                for (int r = 0; r < rv; ++r) ks += src[r * LD + tid];
            }
        } else {
            if (a.relu) load_tile<T, DP, true>(Xs, LD, xb, a.x.sn, a.idx, p0 + c0, rv, kend, D, a.eps, tid, NTHREADS);
            else        load_tile<T, DP, false>(Xs, LD, xb, a.x.sn, a.idx, p0 + c0, rv, kend, D, a.eps, tid, NTHREADS);
            load_tile<T, DP, false>(Ys, LD, yb, a.y.sn, a.idx, p0 + c0, rv, kend, D, 0.f, tid, NTHREADS);
            if (a.normalize) {
                const T* ob = (const T*)a.o.ptr + b * a.o.sb + h * a.o.sh;
                load_tile<T, DP, false>(Es, LD, ob, a.o.sn, a.idx, p0 + c0, rv, kend, D, 0.f, tid, NTHREADS);
                for (int r = tid; r < rv; r += NTHREADS) vecr[r] = a.ninv[((long)bh * a.M + blk) * S + c0 + r];
                if (a.olo) {   // O at fp32 grade: + what its 16-bit store lost (every thread adds to the elements it staged itself)
                    const unsigned short* lo = a.olo + ((long)bh * a.M * S + p0 + c0) * D;
                    for (int v = tid; v < rv * (DP / 4); v += NTHREADS) {
                        const int r = v / (DP / 4), c = (v - r * (DP / 4)) * 4;
                        if (c < D) {
                            const uint2 w = *reinterpret_cast<const uint2*>(lo + (long)r * D + c);
                            float* e = Es + r * LD + c;
                            e[0] += bf16_to_f32((unsigned short)(w.x & 0xffffu)); e[1] += bf16_to_f32((unsigned short)(w.x >> 16));
                            e[2] += bf16_to_f32((unsigned short)(w.y & 0xffffu)); e[3] += bf16_to_f32((unsigned short)(w.y >> 16));
                        }
                    }
                }
            }
            __syncthreads();
            if (a.normalize) {
                for (int r = wave; r < rv; r += 4) {
                    float d = 0.f;
                    for (int c = lane; c < DP; c += 64) d += Ys[r * LD + c] * Es[r * LD + c];
                    d = wave_sum(d);
                    const float ninv = vecr[r];
                    for (int c = lane; c < DP; c += 64) Ys[r * LD + c] *= ninv;
                    if (lane == 0) a.dn[((long)bh * a.M + blk) * S + c0 + r] = -d * ninv;
                }
                __syncthreads();
            }
        }
        xty_accum<DT, NT>(acc, Xs, Ys, LD, kend, wave, lane);
        __syncthreads();
    }

    // KV_j / dG_i -> ws : C layout col = lane & 15, row = (lane >> 4) * 4 + r
    const int DX = MODE == 2 ? a.DX : D, DY = MODE == 2 ? a.DY : D;
    const float alpha = MODE == 2 ? a.alpha : 1.f;
    float* ob = a.out + ((long)bh * a.M + blk) * DX * DY;
    const int r16 = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int t = wave + 4 * i;
        if (t < DT * DT) {
            const int tm = t / DT, tn = t - tm * DT;
            const int col = y0 + tn * 16 + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = x0 + tm * 16 + kq * 4 + r;
                if (row < DX && col < DY) ob[(long)row * DY + col] = alpha * acc[i][r];
            }
        }
    }

    if (MODE == 0 && a.normalize) {
        if (tid < DP) {
            vecd[tid] = ks;
            if (tid < D) a.ksum[((long)bh * a.M + blk) * D + tid] = ks;
        }
        const T* qb = (const T*)a.qd.ptr + b * a.qd.sb + h * a.qd.sh;
        for (int c0 = 0; c0 < S; c0 += CH) {
            const int rv = min(CH, S - c0);
            __syncthreads();
            if (a.relu) load_tile<T, DP, true>(Xs, LD, qb, a.qd.sn, a.idx, p0 + c0, rv, rv, D, a.eps, tid, NTHREADS);
            else        load_tile<T, DP, false>(Xs, LD, qb, a.qd.sn, a.idx, p0 + c0, rv, rv, D, a.eps, tid, NTHREADS);
            __syncthreads();
            for (int r = wave; r < rv; r += 4) {
                float d = 0.f;
                for (int c = lane; c < DP; c += 64) d += Xs[r * LD + c] * vecd[c];
                d = wave_sum(d);
                if (lane == 0) a.zo[((long)bh * a.M + blk) * S + c0 + r] = d;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Out[bh][i][e] = sum_j Wm(i, j) In[bh][j][e]   -- the 1x1 conv over the block axis as a GEMM.
//   TRANS : Wm(i, j) = W[j][i]   (backward: dKV_j = sum_i W[i][j] dG_i)
//   MASK  : 0 none, 1 strictly-lower (causal prefix mixing: j < i ; with TRANS: rows i > j)
// grid (e-strips of 128, i-tiles of 64, bh); 4 waves, wave w owns columns [32 w, 32 w + 32).
// ---------------------------------------------------------------------------------------------
struct MixArgs {
    const float* W;
    int ldw;
    const float* in;
    float* out;
    int M;
    long E;
    long es;   // row stride of `in` / `out` in elements (split.hpp: E + padding; the kernels of this file use E)
};
constexpr int MIX_TI = 64, MIX_TE = 128, MIX_LDW = 80, MIX_LDI = 144;
constexpr int MIX_SMEM_FLOATS = 64 * MIX_LDW + 64 * MIX_LDI;

template <int TRANS, int MASK>
__global__ __launch_bounds__(NTHREADS) void k_mix(const MixArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ws = smem;                 // [64 k][80]   Ws[k][i] = Wm(i0 + i, k0 + k)
    float* Is = Ws + 64 * MIX_LDW;    // [64 k][144]  Is[k][e]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r16 = lane & 15, kq = lane >> 4;
    const long e0 = (long)blockIdx.x * MIX_TE;
    const int i0 = blockIdx.y * MIX_TI, bh = blockIdx.z, M = a.M;
    const float* in = a.in + (long)bh * M * a.E;
    float* out = a.out + (long)bh * M * a.E;
    const bool vec_ok = (a.E & 3) == 0;

    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // reduction range over k (input block index)
    int kbeg = 0, kend_all = M;
    if (MASK == 1 && !TRANS) kend_all = min(M, i0 + MIX_TI);   // k < i
    if (MASK == 1 && TRANS) kbeg = (i0 / 64) * 64;              // k > i

    for (int k0 = kbeg; k0 < kend_all; k0 += 64) {
        const int kv = min(64, M - k0), kpad = (kv + 3) & ~3;
        // W tile (k-major)
        for (int v = tid; v < kpad * 64; v += NTHREADS) {
            const int k = v >> 6, i = v & 63;
            const int gi = i0 + i, gk = k0 + k;
            float w = 0.f;
            if (gi < M && gk < M) {
                const bool keep = (MASK == 0) || (TRANS ? (gk > gi) : (gk < gi));
                if (keep) w = TRANS ? a.W[(long)gk * a.ldw + gi] : a.W[(long)gi * a.ldw + gk];
            }
            Ws[k * MIX_LDW + i] = w;
        }
        // In tile
        for (int v = tid; v < kpad * 32; v += NTHREADS) {
            const int k = v >> 5, c = (v & 31) * 4;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (k < kv && e0 + c < a.E) {
                const float* s = in + (long)(k0 + k) * a.E + e0 + c;
                if (vec_ok && e0 + c + 3 < a.E) x = *reinterpret_cast<const f32x4*>(s);
                else
                    for (int t = 0; t < 4; ++t)
                        if (e0 + c + t < a.E) x[t] = s[t];
            }
            *reinterpret_cast<f32x4*>(Is + k * MIX_LDI + c) = x;
        }
        __syncthreads();
        for (int kk = 0; kk < kpad; kk += 4) {
            float av[4], bv[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = Ws[(kk + kq) * MIX_LDW + i * 16 + r16];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = Is[(kk + kq) * MIX_LDI + wave * 32 + j * 16 + r16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma4(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long col = e0 + wave * 32 + j * 16 + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + i * 16 + kq * 4 + r;
                if (row < M && col < a.E) out[(long)row * a.E + col] = acc[i][j][r];
            }
        }
}

// ---------------------------------------------------------------------------------------------
// O_i = (Q_i G_i) / n_i ,  n_i[s] = sum_j W[i][j] z_j[s] + eps
// ---------------------------------------------------------------------------------------------
struct OutArgs {
    View q;
    MView o;
    const int* idx;
    const float* W;
    int ldw;
    const float* g;   // [bh][M][D][D]
    const float* ninv;   // [bh][M][S]  1 / n  (k_wz<0>)
    int H, M, S, D;
    float eps;
    int relu, normalize;
    const float* rcos;   // fused rotary prologue (split.hpp): q is rotated while it is loaded
    const float* rsin;
    long ldr;
    // fused epilogue (split.hpp, k_sp_out<.., EPI>): y = rmsnorm_D(O) * nw [* silu(gate)] stored in the gate's dtype
    const float* nw;
    float neps;
    View gate;
    long es;   // elements from one block's summary to the next in `g` (split.hpp / split16.hpp)
    // 16-bit tensors at the default arithmetic, when a backward will follow: O - fl(O) as bf16 [bh][M S][D] (block-major token
    // order), so that the backward's row dots dO . O see O at fp32 grade; null: not written.  skip_out: ONLY this residual is
    // written (the backward recomputing it when the forward's workspace was not kept).
    unsigned short* olo;
    int skip_out;
    const float *pro_rq, *pro_wq;   // split.hpp PRO: the q prologue on load (StateArgs)
    long pro_n;
    int nbh;   // split.hpp k_sp_out<.., FLAT>: B * H (the launch's grid no longer says)
};

template <int DT>
__host__ __device__ constexpr int out_smem_floats() {
    return DT * 16 * ld_kmajor(DT * 16) + bm_chunk(DT) * ld_xmajor(DT * 16) + bm_chunk(DT) * (DT * 16 + 4) + bm_chunk(DT);
}

// acc[i] += A B for the output tile t = wave + 4 i -> (tm, tn) = (t / DT, t % DT), reduction length kdim.
//   A: x-major As[row][k] (AT = false) or k-major As[k][row] (AT = true)
//   B: k-major Bs[k][col] (BT = false) or x-major Bs[col][k] (BT = true)
template <int DT, int NT, bool BT, bool AT = false>
__device__ __forceinline__ void ab_accum(f32x4 (&acc)[NT], const float* __restrict__ As, int lda,
                                         const float* __restrict__ Bs, int ldb, int ntiles, int kdim, int wave,
                                         int lane) {
    const int r16 = lane & 15, kq = lane >> 4;
    for (int k0 = 0; k0 < kdim; k0 += 4) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int t = wave + 4 * i;
            if (t < ntiles) {
                const int tm = t / DT, tn = t - tm * DT;
                const float av = AT ? As[(k0 + kq) * lda + tm * 16 + r16] : As[(tm * 16 + r16) * lda + k0 + kq];
                const float bv = BT ? Bs[(tn * 16 + r16) * ldb + k0 + kq] : Bs[(k0 + kq) * ldb + tn * 16 + r16];
                acc[i] = mfma4(av, bv, acc[i]);
            }
        }
    }
}

template <typename T, int DT>
__global__ __launch_bounds__(NTHREADS) void k_bm_out(const OutArgs a) {
    constexpr int DP = DT * 16, CH = bm_chunk(DT), LDG = ld_kmajor(DP), LDQ = ld_xmajor(DP), LDO = DP + 4;
    constexpr int NT = ((CH / 16) * DT + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Gs = smem;               // [DP][LDG]
    float* Qs = Gs + DP * LDG;      // [CH][LDQ]
    float* Os = Qs + CH * LDQ;      // [CH][LDO]
    float* ninv = Os + CH * LDO;    // [CH]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r16 = lane & 15, kq = lane >> 4;
    const int blk = blockIdx.x, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int S = a.S, D = a.D;
    const long p0 = (long)blk * S;
    const T* qb = (const T*)a.q.ptr + b * a.q.sb + h * a.q.sh;
    T* ob = (T*)a.o.ptr + b * a.o.sb + h * a.o.sh;

    load_mat_f32<DP>(Gs, LDG, a.g + ((long)bh * a.M + blk) * D * D, D, D, DP, D, tid, NTHREADS, (D & 3) == 0);

    for (int c0 = 0; c0 < S; c0 += CH) {
        const int rv = min(CH, S - c0), rpad = (rv + 15) & ~15;
        if (a.relu) load_tile<T, DP, true>(Qs, LDQ, qb, a.q.sn, a.idx, p0 + c0, rv, rpad, D, a.eps, tid, NTHREADS);
        else        load_tile<T, DP, false>(Qs, LDQ, qb, a.q.sn, a.idx, p0 + c0, rv, rpad, D, a.eps, tid, NTHREADS);
        for (int r = tid; r < rv; r += NTHREADS) {
            ninv[r] = a.normalize ? a.ninv[((long)bh * a.M + blk) * S + c0 + r] : 1.f;
        }
        __syncthreads();
        f32x4 acc[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int ntiles = (rpad / 16) * DT;
        ab_accum<DT, NT, false>(acc, Qs, LDQ, Gs, LDG, ntiles, DP, wave, lane);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int t = wave + 4 * i;
            if (t < ntiles) {
                const int tm = t / DT, tn = t - tm * DT;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = tm * 16 + kq * 4 + r;
                    Os[row * LDO + tn * 16 + r16] = acc[i][r] * (row < rv ? ninv[row] : 0.f);
                }
            }
        }
        __syncthreads();
        if (!a.skip_out) store_tile<T, DP>(ob, a.o.sn, a.idx, p0 + c0, Os, LDO, rv, D, tid, NTHREADS);
        if (a.olo) {
            unsigned short* lo = a.olo + ((long)bh * a.M * S + p0 + c0) * D;
            for (int v = tid; v < rv * (DP / 4); v += NTHREADS) {
                const int r = v / (DP / 4), c = (v - r * (DP / 4)) * 4;
                if (c < D) *reinterpret_cast<uint2*>(lo + (long)r * D + c) = store_residual4<T>(*reinterpret_cast<const f32x4*>(Os + r * LDO + c));
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// dWp[bh][i][j] = sum_e X[bh][i][e] Y[bh][j][e]  (+ second segment X2/Y2 with row length E2)
// grid (tile pairs, bh, E-slices); MASK 1: only j < i (strictly lower; causal dmix off-diagonal).
// ---------------------------------------------------------------------------------------------
struct DwArgs {
    const float* x;
    const float* y;
    long E;
    const float* x2;
    const float* y2;
    long E2;
    float* out;   // [bh][nsplit][M][M]
    int M, tiles;
    int nsplit;   // the E range (and E2) is cut into nsplit slices, one workgroup each (blockIdx.z)
    long es;      // row stride of x / y in elements (split.hpp's k_sp_dw: E + padding; k_dw uses E)
};
constexpr int DW_LD = 34;
constexpr int DW_SMEM_FLOATS = 2 * 64 * DW_LD;

template <int MASK>
__global__ __launch_bounds__(NTHREADS) void k_dw(const DwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;              // [64 i][34]
    float* Ys = Xs + 64 * DW_LD;   // [64 j][34]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r16 = lane & 15, kq = lane >> 4;
    const int it = blockIdx.x / a.tiles, jt = blockIdx.x - it * a.tiles, bh = blockIdx.y, M = a.M;
    const int i0 = it * 64, j0 = jt * 64;
    const int split = blockIdx.z;
    float* out = a.out + ((long)bh * a.nsplit + split) * M * M;
    if (MASK == 1 && j0 > i0 + 63) return;   // tile entirely above the diagonal (output is masked by the reducer)

    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int seg = 0; seg < 2; ++seg) {
        const float* X = seg ? a.x2 : a.x;
        const float* Y = seg ? a.y2 : a.y;
        const long E = seg ? a.E2 : a.E;
        if (!X || E <= 0) continue;
        X += (long)bh * M * E;
        Y += (long)bh * M * E;
        const bool vec_ok = (E & 3) == 0;
        const long per = ((E + a.nsplit - 1) / a.nsplit + 31) & ~31L;       // slice length, multiple of the 32-wide chunk
        const long ebeg = (long)split * per, eend = ebeg + per < E ? ebeg + per : E;
        for (long e0 = ebeg; e0 < eend; e0 += 32) {
            f32x4 xr[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int v = tid + t * NTHREADS, which = v >> 9, r = (v >> 3) & 63, c = (v & 7) * 4;
                const int g = (which ? j0 : i0) + r;
                const float* src = (which ? Y : X) + (long)g * E + e0 + c;
                xr[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (g < M && e0 + c < eend) {
                    if (vec_ok && e0 + c + 3 < eend) xr[t] = *reinterpret_cast<const f32x4*>(src);
                    else
                        for (int u = 0; u < 4; ++u)
                            if (e0 + c + u < eend) xr[t][u] = src[u];
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int v = tid + t * NTHREADS, which = v >> 9, r = (v >> 3) & 63, c = (v & 7) * 4;
                float* d = (which ? Ys : Xs) + r * DW_LD + c;
                *reinterpret_cast<f32x2*>(d) = f32x2{xr[t][0], xr[t][1]};
                *reinterpret_cast<f32x2*>(d + 2) = f32x2{xr[t][2], xr[t][3]};
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < 32; kk += 4) {
                const float av = Xs[(wave * 16 + r16) * DW_LD + kk + kq];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = mfma4(av, Ys[(j * 16 + r16) * DW_LD + kk + kq], acc[j]);
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = j0 + j * 16 + r16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = i0 + wave * 16 + kq * 4 + r;
            if (row < M && col < M) out[(long)row * M + col] = acc[j][r];
        }
    }
}

// dW[i][j] = sum over parts of dWp[part][i][j]  (parts = (b,h) x E-slices; fixed summation order -> deterministic).
//   MASK 1: j < i from dWp, j == i from diag[bh][i] (nbh rows), j > i WRITTEN AS ZERO -- every entry of the leading [M, M] block of
//   dW is defined after this kernel (mhla_amd/ops.py and the C++ nodes allocate dmix with torch.empty and rely on it).
// 256 threads: EL elements x (256 / EL) part-lanes, then an LDS reduce in lane order.  EL = 64 for large M; EL = 16 when M * M is
// small and the parts are many (M = 16: 4 workgroups each summing 384 values per thread took 20 us).
template <int MASK, int EL = 64>
__global__ __launch_bounds__(256) void k_dw_reduce(const float* __restrict__ dwp, const float* __restrict__ diag,
                                                   float* __restrict__ dW, int ldd, int M, int nparts, int nbh) {
    constexpr int PL = 256 / EL;
    __shared__ float red[PL][EL];
    const int el = threadIdx.x % EL, pl = threadIdx.x / EL;
    const int e = blockIdx.x * EL + el;
    const int i = e < M * M ? e / M : 0, j = e < M * M ? e - i * M : 0;
    float s = 0.f;
    if (e < M * M) {
        if (MASK == 0 || j < i) {
#pragma unroll 8
            for (int p = pl; p < nparts; p += PL) s += dwp[(long)p * M * M + e];
        } else if (j == i) {
            for (int b = pl; b < nbh; b += PL) s += diag[(long)b * M + i];
        }
    }
    red[pl][el] = s;
    __syncthreads();
    if (pl == 0 && e < M * M) {   // MASK: entries above the diagonal are written as zeros (their partial sums are empty)
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < PL; ++q) t += red[q][el];
        dW[(long)i * ldd + j] = t;
    }
}

// ---------------------------------------------------------------------------------------------
// Token gradients, per (block j, bh):
//   dQ_j = dP_j G_j^T + dz_j (x) ksum_j ;  dK_j = V_j dKV_j^T + 1 dksum_j^T ;  dV_j = K_j dKV_j
//   dP = dO / n ; dz_j[s] = sum_i W[i][j] dn_i[s] ; dksum_j = sum_s dz_j[s] Qden_j[s]
// ---------------------------------------------------------------------------------------------
struct TokArgs {
    View q, k, v, qd, kd, dout;
    MView dq, dk, dv, dqd, dkd;
    const int* idx;
    const float* W;
    int ldw;
    const float* g;      // [bh][M][D][D]
    const float* dkv;    // [bh][M][D][D]
    const float* ninv;   // [bh][M][S]  1 / n  (k_wz<0>)
    const float* dz;     // [bh][M][S]  W^T dn (k_wz<1>)
    const float* ksum;   // [bh][M][D]
    float* dks;          // [bh][M][D]  dksum = Qden^T dz (split.hpp: written by the dQ kernel, read by the dK/dV kernel)
    int H, M, S, D;
    float eps;
    int relu, normalize, split;
    // fused rotary prologue in the backward (split.hpp, ROPE variants): q / k are the un-rotated tensors; the gradients of the
    // rotated ones are turned back (transposed rotation) before the normaliser's part is added and they are stored
    const float* rcos;
    const float* rsin;
    long ldr;
    long es;   // elements from one block's summary to the next in `g` / `dkv` (split.hpp / split16.hpp)
};

template <int DT>
__host__ __device__ constexpr int tok_smem_floats() {
    return DT * 16 * ld_xmajor(DT * 16) + 2 * bm_chunk(DT) * ld_xmajor(DT * 16) + bm_chunk(DT) * (DT * 16 + 4) +
           2 * bm_chunk(DT) + 2 * DT * 16;
}

template <typename T, int DT>
__global__ __launch_bounds__(NTHREADS) void k_bm_bwd_tok(const TokArgs a) {
    constexpr int DP = DT * 16, CH = bm_chunk(DT), LD = ld_xmajor(DP), LDO = DP + 4;
    constexpr int NT = ((CH / 16) * DT + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Bs = smem;              // [DP][LD]   G_j, then dKV_j
    float* T1 = Bs + DP * LD;      // [CH][LD]
    float* T2 = T1 + CH * LD;      // [CH][LD]
    float* Os = T2 + CH * LD;      // [CH][LDO]
    float* ninv = Os + CH * LDO;   // [CH]
    float* dz = ninv + CH;         // [CH]
    float* ksum = dz + CH;         // [DP]
    float* dks = ksum + DP;        // [DP]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r16 = lane & 15, kq = lane >> 4;
    const int blk = blockIdx.x, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int S = a.S, D = a.D, M = a.M;
    const long p0 = (long)blk * S;
    const bool vec_ok = (D & 3) == 0;
    auto base = [&](const View& w) { return (const T*)w.ptr + b * w.sb + h * w.sh; };
    auto mbase = [&](const MView& w) { return (T*)w.ptr + b * w.sb + h * w.sh; };

    // ---------------- phase 1: dQ (needs G_j) ----------------
    load_mat_f32<DP>(Bs, LD, a.g + ((long)bh * M + blk) * D * D, D, D, DP, D, tid, NTHREADS, vec_ok);
    if (tid < DP) ksum[tid] = (a.normalize && tid < D) ? a.ksum[((long)bh * M + blk) * D + tid] : 0.f;
    float dks_acc = 0.f;
    for (int c0 = 0; c0 < S; c0 += CH) {
        const int rv = min(CH, S - c0), rpad = (rv + 15) & ~15;
        load_tile<T, DP, false>(T1, LD, base(a.dout), a.dout.sn, a.idx, p0 + c0, rv, rpad, D, 0.f, tid, NTHREADS);
        if (a.relu) load_tile<T, DP, true>(T2, LD, base(a.qd), a.qd.sn, a.idx, p0 + c0, rv, rpad, D, a.eps, tid, NTHREADS);
        else        load_tile<T, DP, false>(T2, LD, base(a.qd), a.qd.sn, a.idx, p0 + c0, rv, rpad, D, a.eps, tid, NTHREADS);
        for (int r = tid; r < rv; r += NTHREADS) {
            ninv[r] = a.normalize ? a.ninv[((long)bh * M + blk) * S + c0 + r] : 1.f;
            dz[r] = a.normalize ? a.dz[((long)bh * M + blk) * S + c0 + r] : 0.f;
        }
        __syncthreads();
        if (a.normalize) {
            for (int v = tid; v < rv * DP; v += NTHREADS) {
                const int r = v / DP, c = v - r * DP;
                T1[r * LD + c] *= ninv[r];
            }
            if (tid < DP) {
                float s = 0.f;
                for (int r = 0; r < rv; ++r) s += dz[r] * T2[r * LD + tid];
                dks_acc += s;
            }
            __syncthreads();
        }
        f32x4 acc[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int ntiles = (rpad / 16) * DT;
        ab_accum<DT, NT, true>(acc, T1, LD, Bs, LD, ntiles, DP, wave, lane);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int t = wave + 4 * i;
            if (t < ntiles) {
                const int tm = t / DT, tn = t - tm * DT, col = tn * 16 + r16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = tm * 16 + kq * 4 + r;
                    float g = acc[i][r];
                    if (a.normalize && !a.split) g += dz[row < CH ? row : 0] * ksum[col];
                    if (a.relu && !(T2[row * LD + col] > a.eps)) g = 0.f;
                    Os[row * LDO + col] = g;
                }
            }
        }
        __syncthreads();
        store_tile<T, DP>(mbase(a.dq), a.dq.sn, a.idx, p0 + c0, Os, LDO, rv, D, tid, NTHREADS);
        if (a.normalize && a.split) {
            __syncthreads();
            for (int v = tid; v < rv * DP; v += NTHREADS) {
                const int r = v / DP, c = v - r * DP;
                Os[r * LDO + c] = dz[r] * ksum[c];
            }
            __syncthreads();
            store_tile<T, DP>(mbase(a.dqd), a.dqd.sn, a.idx, p0 + c0, Os, LDO, rv, D, tid, NTHREADS);
        }
        __syncthreads();
    }
    if (tid < DP) dks[tid] = dks_acc;

    // ---------------- phase 2: dK, dV (needs dKV_j) ----------------
    __syncthreads();
    load_mat_f32<DP>(Bs, LD, a.dkv + ((long)bh * M + blk) * D * D, D, D, DP, D, tid, NTHREADS, vec_ok);
    for (int c0 = 0; c0 < S; c0 += CH) {
        const int rv = min(CH, S - c0), rpad = (rv + 15) & ~15;
        load_tile<T, DP, false>(T1, LD, base(a.v), a.v.sn, a.idx, p0 + c0, rv, rpad, D, 0.f, tid, NTHREADS);
        if (a.relu) load_tile<T, DP, true>(T2, LD, base(a.k), a.k.sn, a.idx, p0 + c0, rv, rpad, D, a.eps, tid, NTHREADS);
        else        load_tile<T, DP, false>(T2, LD, base(a.k), a.k.sn, a.idx, p0 + c0, rv, rpad, D, a.eps, tid, NTHREADS);
        __syncthreads();
        const int ntiles = (rpad / 16) * DT;
        f32x4 acc[NT];
        // dK = V dKV^T (+ dksum)
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        ab_accum<DT, NT, true>(acc, T1, LD, Bs, LD, ntiles, DP, wave, lane);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int t = wave + 4 * i;
            if (t < ntiles) {
                const int tm = t / DT, tn = t - tm * DT, col = tn * 16 + r16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = tm * 16 + kq * 4 + r;
                    float g = acc[i][r];
                    if (a.normalize && !a.split) g += dks[col];
                    if (a.relu && !(T2[row * LD + col] > a.eps)) g = 0.f;
                    Os[row * LDO + col] = g;
                }
            }
        }
        __syncthreads();
        store_tile<T, DP>(mbase(a.dk), a.dk.sn, a.idx, p0 + c0, Os, LDO, rv, D, tid, NTHREADS);
        if (a.normalize && a.split) {
            __syncthreads();
            for (int v = tid; v < rv * DP; v += NTHREADS) {
                const int r = v / DP, c = v - r * DP;
                Os[r * LDO + c] = dks[c];
            }
            __syncthreads();
            store_tile<T, DP>(mbase(a.dkd), a.dkd.sn, a.idx, p0 + c0, Os, LDO, rv, D, tid, NTHREADS);
        }
        __syncthreads();
        // dV = K dKV
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        ab_accum<DT, NT, false>(acc, T2, LD, Bs, LD, ntiles, DP, wave, lane);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int t = wave + 4 * i;
            if (t < ntiles) {
                const int tm = t / DT, tn = t - tm * DT, col = tn * 16 + r16;
#pragma unroll
                for (int r = 0; r < 4; ++r) Os[(tm * 16 + kq * 4 + r) * LDO + col] = acc[i][r];
            }
        }
        __syncthreads();
        store_tile<T, DP>(mbase(a.dv), a.dv.sn, a.idx, p0 + c0, Os, LDO, rv, D, tid, NTHREADS);
        __syncthreads();
    }
}

}  // namespace mhla
