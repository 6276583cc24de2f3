// bf16-MFMA token kernels of the causal chunk-mixing operator (bf16 tensors, K and V multiples of 64).
// Same structure as k_cs_out / k_cs_bwd_tok (causal.hpp): every contraction is a 64 x 64 x 64 tile product, but the
// tiles live in LDS as bf16 ([64][72], row-major) and run on v_mfma_f32_16x16x32_bf16: operands whose reduction
// index is the row index of the staged tile come through the hardware transpose read.
// The chunk summaries S, P, dP, dS are kept as bf16 [bh][n][K][V] in the workspace (half the HBM traffic of the
// generic path's fp32 summaries), produced by k_csf_state (X^T Y per chunk), mixed across chunks by k_csf_mix
// (mixing weights split into bf16 hi + lo so they keep ~16 mantissa bits) and reduced to dmix by k_csf_dw.
// For bf16 inputs all products are exact and accumulate in fp32; intermediates that feed a second contraction
// (tril(QK^T), tril(dO V^T), S, P, dP, dS) carry one bf16 rounding each.
#pragma once
#include "causal.hpp"
#include "fused.hpp"

namespace mhla {
namespace fast {

constexpr int CLD = 72;                 // LDS row stride (bf16) of the 64 x 64 tiles
constexpr int CT = CS * CLD;            // elements per tile

// The chunk summaries S, P, dP, dS of this pipeline are stored tile-major, [bh][n][K / 64][V / 64][64][64]: every 64 x 64 tile
// that a kernel produces or stages is one contiguous 8 KB block (row-major [K][V] summaries made it 64 pieces of 128 B, 2 V bytes
// apart).  The mixing kernels are elementwise across chunks and do not care.
__device__ __forceinline__ long cs_tile_off(int kk0, int v0, int V) { return ((long)(kk0 >> 6) * (V >> 6) + (v0 >> 6)) * (CS * CS); }

// 64 token rows x 64 columns (starting at column c0) of a view -> LDS tile; rows >= rv zero.  256 threads.
__device__ __forceinline__ void cs_stage_tok(u16* __restrict__ dst, const u16* __restrict__ base, long sn, long p0, int rv, int tid) {
    const int r = tid >> 2, c = (tid & 3) * 16;
    uint4 x = make_uint4(0, 0, 0, 0), y = x;
    if (r < rv) {
        const u16* src = base + (p0 + r) * sn + c;
        x = *reinterpret_cast<const uint4*>(src);
        y = *reinterpret_cast<const uint4*>(src + 8);
    }
    *reinterpret_cast<uint4*>(dst + r * CLD + c) = x;
    *reinterpret_cast<uint4*>(dst + r * CLD + c + 8) = y;
}
// 64 x 64 fp32 slice (row stride ld) of a chunk summary -> bf16 LDS tile
__device__ __forceinline__ void cs_stage_state(u16* __restrict__ dst, const float* __restrict__ src, long ld, int tid) {
    const int r = tid >> 2, c = (tid & 3) * 16;
    const float* s = src + (long)r * ld + c;
    f32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(s + 4 * i);
    uint4 x, y;
    x.x = pack_bf16x2(v[0][0], v[0][1]); x.y = pack_bf16x2(v[0][2], v[0][3]);
    x.z = pack_bf16x2(v[1][0], v[1][1]); x.w = pack_bf16x2(v[1][2], v[1][3]);
    y.x = pack_bf16x2(v[2][0], v[2][1]); y.y = pack_bf16x2(v[2][2], v[2][3]);
    y.z = pack_bf16x2(v[3][0], v[3][1]); y.w = pack_bf16x2(v[3][2], v[3][3]);
    *reinterpret_cast<uint4*>(dst + r * CLD + c) = x;
    *reinterpret_cast<uint4*>(dst + r * CLD + c + 8) = y;
}

// 64 x 64 bf16 slice (row stride ld) of a chunk summary -> LDS tile
__device__ __forceinline__ void cs_stage_state(u16* __restrict__ dst, const u16* __restrict__ src, long ld, int tid) {
    const int r = tid >> 2, c = (tid & 3) * 16;
    const u16* s = src + (long)r * ld + c;
    const uint4 x = *reinterpret_cast<const uint4*>(s), y = *reinterpret_cast<const uint4*>(s + 8);
    *reinterpret_cast<uint4*>(dst + r * CLD + c) = x;
    *reinterpret_cast<uint4*>(dst + r * CLD + c + 8) = y;
}

// The two stagings in halves (loads now, LDS writes later), for kernels that fetch the next round's tiles while the current
// round is multiplied.  Rows >= rv read the chunk's first row (a valid address) and are zeroed on the way into LDS.
// A thread moves two 16-byte pieces of a tile: (row tid >> 3, columns 8 (tid & 7) ..) and the same columns 32 rows below, so
// that every load / store instruction of a wave covers eight FULL 128-byte rows (two adjacent pieces per thread made each
// instruction touch 16 bytes of every 32: half-line requests, and partial-line writes for the second instruction to complete).
struct CsTile { uint4 x, y; };
#ifndef CSF_NT_TOK
#define CSF_NT_TOK 0
#endif
#ifndef CSF_NT_STATE
#define CSF_NT_STATE 0
#endif
__device__ __forceinline__ void cs_issue_tok(CsTile& t, const u16* __restrict__ base, long sn, long p0, int rv, int tid) {
    const int r = tid >> 3, c = (tid & 7) * 8;
    const u16* s0 = base + (p0 + (r < rv ? r : 0)) * sn + c;
    const u16* s1 = base + (p0 + (r + 32 < rv ? r + 32 : 0)) * sn + c;
#if CSF_NT_TOK
    t.x = gld_stream16(s0);
    t.y = gld_stream16(s1);
#else
    t.x = gld<uint4>(s0);
    t.y = gld<uint4>(s1);
#endif
}
__device__ __forceinline__ void cs_commit_tok(u16* __restrict__ dst, const CsTile& t, int rv, int tid) {
    const int r = tid >> 3, c = (tid & 7) * 8;
    const bool ok0 = r < rv, ok1 = r + 32 < rv;
    *reinterpret_cast<uint4*>(dst + r * CLD + c) = make_uint4(ok0 ? t.x.x : 0u, ok0 ? t.x.y : 0u, ok0 ? t.x.z : 0u, ok0 ? t.x.w : 0u);
    *reinterpret_cast<uint4*>(dst + (r + 32) * CLD + c) = make_uint4(ok1 ? t.y.x : 0u, ok1 ? t.y.y : 0u, ok1 ? t.y.z : 0u, ok1 ? t.y.w : 0u);
}
__device__ __forceinline__ void cs_issue_state(CsTile& t, const u16* __restrict__ src, long ld, int tid) {
    const int r = tid >> 3, c = (tid & 7) * 8;
    const u16* s = src + (long)r * ld + c;
#if CSF_NT_STATE
    t.x = gld_stream16(s);
    t.y = gld_stream16(s + 32 * ld);
#else
    t.x = gld<uint4>(s);
    t.y = gld<uint4>(s + 32 * ld);
#endif
}
__device__ __forceinline__ void cs_commit_state(u16* __restrict__ dst, const CsTile& t, int tid) {
    const int r = tid >> 3, c = (tid & 7) * 8;
    *reinterpret_cast<uint4*>(dst + r * CLD + c) = t.x;
    *reinterpret_cast<uint4*>(dst + (r + 32) * CLD + c) = t.y;
}

// acc[tn] += A B for output rows 16 wave .. and the four 16-column tiles, reduction length 64.
//   AT false: A[m][k] = Xs[m][k]   AT true: A[m][k] = Xs[k][m]      (Xs, Ys: [64][CLD] bf16 tiles)
//   BT false: B[k][n] = Ys[n][k]   BT true: B[k][n] = Ys[k][n]
template <bool AT, bool BT>
__device__ __forceinline__ void tile_mma(f32x4 (&acc)[4], const u16* __restrict__ Xs, const u16* __restrict__ Ys, int wave, int lane) {
    const int n = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 av = AT ? tr_read8(Xs, CLD, ks * 32, wave * 16, lane)
                             : *reinterpret_cast<const bf16x8*>(Xs + (wave * 16 + n) * CLD + ks * 32 + kg * 8);
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            const bf16x8 bv = BT ? tr_read8(Ys, CLD, ks * 32, tn * 16, lane)
                                 : *reinterpret_cast<const bf16x8*>(Ys + (tn * 16 + n) * CLD + ks * 32 + kg * 8);
            acc[tn] = mfma_bf16(av, bv, acc[tn]);
        }
    }
}
__device__ __forceinline__ void zero4(f32x4 (&x)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}
// C-layout accumulators (row = 16 wave + 4 kg + r, col = 16 tn + n) -> bf16 LDS tile
__device__ __forceinline__ void cs_put(u16* __restrict__ dst, const f32x4 (&x)[4], float mul, int wave, int lane) {
    const int n = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(wave * 16 + kg * 4 + r) * CLD + tn * 16 + n] = cvt_bf16(mul * x[tn][r]);
}
__device__ __forceinline__ void cs_store_tok(u16* __restrict__ base, long sn, long p0, int rv, const u16* __restrict__ Os, int tid) {
    const int r = tid >> 3, c = (tid & 7) * 8;   // (full rows per store instruction, as the loads)
    if (r < rv) *reinterpret_cast<uint4*>(base + (p0 + r) * sn + c) = *reinterpret_cast<const uint4*>(Os + r * CLD + c);
    if (r + 32 < rv) *reinterpret_cast<uint4*>(base + (p0 + r + 32) * sn + c) = *reinterpret_cast<const uint4*>(Os + (r + 32) * CLD + c);
}

// the same with the swish gate applied on the way out: y = staged * g * sigmoid(g)   (gate rows in the output's token layout)
__device__ __forceinline__ void cs_store_tok_gate(u16* __restrict__ base, long sn, const u16* __restrict__ gbase, long gsn, long p0,
                                                  int rv, const u16* __restrict__ Os, int tid) {
    const int c = (tid & 7) * 8;
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
        const int r = (tid >> 3) + 32 * hlf;
        if (r < rv) {
            uint4 x = *reinterpret_cast<const uint4*>(Os + r * CLD + c);
            if (gbase) {
                const uint4 g = gld<uint4>(gbase + (p0 + r) * gsn + c);
                unsigned xw[4] = {x.x, x.y, x.z, x.w};
                const unsigned gw[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float g0 = __uint_as_float(gw[i] << 16), g1 = __uint_as_float(gw[i] & 0xffff0000u);
                    const float y0 = __uint_as_float(xw[i] << 16) * g0 / (1.f + __expf(-g0));
                    const float y1 = __uint_as_float(xw[i] & 0xffff0000u) * g1 / (1.f + __expf(-g1));
                    xw[i] = pack_bf16x2(y0, y1);
                }
                x = make_uint4(xw[0], xw[1], xw[2], xw[3]);
            }
            *reinterpret_cast<uint4*>(base + (p0 + r) * sn + c) = x;
        }
    }
}

constexpr int CSF_OUT_SMEM = 4 * CT * 2;

// O_i = scale (Q_i P_i + m_ii tril(Q_i K_i^T) V_i)      grid (n, bh, ceil(V / 256))
// A workgroup owns up to four 64-wide V slices of a chunk: Q_i, K_i are staged and tril(Q_i K_i^T) is computed once for all
// of them (one workgroup per slice re-read Q, K and redid the score tile per slice: 2x the HBM reads of this kernel).
constexpr int CSF_OUT_VS = 4;
// EPI: the per-head RMSNorm (over the head's V channels) x swish gate of the fla layer applied before the store; needs the
// workgroup to own every V slice of the head (V <= 256, gridDim.z == 1).  The staged normalised tile carries one bf16 rounding
// before the gate (the unfused path rounds o to bf16 first, then normalises: same order of error).
template <typename ST, bool EPI = false>   // ST: element type of the chunk summaries (u16 = bf16, float)
__global__ __launch_bounds__(NTHREADS, 2) void k_csf_out(const CsOutArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Qs = reinterpret_cast<u16*>(smem_raw);   // Q slice, later the output staging
    u16* Ks = Qs + CT;
    u16* Ps = Ks + CT;      // P slice, later the V slice
    u16* As = Ps + CT;      // m_ii tril(QK^T)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, n = lane & 15, kg = lane >> 4;
    const int ci = blockIdx.x, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int vbase = blockIdx.z * 64 * CSF_OUT_VS, nv = min(CSF_OUT_VS, (a.V - vbase) / 64);
    const long p0 = (long)ci * CS;
    const int rv = (int)min((long)CS, a.T - p0);
    const u16* qb = (const u16*)a.q.ptr + b * a.q.sb + h * a.q.sh;
    const u16* kb = (const u16*)a.k.ptr + b * a.k.sb + h * a.k.sh;
    const u16* vb = (const u16*)a.v.ptr + b * a.v.sb + h * a.v.sh;
    u16* ob = (u16*)a.o.ptr + b * a.o.sb + h * a.o.sh;
    const ST* Pi = reinterpret_cast<const ST*>(a.P) + ((long)bh * a.n + ci) * a.K * a.V;

    f32x4 accO[CSF_OUT_VS][4], accA[4];
#pragma unroll
    for (int j = 0; j < CSF_OUT_VS; ++j) zero4(accO[j]);
    zero4(accA);
    // Every tile (the K slice's Q and K rows, each P slice, later each V slice) is requested a step ahead into registers and
    // written to LDS behind the barrier that frees its slot: the products never wait for a load that was only just issued.
    static_assert(sizeof(ST) == 2, "k_csf_out expects bf16 summaries");
    const u16* Pb = reinterpret_cast<const u16*>(Pi);
    CsTile nQ, nK, nP, nVt;
    auto issueP = [&](int ks, int j) { cs_issue_state(nP, Pb + cs_tile_off(ks, vbase + 64 * j, a.V), CS, tid); };
    cs_issue_tok(nQ, qb, a.q.sn, p0, rv, tid);
    cs_issue_tok(nK, kb, a.k.sn, p0, rv, tid);
    issueP(0, 0);
    for (int ks = 0; ks < a.K; ks += 64) {
        cs_commit_tok(Qs, nQ, rv, tid);
        cs_commit_tok(Ks, nK, rv, tid);
        cs_commit_state(Ps, nP, tid);
        __syncthreads();
        const bool more_k = ks + 64 < a.K;   // (uniform)
        if (nv > 1) issueP(ks, 1);
        else if (more_k) issueP(ks + 64, 0);
        if (more_k) {
            cs_issue_tok(nQ, qb + ks + 64, a.q.sn, p0, rv, tid);
            cs_issue_tok(nK, kb + ks + 64, a.k.sn, p0, rv, tid);
        } else {
            cs_issue_tok(nVt, vb + vbase, a.v.sn, p0, rv, tid);   // the first V slice of the second phase
        }
        tile_mma<false, false>(accA, Qs, Ks, wave, lane);   // Q K^T
#pragma unroll
        for (int j = 0; j < CSF_OUT_VS; ++j) {
            if (j < nv) {
                tile_mma<false, true>(accO[j], Qs, Ps, wave, lane);    // Q P
                __syncthreads();
                if (j + 1 < nv) {
                    cs_commit_state(Ps, nP, tid);
                    if (j + 2 < nv) issueP(ks, j + 2);
                    else if (more_k) issueP(ks + 64, 0);
                    __syncthreads();
                }
            }
        }
    }
    const float mii = a.mix[(long)ci * a.ldmix + ci];
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wave * 16 + kg * 4 + r, col = tn * 16 + n;
            As[row * CLD + col] = cvt_bf16(col <= row ? mii * accA[tn][r] : 0.f);
        }
    if constexpr (!EPI) {
#pragma unroll
        for (int j = 0; j < CSF_OUT_VS; ++j) {
            if (j < nv) {
                cs_commit_tok(Ps, nVt, rv, tid);
                __syncthreads();
                if (j + 1 < nv) cs_issue_tok(nVt, vb + vbase + 64 * (j + 1), a.v.sn, p0, rv, tid);
                tile_mma<false, true>(accO[j], As, Ps, wave, lane);        // tril(QK^T) V
                cs_put(Qs, accO[j], a.scale, wave, lane);
                __syncthreads();
                cs_store_tok(ob + vbase + 64 * j, a.o.sn, p0, rv, Qs, tid);
            }
        }
    } else {
        __syncthreads();                                                   // As complete
#pragma unroll
        for (int j = 0; j < CSF_OUT_VS; ++j) {
            if (j < nv) {
                cs_commit_tok(Ps, nVt, rv, tid);
                __syncthreads();
                if (j + 1 < nv) cs_issue_tok(nVt, vb + vbase + 64 * (j + 1), a.v.sn, p0, rv, tid);
                tile_mma<false, true>(accO[j], As, Ps, wave, lane);        // tril(QK^T) V
                __syncthreads();
            }
        }
        // row sums of squares over the head's V channels: lane holds rows 16 wave + 4 kg + r, columns 64 j + 16 tn + n
        float ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < CSF_OUT_VS; ++j)
            if (j < nv)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float x = a.scale * accO[j][tn][r];
                        ss[r] += x * x;
                    }
        u16* yb = (u16*)a.y.ptr + b * a.y.sb + h * a.y.sh;
        const u16* gb = a.gate.ptr ? (const u16*)a.gate.ptr + b * a.gate.sb + h * a.gate.sh : nullptr;
        float rstd[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = ss[r];
            x += __shfl_xor(x, 1, 64); x += __shfl_xor(x, 2, 64); x += __shfl_xor(x, 4, 64); x += __shfl_xor(x, 8, 64);
            rstd[r] = rsqrtf(x / (float)a.V + a.neps);
        }
#pragma unroll
        for (int j = 0; j < CSF_OUT_VS; ++j) {
            if (j < nv) {
                if (a.o.ptr) {   // training: the operator's own output is kept for the norm's backward
                    cs_put(Qs, accO[j], a.scale, wave, lane);
                    __syncthreads();
                    cs_store_tok(ob + vbase + 64 * j, a.o.sn, p0, rv, Qs, tid);
                    __syncthreads();
                }
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) {
                    const float w = a.nw ? gld<float>(a.nw + vbase + 64 * j + tn * 16 + n) : 1.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) accO[j][tn][r] *= a.scale * rstd[r] * w;
                }
                cs_put(Qs, accO[j], 1.f, wave, lane);
                __syncthreads();
                cs_store_tok_gate(yb + vbase + 64 * j, a.y.sn, gb ? gb + vbase + 64 * j : nullptr, a.gate.sn, p0, rv, Qs, tid);
                __syncthreads();
            }
        }
    }
}

constexpr int CSF_TOK_SMEM = 6 * CT * 2 + 16;

template <typename ST>
__global__ __launch_bounds__(NTHREADS) void k_csf_bwd_tok(const CsTokArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* As = reinterpret_cast<u16*>(smem_raw);   // m_ii scale tril(Q K^T)   [c][c']
    u16* dAs = As + CT;                           // m_ii tril(dO V^T)        [c][c']
    u16* X1 = dAs + CT;
    u16* X2 = X1 + CT;
    u16* B1 = X2 + CT;
    u16* B2 = B1 + CT;
    float* red = reinterpret_cast<float*>(B2 + CT);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, n = lane & 15, kg = lane >> 4;
    const int ci = blockIdx.x, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const long p0 = (long)ci * CS;
    const int rv = (int)min((long)CS, a.T - p0);
    const int K = a.K, V = a.V;
    auto base = [&](const View& w) { return (const u16*)w.ptr + b * w.sb + h * w.sh; };
    auto mbase = [&](const MView& w) { return (u16*)w.ptr + b * w.sb + h * w.sh; };
    const u16 *qb = base(a.q), *kb = base(a.k), *vb = base(a.v), *gb = base(a.dout);
    const ST* Pi = reinterpret_cast<const ST*>(a.P) + ((long)bh * a.n + ci) * K * V;
    const ST* dSi = reinterpret_cast<const ST*>(a.dS) + ((long)bh * a.n + ci) * K * V;
    const float mii = a.mix[(long)ci * a.ldmix + ci];

    // ---- step 1: A = tril(Q K^T), dA = tril(dO V^T), diag = scale * sum(A . dA) ----
    f32x4 acc1[4], acc2[4];
    zero4(acc1);
    zero4(acc2);
    for (int ks = 0; ks < K; ks += 64) {
        cs_stage_tok(X1, qb + ks, a.q.sn, p0, rv, tid);
        cs_stage_tok(X2, kb + ks, a.k.sn, p0, rv, tid);
        __syncthreads();
        tile_mma<false, false>(acc1, X1, X2, wave, lane);
        __syncthreads();
    }
    for (int vs = 0; vs < V; vs += 64) {
        cs_stage_tok(X1, gb + vs, a.dout.sn, p0, rv, tid);
        cs_stage_tok(X2, vb + vs, a.v.sn, p0, rv, tid);
        __syncthreads();
        tile_mma<false, false>(acc2, X1, X2, wave, lane);
        __syncthreads();
    }
    float dsum = 0.f;
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wave * 16 + kg * 4 + r, col = tn * 16 + n;
            const bool keep = col <= row;
            const float av = keep ? acc1[tn][r] : 0.f, dv = keep ? acc2[tn][r] : 0.f;
            dsum += av * dv;
            As[row * CLD + col] = cvt_bf16(mii * a.scale * av);
            dAs[row * CLD + col] = cvt_bf16(mii * dv);
        }
    dsum = wave_sum(dsum);
    if (lane == 0) red[wave] = dsum;
    __syncthreads();
    if (tid == 0) a.diag[(long)bh * a.n + ci] = a.scale * (red[0] + red[1] + red[2] + red[3]);

    // ---- step 2: dQ, dK per K slice ----
    for (int ks = 0; ks < K; ks += 64) {
        f32x4 acc3[4];
        zero4(acc1);   // dO P^T + m_ii dA K
        zero4(acc2);   // V dS^T
        zero4(acc3);   // m_ii dA^T Q
        for (int vs = 0; vs < V; vs += 64) {
            cs_stage_tok(X1, gb + vs, a.dout.sn, p0, rv, tid);
            cs_stage_tok(X2, vb + vs, a.v.sn, p0, rv, tid);
            cs_stage_state(B1, Pi + cs_tile_off(ks, vs, V), CS, tid);
            cs_stage_state(B2, dSi + cs_tile_off(ks, vs, V), CS, tid);
            __syncthreads();
            tile_mma<false, false>(acc1, X1, B1, wave, lane);   // dO P^T : B[k = v][n = kk] = P[kk][v]
            tile_mma<false, false>(acc2, X2, B2, wave, lane);   // V dS^T
            __syncthreads();
        }
        cs_stage_tok(X1, kb + ks, a.k.sn, p0, rv, tid);
        cs_stage_tok(X2, qb + ks, a.q.sn, p0, rv, tid);
        __syncthreads();
        tile_mma<false, true>(acc1, dAs, X1, wave, lane);       // dA K      : B[k = c'][n = kk] = K[c'][kk]
        tile_mma<true, true>(acc3, dAs, X2, wave, lane);        // dA^T Q    : A[m = c'][k = c] = dA[c][c']
        __syncthreads();
        cs_put(B1, acc1, a.scale, wave, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc2[i] += a.scale * acc3[i];
        cs_put(B2, acc2, 1.f, wave, lane);
        __syncthreads();
        cs_store_tok(mbase(a.dq) + ks, a.dq.sn, p0, rv, B1, tid);
        cs_store_tok(mbase(a.dk) + ks, a.dk.sn, p0, rv, B2, tid);
        __syncthreads();
    }

    // ---- step 3: dV per V slice ----
    for (int vs = 0; vs < V; vs += 64) {
        zero4(acc1);
        for (int ks = 0; ks < K; ks += 64) {
            cs_stage_tok(X1, kb + ks, a.k.sn, p0, rv, tid);
            cs_stage_state(B1, dSi + cs_tile_off(ks, vs, V), CS, tid);
            __syncthreads();
            tile_mma<false, true>(acc1, X1, B1, wave, lane);    // K dS : B[k = kk][n = v] = dS[kk][v]
            __syncthreads();
        }
        cs_stage_tok(X2, gb + vs, a.dout.sn, p0, rv, tid);
        __syncthreads();
        tile_mma<true, true>(acc1, As, X2, wave, lane);         // A^T dO
        __syncthreads();
        cs_put(B1, acc1, 1.f, wave, lane);
        __syncthreads();
        cs_store_tok(mbase(a.dv) + vs, a.dv.sn, p0, rv, B1, tid);
        __syncthreads();
    }
}


// The same token gradients with dV accumulated alongside dQ / dK: the K slice of the outer loop stays in LDS and every staged
// dS slice also feeds dV[v-slice] += K dS, so dS is read once (not twice) and K is not re-read per V slice.  V / 64 <= NVMAX
// accumulator sets live in registers.
constexpr int CSF_TOK2_SMEM = 7 * CT * 2 + 16;

template <typename ST, int NVMAX>
__global__ __launch_bounds__(NTHREADS) void k_csf_bwd_tok2(const CsTokArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* As = reinterpret_cast<u16*>(smem_raw);   // m_ii scale tril(Q K^T)   [c][c']
    u16* dAs = As + CT;                           // m_ii tril(dO V^T)        [c][c']
    u16* X1 = dAs + CT;
    u16* X2 = X1 + CT;
    u16* B1 = X2 + CT;
    u16* B2 = B1 + CT;
    u16* X3 = B2 + CT;                            // K slice of the outer loop
    float* red = reinterpret_cast<float*>(X3 + CT);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, n = lane & 15, kg = lane >> 4;
    const int ci = blockIdx.x, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const long p0 = (long)ci * CS;
    const int rv = (int)min((long)CS, a.T - p0);
    const int K = a.K, V = a.V, nvs = V / 64;
    auto base = [&](const View& w) { return (const u16*)w.ptr + b * w.sb + h * w.sh; };
    auto mbase = [&](const MView& w) { return (u16*)w.ptr + b * w.sb + h * w.sh; };
    const u16 *qb = base(a.q), *kb = base(a.k), *vb = base(a.v), *gb = base(a.dout);
    const ST* Pi = reinterpret_cast<const ST*>(a.P) + ((long)bh * a.n + ci) * K * V;
    const ST* dSi = reinterpret_cast<const ST*>(a.dS) + ((long)bh * a.n + ci) * K * V;
    const float mii = a.mix[(long)ci * a.ldmix + ci];

    // ---- step 1: A = tril(Q K^T), dA = tril(dO V^T), diag = scale * sum(A . dA) ----
    f32x4 acc1[4], acc2[4];
    zero4(acc1);
    zero4(acc2);
    for (int ks = 0; ks < K; ks += 64) {
        cs_stage_tok(X1, qb + ks, a.q.sn, p0, rv, tid);
        cs_stage_tok(X2, kb + ks, a.k.sn, p0, rv, tid);
        __syncthreads();
        tile_mma<false, false>(acc1, X1, X2, wave, lane);
        __syncthreads();
    }
    for (int vs = 0; vs < V; vs += 64) {
        cs_stage_tok(X1, gb + vs, a.dout.sn, p0, rv, tid);
        cs_stage_tok(X2, vb + vs, a.v.sn, p0, rv, tid);
        __syncthreads();
        tile_mma<false, false>(acc2, X1, X2, wave, lane);
        __syncthreads();
    }
    float dsum = 0.f;
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wave * 16 + kg * 4 + r, col = tn * 16 + n;
            const bool keep = col <= row;
            const float av = keep ? acc1[tn][r] : 0.f, dv = keep ? acc2[tn][r] : 0.f;
            dsum += av * dv;
            As[row * CLD + col] = cvt_bf16(mii * a.scale * av);
            dAs[row * CLD + col] = cvt_bf16(mii * dv);
        }
    dsum = wave_sum(dsum);
    if (lane == 0) red[wave] = dsum;
    __syncthreads();
    if (tid == 0) a.diag[(long)bh * a.n + ci] = a.scale * (red[0] + red[1] + red[2] + red[3]);

    // ---- step 2: dQ, dK per K slice; dV += K dS on the way ----
    f32x4 accV[NVMAX][4];
#pragma unroll
    for (int j = 0; j < NVMAX; ++j) zero4(accV[j]);
    for (int ks = 0; ks < K; ks += 64) {
        f32x4 acc3[4];
        zero4(acc1);   // dO P^T + m_ii dA K
        zero4(acc2);   // V dS^T
        zero4(acc3);   // m_ii dA^T Q
        cs_stage_tok(X3, kb + ks, a.k.sn, p0, rv, tid);
#pragma unroll
        for (int j = 0; j < NVMAX; ++j) {
            if (j < nvs) {
                const int vs = j * 64;
                cs_stage_tok(X1, gb + vs, a.dout.sn, p0, rv, tid);
                cs_stage_tok(X2, vb + vs, a.v.sn, p0, rv, tid);
                cs_stage_state(B1, Pi + cs_tile_off(ks, vs, V), CS, tid);
                cs_stage_state(B2, dSi + cs_tile_off(ks, vs, V), CS, tid);
                __syncthreads();
                tile_mma<false, false>(acc1, X1, B1, wave, lane);       // dO P^T
                tile_mma<false, false>(acc2, X2, B2, wave, lane);       // V dS^T
                tile_mma<false, true>(accV[j], X3, B2, wave, lane);     // K dS : B[k = kk][n = v] = dS[kk][v]
                __syncthreads();
            }
        }
        cs_stage_tok(X2, qb + ks, a.q.sn, p0, rv, tid);
        __syncthreads();
        tile_mma<false, true>(acc1, dAs, X3, wave, lane);       // dA K
        tile_mma<true, true>(acc3, dAs, X2, wave, lane);        // dA^T Q
        __syncthreads();
        cs_put(B1, acc1, a.scale, wave, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc2[i] += a.scale * acc3[i];
        cs_put(B2, acc2, 1.f, wave, lane);
        __syncthreads();
        cs_store_tok(mbase(a.dq) + ks, a.dq.sn, p0, rv, B1, tid);
        cs_store_tok(mbase(a.dk) + ks, a.dk.sn, p0, rv, B2, tid);
        __syncthreads();
    }

    // ---- step 3: dV += A^T dO per V slice ----
#pragma unroll
    for (int j = 0; j < NVMAX; ++j) {
        if (j < nvs) {
            cs_stage_tok(X2, gb + j * 64, a.dout.sn, p0, rv, tid);
            __syncthreads();
            tile_mma<true, true>(accV[j], As, X2, wave, lane);     // A^T dO
            cs_put(B1, accV[j], 1.f, wave, lane);
            __syncthreads();
            cs_store_tok(mbase(a.dv) + j * 64, a.dv.sn, p0, rv, B1, tid);
        }
    }
}

// Token gradients with the V slices in the outer loop (K <= 64 NK): the chunk's K tiles stay in LDS, dQ / dK accumulate in
// registers per K slice, and every V slice stages dO, V, P and dS once and feeds all of dA += dO V^T, dV = A^T dO + K dS,
// dQ += dO P^T, dK += V dS^T from them.  Per chunk the kernel reads Q twice and everything else once (k_csf_bwd_tok2 re-reads
// dO and V per K slice and again for the score tiles: 416 KB instead of 240 KB per chunk at K = 128, V = 256), and V is not
// limited by an accumulator count.
template <int NK>
__host__ __device__ constexpr int csf_tok3_smem() { return (6 + NK) * CT * 2 + 16; }

template <typename ST, int NK>
__global__ __launch_bounds__(NTHREADS, NK <= 2 ? 2 : 1) void k_csf_bwd_tok3(const CsTokArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* As = reinterpret_cast<u16*>(smem_raw);   // m_ii scale tril(Q K^T)   [c][c']
    u16* dAs = As + CT;                           // m_ii tril(dO V^T)        [c][c']
    u16* X1 = dAs + CT;                           // Q slice / dO slice
    u16* X2 = X1 + CT;                            // V slice / Q slice
    u16* B1 = X2 + CT;                            // P slice, output staging
    u16* B2 = B1 + CT;                            // dS slice, output staging
    u16* KT = B2 + CT;                            // the chunk's K tiles [NK]
    float* red = reinterpret_cast<float*>(KT + NK * CT);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, n = lane & 15, kg = lane >> 4;
    const int ci = blockIdx.x, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const long p0 = (long)ci * CS;
    const int rv = (int)min((long)CS, a.T - p0);
    const int K = a.K, V = a.V, nks = K / 64;
    auto base = [&](const View& w) { return (const u16*)w.ptr + b * w.sb + h * w.sh; };
    auto mbase = [&](const MView& w) { return (u16*)w.ptr + b * w.sb + h * w.sh; };
    const u16 *qb = base(a.q), *kb = base(a.k), *vb = base(a.v), *gb = base(a.dout);
    const ST* Pi = reinterpret_cast<const ST*>(a.P) + ((long)bh * a.n + ci) * K * V;
    const ST* dSi = reinterpret_cast<const ST*>(a.dS) + ((long)bh * a.n + ci) * K * V;
    const float mii = a.mix[(long)ci * a.ldmix + ci];

    // A ring of NK register slots per summary set keeps NK (V slice, K slice) rounds of P / dS tiles in flight: the slot a round
    // commits to LDS is refilled with the same K slice of the NEXT V slice as soon as the barrier behind the commit is passed, so
    // NK x 16 KB per workgroup travel while a round multiplies (one round ahead left this kernel waiting for memory in every one of
    // its 2 NK V / 64 rounds: 61 us per wave at K = 256, V = 512).  Before step 1 the same slots carry the chunk's Q and K tiles
    // (all requested up front), and behind the last V slice the Q tiles again for step 3 -- no staging load is ever waited for
    // right after it was issued.
    static_assert(sizeof(ST) == 2, "the prefetching token kernel expects bf16 summaries");
    const u16* Pb = reinterpret_cast<const u16*>(Pi);
    const u16* dSb = reinterpret_cast<const u16*>(dSi);
    CsTile rP[NK], rdS[NK];
    CsTile nG, nV;   // next V slice's dO / V rows
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
        const int kc = kk < nks ? kk : 0;   // (K < 64 NK: the surplus slots repeat the first tile, never committed)
        cs_issue_tok(rP[kk], qb + kc * 64, a.q.sn, p0, rv, tid);
        cs_issue_tok(rdS[kk], kb + kc * 64, a.k.sn, p0, rv, tid);
    }
    cs_issue_tok(nG, gb, a.dout.sn, p0, rv, tid);
    cs_issue_tok(nV, vb, a.v.sn, p0, rv, tid);
    // ---- step 1: A = tril(Q K^T); the K tiles stay ----
    f32x4 accA[4];
    zero4(accA);
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
        if (kk < nks) {
            cs_commit_tok(X1, rP[kk], rv, tid);
            cs_commit_tok(KT + kk * CT, rdS[kk], rv, tid);
            __syncthreads();
            cs_issue_state(rP[kk], Pb + cs_tile_off(kk * 64, 0, V), CS, tid);
            cs_issue_state(rdS[kk], dSb + cs_tile_off(kk * 64, 0, V), CS, tid);
            tile_mma<false, false>(accA, X1, KT + kk * CT, wave, lane);
            __syncthreads();
        }
    }
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wave * 16 + kg * 4 + r, col = tn * 16 + n;
            const float av = col <= row ? accA[tn][r] : 0.f;
            accA[tn][r] = av;   // kept for the diagonal term
            As[row * CLD + col] = cvt_bf16(mii * a.scale * av);
        }

    // ---- step 2: per V slice: dA, dV, and the dQ / dK partials of every K slice ----
    f32x4 accQ[NK][4], accK[NK][4], accdA[4];
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
        zero4(accQ[kk]);
        zero4(accK[kk]);
    }
    zero4(accdA);
    for (int vs = 0; vs < V; vs += 64) {
        f32x4 accV[4];
        zero4(accV);
        const bool last = vs + 64 >= V;   // (uniform)
        cs_commit_tok(X1, nG, rv, tid);
        cs_commit_tok(X2, nV, rv, tid);
        if (!last) {   // the next V slice's rows travel during this slice's rounds
            cs_issue_tok(nG, gb + vs + 64, a.dout.sn, p0, rv, tid);
            cs_issue_tok(nV, vb + vs + 64, a.v.sn, p0, rv, tid);
        }
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            if (kk < nks) {
                cs_commit_state(B1, rP[kk], tid);
                cs_commit_state(B2, rdS[kk], tid);
                __syncthreads();
                if (!last) {
                    cs_issue_state(rP[kk], Pb + cs_tile_off(kk * 64, vs + 64, V), CS, tid);
                    cs_issue_state(rdS[kk], dSb + cs_tile_off(kk * 64, vs + 64, V), CS, tid);
                } else {
                    cs_issue_tok(rP[kk], qb + kk * 64, a.q.sn, p0, rv, tid);   // step 3's Q tile
                }
                if (kk == 0) {
                    tile_mma<false, false>(accdA, X1, X2, wave, lane);      // dO V^T
                    tile_mma<true, true>(accV, As, X1, wave, lane);         // A^T dO
                }
                tile_mma<false, false>(accQ[kk], X1, B1, wave, lane);       // dO P^T
                tile_mma<false, false>(accK[kk], X2, B2, wave, lane);       // V dS^T
                tile_mma<false, true>(accV, KT + kk * CT, B2, wave, lane);  // K dS : B[k = kk][n = v] = dS[kk][v]
                __syncthreads();
            }
        }
        cs_put(B1, accV, 1.f, wave, lane);
        __syncthreads();
        cs_store_tok(mbase(a.dv) + vs, a.dv.sn, p0, rv, B1, tid);
        __syncthreads();
    }

    // ---- step 3: dA tile, diagonal term, the m_ii parts of dQ / dK ----
    float dsum = 0.f;
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wave * 16 + kg * 4 + r, col = tn * 16 + n;
            const float dv = col <= row ? accdA[tn][r] : 0.f;
            dsum += accA[tn][r] * dv;
            dAs[row * CLD + col] = cvt_bf16(mii * dv);
        }
    dsum = wave_sum(dsum);
    if (lane == 0) red[wave] = dsum;
    __syncthreads();
    if (tid == 0) a.diag[(long)bh * a.n + ci] = a.scale * (red[0] + red[1] + red[2] + red[3]);
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
        if (kk < nks) {
            cs_commit_tok(X2, rP[kk], rv, tid);
            __syncthreads();
            f32x4 acc3[4];
            zero4(acc3);
            tile_mma<false, true>(accQ[kk], dAs, KT + kk * CT, wave, lane);   // dA K
            tile_mma<true, true>(acc3, dAs, X2, wave, lane);                  // dA^T Q
            cs_put(B1, accQ[kk], a.scale, wave, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) accK[kk][i] += a.scale * acc3[i];
            cs_put(B2, accK[kk], 1.f, wave, lane);
            __syncthreads();
            cs_store_tok(mbase(a.dq) + kk * 64, a.dq.sn, p0, rv, B1, tid);
            cs_store_tok(mbase(a.dk) + kk * 64, a.dk.sn, p0, rv, B2, tid);
            __syncthreads();
        }
    }
}

// k_csf_bwd_tok4: the same algorithm as k_csf_bwd_tok3 on EIGHT waves -- every wave owns 16 rows x 32 columns of each 64 x 64 product
// (row tile = wave & 3, column half = wave >> 2) instead of 16 x 64.  The accumulator set per wave halves (K = 256: 248 -> ~150
// VGPRs), so two waves share a SIMD and one's LDS operand reads run under the other's MFMAs: with four waves of 248 VGPRs the
// K = 256 kernel had ONE wave per SIMD and spent its rounds in exposed LDS and MFMA latency (11 % MFMA-busy, 61 us per wave; a
// deeper prefetch ring alone moved it from 286 to 265 us).  At K <= 128 the kernel fits 128 VGPRs: two workgroups = 16 waves per CU.
// Every thread moves 16 bytes of a tile (row = tid >> 3), so a ring slot is one uint4.
constexpr int NT4 = 512;
__device__ __forceinline__ void cs8_issue_tok(uint4& t, const u16* __restrict__ base, long sn, long p0, int rv, int tid) {
    const int r = tid >> 3, c = (tid & 7) * 8;
    t = gld<uint4>(base + (p0 + (r < rv ? r : 0)) * sn + c);
}
__device__ __forceinline__ void cs8_commit_tok(u16* __restrict__ dst, const uint4& t, int rv, int tid) {
    const int r = tid >> 3, c = (tid & 7) * 8;
    const bool ok = r < rv;
    *reinterpret_cast<uint4*>(dst + r * CLD + c) = make_uint4(ok ? t.x : 0u, ok ? t.y : 0u, ok ? t.z : 0u, ok ? t.w : 0u);
}
__device__ __forceinline__ void cs8_issue_state(uint4& t, const u16* __restrict__ tile, int tid) { t = gld<uint4>(tile + tid * 8); }   // [64][64] contiguous
__device__ __forceinline__ void cs8_commit_state(u16* __restrict__ dst, const uint4& t, int tid) {
    *reinterpret_cast<uint4*>(dst + (tid >> 3) * CLD + (tid & 7) * 8) = t;
}
#ifndef TOK4_NO_STORE
#define TOK4_NO_STORE 0
#endif
#ifndef TOK4_NO_TOKLOAD
#define TOK4_NO_TOKLOAD 0
#endif
__device__ __forceinline__ void cs8_store_tok(u16* __restrict__ base, long sn, long p0, int rv, const u16* __restrict__ Os, int tid) {
    const int r = tid >> 3, c = (tid & 7) * 8;
#if TOK4_NO_STORE
    if (p0 >= 0) return;
#endif
    if (r < rv) *reinterpret_cast<uint4*>(base + (p0 + r) * sn + c) = *reinterpret_cast<const uint4*>(Os + r * CLD + c);
}
// acc[tn] += A B for output rows 16 rt .. and columns 32 ch + 16 tn ..   (operand conventions of tile_mma)
#ifndef TOK4_NO_MMA
#define TOK4_NO_MMA 0    // ablation switches (tools/build_variant.sh): 1 skips the tile products, TOK4_NO_LOAD the round loads
#endif
#ifndef TOK4_NO_LOAD
#define TOK4_NO_LOAD 0
#endif
template <bool AT, bool BT>
__device__ __forceinline__ void tile_mma8(f32x4 (&acc)[2], const u16* __restrict__ Xs, const u16* __restrict__ Ys, int rt, int ch, int lane) {
    const int n = lane & 15, kg = lane >> 4;
#if TOK4_NO_MMA
    if (Xs != nullptr) return;
#endif
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 av = AT ? tr_read8(Xs, CLD, ks * 32, rt * 16, lane)
                             : *reinterpret_cast<const bf16x8*>(Xs + (rt * 16 + n) * CLD + ks * 32 + kg * 8);
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int c0 = ch * 32 + tn * 16;
            const bf16x8 bv = BT ? tr_read8(Ys, CLD, ks * 32, c0, lane)
                                 : *reinterpret_cast<const bf16x8*>(Ys + (c0 + n) * CLD + ks * 32 + kg * 8);
            acc[tn] = mfma_bf16(av, bv, acc[tn]);
        }
    }
}
// the same with the A operand (the wave's 16 rows x 64 reduction columns) already in registers: operands that several rounds
// share are read from LDS once
__device__ __forceinline__ void tile_a8(bf16x8 (&av)[2], const u16* __restrict__ Xs, int rt, int lane) {
    const int n = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) av[ks] = *reinterpret_cast<const bf16x8*>(Xs + (rt * 16 + n) * CLD + ks * 32 + kg * 8);
}
template <bool BT>
__device__ __forceinline__ void tile_mma8r(f32x4 (&acc)[2], const bf16x8 (&av)[2], const u16* __restrict__ Ys, int ch, int lane) {
    const int n = lane & 15, kg = lane >> 4;
#if TOK4_NO_MMA
    if (Ys != nullptr) return;
#endif
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int c0 = ch * 32 + tn * 16;
            const bf16x8 bv = BT ? tr_read8(Ys, CLD, ks * 32, c0, lane)
                                 : *reinterpret_cast<const bf16x8*>(Ys + (c0 + n) * CLD + ks * 32 + kg * 8);
            acc[tn] = mfma_bf16(av[ks], bv, acc[tn]);
        }
}
__device__ __forceinline__ void zero2(f32x4 (&x)[2]) { x[0] = x[1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ void cs8_put(u16* __restrict__ dst, const f32x4 (&x)[2], float mul, int rt, int ch, int lane) {
    const int n = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(rt * 16 + kg * 4 + r) * CLD + ch * 32 + tn * 16 + n] = cvt_bf16(mul * x[tn][r]);
}

template <int NK>
__host__ __device__ constexpr int csf_tok4_smem() { return (6 + NK + (NK > 2 ? 3 : 0)) * CT * 2 + 32; }   // K > 128: second P / dS buffers + dV staging

template <typename ST, int NK>
__global__ __launch_bounds__(NT4, NK <= 2 ? 4 : 2) void k_csf_bwd_tok4(const CsTokArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* As = reinterpret_cast<u16*>(smem_raw);   // m_ii scale tril(Q K^T)   [c][c']
    u16* dAs = As + CT;                           // m_ii tril(dO V^T)        [c][c']
    u16* X1 = dAs + CT;                           // Q slice / dO slice
    u16* X2 = X1 + CT;                            // V slice / Q slice
    u16* B1 = X2 + CT;                            // P slice, output staging
    u16* B2 = B1 + CT;                            // dS slice, output staging
    u16* KT = B2 + CT;                            // the chunk's K tiles [NK]
    // K > 128 (one workgroup per CU whatever the LDS use): a second pair of P / dS buffers and a dV staging tile of its own, so
    // that a round is commit -> ONE barrier -> refill -> multiply (the products of a round run beside the next round's commit)
    // and a V slice ends with one barrier instead of two: 60 barriers per chunk instead of 100 at K = 256, V = 512.
    constexpr bool DBUF = NK > 2;
    u16* Bx = KT + NK * CT;                       // DBUF: [B1', B2', dV staging]
    float* red = reinterpret_cast<float*>(Bx + (DBUF ? 3 : 0) * CT);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, n = lane & 15, kg = lane >> 4;
    const int rt = wave & 3, ch = wave >> 2;
    const int ci = blockIdx.x, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const long p0 = (long)ci * CS;
    const int rv = (int)min((long)CS, a.T - p0);
    const int K = a.K, V = a.V;   // K == 64 NK
    auto base = [&](const View& w) { return (const u16*)w.ptr + b * w.sb + h * w.sh; };
    auto mbase = [&](const MView& w) { return (u16*)w.ptr + b * w.sb + h * w.sh; };
    const u16 *qb = base(a.q), *kb = base(a.k), *vb = base(a.v), *gb = base(a.dout);
    static_assert(sizeof(ST) == 2, "the prefetching token kernel expects bf16 summaries");
    const u16* Pb = reinterpret_cast<const u16*>(a.P) + ((long)bh * a.n + ci) * K * V;
    const u16* dSb = reinterpret_cast<const u16*>(a.dS) + ((long)bh * a.n + ci) * K * V;
    const float mii = a.mix[(long)ci * a.ldmix + ci];

    // ring of NK register slots per summary set (see k_csf_bwd_tok3): Q / K tiles first, then P / dS a V slice ahead, then Q again
    uint4 rP[NK], rdS[NK], nG, nV;
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
        cs8_issue_tok(rP[kk], qb + kk * 64, a.q.sn, p0, rv, tid);
        cs8_issue_tok(rdS[kk], kb + kk * 64, a.k.sn, p0, rv, tid);
    }
    cs8_issue_tok(nG, gb, a.dout.sn, p0, rv, tid);
    cs8_issue_tok(nV, vb, a.v.sn, p0, rv, tid);
    // ---- step 1: A = tril(Q K^T); the K tiles stay ----
    f32x4 accA[2];
    zero2(accA);
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
        {
            cs8_commit_tok(X1, rP[kk], rv, tid);
            cs8_commit_tok(KT + kk * CT, rdS[kk], rv, tid);
            __syncthreads();
            cs8_issue_state(rP[kk], Pb + cs_tile_off(kk * 64, 0, V), tid);
            cs8_issue_state(rdS[kk], dSb + cs_tile_off(kk * 64, 0, V), tid);
            tile_mma8<false, false>(accA, X1, KT + kk * CT, rt, ch, lane);
            __syncthreads();
        }
    }
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rt * 16 + kg * 4 + r, col = ch * 32 + tn * 16 + n;
            const float av = col <= row ? accA[tn][r] : 0.f;
            accA[tn][r] = av;   // kept for the diagonal term
            As[row * CLD + col] = cvt_bf16(mii * a.scale * av);
        }

    // ---- step 2: per V slice: dA, dV, and the dQ / dK partials of every K slice ----
    // The wave's rows of the K tiles are the A operand of K dS in every round of every V slice, its rows of dO and V of all rounds
    // of one V slice: they are read from LDS once (the kernel is bound by LDS operand traffic: 18 KB per wave and round before).
    constexpr bool KREG = NK > 2;   // (K <= 128 runs two workgroups per CU on 128 VGPRs: no room for the K rows)
    bf16x8 aK[KREG ? NK : 1][2];
    if constexpr (KREG) {
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) tile_a8(aK[kk], KT + kk * CT, rt, lane);
    }
    f32x4 accQ[NK][2], accK[NK][2], accdA[2];
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
        zero2(accQ[kk]);
        zero2(accK[kk]);
    }
    zero2(accdA);
    int rr = 0;   // round counter (DBUF: its parity picks the P / dS buffers)
    for (int vs = 0; vs < V; vs += 64) {
        f32x4 accV[2];
        zero2(accV);
        bf16x8 aG[2], aV[2];
        const bool last = vs + 64 >= V;   // (uniform)
        // (DBUF: the wave's dO / V rows were read into registers in the slice's first round, NK - 1 >= 2 barriers ago)
        cs8_commit_tok(X1, nG, rv, tid);
        cs8_commit_tok(X2, nV, rv, tid);
        // No load of the loop sits behind a branch: hipcc loses count of the loads in flight at every join and waits for ALL of
        // them (s_waitcnt vmcnt(0) before each refill -- the ring then holds one round, whatever its depth).  Behind the last V
        // slice the rows of this slice are requested again (never used) and the P slots receive step 3's Q tiles.
        const int vn = last ? vs : vs + 64;
#if !TOK4_NO_TOKLOAD
        {   // (the fillers behind the last slice read the chunk's first Q tile: lines that step 3 wants anyway)
            const int r = tid >> 3, c = (tid & 7) * 8;
            const long row = p0 + (r < rv ? r : 0);
            const u16* fill = qb + row * a.q.sn + c;
            nG = gld<uint4>(last ? fill : gb + vn + row * a.dout.sn + c);
            nV = gld<uint4>(last ? fill : vb + vn + row * a.v.sn + c);
        }
#endif
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            {
                u16* B1c = DBUF && (rr & 1) ? Bx : B1;
                u16* B2c = DBUF && (rr & 1) ? Bx + CT : B2;
                ++rr;
                cs8_commit_state(B1c, rP[kk], tid);
                cs8_commit_state(B2c, rdS[kk], tid);
                __syncthreads();
#if !TOK4_NO_LOAD
                {
                    const int r = tid >> 3, c = (tid & 7) * 8;
                    const u16* qsrc = qb + kk * 64 + (p0 + (r < rv ? r : 0)) * a.q.sn + c;   // step 3's Q tile
                    const u16* psrc = Pb + cs_tile_off(kk * 64, vn, V) + tid * 8;
                    rP[kk] = gld<uint4>(last ? qsrc : psrc);
                    rdS[kk] = gld<uint4>(last ? qsrc : dSb + cs_tile_off(kk * 64, vn, V) + tid * 8);   // (filler: the same lines)
                }
#endif
                if (kk == 0) {
                    tile_a8(aG, X1, rt, lane);
                    if constexpr (KREG) tile_a8(aV, X2, rt, lane);
                    tile_mma8r<false>(accdA, aG, X2, ch, lane);                // dO V^T
                    tile_mma8<true, true>(accV, As, X1, rt, ch, lane);         // A^T dO
                }
                tile_mma8r<false>(accQ[kk], aG, B1c, ch, lane);                // dO P^T
                if constexpr (KREG) tile_mma8r<false>(accK[kk], aV, B2c, ch, lane);                // V dS^T
                else                tile_mma8<false, false>(accK[kk], X2, B2c, rt, ch, lane);
                if constexpr (KREG) tile_mma8r<true>(accV, aK[kk], B2c, ch, lane);                 // K dS
                else                tile_mma8<false, true>(accV, KT + kk * CT, B2c, rt, ch, lane);
                if constexpr (!DBUF) __syncthreads();
            }
        }
        u16* Vst = DBUF ? Bx + 2 * CT : B1;
        cs8_put(Vst, accV, 1.f, rt, ch, lane);
        __syncthreads();
        cs8_store_tok(mbase(a.dv) + vs, a.dv.sn, p0, rv, Vst, tid);
        if constexpr (!DBUF) __syncthreads();
    }

    // ---- step 3: dA tile, diagonal term, the m_ii parts of dQ / dK ----
    float dsum = 0.f;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rt * 16 + kg * 4 + r, col = ch * 32 + tn * 16 + n;
            const float dv = col <= row ? accdA[tn][r] : 0.f;
            dsum += accA[tn][r] * dv;
            dAs[row * CLD + col] = cvt_bf16(mii * dv);
        }
    dsum = wave_sum(dsum);
    if (lane == 0) red[wave] = dsum;
    __syncthreads();
    if (tid == 0) a.diag[(long)bh * a.n + ci] = a.scale * (((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7])));
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
        {
            cs8_commit_tok(X2, rP[kk], rv, tid);
            __syncthreads();
            f32x4 acc3[2];
            zero2(acc3);
            tile_mma8<false, true>(accQ[kk], dAs, KT + kk * CT, rt, ch, lane);   // dA K
            tile_mma8<true, true>(acc3, dAs, X2, rt, ch, lane);                  // dA^T Q
            cs8_put(B1, accQ[kk], a.scale, rt, ch, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i) accK[kk][i] += a.scale * acc3[i];
            cs8_put(B2, accK[kk], 1.f, rt, ch, lane);
            __syncthreads();
            cs8_store_tok(mbase(a.dq) + kk * 64, a.dq.sn, p0, rv, B1, tid);
            cs8_store_tok(mbase(a.dk) + kk * 64, a.dk.sn, p0, rv, B2, tid);
            __syncthreads();
        }
    }

}

// k_csf_out4: k_csf_out on eight waves (16 rows x 32 columns of every 64 x 64 product per wave, as k_csf_bwd_tok4), with the loads
// of its rounds out of every branch.  k_csf_out chose its next tile with if / else chains: hipcc then waits for ALL loads in flight
// before each staging write (s_waitcnt vmcnt(0)), so its one-step-ahead prefetch was in effect load -> wait -> multiply.  Here a
// round (K slice ki, V slice j) commits slot j of a register ring to one of two LDS buffers, passes ONE barrier, refills the slot
// with the same V slice of the next K slice (behind the last K slice: with the V rows of slice j for the second phase) and
// multiplies; the Q and K tiles of the next K slice travel during the NV rounds of the current one.
//   grid (n, bh, V / (64 NV)); NV = V slices per workgroup (template: the j loop carries no runtime guard)
#ifndef CSF_OUT4_CPW_
#define CSF_OUT4_CPW_ 4
#endif
constexpr int CSF_OUT4_CPW = CSF_OUT4_CPW_;   // chunks per workgroup of k_csf_out4 (plain variant)
template <int NV, bool EPI>
__host__ __device__ constexpr int csf_out4_smem() { return 7 * CT * 2 + (EPI ? 2 * 64 * 4 : 0); }
__device__ __forceinline__ void cs8_store_tok_gate(u16* __restrict__ base, long sn, const u16* __restrict__ gbase, long gsn, long p0,
                                                   int rv, const u16* __restrict__ Os, int tid) {
    const int r = tid >> 3, c = (tid & 7) * 8;
    if (r < rv) {
        uint4 x = *reinterpret_cast<const uint4*>(Os + r * CLD + c);
        if (gbase) {
            const uint4 g = gld<uint4>(gbase + (p0 + r) * gsn + c);
            unsigned xw[4] = {x.x, x.y, x.z, x.w};
            const unsigned gw[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float g0 = __uint_as_float(gw[i] << 16), g1 = __uint_as_float(gw[i] & 0xffff0000u);
                const float y0 = __uint_as_float(xw[i] << 16) * g0 / (1.f + __expf(-g0));
                const float y1 = __uint_as_float(xw[i] & 0xffff0000u) * g1 / (1.f + __expf(-g1));
                xw[i] = pack_bf16x2(y0, y1);
            }
            x = make_uint4(xw[0], xw[1], xw[2], xw[3]);
        }
        *reinterpret_cast<uint4*>(base + (p0 + r) * sn + c) = x;
    }
}

template <int NV, bool EPI>
__global__ __launch_bounds__(NT4, 4) void k_csf_out4(const CsOutArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Qs = reinterpret_cast<u16*>(smem_raw);   // Q tiles [2], later output staging [2]
    u16* Ks = Qs + 2 * CT;                        // K tiles [2]
    u16* Ps = Ks + 2 * CT;                        // P tiles [2], later V tiles [2]
    u16* As = Ps + 2 * CT;                        // m_ii tril(QK^T)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, n = lane & 15, kg = lane >> 4;
    const int rt = wave & 3, ch = wave >> 2;
    // A workgroup walks CPW consecutive chunks (the fused-epilogue variant: one): behind a chunk's last K slice the Q / K slots
    // are refilled with the NEXT chunk's first tiles and, in the second phase, the P slots with its first P tiles -- a chunk's
    // first loads travel during its predecessor's last rounds instead of being waited for cold (one workgroup per chunk lived 14 us,
    // a fifth of it in that first wait).
    constexpr int CPW = EPI ? 1 : CSF_OUT4_CPW;
    const int c0 = blockIdx.x * CPW, c1 = min(a.n, c0 + CPW), bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int vbase = blockIdx.z * 64 * NV;
    const int V = a.V, nks = a.K / 64;
    const u16* qb = (const u16*)a.q.ptr + b * a.q.sb + h * a.q.sh;
    const u16* kb = (const u16*)a.k.ptr + b * a.k.sb + h * a.k.sh;
    const u16* vb = (const u16*)a.v.ptr + b * a.v.sb + h * a.v.sh + vbase;
    u16* ob = (u16*)a.o.ptr + b * a.o.sb + h * a.o.sh + vbase;
    const long KV = (long)a.K * V;
    const int tr = tid >> 3, tc = (tid & 7) * 8;
    // the thread's token row in chunk c (rows past the sequence: the chunk's first row, zeroed on commit)
    auto row_of = [&](int c) { const long p = (long)c * CS; return p + (tr < (int)min((long)CS, a.T - p) ? tr : 0); };

    // The order of these loads must be the order in which the loop re-issues them (Q, K, P slices): hipcc counts the loads in
    // flight per register and, where the entry and the back edge of the loop disagree, waits for the younger position -- with the
    // Q / K loads sunk into the loop's preheader behind the P loads it waited for everything at the top of every K slice.
    uint4 rQ, rK, rP[NV];
    {
        const long trow0 = row_of(c0);
        rQ = gld<uint4>(qb + trow0 * a.q.sn + tc);
        rK = gld<uint4>(kb + trow0 * a.k.sn + tc);
        __builtin_amdgcn_sched_barrier(0);
        const u16* Pb0 = reinterpret_cast<const u16*>(a.P) + ((long)bh * a.n + c0) * KV;
#pragma unroll
        for (int j = 0; j < NV; ++j) cs8_issue_state(rP[j], Pb0 + cs_tile_off(0, vbase + 64 * j, V), tid);
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int ci = c0; ci < c1; ++ci) {
    const long p0 = (long)ci * CS;
    const int rv = (int)min((long)CS, a.T - p0);
    const u16* Pb = reinterpret_cast<const u16*>(a.P) + ((long)bh * a.n + ci) * KV;
    const long trow = row_of(ci);
    const int cn = ci + 1 < c1 ? ci + 1 : ci;   // next chunk (behind the last one: this chunk again -- hot lines, never used)
    const long trown = row_of(cn);
    const u16* Pbn = reinterpret_cast<const u16*>(a.P) + ((long)bh * a.n + cn) * KV;
    if (ci > c0) __syncthreads();   // the previous chunk's staging tiles are dead
    const float mii = gld<float>(a.mix + (long)ci * a.ldmix + ci);   // (requested here: a wait for it later would drain the ring)
    f32x4 accO[NV][2], accA[2];
#pragma unroll
    for (int j = 0; j < NV; ++j) zero2(accO[j]);
    zero2(accA);
    int ki = 0;
    do {   // (K >= 64: no branch around the loop for the Q / K loads to sink under)
        const bool lastk = ki + 1 >= nks;   // (uniform)
        const int kn = lastk ? ki : ki + 1;
        u16* Qc = Qs + (ki & 1) * CT;
        u16* Kc = Ks + (ki & 1) * CT;
        bf16x8 aQ[2];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            u16* Pc = Ps + ((ki * NV + j) & 1) * CT;
            if (j == 0) {
                cs8_commit_tok(Qc, rQ, rv, tid);
                cs8_commit_tok(Kc, rK, rv, tid);
            }
            cs8_commit_state(Pc, rP[j], tid);
            __syncthreads();
            if (j == 0) {   // the next K slice's tiles -- behind the last K slice: the next chunk's first
                const long rown = lastk ? trown : trow;
                const int coln = lastk ? 0 : kn * 64;
                rQ = gld<uint4>(qb + rown * a.q.sn + coln + tc);
                rK = gld<uint4>(kb + rown * a.k.sn + coln + tc);
            }
            {
                const u16* psrc = Pb + cs_tile_off(kn * 64, vbase + 64 * j, V) + tid * 8;
                const u16* vsrc = vb + 64 * j + trow * a.v.sn + tc;   // second phase's V rows
                rP[j] = gld<uint4>(lastk ? vsrc : psrc);
            }
            if (j == 0) {
                tile_a8(aQ, Qc, rt, lane);
                tile_mma8r<false>(accA, aQ, Kc, ch, lane);   // Q K^T
            }
            tile_mma8r<true>(accO[j], aQ, Pc, ch, lane);     // Q P
        }
    } while (++ki < nks);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rt * 16 + kg * 4 + r, col = ch * 32 + tn * 16 + n;
            As[row * CLD + col] = cvt_bf16(col <= row ? mii * accA[tn][r] : 0.f);
        }
    __syncthreads();   // the last round's P tile is dead, As is complete
    if constexpr (!EPI) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            u16* Vc = Ps + (j & 1) * CT;
            u16* Oc = Qs + (j & 1) * CT;
            cs8_commit_tok(Vc, rP[j], rv, tid);
            __syncthreads();
            cs8_issue_state(rP[j], Pbn + cs_tile_off(0, vbase + 64 * j, V), tid);   // the next chunk's first P tiles
            tile_mma8<false, true>(accO[j], As, Vc, rt, ch, lane);        // tril(QK^T) V
            cs8_put(Oc, accO[j], a.scale, rt, ch, lane);
            __syncthreads();
            cs8_store_tok(ob + 64 * j, a.o.sn, p0, rv, Oc, tid);
        }
    } else {
        float* red = reinterpret_cast<float*>(As + CT);   // [2 column halves][64 rows] sums of squares
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            u16* Vc = Ps + (j & 1) * CT;
            cs8_commit_tok(Vc, rP[j], rv, tid);
            __syncthreads();
            tile_mma8<false, true>(accO[j], As, Vc, rt, ch, lane);
        }
        // row sums of squares over the head's V channels: lane holds rows 16 rt + 4 kg + r, columns 64 j + 32 ch + 16 tn + n
        float ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x = a.scale * accO[j][tn][r];
                    ss[r] += x * x;
                }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = ss[r];
            x += __shfl_xor(x, 1, 64); x += __shfl_xor(x, 2, 64); x += __shfl_xor(x, 4, 64); x += __shfl_xor(x, 8, 64);
            if (n == 0) red[ch * 64 + rt * 16 + kg * 4 + r] = x;
        }
        __syncthreads();
        u16* yb = (u16*)a.y.ptr + b * a.y.sb + h * a.y.sh + vbase;
        const u16* gb = a.gate.ptr ? (const u16*)a.gate.ptr + b * a.gate.sb + h * a.gate.sh + vbase : nullptr;
        float rstd[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rt * 16 + kg * 4 + r;
            rstd[r] = rsqrtf((red[row] + red[64 + row]) / (float)V + a.neps);
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            u16* Oc = Qs + (j & 1) * CT;
            u16* Yc = Ps + (j & 1) * CT;
            if (a.o.ptr) {   // training: the operator's own output is kept for the norm's backward
                cs8_put(Oc, accO[j], a.scale, rt, ch, lane);
            }
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const float w = a.nw ? gld<float>(a.nw + vbase + 64 * j + ch * 32 + tn * 16 + n) : 1.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) accO[j][tn][r] *= a.scale * rstd[r] * w;
            }
            cs8_put(Yc, accO[j], 1.f, rt, ch, lane);
            __syncthreads();
            if (a.o.ptr) cs8_store_tok(ob + 64 * j, a.o.sn, p0, rv, Oc, tid);
            cs8_store_tok_gate(yb + 64 * j, a.y.sn, gb ? gb + 64 * j : nullptr, a.gate.sn, p0, rv, Yc, tid);
        }
    }
    }   // chunks of the workgroup
}

// -------------------------------------------------------------------------------------------------
// k_csf_state: out[bh][ci][kk][v] (bf16) = mul * sum_{c in chunk ci} X[c][kk] Y[c][v]        grid (n, bh, K / 64)
//   forward: X = K, Y = V (S_j, naive.py:60);  backward: X = Q, Y = dO, mul = scale (dP_i)
// -------------------------------------------------------------------------------------------------
struct CsfStateArgs {
    View x, y;
    u16* out;
    int H, n, K, V;
    long T;
    float mul;
};
constexpr int CSF_STATE_SMEM = 3 * CT * 2;

__global__ __launch_bounds__(NTHREADS) void k_csf_state(const CsfStateArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Xs = reinterpret_cast<u16*>(smem_raw);
    u16* Ys = Xs + CT;
    u16* Os = Ys + CT;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int ci = blockIdx.x, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H, ks = blockIdx.z * 64;
    const long p0 = (long)ci * CS;
    const int rv = (int)min((long)CS, a.T - p0);
    const u16* xb = (const u16*)a.x.ptr + b * a.x.sb + h * a.x.sh + ks;
    const u16* yb = (const u16*)a.y.ptr + b * a.y.sb + h * a.y.sh;
    const int r = tid >> 2, c = (tid & 3) * 16;

    cs_stage_tok(Xs, xb, a.x.sn, p0, rv, tid);
    uint4 y0 = make_uint4(0, 0, 0, 0), y1 = y0;
    if (r < rv) {
        const u16* src = yb + (p0 + r) * a.y.sn + c;
        y0 = *reinterpret_cast<const uint4*>(src);
        y1 = *reinterpret_cast<const uint4*>(src + 8);
    }
    for (int vs = 0; vs < a.V; vs += 64) {
        *reinterpret_cast<uint4*>(Ys + r * CLD + c) = y0;
        *reinterpret_cast<uint4*>(Ys + r * CLD + c + 8) = y1;
        __syncthreads();
        if (vs + 64 < a.V && r < rv) {   // next V slice in flight during the tile product
            const u16* src = yb + (p0 + r) * a.y.sn + vs + 64 + c;
            y0 = *reinterpret_cast<const uint4*>(src);
            y1 = *reinterpret_cast<const uint4*>(src + 8);
        }
        f32x4 acc[4];
        zero4(acc);
        tile_mma<true, true>(acc, Xs, Ys, wave, lane);   // rows kk, columns v, reduction over the chunk's tokens
        cs_put(Os, acc, a.mul, wave, lane);
        __syncthreads();
        u16* d = a.out + ((long)bh * a.n + ci) * a.K * a.V + cs_tile_off(ks, vs, a.V) + r * CS + c;
        *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(Os + r * CLD + c);
        *reinterpret_cast<uint4*>(d + 8) = *reinterpret_cast<const uint4*>(Os + r * CLD + c + 8);
    }
}

// k_csf_state2: the same summaries, one workgroup per chunk and [128 x 256] block of the summary (the whole summary at
// K = 128, V = 256).  All of the block's token rows are requested at once (48 KB in flight per workgroup instead of one 8 KB V
// slice ahead), one barrier, then every wave multiplies its 32 summary rows against all V columns and streams them out through
// a wave-private staging strip: no further workgroup barrier, 2 KB contiguous per wave store (tile-major layout).
constexpr int ST2_KW = 128, ST2_VW = 256, ST2_LDX = ST2_KW + 8, ST2_LDY = ST2_VW + 8;
constexpr int CSF_STATE2_SMEM = (CS * ST2_LDX + CS * ST2_LDY + 4 * 16 * CLD) * 2;

// A workgroup walks ST2_CPW consecutive chunks with the next chunk's rows in flight in registers while the current one is
// multiplied and stored (one workgroup per chunk lived 8 us, a quarter of it waiting for its first rows with nothing else to do).
constexpr int ST2_CPW = 4;
__global__ __launch_bounds__(NTHREADS, 2) void k_csf_state2(const CsfStateArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Xs = reinterpret_cast<u16*>(smem_raw);
    u16* Ys = Xs + CS * ST2_LDX;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    u16* Ws = Ys + CS * ST2_LDY + wave * 16 * CLD;
    const int c0 = blockIdx.x * ST2_CPW, c1 = min(a.n, c0 + ST2_CPW), bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int nvb = (a.V + ST2_VW - 1) / ST2_VW, kb = blockIdx.z / nvb, vb = blockIdx.z - kb * nvb;
    const int k0 = kb * ST2_KW, v0 = vb * ST2_VW, kw = min(ST2_KW, a.K - k0), vw = min(ST2_VW, a.V - v0);
    const u16* xb = (const u16*)a.x.ptr + b * a.x.sb + h * a.x.sh + k0;
    const u16* yb = (const u16*)a.y.ptr + b * a.y.sb + h * a.y.sh + v0;

    // X: 4 passes of 16 rows x 16 pieces; Y: 8 passes of 8 rows x 32 pieces (pieces past the block's width, rows past the chunk: zeros)
    const int xc = (tid & 15) * 8, yc = (tid & 31) * 8;
    const bool xok = xc < kw, yok = yc < vw;
    uint4 xr[4], yr[8];
    auto issue = [&](int ci, bool filler) {   // (no load behind a branch: rows past the sequence's end read the chunk's first row;
        const long p0 = (long)ci * CS;        //  the filler behind the last chunk reads that one row with every pass)
        const int rv = filler ? 0 : (int)min((long)CS, a.T - p0);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = (tid >> 4) + 16 * p;
            xr[p] = gld_stream16(xb + (p0 + (row < rv ? row : 0)) * a.x.sn + (xok ? xc : 0));
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int row = (tid >> 5) + 8 * p;
            yr[p] = gld_stream16(yb + (p0 + (row < rv ? row : 0)) * a.y.sn + (yok ? yc : 0));
        }
    };
    issue(c0, false);
    for (int ci = c0; ci < c1; ++ci) {
        const int rv = (int)min((long)CS, a.T - (long)ci * CS);
        u16* ob = a.out + ((long)bh * a.n + ci) * a.K * a.V;
        if (ci > c0) __syncthreads();   // the previous chunk's tiles are dead
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = (tid >> 4) + 16 * p;
            const bool ok = row < rv && xok;
            *reinterpret_cast<uint4*>(Xs + row * ST2_LDX + xc) = make_uint4(ok ? xr[p].x : 0u, ok ? xr[p].y : 0u, ok ? xr[p].z : 0u, ok ? xr[p].w : 0u);
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int row = (tid >> 5) + 8 * p;
            const bool ok = row < rv && yok;
            *reinterpret_cast<uint4*>(Ys + row * ST2_LDY + yc) = make_uint4(ok ? yr[p].x : 0u, ok ? yr[p].y : 0u, ok ? yr[p].z : 0u, ok ? yr[p].w : 0u);
        }
        __syncthreads();
        issue(min(ci + 1, c1 - 1), ci + 1 >= c1);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int rt = wave * 2 + rr;               // 16 summary rows kk = k0 + 16 rt ..
            if (rt * 16 < kw) {
                // the product is formed transposed (m = v, n = kk): a lane ends up with four consecutive v of one summary row
                const bf16x8 xa0 = tr_read8(Xs, ST2_LDX, 0, rt * 16, lane), xa1 = tr_read8(Xs, ST2_LDX, 32, rt * 16, lane);
                for (int vt = 0; vt * 64 < vw; ++vt) {
                    f32x4 acc[4];
                    zero4(acc);
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) {
                        acc[tn] = mfma_bf16(tr_read8(Ys, ST2_LDY, 0, vt * 64 + tn * 16, lane), xa0, acc[tn]);
                        acc[tn] = mfma_bf16(tr_read8(Ys, ST2_LDY, 32, vt * 64 + tn * 16, lane), xa1, acc[tn]);
                    }
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn)
                        *reinterpret_cast<uint2*>(Ws + nl * CLD + tn * 16 + kg * 4) =
                            make_uint2(pack_bf16x2(a.mul * acc[tn][0], a.mul * acc[tn][1]), pack_bf16x2(a.mul * acc[tn][2], a.mul * acc[tn][3]));
                    wave_lds_fence();
                    // a store instruction covers eight full 128-byte rows (two half rows per lane pair made it 16 half lines)
                    const int r = lane >> 3, c = (lane & 7) * 8;
                    const uint4 o0 = *reinterpret_cast<const uint4*>(Ws + r * CLD + c), o1 = *reinterpret_cast<const uint4*>(Ws + (r + 8) * CLD + c);
                    u16* d = ob + cs_tile_off(k0 + rt * 16, v0 + vt * 64, a.V) + ((rt * 16) & 63) * CS + r * CS + c;
                    gst<uint4>(d, o0);
                    gst<uint4>(d + 8 * CS, o1);
                    wave_lds_fence();
                }
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
// k_csf_mix: mixing across chunks as a GEMM over the flattened summaries (E = K V elements per chunk):
//   TRANS 0: out[i][e] = sum_{j < i} m[i][j] in[j][e]      (P_i, naive.py:63-66)
//   TRANS 1: out[j][e] = sum_{i > j} m[i][j] in[i][e]      (dS_j)
// grid (E / 256, ceil(n / 64), bh); each wave owns 16 output chunks x 256 elements.  The A operand (mixing
// weights) is built in registers from the fp32 matrix as bf16 hi + lo; the B operand is a [32 chunks x 256] bf16
// tile in LDS read through the transpose read; tiles are double buffered with the next one in flight in registers.
// -------------------------------------------------------------------------------------------------
struct CsfMixArgs {
    const float* W;
    int ldw;
    const u16* in;
    u16* out;
    int n;
    long E;
};
constexpr int MX_TE = 256, MX_LD = MX_TE + 8, MX_KS = 32;
constexpr int CSF_MIX_SMEM = 64 * MX_LD * 2;   // two [32][MX_LD] input tiles, reused as the [64][MX_LD] output staging

template <int TRANS>
__global__ __launch_bounds__(NTHREADS, 2) void k_csf_mix(const CsfMixArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Xs = reinterpret_cast<u16*>(smem_raw);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    const long e0 = (long)blockIdx.x * MX_TE;
    const int i0 = blockIdx.y * 64, bh = blockIdx.z, n = a.n;
    const u16* inb = a.in + (long)bh * n * a.E + e0;
    u16* outb = a.out + (long)bh * n * a.E + e0;
    const int kbeg = TRANS ? i0 : 0, kend = TRANS ? n : min(n, i0 + 64);
    const int steps = (kend - kbeg + MX_KS - 1) / MX_KS;
    const int sr = tid >> 3, sc = (tid & 7) * 8;   // staging: row sr of the tile, 4 x 16 B at columns sc + 64 u

    uint4 pre[4];
    auto fetch = [&](int step) {
        const int row = kbeg + step * MX_KS + sr;
#pragma unroll
        for (int u = 0; u < 4; ++u) pre[u] = make_uint4(0, 0, 0, 0);
        if (row < n) {
            const u16* src = inb + (long)row * a.E + sc;
#pragma unroll
            for (int u = 0; u < 4; ++u) pre[u] = *reinterpret_cast<const uint4*>(src + 64 * u);
        }
    };
    auto commit = [&](int buf) {
        u16* d = Xs + buf * (MX_KS * MX_LD) + sr * MX_LD + sc;
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<uint4*>(d + 64 * u) = pre[u];
    };

    f32x4 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int orow = i0 + wave * 16 + nl;   // output chunk of this lane's A row

    if (steps > 0) {
        fetch(0);
        commit(0);
    }
    for (int step = 0; step < steps; ++step) {
        const int buf = step & 1;
        __syncthreads();
        if (step + 1 < steps) fetch(step + 1);
        // mixing weights for (orow, k0 .. k0 + 7), masked, split into bf16 hi + lo
        const int k0 = kbeg + step * MX_KS + kg * 8;
        bf16x8 ah, al;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int kk = k0 + t;
            float w = 0.f;
            if (orow < n && kk < n && (TRANS ? kk > orow : kk < orow)) w = TRANS ? a.W[(long)kk * a.ldw + orow] : a.W[(long)orow * a.ldw + kk];
            const __bf16 hi = (__bf16)w;
            ah[t] = hi;
            al[t] = (__bf16)(w - (float)hi);
        }
        const u16* tile = Xs + buf * (MX_KS * MX_LD);
#pragma unroll
        for (int t4 = 0; t4 < 16; t4 += 4) {
            bf16x8 bv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) bv[t] = tr_read8(tile, MX_LD, 0, (t4 + t) * 16, lane);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t4 + t] = mfma_bf16(ah, bv[t], acc[t4 + t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t4 + t] = mfma_bf16(al, bv[t], acc[t4 + t]);
        }
        if (step + 1 < steps) commit(buf ^ 1);
    }
    __syncthreads();
    // C layout (row = 16 wave + 4 kg + r, column = 16 t + nl) -> staging -> 512-byte rows
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) Xs[(wave * 16 + kg * 4 + r) * MX_LD + t * 16 + nl] = cvt_bf16(acc[t][r]);
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int row = sr + 32 * rr;
        if (i0 + row < n) {
            u16* d = outb + (long)(i0 + row) * a.E + sc;
#pragma unroll
            for (int u = 0; u < 4; ++u) *reinterpret_cast<uint4*>(d + 64 * u) = *reinterpret_cast<const uint4*>(Xs + row * MX_LD + sc + 64 * u);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// k_csf_dw: strictly-lower part of dmix, dwp[bh][split][i][j] = sum_{e in slice} dP_i[e] S_j[e].
// grid (tile pairs, bh, E-slices) as k_dw<1>; both operands are read straight from HBM in MFMA layout (the
// reduction index e is contiguous in memory for both).  Each wave takes a quarter of the slice for the whole
// 64 x 64 tile; the four partial tiles are summed through LDS in a fixed order.
// -------------------------------------------------------------------------------------------------
struct CsfDwArgs {
    const u16* x;   // dP [bh][n][E]
    const u16* y;   // S  [bh][n][E]
    long E;
    float* out;     // [bh][nsplit][n][n]
    int n, tiles, nsplit;
};
constexpr int CSF_DW_LD = 68;
constexpr int CSF_DW_SMEM = 64 * CSF_DW_LD * 4;

// NT: 16-row tiles per side (4: 64 x 64; 2 / 1 for n <= 32 / 16 chunks, e.g. the fla layer's 2048-token sequences: no loads or
// MFMAs on clamped rows)
template <int NT = 4>
__global__ __launch_bounds__(NTHREADS) void k_csf_dw(const CsfDwArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* Rs = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    // the tile pairs of one (b, h, slice) share their operand rows: consecutive logical indices, kept on one XCD (one L2) and
    // adjacent in dispatch order by xcd_swizzle
    const int np = gridDim.x, nbh = gridDim.y;
    const int L = xcd_swizzle(blockIdx.x + np * (blockIdx.y + nbh * blockIdx.z), np * nbh * gridDim.z);
    const int unit = L / np, pair = L - unit * np, split = unit / nbh, bh = unit - split * nbh, n = a.n;
    const int it = pair / a.tiles, jt = pair - it * a.tiles;
    const int i0 = it * 64, j0 = jt * 64;
    if (j0 > i0 + 63) return;   // tile entirely above the diagonal
    const long per = ((a.E + a.nsplit - 1) / a.nsplit + 255) & ~255L;   // slice: multiple of 4 waves x 64 elements
    const long ebeg = (long)split * per, eend = min(a.E, ebeg + per);
    const long wlen = (eend > ebeg ? (eend - ebeg) : 0) / 4;             // E is a multiple of 4096: wlen multiple of 64
    const long wbeg = ebeg + wave * wlen;
    // each lane covers 16 consecutive e of a 64-element step: two k-steps of 8 (same permutation for both operands)
    const u16* xp[NT];
    const u16* yp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int ri = min(i0 + t * 16 + nl, n - 1), rj = min(j0 + t * 16 + nl, n - 1);
        xp[t] = a.x + ((long)bh * n + ri) * a.E + wbeg + kg * 16;
        yp[t] = a.y + ((long)bh * n + rj) * a.E + wbeg + kg * 16;
    }
    f32x4 acc[NT][NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4 / NT
    for (long e = 0; e < wlen; e += 64) {
        bf16x8 xa[NT][2], yb[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            xa[t][0] = *reinterpret_cast<const bf16x8*>(xp[t] + e);
            xa[t][1] = *reinterpret_cast<const bf16x8*>(xp[t] + e + 8);
            yb[t][0] = *reinterpret_cast<const bf16x8*>(yp[t] + e);
            yb[t][1] = *reinterpret_cast<const bf16x8*>(yp[t] + e + 8);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mfma_bf16(xa[i][s], yb[j][s], acc[i][j]);
    }
    // sum the four waves' tiles in wave order
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* d = Rs + (i * 16 + kg * 4 + r) * CSF_DW_LD + j * 16 + nl;
                        *d = (w == 0 ? 0.f : *d) + acc[i][j][r];
                    }
        }
        __syncthreads();
    }
    float* out = a.out + ((long)bh * a.nsplit + split) * n * n;
    for (int v = tid; v < 64 * 64; v += NTHREADS) {
        const int r = v >> 6, c = v & 63;
        if (i0 + r < n && j0 + c < n) out[(long)(i0 + r) * n + j0 + c] = Rs[r * CSF_DW_LD + c];
    }
}

}  // namespace fast
}  // namespace mhla
