// Small-sequence path of the block-mixing MHLA operator: bf16, block size S = 16, N = 16 M <= 256 tokens, D <= 80
// (the DiT / ViT regime: N = 256, M = 16, D = 64 or 72).  One workgroup per (b, h), ONE launch per direction.
//
// With S = 16 every (query block i, key block j) pair is exactly one 16 x 16 MFMA tile, so the operator is evaluated in
// its attention form      O_i = sum_j W[i][j] (Q_i K_j^T) V_j / n_i
// where the mixing weight is a scalar multiply of the score tile.  At N = 256 this costs ~2.4x the FLOPs of the
// summary form but needs no block summaries at all: Q, K, V are read once, O written once (the summary form moves
// 6.75x the token bytes through the KV / G workspaces at this shape).
//
// Data flow of one block row i (owned by one wave), flash-attention style without softmax:
//   S^T tile (j, i) = K_j Q_i^T   (A = rows of K from LDS, B = Q_i held in registers; C layout: lane = (s, 4 t's))
//   scale by W[i][j], pack two tiles (j0, j1) to bf16  ->  this IS the A operand (m = s, 32 k-slots = the 2 x 16 t's) of
//   O_i += P V     with B = V rows fetched by two hardware transpose reads that follow the same k-slot order.
// No shuffle or LDS round trip sits between the two contractions.
#pragma once
#include "fused.hpp"

namespace mhla {
namespace fast {

// forward: 16 waves, wave w owns block w (M <= 16) -- its phases are latency chains (LDS read -> MFMA -> pack -> MFMA per block
// pair) that 16 waves hide better than 8 waves with two blocks each (42.7 -> 37.8 us at the DiT-XL/2 shape).
// backward: 8 waves, wave w owns blocks w and w + 8 and processes them JOINTLY (every (i, j) pair re-reads the K / V / Q / dO' rows of
// block j, and two blocks that share each fetched operand halve that LDS traffic) -- the reduced-precision kernel; the default-arithmetic
// one (hi + lo score tiles: 187 / 230 registers with two blocks) runs 16 waves with one block each (template parameter NB), 3.5 % faster at
// C3: per block pair and SIMD its MFMAs (~1 500 cycles), its VALU work (~800) and the LDS reads (~1 000 - 2 000) add up to the ~3 500
// measured either way -- the passes are bound by the SUM of the three pipes, not by latency (tools/trace_smalln.py).
constexpr int SN_T = 1024;
constexpr int SN_W = SN_T / 64;
constexpr int SN_TB = 512;
constexpr int SN_WB = SN_TB / 64;

struct SnArgs {
    View q, k, v, o, dout;
    MView out, dq, dk, dv;
    const int* idx;
    const float* W;
    int ldw;
    float* dwp;     // [bh][M][M] partial dW (backward)
    int H, M, D;
    float eps;
    int relu, normalize;
    unsigned long long* trace;   // debugging aid (mhla_debug_set_trace): phase timestamps of k_sn_bwd
};

template <int DT>
// LDS row stride (bf16) of the staged token tiles: 80 elements = 160 bytes for D <= 64 and for D <= 80 alike.  In gfx950's
// 64-bank lane-group model (MI355X_MICROARCH.md; tools/lds_conflicts.py) 160-byte rows make the two operand patterns of these
// kernels -- 16-byte row reads of 16 consecutive rows, and transpose reads of two 16-row tiles -- conflict-free; the strides of
// rounds 1-3 (DT * 16 + 8: 144 / 176 bytes, derived for 32 banks) were two-way conflicted in both (counters: 0.33).
__host__ __device__ constexpr int sn_ldr() { static_assert(DT * 16 <= 80, "row of at most 80 elements"); return 80; }

// stage `nrows` token rows (D valid columns, zero up to DP) into an LDS tile [nrows][LDR]; optionally every row scaled by
// rowscale[r] (fp32, LDS).  All loads are issued before any is used, from clamped (always valid) addresses with no branch around them
// -- a load inside `if (r < nrows && p < dv)` makes the compiler wait for it at the end of the branch, one memory latency per piece.
__device__ __forceinline__ uint4 sel4(bool c, uint4 v) { return make_uint4(c ? v.x : 0u, c ? v.y : 0u, c ? v.z : 0u, c ? v.w : 0u); }
template <int DT, int NT = SN_T, bool SCALE = false>
__device__ __forceinline__ void sn_stage(u16* __restrict__ dst, const u16* __restrict__ base, long sn, const int* __restrict__ idx,
                                         int nrows, int D, float eps, int tid, bool relu, const float* __restrict__ rowscale = nullptr) {
    constexpr int LDR = sn_ldr<DT>(), PV = DT * 2;   // 16-byte pieces per padded row
    const int dv = D >> 3;
    constexpr int MAXIT = (256 * PV + NT - 1) / NT;
    uint4 reg[MAXIT];
#pragma unroll
    for (int t = 0; t < MAXIT; ++t) {
        const int v = tid + t * NT, r = v / PV, p = v - r * PV;
        reg[t] = gld<uint4>(base + tok_row(idx, min(r, nrows - 1)) * sn + min(p, dv - 1) * 8);
    }
#pragma unroll
    for (int t = 0; t < MAXIT; ++t) {
        const int v = tid + t * NT, r = v / PV, p = v - r * PV;
        if (r < nrows) {
            uint4 x = reg[t];
            if (relu) x = relu_eps8(x, eps);   // (uniform)
            if (SCALE) {
                const float sc = rowscale[r];
                unsigned w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    w[u] = pack_bf16x2(__uint_as_float(w[u] << 16) * sc, __uint_as_float(w[u] & 0xffff0000u) * sc);
                x = make_uint4(w[0], w[1], w[2], w[3]);
            }
            *reinterpret_cast<uint4*>(dst + r * LDR + p * 8) = sel4(p < dv, x);
        }
    }
}

// Two tiles at once: BOTH tensors' loads are issued before either is written to LDS.  Two sn_stage calls in a row cost two
// memory round trips (the second tensor's loads wait behind the first tensor's s_waitcnt): with an otherwise empty GPU a backward
// workgroup spent 6.6 us on "stage K, V" -- own-row loads drained for the weights' LDS write, then K, then V, three round trips.
template <int DT, int NT, bool SCALE1>
__device__ __forceinline__ void sn_stage2(u16* __restrict__ dst0, const u16* __restrict__ base0, long sn0, float eps0, bool relu0,
                                          u16* __restrict__ dst1, const u16* __restrict__ base1, long sn1,
                                          const int* __restrict__ idx, int nrows, int D, int tid, const float* __restrict__ rowscale1 = nullptr) {
    constexpr int LDR = sn_ldr<DT>(), PV = DT * 2;
    const int dv = D >> 3;
    constexpr int MAXIT = (256 * PV + NT - 1) / NT;
    uint4 r0[MAXIT], r1[MAXIT];
#pragma unroll
    for (int t = 0; t < MAXIT; ++t) {
        const int v = tid + t * NT, r = v / PV, p = v - r * PV;
        const long row = tok_row(idx, min(r, nrows - 1));
        r0[t] = gld<uint4>(base0 + row * sn0 + min(p, dv - 1) * 8);
        r1[t] = gld<uint4>(base1 + row * sn1 + min(p, dv - 1) * 8);
    }
#pragma unroll
    for (int t = 0; t < MAXIT; ++t) {
        const int v = tid + t * NT, r = v / PV, p = v - r * PV;
        if (r < nrows) {
            uint4 x = r0[t], y = r1[t];
            if (relu0) x = relu_eps8(x, eps0);   // (uniform)
            if (SCALE1) {
                const float sc = rowscale1[r];
                unsigned w[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    w[u] = pack_bf16x2(__uint_as_float(w[u] << 16) * sc, __uint_as_float(w[u] & 0xffff0000u) * sc);
                y = make_uint4(w[0], w[1], w[2], w[3]);
            }
            *reinterpret_cast<uint4*>(dst0 + r * LDR + p * 8) = sel4(p < dv, x);
            *reinterpret_cast<uint4*>(dst1 + r * LDR + p * 8) = sel4(p < dv, y);
        }
    }
}

// 16 rows x KS k-steps of an MFMA operand straight from global: lane (m = lane & 15, kg) -> row0 + m, cols 32 ks + 8 kg ..
// (columns beyond D: a clamped address, zeroed on arrival).  In two halves, so that a kernel can request all its rows first and touch
// them (relu + eps, zeroing) only where they are consumed.
template <int KS>
__device__ __forceinline__ void sn_issue_rows(uint4 (&v)[KS], const u16* __restrict__ base, long sn, const int* __restrict__ idx,
                                              int row0, int D, int lane) {
    const int m = lane & 15, kg = lane >> 4;
    const u16* src = base + tok_row(idx, row0 + m) * sn;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) v[ks] = gld<uint4>(src + min(ks * 32 + kg * 8, D - 8));
}
template <int KS>
__device__ __forceinline__ void sn_finish_rows(bf16x8 (&a)[KS], const uint4 (&v)[KS], int D, float eps, int lane, bool relu) {
    const int kg = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        uint4 x = v[ks];
        if (relu) x = relu_eps8(x, eps);   // (uniform; no memory operation inside the branch)
        a[ks] = __builtin_bit_cast(bf16x8, sel4(ks * 32 + kg * 8 < D, x));
    }
}
template <int KS>
__device__ __forceinline__ void sn_load_rows(bf16x8 (&a)[KS], const u16* __restrict__ base, long sn, const int* __restrict__ idx,
                                             int row0, int D, float eps, int lane, bool relu = false) {
    uint4 v[KS];
    sn_issue_rows<KS>(v, base, sn, idx, row0, D, lane);
    sn_finish_rows<KS>(a, v, D, eps, lane, relu);
}
// the same from an LDS tile [rows][LDR]
template <int KS>
__device__ __forceinline__ void sn_lds_rows(bf16x8 (&a)[KS], const u16* __restrict__ tile, int ldr, int row0, int D, int lane) {
    const int m = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ks * 32 + kg * 8 < D) v = *reinterpret_cast<const uint4*>(tile + (row0 + m) * ldr + ks * 32 + kg * 8);
        a[ks] = __builtin_bit_cast(bf16x8, v);
    }
}
// B operand whose 32 k-slots are (tile t0 rows 4 kg .. +3, tile t1 rows 4 kg .. +3): two transpose reads
__device__ __forceinline__ bf16x8 sn_tr_pair(const u16* __restrict__ tile, int ldr, int row_t0, int row_t1, int c0, int lane) {
    const int g = lane >> 4, li = lane & 15;
    const u16* p0 = tile + (row_t0 + g * 4 + (li >> 2)) * ldr + c0 + (li & 3) * 4;
    const u16* p1 = tile + (row_t1 + g * 4 + (li >> 2)) * ldr + c0 + (li & 3) * 4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(p1));
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ bf16x8 sn_pack_pair(f32x4 c0, f32x4 c1) {
    uint4 v;
    v.x = pack_bf16x2(c0[0], c0[1]); v.y = pack_bf16x2(c0[2], c0[3]);
    v.z = pack_bf16x2(c1[0], c1[1]); v.w = pack_bf16x2(c1[2], c1[3]);
    return __builtin_bit_cast(bf16x8, v);
}

// the same pair as bf16 hi + lo parts (x = hi + lo, 16 significand bits): the score tile keeps the reference's fp32 accuracy on
// its way into the second contraction, which then runs on both parts (HL kernels)
__device__ __forceinline__ void sn_pack_pair_hl(const f32x4& c0, const f32x4& c1, bf16x8& hi, bf16x8& lo) {
    const float x[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = pack_bf16x2(x[2 * i], x[2 * i + 1]);
        l[i] = pack_bf16x2(x[2 * i] - __uint_as_float(h[i] << 16), x[2 * i + 1] - __uint_as_float(h[i] & 0xffff0000u));
    }
    hi = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
    lo = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
}

// Sums 16 per-lane values over the 64 lanes with 17 shuffles (instead of 16 x 6): at each of the first four butterfly steps a
// lane keeps the half of the values selected by its own lane bit and sends the other half.  Returns the total of value
// j = 8 b5 + 4 b4 + 2 b3 + b2 (b_k = bit k of the lane); the four lanes of a quad hold the same total.
__device__ __forceinline__ float wave_reduce16(float (&v)[16], int lane) {
    {
        const bool up = lane & 32;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float keep = up ? v[k + 8] : v[k], send = up ? v[k] : v[k + 8];
            v[k] = keep + __shfl_xor(send, 32, 64);
        }
    }
    {
        const bool up = lane & 16;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float keep = up ? v[k + 4] : v[k], send = up ? v[k] : v[k + 4];
            v[k] = keep + __shfl_xor(send, 16, 64);
        }
    }
    {
        const bool up = lane & 8;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float keep = up ? v[k + 2] : v[k], send = up ? v[k] : v[k + 2];
            v[k] = keep + __shfl_xor(send, 8, 64);
        }
    }
    const bool up = lane & 4;
    float r = (up ? v[1] : v[0]) + __shfl_xor(up ? v[0] : v[1], 4, 64);
    r += __shfl_xor(r, 2, 64);
    r += __shfl_xor(r, 1, 64);
    return r;
}

template <int DT>
__host__ __device__ constexpr int sn_fwd_smem() {
    return 2 * 256 * sn_ldr<DT>() * 2 + (16 * DT * 16 + 256) * 4 + SN_W * 16 * sn_ldr<DT>() * 2;
}

// ------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------
// HL (the default arithmetic): the weighted score tile enters the second contraction as bf16 hi + lo parts (two MFMAs per feature
// tile) -- the reference keeps it in fp32 (mhla_dit/mhla/mhla.py:262-268 under mhla_dit/train.py:12-13); false: one bf16 value
// (MHLA_FLAG_BF16_SUMMARIES)
template <int DT, bool GATHER, bool M16 = false, bool HL = true>
__global__ __launch_bounds__(SN_T, SN_T / 256) void k_sn_fwd(const SnArgs a) {
    // GATHER: the launch has a block_index map.  As a template parameter the row lookups carry no branch: with `idx ? idx[p] : p`
    // decided at run time hipcc branched around every map load and waited for ALL loads in flight at each join (s_waitcnt
    // vmcnt(0) after every group of row loads: the staging became a chain of dependent round trips).
    const int* const idx = GATHER ? a.idx : nullptr;
    if constexpr (GATHER) __builtin_assume(idx != nullptr);
    constexpr int DP = DT * 16, LDR = sn_ldr<DT>(), KS = (DP + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Ks = reinterpret_cast<u16*>(smem_raw);          // [N][LDR]
    u16* Vs = Ks + 256 * LDR;                            // [N][LDR]
    float* ksum_s = reinterpret_cast<float*>(Vs + 256 * LDR);   // [M][DP]
    float* zs = ksum_s + 16 * DP;                        // [M][16]
    u16* Ost = reinterpret_cast<u16*>(zs + 256);         // [8 waves][16][LDR] output staging
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, n = lane & 15, kg = lane >> 4;
    // (b,h) pairs in XCD-contiguous order: a token's heads are adjacent in memory (144-byte rows for D = 72: neighbouring heads share
    // cache lines), so the heads of one batch element go to workgroups of ONE XCD, running side by side, and each line is fetched once
    const int bh = xcd_swizzle(blockIdx.x, gridDim.x), b = bh / a.H, h = bh - b * a.H;
    const int M = M16 ? 16 : a.M, D = a.D, N = M * 16;
    const u16* qb = (const u16*)a.q.ptr + b * a.q.sb + h * a.q.sh;
    const u16* kb = (const u16*)a.k.ptr + b * a.k.sb + h * a.k.sh;
    const u16* vb = (const u16*)a.v.ptr + b * a.v.sb + h * a.v.sh;
    u16* ob = (u16*)a.out.ptr + b * a.out.sb + h * a.out.sh;

    __shared__ float Wsh[16 * 17];   // mixing weights (M <= 16): read in the inner loops, so kept off the global-load path
    // the thread's weight is requested FIRST and parked in LDS behind the staging (a wait for the youngest load drains every load)
    const int wi = (tid >> 4) & 15, wj = tid & 15;
    const bool wok = wi < M && wj < M;
    const float wreg = gld<float>(a.W + (long)(wok ? wi : 0) * a.ldw + (wok ? wj : 0));
    // the wave's own Q rows are requested next: they travel while K, V are staged
    uint4 qraw[2][KS];
#pragma unroll
    for (int x = 0; x < 2; ++x) sn_issue_rows<KS>(qraw[x], qb, a.q.sn, idx, min(wave + SN_W * x, M - 1) * 16, D, lane);
    sn_stage2<DT, SN_T, false>(Ks, kb, a.k.sn, a.eps, a.relu != 0, Vs, vb, a.v.sn, idx, N, D, tid);
    if (tid < 256) Wsh[wi * 17 + wj] = wok ? wreg : 0.f;
    __syncthreads();
    if (a.normalize) {
        for (int v = tid; v < M * DP; v += SN_T) {
            const int j = v / DP, d = v - j * DP;
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += bf(Ks[(j * 16 + r) * LDR + d]);
            ksum_s[v] = s;
        }
        __syncthreads();
    }
    // own blocks: i = wave, wave + 4, ...   Q_i rows as MFMA operands; z_i
    bf16x8 qa[2][KS];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const int i = wave + SN_W * x;
        if (i < M) {
            sn_finish_rows<KS>(qa[x], qraw[x], D, a.eps, lane, a.relu != 0);
            if (a.normalize) {
                float z = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const s16x8 qs = __builtin_bit_cast(s16x8, qa[x][ks]);
                    if (ks * 32 + kg * 8 < D) {
#pragma unroll
                        for (int t = 0; t < 8; ++t) z += bf((u16)qs[t]) * ksum_s[i * DP + ks * 32 + kg * 8 + t];
                    }
                }
                z += __shfl_xor(z, 16, 64);
                z += __shfl_xor(z, 32, 64);
                if (kg == 0) zs[i * 16 + n] = z;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const int i = wave + SN_W * x;
        if (i >= M) continue;
        float ninv = 1.f;   // lane n = row s of the block
        if (a.normalize) {
            float nn = a.eps;
            for (int j = 0; j < M; ++j) nn += Wsh[i * 17 + j] * zs[j * 16 + n];
            ninv = 1.f / nn;
        }
        f32x4 acc[DT];
#pragma unroll
        for (int tn = 0; tn < DT; ++tn) acc[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int j0 = 0; j0 < M; j0 += 2) {
            const int j1 = j0 + 1 < M ? j0 + 1 : j0;
            const float w0 = Wsh[i * 17 + j0], w1 = j0 + 1 < M ? Wsh[i * 17 + j0 + 1] : 0.f;
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
            bf16x8 ka[KS], kb2[KS];
            sn_lds_rows<KS>(ka, Ks, LDR, j0 * 16, D, lane);
            sn_lds_rows<KS>(kb2, Ks, LDR, j1 * 16, D, lane);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                c0 = mfma_bf16(ka[ks], qa[x][ks], c0);     // S^T tile (j0, i): rows t, cols s
                c1 = mfma_bf16(kb2[ks], qa[x][ks], c1);
            }
            if constexpr (HL) {
                bf16x8 ph, pl;
                sn_pack_pair_hl(c0 * w0, c1 * w1, ph, pl);
#pragma unroll
                for (int tn = 0; tn < DT; ++tn) {
                    const bf16x8 bv = sn_tr_pair(Vs, LDR, j0 * 16, j1 * 16, tn * 16, lane);
                    acc[tn] = mfma_bf16(ph, bv, acc[tn]);
                    acc[tn] = mfma_bf16(pl, bv, acc[tn]);
                }
            } else {
                const bf16x8 pa = sn_pack_pair(c0 * w0, c1 * w1);
#pragma unroll
                for (int tn = 0; tn < DT; ++tn) acc[tn] = mfma_bf16(pa, sn_tr_pair(Vs, LDR, j0 * 16, j1 * 16, tn * 16, lane), acc[tn]);
            }
        }
        // O rows: C layout lane (col d2 = 16 tn + n, rows s = 4 kg + r); scale by 1/n[s]; stage; coalesced store
        u16* Os = Ost + wave * 16 * LDR;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ni = __shfl(ninv, kg * 4 + r, 64);
#pragma unroll
            for (int tn = 0; tn < DT; ++tn) Os[(kg * 4 + r) * LDR + tn * 16 + n] = cvt_bf16(acc[tn][r] * ni);
        }
        wave_lds_fence();
        const int dv = D >> 3;
        for (int v = lane; v < 16 * dv; v += 64) {
            const int r = v / dv, p = v - r * dv;
            *reinterpret_cast<uint4*>(ob + tok_row(idx, i * 16 + r) * a.out.sn + p * 8) = *reinterpret_cast<const uint4*>(Os + r * LDR + p * 8);
        }
        wave_lds_fence();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// backward: dQ, dK, dV and the per-(b,h) partial of dW, one launch.
//   pass A (wave owns query blocks i, K / V tiles in LDS):  S^T, dP^T tiles -> dW[i][:], dS -> dQ_i ; dksum_i
//   pass B (wave owns key blocks j, Q / dO' tiles in LDS):  S, dP tiles -> P^T dO' = dV_j ; dS^T Q = dK_j
// with S = Q K^T, P = W (.) S, dO' = dO / n, dP = dO' V^T, dS = W (.) dP, dn = -(dO . O) / n, dz = W^T dn.
// ------------------------------------------------------------------------------------------------------------------
template <int DT, int NB = 2>
__host__ __device__ constexpr int sn_bwd_smem() {
    return 2 * 256 * sn_ldr<DT>() * 2 + (2 * 16 * DT * 16 + 5 * 256) * 4 + (NB == 2 ? SN_WB : 2 * SN_WB) * 16 * sn_ldr<DT>() * 2;
}

template <bool MASK>
__device__ __forceinline__ void sn_store16(u16* __restrict__ base, long sn, const int* __restrict__ idx, int row0, int D,
                                           const u16* __restrict__ Os, int ldr, const u16* __restrict__ mbase, long msn, int lane) {
    const int dv = D >> 3;
    for (int v = lane; v < 16 * dv; v += 64) {
        const int r = v / dv, p = v - r * dv;
        const long tr = tok_row(idx, row0 + r);
        uint4 x = *reinterpret_cast<const uint4*>(Os + r * ldr + p * 8);
        if (MASK) x = mask_pos8(x, *reinterpret_cast<const uint4*>(mbase + tr * msn + p * 8));
        *reinterpret_cast<uint4*>(base + tr * sn + p * 8) = x;
    }
}

// M16: the launch has exactly 16 blocks (the DiT 256^2 grid): the block loops carry no run-time guard, straight-line code
// HL (the default arithmetic): no intermediate is rounded to bf16 on its way into a second contraction -- dO stays the exact tensor
// (its 1 / n factor is applied to the fp32 tiles it produces: per column in pass A, per row in pass B) and the weighted score tiles
// P, dS enter the second contractions as bf16 hi + lo parts; false: dO' = dO / n and the tiles as single bf16 values
// NB: blocks per wave.  2: eight waves, wave w owns blocks w and w + 8 and works on them jointly (each fetched operand serves both);
// 1: sixteen waves, one block each -- twice the LDS operand traffic, but four waves per SIMD to overlap each other's LDS read -> MFMA ->
// pack -> MFMA chains (the choice the forward made)
template <int DT, bool GATHER, bool M16 = false, bool HL = true, int NB = 2>
__global__ __launch_bounds__(NB == 2 ? SN_TB : 2 * SN_TB, NB == 2 ? 2 : 4) void k_sn_bwd(const SnArgs a) {
    constexpr int NTH = NB == 2 ? SN_TB : 2 * SN_TB, NWV = NTH / 64;
    // GATHER: the launch has a block_index map.  As a template parameter the row lookups carry no branch: with `idx ? idx[p] : p`
    // decided at run time hipcc branched around every map load and waited for ALL loads in flight at each join (s_waitcnt
    // vmcnt(0) after every group of row loads: the staging became a chain of dependent round trips).
    const int* const idx = GATHER ? a.idx : nullptr;
    if constexpr (GATHER) __builtin_assume(idx != nullptr);
    constexpr int DP = DT * 16, LDR = sn_ldr<DT>(), KS = (DP + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* T0 = reinterpret_cast<u16*>(smem_raw);          // K, later Q          [N][LDR]
    u16* T1 = T0 + 256 * LDR;                            // V, later dO' = dO/n
    float* ksum_s = reinterpret_cast<float*>(T1 + 256 * LDR);   // [M][DP]
    float* dks_s = ksum_s + 16 * DP;                     // [M][DP]
    float* zs = dks_s + 16 * DP;                         // [M][16]
    float* rds = zs + 256;                               // row dots dO . O
    float* nis = rds + 256;                              // 1 / n
    float* dns = nis + 256;                              // dn
    float* dzs = dns + 256;                              // dz
    u16* Ost = reinterpret_cast<u16*>(dzs + 256);        // [NWV waves][16][LDR]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, n = lane & 15, kg = lane >> 4;
    // (b,h) pairs in XCD-contiguous order: a token's heads are adjacent in memory (144-byte rows for D = 72: neighbouring heads share
    // cache lines), so the heads of one batch element go to workgroups of ONE XCD, running side by side, and each line is fetched once
    const int bh = xcd_swizzle(blockIdx.x, gridDim.x), b = bh / a.H, h = bh - b * a.H;
    const int M = M16 ? 16 : a.M, D = a.D, N = M * 16;
    auto base = [&](const View& w) { return (const u16*)w.ptr + b * w.sb + h * w.sh; };
    auto mbase = [&](const MView& w) { return (u16*)w.ptr + b * w.sb + h * w.sh; };
    const u16 *qb = base(a.q), *kb = base(a.k), *vb = base(a.v), *ob = base(a.o), *gb = base(a.dout);
    u16 *dqb = mbase(a.dq), *dkb = mbase(a.dk), *dvb = mbase(a.dv);
    float* dwp = a.dwp + (long)bh * M * M;
    u16* Os = Ost + wave * 16 * LDR;
    // the wave's two blocks; with M <= 8 the second one does not exist: it is clamped onto the first (its results are dropped)
    const int bA = wave, bB = wave + NWV;
    const bool hasA = bA < M, hasB = NB == 2 && bB < M;
    int blk[NB];
    blk[0] = hasA ? bA : 0;
    if constexpr (NB == 2) blk[1] = hasB ? bB : (hasA ? bA : 0);
    auto has = [&](int x) { return x == 0 ? hasA : hasB; };
    auto load_k = [&](bf16x8 (&r)[KS], int j) {
        sn_load_rows<KS>(r, kb, a.k.sn, idx, j * 16, D, a.eps, lane, a.relu != 0);
    };
    // dO rows scaled by 1/n (row = lane & 15), rounded to bf16
    auto scale_dop = [&](bf16x8 (&r)[KS], int i) {
        if (a.normalize) {
            const float ni = nis[i * 16 + n];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const uint4 v = __builtin_bit_cast(uint4, r[ks]);
                unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    w[t] = pack_bf16x2(__uint_as_float(w[t] << 16) * ni, __uint_as_float(w[t] & 0xffff0000u) * ni);
                r[ks] = __builtin_bit_cast(bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
            }
        }
    };

    trace_mark(a.trace, 0);
    // the thread's mixing weight first (parked in LDS behind the staging: a wait for the youngest load drains every load), then
    // the wave's own rows of Q and dO (operands of P2 and of pass A; and, for the row dots, of O): they travel while K, V are staged
    const int wi = (tid >> 4) & 15, wj = tid & 15;
    const bool wok = wi < M && wj < M;
    const float wreg = gld<float>(a.W + (long)(wok ? wi : 0) * a.ldw + (wok ? wj : 0));
    bf16x8 qa[NB][KS], ga[NB][KS];
    uint4 qraw[NB][KS], graw[NB][KS], oraw[NB][KS];
#pragma unroll
    for (int x = 0; x < NB; ++x) {
        sn_issue_rows<KS>(qraw[x], qb, a.q.sn, idx, blk[x] * 16, D, lane);
        sn_issue_rows<KS>(graw[x], gb, a.dout.sn, idx, blk[x] * 16, D, lane);
        if (a.normalize && !HL) sn_issue_rows<KS>(oraw[x], ob, a.o.sn, idx, blk[x] * 16, D, lane);   // (HL: the row dots come out of pass A)
    }
    // ---- P0 / P1: K, V tiles; ksum ----
    __shared__ float Wsh[16 * 17];   // mixing weights (M <= 16), read in every inner loop
    sn_stage2<DT, NTH, false>(T0, kb, a.k.sn, a.eps, a.relu != 0, T1, vb, a.v.sn, idx, N, D, tid);
    if (tid < 256) Wsh[wi * 17 + wj] = wok ? wreg : 0.f;
#pragma unroll
    for (int x = 0; x < NB; ++x) {
        sn_finish_rows<KS>(qa[x], qraw[x], D, a.eps, lane, a.relu != 0);
        sn_finish_rows<KS>(ga[x], graw[x], D, 0.f, lane, false);
    }
    __syncthreads();
    trace_mark(a.trace, 1);
    if (a.normalize) {
        for (int v = tid; v < M * DP; v += NTH) {
            const int j = v / DP, d = v - j * DP;
            float sacc = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc += bf(T0[(j * 16 + r) * LDR + d]);
            ksum_s[v] = sacc;
        }
        __syncthreads();
        trace_mark(a.trace, 2);
        // ---- P2: z_i, row dots (own blocks) ----
#pragma unroll
        for (int x = 0; x < NB; ++x) {
            if (has(x)) {
                const int i = blk[x];
                bf16x8 oa[KS];
                if constexpr (!HL) sn_finish_rows<KS>(oa, oraw[x], D, 0.f, lane, false);
                float z = 0.f, rd = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks * 32 + kg * 8 < D) {
                        const s16x8 qs = __builtin_bit_cast(s16x8, qa[x][ks]), gs = __builtin_bit_cast(s16x8, ga[x][ks]);
#pragma unroll
                        for (int t = 0; t < 8; ++t) z += bf((u16)qs[t]) * ksum_s[i * DP + ks * 32 + kg * 8 + t];
                        if constexpr (!HL) {
                            const s16x8 os = __builtin_bit_cast(s16x8, oa[ks]);
#pragma unroll
                            for (int t = 0; t < 8; ++t) rd += bf((u16)gs[t]) * bf((u16)os[t]);
                        }
                    }
                }
                z += __shfl_xor(z, 16, 64); z += __shfl_xor(z, 32, 64);
                rd += __shfl_xor(rd, 16, 64); rd += __shfl_xor(rd, 32, 64);
                if (kg == 0) { zs[i * 16 + n] = z; rds[i * 16 + n] = rd; }
            }
        }
        __syncthreads();
        trace_mark(a.trace, 3);
        // ---- P3: 1/n, dn ----
        for (int v = tid; v < N; v += NTH) {
            const int i = v >> 4, sx = v & 15;
            float nn = a.eps;
            for (int j = 0; j < M; ++j) nn += Wsh[i * 17 + j] * zs[j * 16 + sx];
            const float ni = 1.f / nn;
            nis[v] = ni;
            dns[v] = -rds[v] * ni;
        }
        __syncthreads();
        // ---- P4: dz = W^T dn ----   (HL: dn = -(dO . O) / n is formed in pass A, from the fp32 score tiles -- the stored, rounded O
        // would cost the row dots 2e-3 -- and dz follows it)
        if constexpr (!HL) {
            for (int v = tid; v < N; v += NTH) {
                const int j = v >> 4, sx = v & 15;
                float dz = 0.f;
                for (int i = 0; i < M; ++i) dz += Wsh[i * 17 + j] * dns[i * 16 + sx];
                dzs[v] = dz;
            }
            __syncthreads();
        }
    }
    trace_mark(a.trace, 4);

    // ---- pass A: dQ_i, dW[i][:] for the wave's two query blocks at once (K, V rows and K^T operands fetched once for both) ----
    {
        if constexpr (!HL) {
#pragma unroll
            for (int x = 0; x < NB; ++x) scale_dop(ga[x], blk[x]);
        }
        // HL: the factor 1 / n_i[s] of dP^T = V_j (dO_i / n_i)^T is a per-column (= per-lane) scale of the fp32 tile
        float niA[NB];
#pragma unroll
        for (int x = 0; x < NB; ++x) niA[x] = (HL && a.normalize) ? nis[blk[x] * 16 + n] : 1.f;
        f32x4 acc[NB][DT];
#pragma unroll
        for (int x = 0; x < NB; ++x)
#pragma unroll
            for (int tn = 0; tn < DT; ++tn) acc[x][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
        float ew[NB][16];   // per-lane partials of dW[i][0..15]
#pragma unroll
        for (int jp = 0; jp < 8; ++jp) {
            const int j0 = 2 * jp;
            #pragma unroll
            for (int x = 0; x < NB; ++x) ew[x][j0] = ew[x][j0 + 1] = 0.f;
            if (j0 < M) {
                const bool has1 = j0 + 1 < M;
                const int j1 = has1 ? j0 + 1 : j0;
                f32x4 s0[NB], s1[NB], p0[NB], p1[NB];
#pragma unroll
                for (int x = 0; x < NB; ++x) s0[x] = s1[x] = p0[x] = p1[x] = f32x4{0.f, 0.f, 0.f, 0.f};
                bf16x8 t0[KS], t1[KS];
                sn_lds_rows<KS>(t0, T0, LDR, j0 * 16, D, lane);
                sn_lds_rows<KS>(t1, T0, LDR, j1 * 16, D, lane);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int x = 0; x < NB; ++x) { s0[x] = mfma_bf16(t0[ks], qa[x][ks], s0[x]); s1[x] = mfma_bf16(t1[ks], qa[x][ks], s1[x]); }   // S^T
                sn_lds_rows<KS>(t0, T1, LDR, j0 * 16, D, lane);
                sn_lds_rows<KS>(t1, T1, LDR, j1 * 16, D, lane);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int x = 0; x < NB; ++x) { p0[x] = mfma_bf16(t0[ks], ga[x][ks], p0[x]); p1[x] = mfma_bf16(t1[ks], ga[x][ks], p1[x]); }   // dP^T
                bf16x8 da[NB], dl[HL ? NB : 1];
#pragma unroll
                for (int x = 0; x < NB; ++x) {
                    const int i = blk[x];
                    if constexpr (HL) { p0[x] *= niA[x]; p1[x] *= niA[x]; }
                    // dW[i][j] = sum(dP . S) + sum_s dn_i[s] z_j[s]: lane partials, reduced once per query block below
                    float e0 = s0[x][0] * p0[x][0] + s0[x][1] * p0[x][1] + s0[x][2] * p0[x][2] + s0[x][3] * p0[x][3];
                    float e1 = s1[x][0] * p1[x][0] + s1[x][1] * p1[x][1] + s1[x][2] * p1[x][2] + s1[x][3] * p1[x][3];
                    const float w0 = Wsh[i * 17 + j0], w1 = has1 ? Wsh[i * 17 + j1] : 0.f;
                    if (!HL && a.normalize && kg == 0) {
                        e0 += dns[i * 16 + n] * zs[j0 * 16 + n];
                        e1 += dns[i * 16 + n] * zs[j1 * 16 + n];
                    }
                    ew[x][j0] = e0;
                    ew[x][j0 + 1] = has1 ? e1 : 0.f;
                    if constexpr (HL) sn_pack_pair_hl(p0[x] * w0, p1[x] * w1, da[x], dl[x]);
                    else da[x] = sn_pack_pair(p0[x] * w0, p1[x] * w1);      // dS^T pair -> A operand (m = s, k-slots = t)
                }
#pragma unroll
                for (int tn = 0; tn < DT; ++tn) {
                    const bf16x8 bk = sn_tr_pair(T0, LDR, j0 * 16, j1 * 16, tn * 16, lane);
#pragma unroll
                    for (int x = 0; x < NB; ++x) acc[x][tn] = mfma_bf16(da[x], bk, acc[x][tn]);
                    if constexpr (HL) {
#pragma unroll
                        for (int x = 0; x < NB; ++x) acc[x][tn] = mfma_bf16(dl[x], bk, acc[x][tn]);
                    }
                }
            }
        }
        // HL: per-lane partials of the row dot (dO . O)[s = n] = sum_j W_ij sum_t S[s][t] dP[s][t] -- the lane's dW partials, weighted
        float rdl[NB] = {};
        if constexpr (HL) {
            asm volatile("" ::: "memory");   // (keeps the 32 weight reads below from being hoisted above the block loop: +32 live registers there, 131 spilled)
#pragma unroll
            for (int x = 0; x < NB; ++x)
#pragma unroll
                for (int j = 0; j < 16; ++j) rdl[x] += Wsh[blk[x] * 17 + j] * ew[x][j];
        }
        float totx[NB];
#pragma unroll
        for (int x = 0; x < NB; ++x) totx[x] = wave_reduce16(ew[x], lane);
        const int jw = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
        if constexpr (HL) {
            if (a.normalize) {   // (uniform)
#pragma unroll
                for (int x = 0; x < NB; ++x) {
                    float rd = rdl[x];
                    rd += __shfl_xor(rd, 16, 64);
                    rd += __shfl_xor(rd, 32, 64);
                    if (has(x) && kg == 0) dns[blk[x] * 16 + n] = -rd * nis[blk[x] * 16 + n];
                }
                __syncthreads();
                for (int v = tid; v < N; v += NTH) {   // dz = W^T dn
                    const int j = v >> 4, sx = v & 15;
                    float dz = 0.f;
                    for (int i = 0; i < M; ++i) dz += Wsh[i * 17 + j] * dns[i * 16 + sx];
                    dzs[v] = dz;
                }
                __syncthreads();
#pragma unroll
                for (int x = 0; x < NB; ++x) {   // dW[i][jw] += <dn_i, z_jw>: the four lanes of a quad take four positions each
                    const int i = blk[x], jc = min(jw, M - 1), s4 = (lane & 3) * 4;
                    float t = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) t += dns[i * 16 + s4 + r] * zs[jc * 16 + s4 + r];
                    t += __shfl_xor(t, 1, 64);
                    t += __shfl_xor(t, 2, 64);
                    totx[x] += t;
                }
            }
        }
#pragma unroll
        for (int x = 0; x < NB; ++x) {
            const int i = blk[x];
            const bool live = has(x);
            const float tot = totx[x];
            if (live && (lane & 3) == 0 && jw < M) dwp[i * M + jw] = tot;
            if (live) {
                // epilogue: + dz (x) ksum ; stage ; masked store
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dz = a.normalize ? dzs[i * 16 + kg * 4 + r] : 0.f;
#pragma unroll
                    for (int tn = 0; tn < DT; ++tn) {
                        const float kk = a.normalize ? ksum_s[i * DP + tn * 16 + n] : 0.f;
                        Os[(kg * 4 + r) * LDR + tn * 16 + n] = cvt_bf16(acc[x][tn][r] + dz * kk);
                    }
                }
                wave_lds_fence();
                if (a.relu) sn_store16<true>(dqb, a.dq.sn, idx, i * 16, D, Os, LDR, qb, a.q.sn, lane);
                else        sn_store16<false>(dqb, a.dq.sn, idx, i * 16, D, Os, LDR, nullptr, 0, lane);
                wave_lds_fence();
            }
        }
    }
    trace_mark(a.trace, 5);
    // the wave's key-block rows for pass B are requested before the tiles are re-staged
    bf16x8 ka[NB][KS], va[NB][KS];
#pragma unroll
    for (int x = 0; x < NB; ++x) {
        load_k(ka[x], blk[x]);
        sn_load_rows<KS>(va[x], vb, a.v.sn, idx, blk[x] * 16, D, 0.f, lane);
    }
    __syncthreads();
    trace_mark(a.trace, 6);

    // ---- P5: Q and dO' tiles replace K and V ----
    // (HL: dO itself is staged; 1 / n scales the rows of the fp32 tiles below)
    if (a.normalize && !HL) sn_stage2<DT, NTH, true>(T0, qb, a.q.sn, a.eps, a.relu != 0, T1, gb, a.dout.sn, idx, N, D, tid, nis);
    else                    sn_stage2<DT, NTH, false>(T0, qb, a.q.sn, a.eps, a.relu != 0, T1, gb, a.dout.sn, idx, N, D, tid);
    __syncthreads();
    trace_mark(a.trace, 7);

    // ---- pass B: dK_j, dV_j for the wave's two key blocks at once (Q, dO' rows and their transposes fetched once for both) ----
    {
        f32x4 accK[NB][DT], accV[NB][DT];
#pragma unroll
        for (int x = 0; x < NB; ++x)
#pragma unroll
            for (int tn = 0; tn < DT; ++tn) accK[x][tn] = accV[x][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int i0 = 0; i0 < M; i0 += 2) {
            const bool has1 = i0 + 1 < M;
            const int i1 = has1 ? i0 + 1 : i0;
            f32x4 s0[NB], s1[NB], p0[NB], p1[NB];
#pragma unroll
            for (int x = 0; x < NB; ++x) s0[x] = s1[x] = p0[x] = p1[x] = f32x4{0.f, 0.f, 0.f, 0.f};
            bf16x8 t0[KS], t1[KS];
            sn_lds_rows<KS>(t0, T0, LDR, i0 * 16, D, lane);
            sn_lds_rows<KS>(t1, T0, LDR, i1 * 16, D, lane);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int x = 0; x < NB; ++x) { s0[x] = mfma_bf16(t0[ks], ka[x][ks], s0[x]); s1[x] = mfma_bf16(t1[ks], ka[x][ks], s1[x]); }   // S: rows s, cols t
            sn_lds_rows<KS>(t0, T1, LDR, i0 * 16, D, lane);
            sn_lds_rows<KS>(t1, T1, LDR, i1 * 16, D, lane);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int x = 0; x < NB; ++x) { p0[x] = mfma_bf16(t0[ks], va[x][ks], p0[x]); p1[x] = mfma_bf16(t1[ks], va[x][ks], p1[x]); }   // dP
            bf16x8 pa[NB], da[NB], pl[HL ? NB : 1], dl[HL ? NB : 1];
            // HL: 1 / n_i[s] of the tiles' rows s = 4 kg + r (dP = diag(1/n) dO V^T; dV = (P diag-scaled)^T dO)
            f32x4 ni0 = {1.f, 1.f, 1.f, 1.f}, ni1 = ni0;
            if constexpr (HL) {
                if (a.normalize) {
                    ni0 = *reinterpret_cast<const f32x4*>(nis + i0 * 16 + kg * 4);
                    ni1 = *reinterpret_cast<const f32x4*>(nis + i1 * 16 + kg * 4);
                }
            }
#pragma unroll
            for (int x = 0; x < NB; ++x) {
                const float w0 = Wsh[i0 * 17 + blk[x]], w1 = has1 ? Wsh[i1 * 17 + blk[x]] : 0.f;
                if constexpr (HL) {
                    const f32x4 f0 = ni0 * w0, f1 = ni1 * w1;
                    sn_pack_pair_hl(s0[x] * f0, s1[x] * f1, pa[x], pl[x]);   // P diag(1/n) pair -> A operand (m = t, k-slots = s)
                    sn_pack_pair_hl(p0[x] * f0, p1[x] * f1, da[x], dl[x]);   // dS pair
                } else {
                    pa[x] = sn_pack_pair(s0[x] * w0, s1[x] * w1);      // P pair  -> A operand (m = t, k-slots = s)
                    da[x] = sn_pack_pair(p0[x] * w0, p1[x] * w1);      // dS pair
                }
            }
#pragma unroll
            for (int tn = 0; tn < DT; ++tn) {
                const bf16x8 bv = sn_tr_pair(T1, LDR, i0 * 16, i1 * 16, tn * 16, lane);
                const bf16x8 bq = sn_tr_pair(T0, LDR, i0 * 16, i1 * 16, tn * 16, lane);
#pragma unroll
                for (int x = 0; x < NB; ++x) {
                    accV[x][tn] = mfma_bf16(pa[x], bv, accV[x][tn]);   // dV += P^T dO'
                    accK[x][tn] = mfma_bf16(da[x], bq, accK[x][tn]);   // dK += dS^T Q
                    if constexpr (HL) {
                        accV[x][tn] = mfma_bf16(pl[x], bv, accV[x][tn]);
                        accK[x][tn] = mfma_bf16(dl[x], bq, accK[x][tn]);
                    }
                }
            }
        }
#pragma unroll
        for (int x = 0; x < NB; ++x) {
            if (!has(x)) continue;
            const int j = blk[x];
            // dksum_j[d] = sum_s dz_j[s] q_j[s][d] as one MFMA per feature tile: every row of the A operand is dz_j (k-slots of the
            // first tile of the pair, hi + lo bf16), the B operand the transposed Q_j tile; all rows of the result are equal
            f32x4 dks[DT];
#pragma unroll
            for (int tn = 0; tn < DT; ++tn) dks[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (a.normalize) {
                f32x4 dz4, dzl;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dz4[r] = dzs[j * 16 + kg * 4 + r];
                    dzl[r] = dz4[r] - bf(cvt_bf16(dz4[r]));
                }
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                const bf16x8 ah = sn_pack_pair(dz4, zero), al = sn_pack_pair(dzl, zero);
#pragma unroll
                for (int tn = 0; tn < DT; ++tn) {
                    const bf16x8 bq = sn_tr_pair(T0, LDR, j * 16, j * 16, tn * 16, lane);
                    dks[tn] = mfma_bf16(ah, bq, dks[tn]);
                    dks[tn] = mfma_bf16(al, bq, dks[tn]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int tn = 0; tn < DT; ++tn) Os[(kg * 4 + r) * LDR + tn * 16 + n] = cvt_bf16(accK[x][tn][r] + dks[tn][0]);
            wave_lds_fence();
            if (a.relu) sn_store16<true>(dkb, a.dk.sn, idx, j * 16, D, Os, LDR, kb, a.k.sn, lane);
            else        sn_store16<false>(dkb, a.dk.sn, idx, j * 16, D, Os, LDR, nullptr, 0, lane);
            wave_lds_fence();
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int tn = 0; tn < DT; ++tn) Os[(kg * 4 + r) * LDR + tn * 16 + n] = cvt_bf16(accV[x][tn][r]);
            wave_lds_fence();
            sn_store16<false>(dvb, a.dv.sn, idx, j * 16, D, Os, LDR, nullptr, 0, lane);
            wave_lds_fence();
        }
    }
    trace_mark(a.trace, 8);
    if (a.trace) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); trace_mark(a.trace, 9); }
}

// dW[i][j] = sum_bh dWp[bh][i][j] : one workgroup per element, fixed-order tree (deterministic)
__global__ __launch_bounds__(256) void k_sn_dw_reduce(const float* __restrict__ dwp, float* __restrict__ dW, int MM, int BH) {
    __shared__ float red[4];
    const int e = blockIdx.x, tid = threadIdx.x;
    float s = 0.f;
    for (int b = tid; b < BH; b += 256) s += dwp[(long)b * MM + e];
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) dW[e] = red[0] + red[1] + red[2] + red[3];
}

}  // namespace fast
}  // namespace mhla
