// Chunk mixing of the causal operator with every chunk of a sequence resident in one workgroup (n <= 256 chunks: 16 384 tokens).
//
//   forward  k_csf_mixf:  P_i[e]  = sum_{j < i} m_ij S_j[e]                         (naive.py:73-75)
//   backward k_csf_mixb:  dS_j[e] = sum_{i > j} m_ij dP_i[e]   and, from the same two tiles,
//                         dmix_ij = sum_e dP_i[e] S_j[e]  (j < i)                   (autograd of naive.py:73-78)
//
// The summaries are [bh][n][tiles][planes][4096] bf16 (causal_bf16.hpp; planes = hi, lo with HL; E = K V logical elements per
// chunk).  A workgroup owns slices of TE logical elements: the slice's rows of ALL chunks (n x planes x 2 TE bytes) sit in LDS,
// so every summary byte is read from HBM exactly once per direction.  One wave per 16 output chunks; the mixing weights of its
// chunks live in its registers as bf16 hi + lo for the whole launch; the summaries enter the MFMA through the hardware
// transpose read as the A operand, so a lane ends up with four consecutive elements of one output chunk (8-byte staging
// writes, full rows out).  HL: x = hi + lo on both sides -- m_hi x_hi + m_lo x_hi + m_hi x_lo, fp32 accumulation, and the
// result is split into hi + lo again on its way out (>= 16 significand bits end to end, as the reference's fp32 loop keeps).
// A workgroup walks `spw` consecutive slices with the next slice's rows in flight in registers while the current one is
// multiplied and stored.
// dmix: the 16 x 16 tiles on and below the diagonal are dealt round-robin to the waves and accumulated over all slices of the
// workgroup -- one [n][n] partial per workgroup, summed in a fixed order by k_dw_reduce<1> (deterministic, no atomics).
#pragma once
#include "causal_bf16.hpp"

namespace mhla {
namespace fast {

// logical elements per slice: the slice's rows of all 16 NW chunks must fit the LDS (forward: input + staging, two workgroups per CU
// up to 128 chunks; backward: dP, S and the dS staging) -- halved for sequences of 129..256 chunks (NW = 16)
// HL: the summaries' format (causal_bf16.hpp): 0 single bf16, 1 bf16 hi + lo planes, 2 h16 (fp16 payload, one multiplier per 16-row strip
// of a chunk tile).  h16 is mixed ON THE PAYLOAD with the fp16 MFMA (as k_sp_mixh of the block-mixing operator, mixh.hpp): the slice goes
// to LDS as it is, the wave's mixing weights are rescaled per slice, w'(i, j) = m_ij mult_j / mult_i as fp16 hi + lo, with the output
// strip's multiplier mult_i = the power of two >= sum_j |m_ij| mult_j (the same in every workgroup that holds a slice of the strip; the
// one with the strip's first slice stores it); dmix_ij += mult_i mult'_j sum_e pay_i[e] pay'_j[e], one MFMA per product.
template <int NW, int HL> __host__ __device__ constexpr int mix_te() { return (HL == 1 ? 64 : 128) / (NW > 8 ? 2 : 1); }

struct CsfMix2Args {
    const float* W;     // mixing matrix [n][ldw]
    int ldw;
    const u16* in;      // forward: S        backward: dP
    const u16* in2;     // backward: S
    u16* out;           // forward: P        backward: dS
    float* dwp;         // backward: [gridDim.x][n][n] partials of dmix
    int n;
    long E;             // logical elements per chunk summary (a multiple of 4096)
    long total;         // slices = bh * E / TE
    int spw;            // slices per workgroup
};

// Rows of one slice: every row is PPR = planes TE / 8 pieces of 16 bytes (the hi pieces, then the lo pieces, 8 KB apart in
// memory); NTHR threads move ROWS rows in ROWS PPR / NTHR passes.  The thread's byte offsets inside a (b,h)'s summaries do not
// depend on the slice: computed once, 32 bits each, added to a wave-uniform base.
template <int NTHR, int ROWS, int TE, int HL>
struct MixRows {
    static constexpr int P = cs_mplanes(HL), PPP = TE / 8, PPR = P * PPP, RPP = NTHR / PPR, NP = ROWS / RPP, MF_LD = TE + 8;
    static_assert(NTHR % PPR == 0 && ROWS % RPP == 0, "row mover: threads must tile the rows");
    uint4 v[NP];
    unsigned goff[NP];
    float m[HL == 2 ? NP : 1];   // h16: the multiplier of the row's strip
    // h16: the slice's strip multipliers; mrel: bytes from the slice's first element to its strip's multiplier in the same chunk tile
    __device__ __forceinline__ void issue_mult(const u16* __restrict__ base, long mrel, int tid) {
        const int c2 = ((tid % PPR) % PPP) * 16;
#pragma unroll
        for (int p = 0; p < NP; ++p) m[p] = gld<float>(reinterpret_cast<const char*>(base) + goff[p] - c2 + mrel);
    }
    // ... into LDS (rows past the last chunk: 0), by the thread with the row's first piece
    __device__ __forceinline__ void commit_mult(float* __restrict__ ms, int n, int tid) const {
        const int r0 = tid / PPR, q = tid % PPR;
        if (q == 0) {
#pragma unroll
            for (int p = 0; p < NP; ++p) ms[p * RPP + r0] = (p * RPP + r0 < n) ? m[p] : 0.f;
        }
    }
    __device__ __forceinline__ void store_mult(u16* __restrict__ base, long mrel, int n, const float* __restrict__ mo, int tid) const {
        const int r0 = tid / PPR, q = tid % PPR;
        if (q == 0) {
#pragma unroll
            for (int p = 0; p < NP; ++p)
                if (p * RPP + r0 < n) gst<float>(reinterpret_cast<char*>(base) + goff[p] + mrel, mo[p * RPP + r0]);
        }
    }
    __device__ __forceinline__ void offsets(long CSZ, int n, int tid) {   // CSZ: u16 elements from one chunk's tile to the next (CsLayout::cst)
        const int r0 = tid / PPR, q = tid % PPR, pl = q / PPP, c = (q % PPP) * 8;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int row = p * RPP + r0;   // rows past the last chunk read the last chunk's row: a valid address
            goff[p] = (unsigned)(((long)(row < n ? row : n - 1) * CSZ + pl * CTE + c) * 2);
        }
    }
    __device__ __forceinline__ void issue(const u16* __restrict__ base) {
#pragma unroll
        for (int p = 0; p < NP; ++p) v[p] = gld_stream16(reinterpret_cast<const char*>(base) + goff[p]);
    }
    // tiles: [P][ROWS][MF_LD]
    __device__ __forceinline__ void commit(u16* __restrict__ tiles, int n, int tid) const {
        const int r0 = tid / PPR, q = tid % PPR, pl = q / PPP, c = (q % PPP) * 8;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int row = p * RPP + r0;
            const bool ok = row < n;   // rows past the last chunk: zeros (their weights are zero too, but 0 x NaN is not)
            *reinterpret_cast<uint4*>(tiles + (pl * ROWS + row) * MF_LD + c) = make_uint4(ok ? v[p].x : 0u, ok ? v[p].y : 0u, ok ? v[p].z : 0u, ok ? v[p].w : 0u);
        }
    }
    __device__ __forceinline__ void store(u16* __restrict__ base, int n, const u16* __restrict__ tiles, int tid) const {
        const int r0 = tid / PPR, q = tid % PPR, pl = q / PPP, c = (q % PPP) * 8;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int row = p * RPP + r0;
            if (row < n) gst<uint4>(reinterpret_cast<char*>(base) + goff[p], *reinterpret_cast<const uint4*>(tiles + (pl * ROWS + row) * MF_LD + c));
        }
    }
};
// offset (u16 elements) of slice s inside the summaries: (b,h) s / nsl, logical elements (s % nsl) TE ..
template <int TE, int HL>
__device__ __forceinline__ long mix_slice_off(long s, long nsl, const CsLayout& L) {
    const long bh = s / nsl, e0 = (s - bh * nsl) * TE;
    return bh * L.bhs + (e0 / CTE) * L.ts + (e0 % CTE);
}
// h16: bytes from slice s's first element to its strip's multiplier (the four floats behind the chunk tile's payload plane), and whether
// the slice is the first of its strip (its workgroup stores the output strip's multiplier)
template <int TE>
__device__ __forceinline__ long mix_mult_rel(long s, long nsl, bool& first) {
    const long t0 = ((s % nsl) * TE) % CTE;   // the slice's first element inside its tile
    first = (t0 % (16 * CS)) == 0;
    return (CTE - t0) * 2 + (t0 / (16 * CS)) * 4;
}
// transposed-product accumulators (lane: elements 16 t + 4 kg .. + 3 of chunk row0 + nl) -> staging tiles [P][ROWS][MF_LD]
template <int TE, int ROWS, int HL>
__device__ __forceinline__ void mix_stage4(u16* __restrict__ tiles, const f32x4& acc, int row, int col) {
    if constexpr (HL == 2) {   // (the accumulators are the output payload)
        *reinterpret_cast<uint2*>(tiles + row * (TE + 8) + col) = make_uint2(h16_pack2(acc[0], acc[1]), h16_pack2(acc[2], acc[3]));
        return;
    }
    unsigned h0, h1, l0, l1;
    split_pack2(acc[0], acc[1], h0, l0);
    split_pack2(acc[2], acc[3], h1, l1);
    *reinterpret_cast<uint2*>(tiles + row * (TE + 8) + col) = make_uint2(h0, h1);
    if constexpr (HL == 1) *reinterpret_cast<uint2*>(tiles + (ROWS + row) * (TE + 8) + col) = make_uint2(l0, l1);
}
__device__ __forceinline__ void mix_split(const float (&w)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const __bf16 h = (__bf16)w[t];
        hi[t] = h;
        lo[t] = (__bf16)(w[t] - (float)h);
    }
}
// h16: the output multiplier of the lane's chunk for this slice -- the power of two >= sum_k |w[k]| mult[k] (the four kg lanes of a chunk
// agree) -- and one reduction step's weights as fp16 hi + lo B operands, w'[t] = w[t] mult[k-row] / mult_out (ms = the input rows'
// multipliers in LDS; they are read again per step rather than kept: 8 NK registers less)
template <int NK>
__device__ __forceinline__ float mix_h16_out_mult(const float (&w)[NK][8], const float* __restrict__ ms, int kg) {
    float beta = 0.f;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        const f32x4 m0 = *reinterpret_cast<const f32x4*>(ms + ks * 32 + kg * 8), m1 = *reinterpret_cast<const f32x4*>(ms + ks * 32 + kg * 8 + 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) beta += fabsf(w[ks][t] * m0[t]) + fabsf(w[ks][4 + t] * m1[t]);
    }
    beta += __shfl_xor(beta, 16, 64);
    beta += __shfl_xor(beta, 32, 64);
    return h16_mult_from_bound(beta);
}
__device__ __forceinline__ void mix_h16_step_weights(const float (&w)[8], const float* __restrict__ ms8, float oinv, f16x8& wh, f16x8& wl) {
    const f32x4 m0 = *reinterpret_cast<const f32x4*>(ms8), m1 = *reinterpret_cast<const f32x4*>(ms8 + 4);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const float x = w[t] * (t < 4 ? m0[t & 3] : m1[t & 3]) * oinv;
        const _Float16 h = (_Float16)x;
        wh[t] = h;
        wl[t] = (_Float16)(x - (float)h);
    }
}
template <int NW, int HL> __host__ __device__ constexpr int mixf_smem() { return 2 * cs_mplanes(HL) * 16 * NW * (mix_te<NW, HL>() + 8) * 2 + (HL == 2 ? 2 * 16 * NW * 4 : 0); }

// NW waves = 16 NW chunk rows (n <= 16 NW); NK = reduction steps of 32 chunks
template <int NW, int HL>
__global__ __launch_bounds__(64 * NW, 4) void k_csf_mixf(const CsfMix2Args a) {
    constexpr int TE = mix_te<NW, HL>(), P = cs_mplanes(HL), NK = (NW + 1) / 2, MF_LD = TE + 8, NT = TE / 16, ROWS = 16 * NW;
    constexpr int TB = NT < 4 ? NT : 4;   // element tiles multiplied together
    constexpr bool H16 = HL == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Ts = reinterpret_cast<u16*>(smem_raw);   // [P][ROWS][MF_LD]
    u16* Os = Ts + P * ROWS * MF_LD;              // [P][ROWS][MF_LD]
    float* ms = reinterpret_cast<float*>(Os + P * ROWS * MF_LD);   // h16: the input rows' strip multipliers [ROWS], then the output rows'
    float* mo = ms + ROWS;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    const int n = a.n;
    const long nsl = a.E / TE;
    const CsLayout L = cs_layout(n, a.E, P);
    const long CSZ = L.cst;
    const long s0 = (long)blockIdx.x * a.spw;
    const int cnt = (int)min((long)a.spw, a.total - s0);
    if (cnt <= 0) return;
    // B operand: B[k = j][n = i] = m_ij for the wave's output chunks i = 16 wave + nl, j = 32 ks + 8 kg + t, j < i
    bf16x8 wh[H16 ? 1 : NK], wl[H16 ? 1 : NK];
    float wf[H16 ? NK : 1][8];
    const int irow = wave * 16 + nl;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        float w[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int j = ks * 32 + kg * 8 + t;
            const bool ok = irow < n && j < irow;
            const float x = gld<float>(a.W + (long)(ok ? irow : 0) * a.ldw + (ok ? j : 0));
            w[t] = ok ? x : 0.f;
        }
        if constexpr (H16) {
#pragma unroll
            for (int t = 0; t < 8; ++t) wf[ks][t] = w[t];
        } else {
            mix_split(w, wh[ks], wl[ks]);
        }
    }
    MixRows<64 * NW, ROWS, TE, HL> pre;
    pre.offsets(CSZ, n, tid);
    bool fst = false, pfst = false;
    long mrel = 0, pmrel = 0;
    pre.issue(a.in + mix_slice_off<TE, HL>(s0, nsl, L));
    if constexpr (H16) {
        mrel = mix_mult_rel<TE>(s0, nsl, fst);
        pre.issue_mult(a.in + mix_slice_off<TE, HL>(s0, nsl, L), mrel, tid);
    }
    const int kmax = min(wave / 2, (n - 1) / 32);   // last reduction step with a chunk j < i for this wave's rows
    for (int it = 0; it < cnt; ++it) {
        const long off = mix_slice_off<TE, HL>(s0 + it, nsl, L);
        pre.commit(Ts, n, tid);
        if constexpr (H16) {
            pre.commit_mult(ms, n, tid);
            pmrel = mrel;
            pfst = fst;
        }
        __syncthreads();
        if (it + 1 < cnt) {
            pre.issue(a.in + mix_slice_off<TE, HL>(s0 + it + 1, nsl, L));
            if constexpr (H16) {
                mrel = mix_mult_rel<TE>(s0 + it + 1, nsl, fst);
                pre.issue_mult(a.in + mix_slice_off<TE, HL>(s0 + it + 1, nsl, L), mrel, tid);
            }
        }
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (H16) {
            const float om = mix_h16_out_mult<NK>(wf, ms, kg), oinv = h16_inv(om);
            if (kg == 0) mo[wave * 16 + nl] = om;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                if (ks <= kmax) {
                    f16x8 hh, hl;
                    mix_h16_step_weights(wf[ks], ms + ks * 32 + kg * 8, oinv, hh, hl);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const f16x8 sv = as_f16x8(tr_read8(Ts, MF_LD, ks * 32, t * 16, lane));
                        acc[t] = mfma_f16(sv, hh, acc[t]);
                        acc[t] = mfma_f16(sv, hl, acc[t]);
                    }
                }
            }
        } else {
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            if (ks <= kmax) {
#pragma unroll
                for (int t4 = 0; t4 < NT; t4 += TB) {
                    bf16x8 sv[TB], sl[TB];
#pragma unroll
                    for (int t = 0; t < TB; ++t) sv[t] = tr_read8(Ts, MF_LD, ks * 32, (t4 + t) * 16, lane);
                    if constexpr (HL == 1) {
#pragma unroll
                        for (int t = 0; t < TB; ++t) sl[t] = tr_read8(Ts + ROWS * MF_LD, MF_LD, ks * 32, (t4 + t) * 16, lane);
                    }
#pragma unroll
                    for (int t = 0; t < TB; ++t) acc[t4 + t] = mfma_bf16(sv[t], wh[ks], acc[t4 + t]);
#pragma unroll
                    for (int t = 0; t < TB; ++t) acc[t4 + t] = mfma_bf16(sv[t], wl[ks], acc[t4 + t]);
                    if constexpr (HL == 1) {
#pragma unroll
                        for (int t = 0; t < TB; ++t) acc[t4 + t] = mfma_bf16(sl[t], wh[ks], acc[t4 + t]);
                    }
                }
            }
        }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) mix_stage4<TE, ROWS, HL>(Os, acc[t], wave * 16 + nl, t * 16 + kg * 4);
        __syncthreads();
        pre.store(a.out + off, n, Os, tid);
        if constexpr (H16) {
            if (pfst) pre.store_mult(a.out + off, pmrel, n, mo, tid);   // (uniform)
        }
    }
}

// lower-triangular 16 x 16 tile number idx -> (row tile, column tile)
__device__ __forceinline__ void tri_tile(int idx, int& it, int& jt) {
    it = 0;
    while (idx > it) { idx -= it + 1; ++it; }
    jt = idx;
}

// Backward: 2 NW waves.  Waves [0, NW) form dS for their 16 chunks j (as the forward does for P); waves [NW, 2 NW) accumulate
// the dmix tiles -- two roles on disjoint register budgets (the one-role version needed ~150 VGPRs and spilled at the 128 that
// four waves per SIMD leave), multiplying side by side between the same two barriers.  Every thread helps moving the rows.
template <int NW, int HL> __host__ __device__ constexpr int mixb_smem() { return 3 * cs_mplanes(HL) * 16 * NW * (mix_te<NW, HL>() + 8) * 2 + (HL == 2 ? 3 * 16 * NW * 4 : 0); }

// ROLE 0: both roles in one workgroup of 2 NW waves (NW <= 8).  Sequences of 129..256 chunks (NW = 16) would need 32 waves: the two
// roles run as two launches of NW waves each, ROLE 1 (dS) and ROLE 2 (dmix); dP is then read twice.
template <int NW, int HL, int ROLE = 0>
__global__ __launch_bounds__((ROLE == 0 ? 128 : 64) * NW) void k_csf_mixb(const CsfMix2Args a) {
    constexpr int TE = mix_te<NW, HL>(), P = cs_mplanes(HL), NK = (NW + 1) / 2, MF_LD = TE + 8, NT = TE / 16, ROWS = 16 * NW;
    constexpr int TB = NT < 4 ? NT : 4, NTHR = (ROLE == 0 ? 128 : 64) * NW;
    constexpr bool H16 = HL == 2;
    static_assert(ROLE != 0 || NW <= 8, "both roles in one workgroup: at most 16 waves");
    constexpr int NTL = NW * (NW + 1) / 2, TPW = (NTL + NW - 1) / NW;   // dmix tiles, tiles per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Tp = reinterpret_cast<u16*>(smem_raw);   // [P][ROWS][MF_LD] dP rows of the slice
    u16* Tq = Tp + P * ROWS * MF_LD;               // S rows of the slice
    u16* Os = Tq + P * ROWS * MF_LD;               // dS staging
    float* msp = reinterpret_cast<float*>(Os + P * ROWS * MF_LD);   // h16: strip multipliers of the dP rows, of the S rows, of the dS rows
    float* msq = msp + ROWS;
    float* mo = msq + ROWS;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    const bool mixer = ROLE == 1 || (ROLE == 0 && wave < NW);   // (uniform) dS role; the others: dmix role
    const int rw = (ROLE == 0 && !mixer) ? wave - NW : wave;
    const int n = a.n;
    const long nsl = a.E / TE;
    const CsLayout L = cs_layout(n, a.E, P);
    const long CSZ = L.cst;
    const long s0 = (long)blockIdx.x * a.spw;
    const int cnt = (int)min((long)a.spw, a.total - s0);
    float* part = a.dwp + (long)blockIdx.x * n * n;
    MixRows<NTHR, ROWS, TE, HL> pp, pq;
    pp.offsets(CSZ, n, tid);
#pragma unroll
    for (int p = 0; p < pp.NP; ++p) pq.goff[p] = pp.goff[p];
    bool fst = false, pfst = false;
    long mrel = 0, pmrel = 0;
    auto issue_all = [&](long s) __attribute__((always_inline)) {
        const long off = mix_slice_off<TE, HL>(s, nsl, L);
        pp.issue(a.in + off);
        if constexpr (ROLE != 1) pq.issue(a.in2 + off);
        if constexpr (H16) {
            mrel = mix_mult_rel<TE>(s, nsl, fst);
            pp.issue_mult(a.in + off, mrel, tid);
            if constexpr (ROLE != 1) pq.issue_mult(a.in2 + off, mrel, tid);
        }
    };
    auto commit_all = [&]() __attribute__((always_inline)) {
        pp.commit(Tp, n, tid);
        if constexpr (ROLE != 1) pq.commit(Tq, n, tid);
        if constexpr (H16) {
            pp.commit_mult(msp, n, tid);
            if constexpr (ROLE != 1) pq.commit_mult(msq, n, tid);
            pmrel = mrel;
            pfst = fst;
        }
    };
    if (cnt > 0) issue_all(s0);
    if (mixer) {
        // B operand: B[k = i][n = j] = m_ij for the wave's chunks j = 16 rw + nl, i = 32 ks + 8 kg + t, i > j
        bf16x8 wh[H16 ? 1 : NK], wl[H16 ? 1 : NK];
        float wf[H16 ? NK : 1][8];
        const int jrow = rw * 16 + nl;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            float w[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int i = ks * 32 + kg * 8 + t;
                const bool ok = i < n && i > jrow;
                const float x = gld<float>(a.W + (long)(ok ? i : 0) * a.ldw + (ok ? jrow : 0));
                w[t] = ok ? x : 0.f;
            }
            if constexpr (H16) {
#pragma unroll
                for (int t = 0; t < 8; ++t) wf[ks][t] = w[t];
            } else {
                mix_split(w, wh[ks], wl[ks]);
            }
        }
        const int kmin = rw / 2, kend = (n + 31) / 32;   // reduction steps that hold a chunk i > j for this wave's rows
        for (int it = 0; it < cnt; ++it) {
            commit_all();
            __syncthreads();
            if (it + 1 < cnt) issue_all(s0 + it + 1);
            if constexpr (H16) {
                const float om = mix_h16_out_mult<NK>(wf, msp, kg), oinv = h16_inv(om);
                if (kg == 0) mo[rw * 16 + nl] = om;
                f32x4 acc[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) {
                    if (ks >= kmin && ks < kend) {
                        f16x8 hh, hl;
                        mix_h16_step_weights(wf[ks], msp + ks * 32 + kg * 8, oinv, hh, hl);
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const f16x8 sv = as_f16x8(tr_read8(Tp, MF_LD, ks * 32, t * 16, lane));
                            acc[t] = mfma_f16(sv, hh, acc[t]);
                            acc[t] = mfma_f16(sv, hl, acc[t]);
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) mix_stage4<TE, ROWS, HL>(Os, acc[t], rw * 16 + nl, t * 16 + kg * 4);
            } else {
#pragma unroll
            for (int t4 = 0; t4 < NT; t4 += TB) {
                f32x4 acc[TB];
#pragma unroll
                for (int t = 0; t < TB; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) {
                    if (ks >= kmin && ks < kend) {
                        bf16x8 sv[TB], sl[TB];
#pragma unroll
                        for (int t = 0; t < TB; ++t) sv[t] = tr_read8(Tp, MF_LD, ks * 32, (t4 + t) * 16, lane);
                        if constexpr (HL == 1) {
#pragma unroll
                            for (int t = 0; t < TB; ++t) sl[t] = tr_read8(Tp + ROWS * MF_LD, MF_LD, ks * 32, (t4 + t) * 16, lane);
                        }
#pragma unroll
                        for (int t = 0; t < TB; ++t) acc[t] = mfma_bf16(sv[t], wh[ks], acc[t]);
#pragma unroll
                        for (int t = 0; t < TB; ++t) acc[t] = mfma_bf16(sv[t], wl[ks], acc[t]);
                        if constexpr (HL == 1) {
#pragma unroll
                            for (int t = 0; t < TB; ++t) acc[t] = mfma_bf16(sl[t], wh[ks], acc[t]);
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < TB; ++t) mix_stage4<TE, ROWS, HL>(Os, acc[t], rw * 16 + nl, (t4 + t) * 16 + kg * 4);
            }
            }
            __syncthreads();
            pp.store(a.out + mix_slice_off<TE, HL>(s0 + it, nsl, L), n, Os, tid);
            if constexpr (H16) {
                if (pfst) pp.store_mult(a.out + mix_slice_off<TE, HL>(s0 + it, nsl, L), pmrel, n, mo, tid);   // (uniform)
            }
        }
    } else {
        int tit[TPW], tjt[TPW];
        f32x4 dacc[TPW];
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const int idx = rw + u * NW;
            tri_tile(idx < NTL ? idx : 0, tit[u], tjt[u]);
            if (idx >= NTL || tit[u] * 16 >= n) tit[u] = -1;   // no such tile, or beyond the last chunk
            dacc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int it = 0; it < cnt; ++it) {
            commit_all();
            __syncthreads();
            if (it + 1 < cnt) issue_all(s0 + it + 1);
            // dmix tiles: A[m = i][k = e] = dP_i[e], B[k = e][n = j] = S_j[e], both 16-byte row reads (HL: hi hi + hi lo + lo hi)
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                if (tit[u] >= 0) {
                    const u16* ap = Tp + (tit[u] * 16 + nl) * MF_LD + kg * 8;
                    const u16* bp = Tq + (tjt[u] * 16 + nl) * MF_LD + kg * 8;
                    if constexpr (H16) {   // on the payloads, scaled by the two strips' multipliers
                        f32x4 tmp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int ks = 0; ks < TE / 32; ++ks)
                            tmp = mfma_f16(as_f16x8(*reinterpret_cast<const bf16x8*>(ap + ks * 32)), as_f16x8(*reinterpret_cast<const bf16x8*>(bp + ks * 32)), tmp);
                        const f32x4 mi = *reinterpret_cast<const f32x4*>(msp + tit[u] * 16 + kg * 4);
                        const float mj = msq[tjt[u] * 16 + nl];
#pragma unroll
                        for (int r = 0; r < 4; ++r) dacc[u][r] += tmp[r] * (mi[r] * mj);
                    } else {
#pragma unroll
                    for (int ks = 0; ks < TE / 32; ++ks) {
                        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ap + ks * 32), bh_ = *reinterpret_cast<const bf16x8*>(bp + ks * 32);
                        dacc[u] = mfma_bf16(ah, bh_, dacc[u]);
                        if constexpr (HL == 1) {
                            const bf16x8 al = *reinterpret_cast<const bf16x8*>(ap + ROWS * MF_LD + ks * 32);
                            const bf16x8 bl = *reinterpret_cast<const bf16x8*>(bp + ROWS * MF_LD + ks * 32);
                            dacc[u] = mfma_bf16(ah, bl, dacc[u]);
                            dacc[u] = mfma_bf16(al, bh_, dacc[u]);
                        }
                    }
                    }
                }
            }
            __syncthreads();
            if constexpr (ROLE == 0) {
                pp.store(a.out + mix_slice_off<TE, HL>(s0 + it, nsl, L), n, Os, tid);
                if constexpr (H16) {
                    if (pfst) pp.store_mult(a.out + mix_slice_off<TE, HL>(s0 + it, nsl, L), pmrel, n, mo, tid);   // (uniform)
                }
            }
        }
        // the workgroup's partial of dmix: C[m = i][n = j], lane (i = 16 it + 4 kg + r, j = 16 jt + nl); entries with j >= i are never read
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            if (tit[u] >= 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = tit[u] * 16 + kg * 4 + r, j = tjt[u] * 16 + nl;
                    if (i < n && j < n) part[(long)i * n + j] = dacc[u][r];
                }
            }
        }
    }
}

}  // namespace fast
}  // namespace mhla
