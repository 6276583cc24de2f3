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
template <int NW, bool HL> __host__ __device__ constexpr int mix_te() { return (HL ? 64 : 128) / (NW > 8 ? 2 : 1); }

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
template <int NTHR, int ROWS, int TE, bool HL>
struct MixRows {
    static constexpr int P = HL ? 2 : 1, PPP = TE / 8, PPR = P * PPP, RPP = NTHR / PPR, NP = ROWS / RPP, MF_LD = TE + 8;
    static_assert(NTHR % PPR == 0 && ROWS % RPP == 0, "row mover: threads must tile the rows");
    uint4 v[NP];
    unsigned goff[NP];
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
template <int TE, bool HL>
__device__ __forceinline__ long mix_slice_off(long s, long nsl, const CsLayout& L) {
    const long bh = s / nsl, e0 = (s - bh * nsl) * TE;
    return bh * L.bhs + (e0 / CTE) * L.ts + (e0 % CTE);
}
// transposed-product accumulators (lane: elements 16 t + 4 kg .. + 3 of chunk row0 + nl) -> staging tiles [P][ROWS][MF_LD]
template <int TE, int ROWS, bool HL>
__device__ __forceinline__ void mix_stage4(u16* __restrict__ tiles, const f32x4& acc, int row, int col) {
    unsigned h0, h1, l0, l1;
    split_pack2(acc[0], acc[1], h0, l0);
    split_pack2(acc[2], acc[3], h1, l1);
    *reinterpret_cast<uint2*>(tiles + row * (TE + 8) + col) = make_uint2(h0, h1);
    if constexpr (HL) *reinterpret_cast<uint2*>(tiles + (ROWS + row) * (TE + 8) + col) = make_uint2(l0, l1);
}
__device__ __forceinline__ void mix_split(const float (&w)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const __bf16 h = (__bf16)w[t];
        hi[t] = h;
        lo[t] = (__bf16)(w[t] - (float)h);
    }
}

template <int NW, bool HL> __host__ __device__ constexpr int mixf_smem() { return 2 * (HL ? 2 : 1) * 16 * NW * (mix_te<NW, HL>() + 8) * 2; }

// NW waves = 16 NW chunk rows (n <= 16 NW); NK = reduction steps of 32 chunks
template <int NW, bool HL>
__global__ __launch_bounds__(64 * NW, 4) void k_csf_mixf(const CsfMix2Args a) {
    constexpr int TE = mix_te<NW, HL>(), P = HL ? 2 : 1, NK = (NW + 1) / 2, MF_LD = TE + 8, NT = TE / 16, ROWS = 16 * NW;
    constexpr int TB = NT < 4 ? NT : 4;   // element tiles multiplied together
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Ts = reinterpret_cast<u16*>(smem_raw);   // [P][ROWS][MF_LD]
    u16* Os = Ts + P * ROWS * MF_LD;              // [P][ROWS][MF_LD]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    const int n = a.n;
    const long nsl = a.E / TE;
    const CsLayout L = cs_layout(n, a.E, P);
    const long CSZ = L.cst;
    const long s0 = (long)blockIdx.x * a.spw;
    const int cnt = (int)min((long)a.spw, a.total - s0);
    if (cnt <= 0) return;
    // B operand: B[k = j][n = i] = m_ij for the wave's output chunks i = 16 wave + nl, j = 32 ks + 8 kg + t, j < i
    bf16x8 wh[NK], wl[NK];
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
        mix_split(w, wh[ks], wl[ks]);
    }
    MixRows<64 * NW, ROWS, TE, HL> pre;
    pre.offsets(CSZ, n, tid);
    pre.issue(a.in + mix_slice_off<TE, HL>(s0, nsl, L));
    const int kmax = min(wave / 2, (n - 1) / 32);   // last reduction step with a chunk j < i for this wave's rows
    for (int it = 0; it < cnt; ++it) {
        const long off = mix_slice_off<TE, HL>(s0 + it, nsl, L);
        pre.commit(Ts, n, tid);
        __syncthreads();
        if (it + 1 < cnt) pre.issue(a.in + mix_slice_off<TE, HL>(s0 + it + 1, nsl, L));
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            if (ks <= kmax) {
#pragma unroll
                for (int t4 = 0; t4 < NT; t4 += TB) {
                    bf16x8 sv[TB], sl[TB];
#pragma unroll
                    for (int t = 0; t < TB; ++t) sv[t] = tr_read8(Ts, MF_LD, ks * 32, (t4 + t) * 16, lane);
                    if constexpr (HL) {
#pragma unroll
                        for (int t = 0; t < TB; ++t) sl[t] = tr_read8(Ts + ROWS * MF_LD, MF_LD, ks * 32, (t4 + t) * 16, lane);
                    }
#pragma unroll
                    for (int t = 0; t < TB; ++t) acc[t4 + t] = mfma_bf16(sv[t], wh[ks], acc[t4 + t]);
#pragma unroll
                    for (int t = 0; t < TB; ++t) acc[t4 + t] = mfma_bf16(sv[t], wl[ks], acc[t4 + t]);
                    if constexpr (HL) {
#pragma unroll
                        for (int t = 0; t < TB; ++t) acc[t4 + t] = mfma_bf16(sl[t], wh[ks], acc[t4 + t]);
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) mix_stage4<TE, ROWS, HL>(Os, acc[t], wave * 16 + nl, t * 16 + kg * 4);
        __syncthreads();
        pre.store(a.out + off, n, Os, tid);
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
template <int NW, bool HL> __host__ __device__ constexpr int mixb_smem() { return 3 * (HL ? 2 : 1) * 16 * NW * (mix_te<NW, HL>() + 8) * 2; }

// ROLE 0: both roles in one workgroup of 2 NW waves (NW <= 8).  Sequences of 129..256 chunks (NW = 16) would need 32 waves: the two
// roles run as two launches of NW waves each, ROLE 1 (dS) and ROLE 2 (dmix); dP is then read twice.
template <int NW, bool HL, int ROLE = 0>
__global__ __launch_bounds__((ROLE == 0 ? 128 : 64) * NW) void k_csf_mixb(const CsfMix2Args a) {
    constexpr int TE = mix_te<NW, HL>(), P = HL ? 2 : 1, NK = (NW + 1) / 2, MF_LD = TE + 8, NT = TE / 16, ROWS = 16 * NW;
    constexpr int TB = NT < 4 ? NT : 4, NTHR = (ROLE == 0 ? 128 : 64) * NW;
    static_assert(ROLE != 0 || NW <= 8, "both roles in one workgroup: at most 16 waves");
    constexpr int NTL = NW * (NW + 1) / 2, TPW = (NTL + NW - 1) / NW;   // dmix tiles, tiles per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Tp = reinterpret_cast<u16*>(smem_raw);   // [P][ROWS][MF_LD] dP rows of the slice
    u16* Tq = Tp + P * ROWS * MF_LD;               // S rows of the slice
    u16* Os = Tq + P * ROWS * MF_LD;               // dS staging
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
    if (cnt > 0) {
        pp.issue(a.in + mix_slice_off<TE, HL>(s0, nsl, L));
        if constexpr (ROLE != 1) pq.issue(a.in2 + mix_slice_off<TE, HL>(s0, nsl, L));
    }
    if (mixer) {
        // B operand: B[k = i][n = j] = m_ij for the wave's chunks j = 16 rw + nl, i = 32 ks + 8 kg + t, i > j
        bf16x8 wh[NK], wl[NK];
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
            mix_split(w, wh[ks], wl[ks]);
        }
        const int kmin = rw / 2, kend = (n + 31) / 32;   // reduction steps that hold a chunk i > j for this wave's rows
        for (int it = 0; it < cnt; ++it) {
            pp.commit(Tp, n, tid);
            if constexpr (ROLE != 1) pq.commit(Tq, n, tid);
            __syncthreads();
            if (it + 1 < cnt) {
                pp.issue(a.in + mix_slice_off<TE, HL>(s0 + it + 1, nsl, L));
                if constexpr (ROLE != 1) pq.issue(a.in2 + mix_slice_off<TE, HL>(s0 + it + 1, nsl, L));
            }
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
                        if constexpr (HL) {
#pragma unroll
                            for (int t = 0; t < TB; ++t) sl[t] = tr_read8(Tp + ROWS * MF_LD, MF_LD, ks * 32, (t4 + t) * 16, lane);
                        }
#pragma unroll
                        for (int t = 0; t < TB; ++t) acc[t] = mfma_bf16(sv[t], wh[ks], acc[t]);
#pragma unroll
                        for (int t = 0; t < TB; ++t) acc[t] = mfma_bf16(sv[t], wl[ks], acc[t]);
                        if constexpr (HL) {
#pragma unroll
                            for (int t = 0; t < TB; ++t) acc[t] = mfma_bf16(sl[t], wh[ks], acc[t]);
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < TB; ++t) mix_stage4<TE, ROWS, HL>(Os, acc[t], rw * 16 + nl, (t4 + t) * 16 + kg * 4);
            }
            __syncthreads();
            pp.store(a.out + mix_slice_off<TE, HL>(s0 + it, nsl, L), n, Os, tid);
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
            pp.commit(Tp, n, tid);
            pq.commit(Tq, n, tid);
            __syncthreads();
            if (it + 1 < cnt) {
                pp.issue(a.in + mix_slice_off<TE, HL>(s0 + it + 1, nsl, L));
                pq.issue(a.in2 + mix_slice_off<TE, HL>(s0 + it + 1, nsl, L));
            }
            // dmix tiles: A[m = i][k = e] = dP_i[e], B[k = e][n = j] = S_j[e], both 16-byte row reads (HL: hi hi + hi lo + lo hi)
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                if (tit[u] >= 0) {
                    const u16* ap = Tp + (tit[u] * 16 + nl) * MF_LD + kg * 8;
                    const u16* bp = Tq + (tjt[u] * 16 + nl) * MF_LD + kg * 8;
#pragma unroll
                    for (int ks = 0; ks < TE / 32; ++ks) {
                        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ap + ks * 32), bh_ = *reinterpret_cast<const bf16x8*>(bp + ks * 32);
                        dacc[u] = mfma_bf16(ah, bh_, dacc[u]);
                        if constexpr (HL) {
                            const bf16x8 al = *reinterpret_cast<const bf16x8*>(ap + ROWS * MF_LD + ks * 32);
                            const bf16x8 bl = *reinterpret_cast<const bf16x8*>(bp + ROWS * MF_LD + ks * 32);
                            dacc[u] = mfma_bf16(ah, bl, dacc[u]);
                            dacc[u] = mfma_bf16(al, bh_, dacc[u]);
                        }
                    }
                }
            }
            __syncthreads();
            if constexpr (ROLE == 0) pp.store(a.out + mix_slice_off<TE, HL>(s0 + it, nsl, L), n, Os, tid);
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
