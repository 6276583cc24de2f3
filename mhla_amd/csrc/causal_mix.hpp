// Chunk mixing of the causal operator with every chunk of a sequence resident in one workgroup (n <= 128 chunks).
//
//   forward  k_csf_mixf:  P_i[e]  = sum_{j < i} m_ij S_j[e]                         (naive.py:73-75)
//   backward k_csf_mixb:  dS_j[e] = sum_{i > j} m_ij dP_i[e]   and, from the same two tiles,
//                         dmix_ij = sum_e dP_i[e] S_j[e]  (j < i)                   (autograd of naive.py:73-78)
//
// The summaries are [bh][n][E] (E = K V elements per chunk, bf16).  A workgroup owns slices of MF_TE = 256 elements: the slice's
// rows of ALL chunks (n x 512 B) sit in LDS, so every summary byte is read from HBM exactly once per direction -- k_csf_mix read
// a slice's rows once per 64-chunk output tile (1.5x at n = 128) and k_csf_dw read dP and S again for dmix (403 MB at C5).
// One wave per 16 output chunks; the mixing weights of its chunks live in its registers as bf16 hi + lo (two MFMAs, ~16 mantissa
// bits) for the whole launch; the summaries enter the MFMA through the hardware transpose read as the A operand, so a lane ends
// up with four consecutive elements of one output chunk (8-byte staging writes, 512-byte rows out).  A workgroup walks `spw`
// consecutive slices with the next slice's rows in flight in registers while the current one is multiplied and stored.
// dmix: the 16 x 16 tiles on and below the diagonal are dealt round-robin to the waves and accumulated over all slices of the
// workgroup -- one [n][n] partial per workgroup, summed in a fixed order by k_dw_reduce<1> (deterministic, no atomics).
#pragma once
#include "causal_bf16.hpp"

namespace mhla {
namespace fast {

constexpr int MF_TE = 128;   // smallest slice (the divisibility the dispatcher checks)

struct CsfMix2Args {
    const float* W;     // mixing matrix [n][ldw]
    int ldw;
    const u16* in;      // forward: S        backward: dP
    const u16* in2;     // backward: S
    u16* out;           // forward: P        backward: dS
    float* dwp;         // backward: [gridDim.x][n][n] partials of dmix
    int n;
    long E;             // elements per chunk summary (multiple of MF_TE)
    long total;         // slices = bh * E / MF_TE
    int spw;            // slices per workgroup
};

template <int NW, int TE> __host__ __device__ constexpr int mixf_smem() { return 2 * 16 * NW * (TE + 8) * 2; }

// rows of one slice: TE / 32 passes of (512 NW / TE rows x 2 TE bytes); a thread moves 16 bytes per pass.  The thread's byte
// offsets inside a (b,h)'s summaries do not depend on the slice: computed once, 32 bits each, added to a wave-uniform base
// (global_load with an SGPR base: no 64-bit address registers per load).
template <int NW, int TE>
struct MixRows {
    static constexpr int NP = TE / 32, TPR = TE / 8, RPP = 64 * NW / TPR, MF_LD = TE + 8;
    uint4 v[NP];
    __device__ __forceinline__ void issue(const u16* __restrict__ base, const unsigned (&goff)[NP]) {
#pragma unroll
        for (int p = 0; p < NP; ++p) v[p] = gld_stream16(reinterpret_cast<const char*>(base) + goff[p]);
    }
    __device__ __forceinline__ void commit(u16* __restrict__ tile, int n, int tid) const {
        const int r0 = tid / TPR, c = (tid % TPR) * 8;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int row = p * RPP + r0;
            const bool ok = row < n;   // rows past the last chunk: zeros (their weights are zero too, but 0 x NaN is not)
            *reinterpret_cast<uint4*>(tile + row * MF_LD + c) = make_uint4(ok ? v[p].x : 0u, ok ? v[p].y : 0u, ok ? v[p].z : 0u, ok ? v[p].w : 0u);
        }
    }
};
// byte offsets of the thread's pieces (rows past the last chunk read the last chunk's row: a valid address)
template <int NW, int TE>
__device__ __forceinline__ void mix_row_offsets(unsigned (&goff)[TE / 32], long E, int n, int tid) {
    constexpr int NP = TE / 32, TPR = TE / 8, RPP = 64 * NW / TPR;
    const int r0 = tid / TPR, c = (tid % TPR) * 8;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int row = p * RPP + r0;
        goff[p] = (unsigned)(((long)(row < n ? row : n - 1) * E + c) * 2);
    }
}
template <int NW, int TE>
__device__ __forceinline__ void mix_store_rows(u16* __restrict__ base, const unsigned (&goff)[TE / 32], int n, const u16* __restrict__ tile, int tid) {
    constexpr int NP = TE / 32, TPR = TE / 8, RPP = 64 * NW / TPR, MF_LD = TE + 8;
    const int r0 = tid / TPR, c = (tid % TPR) * 8;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int row = p * RPP + r0;
        if (row < n) gst<uint4>(reinterpret_cast<char*>(base) + goff[p], *reinterpret_cast<const uint4*>(tile + row * MF_LD + c));
    }
}
// transposed-product accumulators (lane: elements 16 t + 4 kg .. + 3 of chunk 16 wave + nl) -> staging tile [chunk][element]
template <int TE>
__device__ __forceinline__ void mix_stage(u16* __restrict__ tile, const f32x4 (&acc)[TE / 16], int wave, int lane) {
    constexpr int MF_LD = TE + 8;
    const int nl = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int t = 0; t < TE / 16; ++t)
        *reinterpret_cast<uint2*>(tile + (wave * 16 + nl) * MF_LD + t * 16 + kg * 4) =
            make_uint2(pack_bf16x2(acc[t][0], acc[t][1]), pack_bf16x2(acc[t][2], acc[t][3]));
}
__device__ __forceinline__ void mix_split(const float (&w)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const __bf16 h = (__bf16)w[t];
        hi[t] = h;
        lo[t] = (__bf16)(w[t] - (float)h);
    }
}

// NW waves = 16 NW chunk rows (n <= 16 NW); NK = reduction steps of 32 chunks
template <int NW, int TE>
__global__ __launch_bounds__(64 * NW, TE == 128 ? 4 : 2) void k_csf_mixf(const CsfMix2Args a) {
    constexpr int NK = (NW + 1) / 2, MF_LD = TE + 8, NT = TE / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Ts = reinterpret_cast<u16*>(smem_raw);
    u16* Os = Ts + 16 * NW * MF_LD;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    const int n = a.n;
    const long nsl = a.E / TE;
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
    auto slice_off = [&](long s) { const long bh = s / nsl, es = s - bh * nsl; return bh * n * a.E + es * TE; };
    unsigned goff[TE / 32];
    mix_row_offsets<NW, TE>(goff, a.E, n, tid);
    MixRows<NW, TE> pre;
    pre.issue(a.in + slice_off(s0), goff);
    const int kmax = min(wave / 2, (n - 1) / 32);   // last reduction step with a chunk j < i for this wave's rows
    for (int it = 0; it < cnt; ++it) {
        const long off = slice_off(s0 + it);
        pre.commit(Ts, n, tid);
        __syncthreads();
        if (it + 1 < cnt) pre.issue(a.in + slice_off(s0 + it + 1), goff);
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            if (ks <= kmax) {
#pragma unroll
                for (int t4 = 0; t4 < NT; t4 += 4) {
                    bf16x8 sv[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) sv[t] = tr_read8(Ts, MF_LD, ks * 32, (t4 + t) * 16, lane);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t4 + t] = mfma_bf16(sv[t], wh[ks], acc[t4 + t]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t4 + t] = mfma_bf16(sv[t], wl[ks], acc[t4 + t]);
                }
            }
        }
        mix_stage<TE>(Os, acc, wave, lane);
        __syncthreads();
        mix_store_rows<NW, TE>(a.out + off, goff, n, Os, tid);
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
template <int NW> __host__ __device__ constexpr int mixb_smem() { return 3 * 16 * NW * (128 + 8) * 2; }

template <int NW>
__global__ __launch_bounds__(128 * NW) void k_csf_mixb(const CsfMix2Args a) {
    constexpr int TE = 128, NK = (NW + 1) / 2, MF_LD = TE + 8, NT = TE / 16, NWT = 2 * NW;   // NWT: waves that move rows
    constexpr int NTL = NW * (NW + 1) / 2, TPW = (NTL + NW - 1) / NW;   // dmix tiles, tiles per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Tp = reinterpret_cast<u16*>(smem_raw);   // dP rows of the slice
    u16* Tq = Tp + 16 * NW * MF_LD;                // S rows of the slice
    u16* Os = Tq + 16 * NW * MF_LD;                // dS staging
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    const bool mixer = wave < NW;                  // (uniform) dS role; the others: dmix role
    const int rw = mixer ? wave : wave - NW;
    const int n = a.n;
    const long nsl = a.E / TE;
    const long s0 = (long)blockIdx.x * a.spw;
    const int cnt = (int)min((long)a.spw, a.total - s0);
    float* part = a.dwp + (long)blockIdx.x * n * n;
    auto slice_off = [&](long s) { const long bh = s / nsl, es = s - bh * nsl; return bh * n * a.E + es * TE; };
    unsigned goff[TE / 32 / 2];
    {   // 2 NW waves move the 16 NW rows: TE / 64 passes
        constexpr int TPR = TE / 8, RPP = 64 * NWT / TPR;
        const int r0 = tid / TPR, c = (tid % TPR) * 8;
#pragma unroll
        for (int p = 0; p < TE / 64; ++p) {
            const int row = p * RPP + r0;
            goff[p] = (unsigned)(((long)(row < n ? row : n - 1) * a.E + c) * 2);
        }
    }
    MixRows<NWT, TE / 2> pp, pq;   // (the struct only sees passes x threads: TE / 64 passes of 64 NWT threads)
    if (cnt > 0) {
        pp.issue(a.in + slice_off(s0), goff);
        pq.issue(a.in2 + slice_off(s0), goff);
    }
    auto commit = [&](u16* tile, const MixRows<NWT, TE / 2>& r) {
        constexpr int TPR = TE / 8, RPP = 64 * NWT / TPR;
        const int r0 = tid / TPR, c = (tid % TPR) * 8;
#pragma unroll
        for (int p = 0; p < TE / 64; ++p) {
            const int row = p * RPP + r0;
            const bool ok = row < n;
            *reinterpret_cast<uint4*>(tile + row * MF_LD + c) = make_uint4(ok ? r.v[p].x : 0u, ok ? r.v[p].y : 0u, ok ? r.v[p].z : 0u, ok ? r.v[p].w : 0u);
        }
    };
    auto store_rows = [&](u16* base) {
        constexpr int TPR = TE / 8, RPP = 64 * NWT / TPR;
        const int r0 = tid / TPR, c = (tid % TPR) * 8;
#pragma unroll
        for (int p = 0; p < TE / 64; ++p) {
            const int row = p * RPP + r0;
            if (row < n) gst<uint4>(reinterpret_cast<char*>(base) + goff[p], *reinterpret_cast<const uint4*>(Os + row * MF_LD + c));
        }
    };
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
            commit(Tp, pp);
            commit(Tq, pq);
            __syncthreads();
            if (it + 1 < cnt) {
                pp.issue(a.in + slice_off(s0 + it + 1), goff);
                pq.issue(a.in2 + slice_off(s0 + it + 1), goff);
            }
#pragma unroll
            for (int t4 = 0; t4 < NT; t4 += 4) {
                f32x4 acc[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) {
                    if (ks >= kmin && ks < kend) {
                        bf16x8 sv[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) sv[t] = tr_read8(Tp, MF_LD, ks * 32, (t4 + t) * 16, lane);
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[t] = mfma_bf16(sv[t], wh[ks], acc[t]);
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[t] = mfma_bf16(sv[t], wl[ks], acc[t]);
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    *reinterpret_cast<uint2*>(Os + (rw * 16 + nl) * MF_LD + (t4 + t) * 16 + kg * 4) =
                        make_uint2(pack_bf16x2(acc[t][0], acc[t][1]), pack_bf16x2(acc[t][2], acc[t][3]));
            }
            __syncthreads();
            store_rows(a.out + slice_off(s0 + it));
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
            commit(Tp, pp);
            commit(Tq, pq);
            __syncthreads();
            if (it + 1 < cnt) {
                pp.issue(a.in + slice_off(s0 + it + 1), goff);
                pq.issue(a.in2 + slice_off(s0 + it + 1), goff);
            }
            // dmix tiles: A[m = i][k = e] = dP_i[e], B[k = e][n = j] = S_j[e], both 16-byte row reads
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                if (tit[u] >= 0) {
                    const u16* ap = Tp + (tit[u] * 16 + nl) * MF_LD + kg * 8;
                    const u16* bp = Tq + (tjt[u] * 16 + nl) * MF_LD + kg * 8;
#pragma unroll
                    for (int ks = 0; ks < TE / 32; ++ks)
                        dacc[u] = mfma_bf16(*reinterpret_cast<const bf16x8*>(ap + ks * 32), *reinterpret_cast<const bf16x8*>(bp + ks * 32), dacc[u]);
                }
            }
            __syncthreads();
            store_rows(a.out + slice_off(s0 + it));
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
