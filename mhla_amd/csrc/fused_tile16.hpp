// Output tiles of the bf16 fast path (see fused.hpp): a workgroup mixes the (b,h)'s summaries for a tile of TTP blocks into LDS
// (9 KB per block) and its waves then turn token rows into output rows block by block.  512 threads.
//   TTP = 16: one workgroup per CU (160 KB LDS, <= 256 VGPRs), wave w owns blocks w and w + 8; the state is streamed four times per
//             (b,h) (from L2 after the first touch) and all 16 MFMA columns carry blocks.
//   TTP = 8 : two workgroups per CU (80 KB LDS, <= 128 VGPRs), wave w owns block w; the state is streamed eight times per (b,h); the
//             16 MFMA columns carry the 8 blocks' bf16 hi and lo weight parts side by side (one MFMA instead of two, the halves
//             added across lanes), so the matrix work per block is the same.  Measured (round 2): the output kernel 52 us against
//             49 us with 16-block tiles -- the mixing phase is bound by the CU's L2 read rate (about 25 B / clk: 512 KB per tile
//             workgroup whatever the tile size), so twice the tiles cost what the second workgroup's overlap gains; the two
//             gradient kernels do not fit 128 VGPRs.  Only TTP = 16 is instantiated.
#pragma once
#include "fused.hpp"

namespace mhla {
namespace fast {

constexpr int GSLOT = FD * GLD + 8;                  // elements per mixed-summary slot (+16 B: the 16-byte mixing stores of 8 lanes = 8 blocks cover all banks)
// LDS: the tile's mixed summaries, then three [TTP][64] fp32 arrays of per-block side values (1 / n, dz, ksum or dksum)
template <int TTP> constexpr int tile_gt_bytes() { return TTP * GSLOT * 2; }
template <int TTP> constexpr int tile_smem() { return tile_gt_bytes<TTP>() + 3 * TTP * 64 * 4; }   // 160000 B (16) / 80000 B (8)

__device__ __forceinline__ float dpp_row_ror8(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));
}

// Eight waves stream the (b,h) state through a ring of NBUF register batches (UN e'-tiles of 16 x 64 blocks each): with plain
// global loads (address space 1, in-order vmcnt) NBUF - 1 batches stay in flight while one is multiplied.
template <int TTP, int TRANSW, bool TWO, int NBUF, int UN>
__device__ __forceinline__ void mix_tile_impl(u16* __restrict__ Gt, const u16* __restrict__ state_bh, int njg,
                                              const float* __restrict__ W, int ldw, int M, int i0, int tid) {
    const int wave = tid >> 6, lane = tid & 63, n = lane & 15, kg = lane >> 4;
    constexpr int NW = FT8 / 64;
    constexpr int NB = FE / 16 / NW / UN;   // batches per wave
    constexpr bool HL = TTP == 8;           // columns 0..7: the blocks' hi weight parts, 8..15: their lo parts
    const int nb = HL ? (n & 7) : n;        // block column of this lane
    // block groups beyond the last one are clamped onto it: their mixing weights are zero and every stored summary is finite
    // (groups are written completely, blocks beyond M as zeros), so no select is needed
    const u16* g0 = state_bh + (long)min(kg, njg - 1) * FE * IT + n * IT;
    const u16* g1 = state_bh + (long)min(4 + kg, njg - 1) * FE * IT + n * IT;
    auto load_batch = [&](uint4 (&av)[UN][2], int bt) {
        const long et0 = (long)(wave + NW * bt) * UN;
#pragma unroll
        for (int u = 0; u < UN; ++u) {
#ifdef T16_NOMIXLOAD
            av[u][0] = make_uint4(tid, bt, u, 0x3f803f80u);
            if (TWO) av[u][1] = make_uint4(tid, bt, u, 0x3f803f80u);
#else
            av[u][0] = gld<uint4>(g0 + (et0 + u) * 16 * IT);
            if (TWO) av[u][1] = gld<uint4>(g1 + (et0 + u) * 16 * IT);
#endif
        }
    };
    uint4 buf[NBUF][UN][2];
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b) load_batch(buf[b], b);   // the state stream starts before the weights are fetched

    bf16x8 bhi[2], blo[2];                  // (HL: bhi carries this lane's part, blo is unused)
    {   // mixing weights of this lane: rows / columns beyond M are clamped for the load and zeroed after (no branches)
        float wv[2][8];
        const int i = i0 + nb, ic = min(i, M - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int jc = min(ks * 32 + kg * 8 + t, M - 1);
                wv[ks][t] = gld<float>(TRANSW ? W + (long)jc * ldw + ic : W + (long)ic * ldw + jc);
            }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            s16x8 hi, lo;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int j = ks * 32 + kg * 8 + t;
                const float w = (i < M && j < M) ? wv[ks][t] : 0.f;
                const u16 h = cvt_bf16(w), l = cvt_bf16(w - bf(h));
                hi[t] = (short)((HL && n >= 8) ? l : h);
                lo[t] = (short)l;
            }
            bhi[ks] = __builtin_bit_cast(bf16x8, hi);
            blo[ks] = __builtin_bit_cast(bf16x8, lo);
        }
    }
    auto do_batch = [&](const uint4 (&av)[UN][2], int bt) {
        // UN independent accumulator chains, interleaved step by step (no back-to-back dependent MFMAs)
        f32x4 c[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) c[u] = mfma_bf16(__builtin_bit_cast(bf16x8, av[u][0]), bhi[0], f32x4{0.f, 0.f, 0.f, 0.f});
        if (!HL) {
#pragma unroll
            for (int u = 0; u < UN; ++u) c[u] = mfma_bf16(__builtin_bit_cast(bf16x8, av[u][0]), blo[0], c[u]);
        }
        if (TWO) {
#pragma unroll
            for (int u = 0; u < UN; ++u) c[u] = mfma_bf16(__builtin_bit_cast(bf16x8, av[u][1]), bhi[1], c[u]);
            if (!HL) {
#pragma unroll
                for (int u = 0; u < UN; ++u) c[u] = mfma_bf16(__builtin_bit_cast(bf16x8, av[u][1]), blo[1], c[u]);
            }
        }
        uint2 pk[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (HL) {
#pragma unroll
                for (int r = 0; r < 4; ++r) c[u][r] += dpp_row_ror8(c[u][r]);   // hi part (column nb) + lo part (column nb + 8)
            }
            pk[u].x = pack_bf16x2(c[u][0], c[u][1]);
            pk[u].y = pack_bf16x2(c[u][2], c[u][3]);
        }
        if constexpr (!HL && UN % 2 == 0) {
            // element tiles et = (wave + NW bt) UN + u: with UN = 4 the batch is the four column tiles of summary row d2
#pragma unroll
            for (int u = 0; u < UN; u += 2) {
                const int et = (wave + NW * bt) * UN + u + (kg & 1), d2 = et >> 2, d1 = (et & 3) * 16 + (kg >> 1) * 8;
                *reinterpret_cast<uint4*>(Gt + (long)nb * GSLOT + gt_off(d2, d1)) = pair_pieces(pk[u], pk[u + 1]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int et = (wave + NW * bt) * UN + u, d2 = et >> 2, d1 = (et & 3) * 16 + kg * 4;
                if (!HL || n < 8) *reinterpret_cast<uint2*>(Gt + (long)nb * GSLOT + gt_off(d2, d1)) = pk[u];
            }
        }
    };
#pragma unroll
    for (int bt = 0; bt < NB; ++bt) {   // fully unrolled: the ring slots are registers
        if (bt + NBUF - 1 < NB) load_batch(buf[(bt + NBUF - 1) % NBUF], bt + NBUF - 1);
        do_batch(buf[bt % NBUF], bt);
    }
}

template <int TTP, int TRANSW, int NBUF>
__device__ __forceinline__ void mix_tile_to_lds(u16* __restrict__ Gt, const u16* __restrict__ state_bh, int njg,
                                                const float* __restrict__ W, int ldw, int M, int i0, int tid) {
    constexpr int UN = TTP == 16 ? 4 : 2;
    if (njg > 4) mix_tile_impl<TTP, TRANSW, true, NBUF, UN>(Gt, state_bh, njg, W, ldw, M, i0, tid);
    else         mix_tile_impl<TTP, TRANSW, false, NBUF, UN>(Gt, state_bh, njg, W, ldw, M, i0, tid);
}
// (starting each of a (b,h)'s tile workgroups at a different eighth of the state -- so that the lines in flight towards HBM differ --
// was measured neutral to slightly slower)

// Token rows of a 64-row chunk straight from a token view, in MFMA operand layout: a[st][ks] = row 16 st + (lane & 15), columns
// 32 ks + 8 (lane >> 4) .. + 7.  Rows beyond rv read the block's first row (valid memory, finite values): every consumer of these
// operands produces output rows from operand rows one to one and never stores rows >= rv, and the row weights of the dksum sum
// (dz) are zero there -- so there is no zeroing, and nothing depends on the loaded data until the MFMAs that consume it.
__device__ __forceinline__ void load_a64(bf16x8 (&a)[4][2], const u16* __restrict__ base, long sn,
                                         const int* __restrict__ idx, long p0, int rv, int lane) {
    const int m = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const int row = st * 16 + m;
        const u16* src = base + tok_row(idx, p0 + (row < rv ? row : 0)) * sn + kg * 8;
#ifdef T16_NOTOK
        for (int ks = 0; ks < 2; ++ks) a[st][ks] = __builtin_bit_cast(bf16x8, make_uint4(lane, (unsigned)(uintptr_t)src, ks, 0x3f803f80u));
#else
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a[st][ks] = __builtin_bit_cast(bf16x8, gld_stream16(src + ks * 32));
#endif
    }
}
// relu(x) + eps on loaded operands (MHLA_FLAG_RELU_EPS), applied where they are consumed
__device__ __forceinline__ void relu_a64(bf16x8 (&a)[4][2], float eps) {
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a[st][ks] = __builtin_bit_cast(bf16x8, relu_eps8(__builtin_bit_cast(uint4, a[st][ks]), eps));
}

// acc[st][tn] += (X[st] x B)^T for a 64-row chunk X of token rows and one mixed summary Gb[d2][d1] (GLD stride):
//   TRB false: B[k = d1][c = d2] = Gb[c][k]  (k contiguous: plain 16-byte LDS reads)
//   TRB true : B[k = d2][c = d1] = Gb[k][c]  (hardware transpose reads)
// The summary is the MFMA's A operand and the token rows its B operand, i.e. the product comes out TRANSPOSED: lane (n = lane & 15,
// kg = lane >> 4) holds, for token row 16 st + n, the four consecutive columns 16 tn + 4 kg + 0..3.  Per-row factors (1 / n, dz) are
// then per-lane scalars, a result row packs into 8-byte pieces (16 LDS stores per block instead of 64 two-byte ones), and the
// per-column terms (ksum, dksum) are 16-byte reads.
template <bool TRB>
__device__ __forceinline__ void chunk_times_gt(f32x4 (&acc)[4][4], const bf16x8 (&a)[4][2], const u16* __restrict__ Gb, int lane) {
    lane = opaque_lane(lane);
    const int n = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 bv = TRB ? tr_read8_gt(Gb, ks * 32, tn * 16, lane)
                                  : *reinterpret_cast<const bf16x8*>(Gb + gt_off(tn * 16 + n, ks * 32 + kg * 8));
#pragma unroll
            for (int st = 0; st < 4; ++st) acc[st][tn] = mfma_bf16(bv, a[st][ks], acc[st][tn]);
        }
    }
}
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[4][4]) {
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[st][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ uint2 pack4(const f32x4& v) { return make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])); }

// Wave-private staging of a 64 x 64 result in the transposed-product layout as bf16 rows of a swizzled slot (gt_off): neighbouring
// column tiles are paired into 16-byte pieces (pair_pieces) -- 8 conflict-free 16-byte stores per block instead of 16 two-way
// conflicted 8-byte ones.
__device__ __forceinline__ void stage64_packed(u16* __restrict__ Os, const uint2 (&pv)[4][4], int lane) {
    lane = opaque_lane(lane);
    const int n = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int tn = 0; tn < 4; tn += 2)
            *reinterpret_cast<uint4*>(Os + gt_off(st * 16 + n, (tn + (kg & 1)) * 16 + (kg >> 1) * 8)) = pair_pieces(pv[st][tn], pv[st][tn + 1]);
}
__device__ __forceinline__ void stage64(u16* __restrict__ Os, const f32x4 (&acc)[4][4], int lane) {
    lane = opaque_lane(lane);
    const int n = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int tn = 0; tn < 4; tn += 2)
            *reinterpret_cast<uint4*>(Os + gt_off(st * 16 + n, (tn + (kg & 1)) * 16 + (kg >> 1) * 8)) = pair_pieces(pack4(acc[st][tn]), pack4(acc[st][tn + 1]));
}
// zero the bf16 lanes of v where the corresponding element of m is <= 0 (relu gradient mask), 4 elements
__device__ __forceinline__ uint2 mask_pos4(uint2 v, uint2 m) {
    const uint4 r = mask_pos8(make_uint4(v.x, v.y, 0, 0), make_uint4(m.x, m.y, 0, 0));
    return make_uint2(r.x, r.y);
}
// narrow fallback (no free staging slot: inner chunks of multi-chunk blocks): direct 8-byte stores from the packed result
template <bool MASK>
__device__ __forceinline__ void store64_direct(u16* __restrict__ base, long sn, const int* __restrict__ idx, long p0, int rv,
                                               const uint2 (&pv)[4][4], const u16* __restrict__ mbase, long msn, int lane) {
    const int n = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const int row = st * 16 + n;
        if (row < rv) {
            const long tr = tok_row(idx, p0 + row);
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
                uint2 v = pv[st][tn];
                if (MASK) v = mask_pos4(v, *reinterpret_cast<const uint2*>(mbase + tr * msn + tn * 16 + kg * 4));
                *reinterpret_cast<uint2*>(base + tr * sn + tn * 16 + kg * 4) = v;
            }
        }
    }
}
__device__ __forceinline__ void pack_acc(uint2 (&pv)[4][4], const f32x4 (&acc)[4][4]) {
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) pv[st][tn] = pack4(acc[st][tn]);
}

// Side values of the tile's blocks ([TTP][64] fp32 in LDS): fetched by the whole workgroup before the mixing (TTP / 8 elements per
// thread: block (tid >> 6) + 8 t, column tid & 63) and written to LDS just before the barrier that follows it.
template <int TTP> struct SideRegs { float v[TTP / 8]; };
template <int TTP, bool COHERENT = false>
__device__ __forceinline__ void side_issue(SideRegs<TTP>& r, const float* __restrict__ src_bh, int stride, int cols, int blk0, int M, int tid) {
#pragma unroll
    for (int t = 0; t < TTP / 8; ++t) {
        const int blk = min(blk0 + (tid >> 6) + 8 * t, M - 1), c = min(tid & 63, cols - 1);
        r.v[t] = COHERENT ? coherent_load(src_bh + (long)blk * stride + c) : gld<float>(src_bh + (long)blk * stride + c);
    }
}
template <int TTP>
__device__ __forceinline__ void side_commit(float* __restrict__ dst, const SideRegs<TTP>& r, int cols, int tid) {
#pragma unroll
    for (int t = 0; t < TTP / 8; ++t) dst[((tid >> 6) + 8 * t) * 64 + (tid & 63)] = (tid & 63) < cols ? r.v[t] : 0.f;
}

// MC: blocks of several 64-token chunks with `a.cs` workgroups per tile (the single-chunk instantiation carries no trace of it)
template <int TTP, bool MC = false>
__global__ __launch_bounds__(FT8, TTP == 16 ? 2 : 4) void k_tile_out(const FsOutArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Gt = reinterpret_cast<u16*>(smem_raw);   // [TTP][64 d2][72]
    float* sideN = reinterpret_cast<float*>(smem_raw + tile_gt_bytes<TTP>());   // [TTP][64]  1 / n
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, n = lane & 15;
    const int Lp = xcd_swizzle(blockIdx.x, gridDim.x), cs = MC ? a.cs : 1, L = Lp / cs, part = Lp - L * cs;   // (parts of a tile are neighbours: one XCD's L2 serves their state streams)
    const int ntt = (a.njg * IT + TTP - 1) / TTP, bh = L / ntt, it = L - bh * ntt, b = bh / a.H, h = bh - b * a.H;
    const int S = a.S, M = a.M;
    const u16* qb = (const u16*)a.q.ptr + b * a.q.sb + h * a.q.sh;
    u16* ob = (u16*)a.o.ptr + b * a.o.sb + h * a.o.sh;
    const float* ninv_bh = a.ninv + (long)bh * M * S;
    const u16* state_bh = a.state + (long)bh * a.njg * FE * IT;
    constexpr int NBUF = 3;

    auto compute_store = [&](bf16x8 (&av)[4][2], int bi, int i, int c0, int rv) {
        const long p0 = (long)i * S + c0;
        u16* Gb = Gt + bi * GSLOT;
        if (a.relu) relu_a64(av, a.eps);
        f32x4 acc[4][4];
        zero_acc(acc);
        chunk_times_gt<false>(acc, av, Gb, lane);
        if (a.normalize) {
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const float ni = sideN[bi * 64 + st * 16 + n];
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) acc[st][tn] *= ni;
            }
        }
        if (c0 + 64 * cs >= S) {   // this workgroup's last chunk of the block: its Gt slot is dead for this wave -> staging buffer
            wave_lds_fence();
            stage64(Gb, acc, lane);
            wave_lds_fence();
            store64<false, false>(ob, a.o.sn, a.idx, p0, rv, Gb, nullptr, 0, lane);
        } else {
            uint2 pv[4][4];
            pack_acc(pv, acc);
            store64_direct<false>(ob, a.o.sn, a.idx, p0, rv, pv, nullptr, 0, lane);
        }
    };

    if (S <= 64) {
        // each wave owns block wave (and wave + 8); their operands are fetched before the mixing, whose L2 / MALL-bound phase
        // hides their HBM latency
        const int iA = it * TTP + wave, iB = iA + 8;
        bf16x8 avA[4][2], avB[TTP == 16 ? 4 : 1][2];
        SideRegs<TTP> sn;
        trace_mark(a.trace, 0);
        if (a.normalize) side_issue<TTP>(sn, ninv_bh, S, S, it * TTP, M, tid);
        load_a64(avA, qb, a.q.sn, a.idx, (long)min(iA, M - 1) * S, S, lane);
        if constexpr (TTP == 16) load_a64(avB, qb, a.q.sn, a.idx, (long)min(iB, M - 1) * S, S, lane);
        trace_mark(a.trace, 1);
        mix_tile_to_lds<TTP, 0, NBUF>(Gt, state_bh, a.njg, a.W, a.ldw, M, it * TTP, tid);
        if (a.normalize) side_commit<TTP>(sideN, sn, S, tid);
        trace_mark(a.trace, 2);
        __syncthreads();
        trace_mark(a.trace, 3);
        if (iA < M) compute_store(avA, wave, iA, 0, S);
        trace_mark(a.trace, 4);
        if constexpr (TTP == 16) {
            if (iB < M) compute_store(avB, wave + 8, iB, 0, S);
        }
        trace_mark(a.trace, 5);
        if (a.trace) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); trace_mark(a.trace, 6); }
        return;
    }

    mix_tile_to_lds<TTP, 0, NBUF>(Gt, state_bh, a.njg, a.W, a.ldw, M, it * TTP, tid);
    __syncthreads();
    for (int bi = wave; bi < TTP; bi += 8) {
        const int i = it * TTP + bi;
        if (i >= M) continue;
        for (int c0 = part * 64; c0 < S; c0 += 64 * cs) {
            const int rv = min(64, S - c0);
            bf16x8 av[4][2];
            load_a64(av, qb, a.q.sn, a.idx, (long)i * S + c0, rv, lane);
            if (a.normalize) {   // the chunk's 1 / n into the wave's own side slot
                wave_lds_fence();
                sideN[bi * 64 + lane] = gld<float>(ninv_bh + (long)i * S + c0 + min(lane, rv - 1));
                wave_lds_fence();
            }
            compute_store(av, bi, i, c0, rv);
        }
    }
}

// dQ role of k_tile_bwd: workgroup wg of nwg
template <int TTP, bool MC>
__device__ __forceinline__ void tile_bwd_dq_body(const FsTokArgs& a, unsigned char* smem_raw, int wg, int nwg) {
    u16* Gt = reinterpret_cast<u16*>(smem_raw);
    float* sideN = reinterpret_cast<float*>(smem_raw + tile_gt_bytes<TTP>());   // [TTP][64]  1 / n
    float* sideZ = sideN + TTP * 64;                                             // [TTP][64]  dz (zero beyond the block's rows)
    float* sideK = sideZ + TTP * 64;                                             // [TTP][64]  ksum
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, n = lane & 15, kg = lane >> 4;
    const int Lp = xcd_swizzle(wg, nwg), cs = MC ? a.cs : 1, L = Lp / cs, part = Lp - L * cs;
    const int ntt = (a.njg * IT + TTP - 1) / TTP, bh = L / ntt, jgx = L - bh * ntt, b = bh / a.H, h = bh - b * a.H;
    const int S = a.S, M = a.M;
    auto base = [&](const View& w) { return (const u16*)w.ptr + b * w.sb + h * w.sh; };
    auto mbase = [&](const MView& w) { return (u16*)w.ptr + b * w.sb + h * w.sh; };
    const u16 *qb = base(a.q), *gb = base(a.dout);
    u16* dqb = mbase(a.dq);
    const long sofs = (long)bh * a.njg * FE * IT;
    constexpr int NBUF = TTP == 16 ? 2 : 3;

    // one 64-row chunk in two steps: the products (dO G^T)[s][d1] (B[k = d2][c = d1] = Gt[d2][d1]) ...
    auto products = [&](f32x4 (&acc)[4][4], const bf16x8 (&gv)[4][2], int bi) {
        zero_acc(acc);
        chunk_times_gt<true>(acc, gv, Gt + bi * GSLOT, lane);
    };
    // ... then dQ rows (scaling by 1 / n, + dz (x) ksum, store) and the chunk's contribution to dksum (per-lane partials: this
    // lane's four rows, columns 32 ks + 8 kg + t)
    auto finish_store = [&](f32x4 (&acc)[4][4], bf16x8 (&qv)[4][2], float (&dks_acc)[2][8], int bi, int j, int c0, int rv, auto after_q) {
        const long p0 = (long)j * S + c0;
        u16* Gb = Gt + bi * GSLOT;
        if (a.normalize) {
            if (a.relu) relu_a64(qv, a.eps);
            f32x4 ks4[4];
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) ks4[tn] = *reinterpret_cast<const f32x4*>(sideK + bi * 64 + tn * 16 + kg * 4);
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const float dzr = sideZ[bi * 64 + st * 16 + n], ni = sideN[bi * 64 + st * 16 + n];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const s16x8 qs = __builtin_bit_cast(s16x8, qv[st][ks]);
#pragma unroll
                    for (int t = 0; t < 8; ++t) dks_acc[ks][t] += dzr * bf((u16)qs[t]);
                }
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) acc[st][tn] = acc[st][tn] * ni + dzr * ks4[tn];
            }
        }
        after_q();   // the chunk's Q rows are dead from here on: the caller may request the next ones into their registers
        if (c0 + 64 * cs >= S) {
            wave_lds_fence();
            stage64(Gb, acc, lane);
            wave_lds_fence();
            if (a.relu) store64<true>(dqb, a.dq.sn, a.idx, p0, rv, Gb, qb, a.q.sn, lane);
            else        store64<false>(dqb, a.dq.sn, a.idx, p0, rv, Gb, nullptr, 0, lane);
        } else {
            uint2 pv[4][4];
            pack_acc(pv, acc);
            if (a.relu) store64_direct<true>(dqb, a.dq.sn, a.idx, p0, rv, pv, qb, a.q.sn, lane);
            else        store64_direct<false>(dqb, a.dq.sn, a.idx, p0, rv, pv, nullptr, 0, lane);
        }
    };
    // dksum[col]: sum the per-lane partials over the 16 row-lanes n.  Halving butterfly: each step a lane hands half of its
    // values to its partner and keeps (and completes) the other half -- 15 exchanges instead of 64, and lane n ends up with the
    // total of value index n = 8 ks + t, i.e. column 32 (n >> 3) + 8 kg + (n & 7): one coalesced store.
    auto finish_dks = [&](const float (&dks_acc)[2][8], int j) {
        float v8[8], v4[4], v2[2];
        const bool b8 = n & 8, b4 = n & 4, b2 = n & 2, b1 = n & 1;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float send = b8 ? dks_acc[0][t] : dks_acc[1][t], keep = b8 ? dks_acc[1][t] : dks_acc[0][t];
            v8[t] = keep + __shfl_xor(send, 8, 64);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float send = b4 ? v8[t] : v8[t + 4], keep = b4 ? v8[t + 4] : v8[t];
            v4[t] = keep + __shfl_xor(send, 4, 64);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float send = b2 ? v4[t] : v4[t + 2], keep = b2 ? v4[t + 2] : v4[t];
            v2[t] = keep + __shfl_xor(send, 2, 64);
        }
        const float send = b1 ? v2[0] : v2[1], keep = b1 ? v2[1] : v2[0];
        coherent_store(a.dksum + part * a.dks_part + ((long)bh * M + j) * 64 + 32 * (n >> 3) + 8 * kg + (n & 7), keep + __shfl_xor(send, 1, 64));
    };
    auto zero_dks = [](float (&d)[2][8]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < 8; ++t) d[ks][t] = 0.f;
    };

    if (S <= 64) {
        // TTP = 16: the first block's operands (dO, Q rows) before the mixing; the second block's dO rows right after the barrier
        // (they travel while the first block is processed) and its Q rows once the first block is done (256 VGPRs: no room earlier).
        // TTP = 8 (128 VGPRs): dO rows before the mixing, Q rows right after it.
        const int jA = jgx * TTP + wave, jB = jA + 8, jAc = min(jA, M - 1), jBc = min(jB, M - 1);
        bf16x8 gvA[4][2], qvA[4][2];
        SideRegs<TTP> sn, sz, sk;
        trace_mark(a.trace, 0);
        if (a.normalize) {
            side_issue<TTP>(sn, a.ninv + (long)bh * M * S, S, S, jgx * TTP, M, tid);
            side_issue<TTP>(sz, a.dz + (long)bh * M * S, S, S, jgx * TTP, M, tid);
            side_issue<TTP>(sk, a.ksum + (long)bh * M * 64, 64, 64, jgx * TTP, M, tid);
        }
        load_a64(gvA, gb, a.dout.sn, a.idx, (long)jAc * S, S, lane);
        if (TTP == 16 && a.normalize) load_a64(qvA, qb, a.q.sn, a.idx, (long)jAc * S, S, lane);
        trace_mark(a.trace, 1);
        mix_tile_to_lds<TTP, 0, NBUF>(Gt, a.state + sofs, a.njg, a.W, a.ldw, M, jgx * TTP, tid);
        if (TTP == 8 && a.normalize) load_a64(qvA, qb, a.q.sn, a.idx, (long)jAc * S, S, lane);
        if (a.normalize) {
            side_commit<TTP>(sideN, sn, S, tid);
            side_commit<TTP>(sideZ, sz, S, tid);
            side_commit<TTP>(sideK, sk, 64, tid);
        }
        trace_mark(a.trace, 2);
        __syncthreads();
        trace_mark(a.trace, 3);
        f32x4 acc[4][4];
        float dks_acc[2][8];
        if constexpr (TTP == 16) {
            bf16x8 gvB[4][2], qvB[4][2];
            load_a64(gvB, gb, a.dout.sn, a.idx, (long)jBc * S, S, lane);
            products(acc, gvA, wave);
            zero_dks(dks_acc);
            // the second block's Q rows are requested as soon as the first block's are dead (before its staging and stores): they travel
            // during those and the second block's products
            auto fetch_qB = [&]() { if (a.normalize) load_a64(qvB, qb, a.q.sn, a.idx, (long)jBc * S, S, lane); };
            auto nothing = []() {};
            if (jA < M) { finish_store(acc, qvA, dks_acc, wave, jA, 0, S, fetch_qB); if (a.normalize) finish_dks(dks_acc, jA); }
            else fetch_qB();
            trace_mark(a.trace, 4);
            products(acc, gvB, wave + 8);
            zero_dks(dks_acc);
            if (jB < M) { finish_store(acc, qvB, dks_acc, wave + 8, jB, 0, S, nothing); if (a.normalize) finish_dks(dks_acc, jB); }
        } else {
            products(acc, gvA, wave);
            zero_dks(dks_acc);
            if (jA < M) { finish_store(acc, qvA, dks_acc, wave, jA, 0, S, []() {}); if (a.normalize) finish_dks(dks_acc, jA); }
            trace_mark(a.trace, 4);
        }
        trace_mark(a.trace, 5);
        if (a.trace) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); trace_mark(a.trace, 6); }
        if (a.normalize && !a.drop_signal) tile_signal(a.done + L, tid);   // this tile's dksum rows are written
        return;
    }
    mix_tile_to_lds<TTP, 0, NBUF>(Gt, a.state + sofs, a.njg, a.W, a.ldw, M, jgx * TTP, tid);
    __syncthreads();
    for (int bi = wave; bi < TTP; bi += 8) {
        const int j = jgx * TTP + bi;
        if (j >= M) continue;
        float dks_acc[2][8];
        zero_dks(dks_acc);
        if (a.normalize) sideK[bi * 64 + lane] = gld<float>(a.ksum + ((long)bh * M + j) * 64 + lane);
        for (int c0 = part * 64; c0 < S; c0 += 64 * cs) {
            const int rv = min(64, S - c0);
            bf16x8 gv[4][2], qv[4][2];
            f32x4 acc[4][4];
            load_a64(gv, gb, a.dout.sn, a.idx, (long)j * S + c0, rv, lane);
            if (a.normalize) {
                load_a64(qv, qb, a.q.sn, a.idx, (long)j * S + c0, rv, lane);
                const long so = ((long)bh * M + j) * S + c0 + min(lane, rv - 1);
                const float ni = gld<float>(a.ninv + so), dz = gld<float>(a.dz + so);
                wave_lds_fence();
                sideN[bi * 64 + lane] = ni;
                sideZ[bi * 64 + lane] = lane < rv ? dz : 0.f;   // rows beyond the chunk carry another row's data: no weight in dksum
                wave_lds_fence();
            }
            products(acc, gv, bi);
            finish_store(acc, qv, dks_acc, bi, j, c0, rv, []() {});
        }
        if (a.normalize) finish_dks(dks_acc, j);
    }
    if (a.normalize && !a.drop_signal) tile_signal(a.done + Lp, tid);   // this part's share of the tile's dksum rows is written
}

// dK / dV role of k_tile_bwd: workgroup wg of nwg.  The only thing it needs from the dQ role is the tile's dksum rows (awaited just
// before they are fetched, after the mixing).
template <int TTP, bool MC>
__device__ __forceinline__ void tile_bwd_dkv_body(const FsTokArgs& a, unsigned char* smem_raw, int wg, int nwg) {
    u16* Gt = reinterpret_cast<u16*>(smem_raw);
    float* sideK = reinterpret_cast<float*>(smem_raw + tile_gt_bytes<TTP>());   // [TTP][64]  dksum (dQ role)
    int* wait_word = reinterpret_cast<int*>(sideK + TTP * 64);                   // (the dQ role's second side array: unused here)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, kg = lane >> 4;
    const int Lp = xcd_swizzle(wg, nwg), cs = MC ? a.cs : 1, L = Lp / cs, part = Lp - L * cs;
    const int ntt = (a.njg * IT + TTP - 1) / TTP, bh = L / ntt, jgx = L - bh * ntt, b = bh / a.H, h = bh - b * a.H;
    const int S = a.S, M = a.M;
    auto base = [&](const View& w) { return (const u16*)w.ptr + b * w.sb + h * w.sh; };
    auto mbase = [&](const MView& w) { return (u16*)w.ptr + b * w.sb + h * w.sh; };
    const u16 *kb = base(a.k), *vb = base(a.v);
    u16 *dkb = mbase(a.dk), *dvb = mbase(a.dv);
    const long sofs = (long)bh * a.njg * FE * IT;
    constexpr int NBUF = 3;
    auto compute_store = [&](bf16x8 (&kv)[4][2], const bf16x8 (&vv)[4][2], int bi, int j, int c0, int rv) {
        const long p0 = (long)j * S + c0;
        u16* Gb = Gt + bi * GSLOT;
        const bool last = c0 + 64 * cs >= S;
        if (a.relu) relu_a64(kv, a.eps);
        // dV first, kept packed as bf16 while dK is computed (both need the intact Gb)
        uint2 pv[4][4];
        {
            f32x4 accV[4][4];
            zero_acc(accV);
            chunk_times_gt<false>(accV, kv, Gb, lane);   // dV[s][d2] = sum_d1 K[s][d1] dKVt[d2][d1]
            pack_acc(pv, accV);
        }
        f32x4 accK[4][4];
        zero_acc(accK);
        chunk_times_gt<true>(accK, vv, Gb, lane);        // dK[s][d1] = sum_d2 V[s][d2] dKVt[d2][d1]
        if (a.normalize) {
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
                const f32x4 dk = *reinterpret_cast<const f32x4*>(sideK + bi * 64 + tn * 16 + kg * 4);
#pragma unroll
                for (int st = 0; st < 4; ++st) accK[st][tn] += dk;
            }
        }
        if (last) {
            wave_lds_fence();
            stage64(Gb, accK, lane);
            wave_lds_fence();
            if (a.relu) store64<true>(dkb, a.dk.sn, a.idx, p0, rv, Gb, kb, a.k.sn, lane);
            else        store64<false>(dkb, a.dk.sn, a.idx, p0, rv, Gb, nullptr, 0, lane);
            wave_lds_fence();
            stage64_packed(Gb, pv, lane);
            wave_lds_fence();
            store64<false>(dvb, a.dv.sn, a.idx, p0, rv, Gb, nullptr, 0, lane);
        } else {
            uint2 pk[4][4];
            pack_acc(pk, accK);
            if (a.relu) store64_direct<true>(dkb, a.dk.sn, a.idx, p0, rv, pk, kb, a.k.sn, lane);
            else        store64_direct<false>(dkb, a.dk.sn, a.idx, p0, rv, pk, nullptr, 0, lane);
            store64_direct<false>(dvb, a.dv.sn, a.idx, p0, rv, pv, nullptr, 0, lane);
        }
    };

    if (S <= 64) {
        // TTP = 16: the first block's K and V rows before the mixing, the second block's right after the barrier.
        // TTP = 8 (128 VGPRs): K rows before the mixing, V rows right after it.
        const int jA = jgx * TTP + wave, jB = jA + 8, jAc = min(jA, M - 1), jBc = min(jB, M - 1);
        bf16x8 kvA[4][2], vvA[4][2];
        SideRegs<TTP> sk;
        trace_mark(a.trace, 0);
        load_a64(kvA, kb, a.k.sn, a.idx, (long)jAc * S, S, lane);
        if (TTP == 16) load_a64(vvA, vb, a.v.sn, a.idx, (long)jAc * S, S, lane);
        trace_mark(a.trace, 1);
        mix_tile_to_lds<TTP, 1, NBUF>(Gt, a.dstate + sofs, a.njg, a.W, a.ldw, M, jgx * TTP, tid);
        if (TTP == 8) load_a64(vvA, vb, a.v.sn, a.idx, (long)jAc * S, S, lane);
        if (a.normalize) {
            const bool expired = tile_wait(a.done + L, a.err, wait_word, tid);
            side_issue<TTP, true>(sk, a.dksum + (long)bh * M * 64, 64, 64, jgx * TTP, M, tid);
#pragma unroll
            for (int t = 0; t < TTP / 8; ++t) sk.v[t] = expired ? __builtin_nanf("") : sk.v[t];   // (no hand-over: dk = NaN, not garbage)
            side_commit<TTP>(sideK, sk, 64, tid);
        }
        trace_mark(a.trace, 2);
        __syncthreads();
        trace_mark(a.trace, 3);
        if constexpr (TTP == 16) {
            bf16x8 kvB[4][2], vvB[4][2];
            load_a64(kvB, kb, a.k.sn, a.idx, (long)jBc * S, S, lane);
            load_a64(vvB, vb, a.v.sn, a.idx, (long)jBc * S, S, lane);
            if (jA < M) compute_store(kvA, vvA, wave, jA, 0, S);
            trace_mark(a.trace, 4);
            if (jB < M) compute_store(kvB, vvB, wave + 8, jB, 0, S);
        } else {
            if (jA < M) compute_store(kvA, vvA, wave, jA, 0, S);
            trace_mark(a.trace, 4);
        }
        trace_mark(a.trace, 5);
        if (a.trace) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); trace_mark(a.trace, 6); }
        return;
    }
    mix_tile_to_lds<TTP, 1, NBUF>(Gt, a.dstate + sofs, a.njg, a.W, a.ldw, M, jgx * TTP, tid);
    const bool expired = a.normalize ? tile_wait_n(a.done + L * cs, cs, a.err, wait_word, tid) : false;   // every dQ part of the tile
    __syncthreads();
    for (int bi = wave; bi < TTP; bi += 8) {
        const int j = jgx * TTP + bi;
        if (j >= M) continue;
        if (a.normalize) {
            float dks = 0.f;
            for (int p = 0; p < cs; ++p) dks += coherent_load(a.dksum + p * a.dks_part + ((long)bh * M + j) * 64 + lane);   // (part order: deterministic)
            sideK[bi * 64 + lane] = expired ? __builtin_nanf("") : dks;
        }
        for (int c0 = part * 64; c0 < S; c0 += 64 * cs) {
            const int rv = min(64, S - c0);
            bf16x8 kv[4][2], vv[4][2];
            load_a64(kv, kb, a.k.sn, a.idx, (long)j * S + c0, rv, lane);
            load_a64(vv, vb, a.v.sn, a.idx, (long)j * S + c0, rv, lane);
            wave_lds_fence();
            compute_store(kv, vv, bi, j, c0, rv);
        }
    }
}

// The backward's token gradients in ONE launch: workgroups [0, ntiles) compute dQ (and dksum), [ntiles, 2 ntiles) dK and dV, the
// last DWR_WGS reduce the dW partials.  dQ and dK/dV tiles of 25-30 us each in two rounds per kernel left the chip waiting for the
// slowest workgroup twice (about 20 us per kernel); as one launch of four rounds the dK/dV tiles fill the dQ tail.
template <int TTP, bool MC = false>
__global__ __launch_bounds__(FT8, TTP == 16 ? 2 : 4) void k_tile_bwd(const FsTokArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    static_assert(TTP == 16, "the flag array and the role ranges are sized for 16-block tiles (tiles_per_bh(njg, 16))");
    const int x = blockIdx.x + a.x0;
    if (x < a.ntiles) tile_bwd_dq_body<TTP, MC>(a, smem_raw, x, a.ntiles);
    else if (x < 2 * a.ntiles) tile_bwd_dkv_body<TTP, MC>(a, smem_raw, x - a.ntiles, a.ntiles);
    else dw_reduce_body(reinterpret_cast<float*>(smem_raw), a.dwp, a.dW, a.M, a.nparts, x - 2 * a.ntiles, threadIdx.x);
}

// Measured without gain in round 3 (code not kept): the same three roles behind per-XCD work queues -- one persistent workgroup
// per CU taking the next tile (atomicAdd on its XCD's counter) when it has finished one, the roles as noinline calls so that the
// loop does not share their register allocation (inlined: 163 spilled VGPRs).  173.8 us against 130.7 us for the static launch
// at C2: what the queue gives back in balance it loses several times over in the call frames (the argument struct goes through
// scratch) and in tiles that no longer start while their predecessors drain their stores.

}  // namespace fast
}  // namespace mhla
