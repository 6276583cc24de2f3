// 16-block output tiles for the bf16 fast path (see fused.hpp).  The mixing phase is bound by L2 traffic
// (every tile workgroup streams the whole (b,h) state: 512 KB); tiles of 16 blocks halve that traffic and
// use all 16 MFMA columns.  512 threads, one workgroup per CU (LDS: 16 x 9 KB mixed summaries), wave w owns
// blocks w and w + 8 of the tile; everything else as in the 8-block kernels.
#pragma once
#include "fused.hpp"

namespace mhla {
namespace fast {

constexpr int TT = 16;                              // blocks per tile
constexpr int GSLOT = FD * GLD + 8;                  // elements per mixed-summary slot (+16 B: spreads the 16 blocks over banks)
constexpr int FS_GT16_BYTES = TT * GSLOT * 2;       // 147712 B

template <int TRANSW>
__device__ __forceinline__ void mix16_tile_to_lds(u16* __restrict__ Gt, const u16* __restrict__ state_bh, int njg,
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
            if (i < M && j < M) w = TRANSW ? W[(long)j * ldw + i] : W[(long)i * ldw + j];
            const u16 h = cvt_bf16(w);
            hi[t] = (short)h;
            lo[t] = (short)cvt_bf16(w - bf(h));
        }
        bhi[ks] = __builtin_bit_cast(bf16x8, hi);
        blo[ks] = __builtin_bit_cast(bf16x8, lo);
    }
    const bool two = njg > 4;
    constexpr int UN = 4, NW = FT8 / 64;
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
        // UN independent accumulator chains, interleaved step by step (no back-to-back dependent MFMAs)
        f32x4 c[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) c[u] = mfma_bf16(__builtin_bit_cast(bf16x8, av[u][0]), bhi[0], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int u = 0; u < UN; ++u) c[u] = mfma_bf16(__builtin_bit_cast(bf16x8, av[u][0]), blo[0], c[u]);
        if (two) {
#pragma unroll
            for (int u = 0; u < UN; ++u) c[u] = mfma_bf16(__builtin_bit_cast(bf16x8, av[u][1]), bhi[1], c[u]);
#pragma unroll
            for (int u = 0; u < UN; ++u) c[u] = mfma_bf16(__builtin_bit_cast(bf16x8, av[u][1]), blo[1], c[u]);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int et = et0 + u, d2 = et >> 2, d1 = (et & 3) * 16 + kg * 4;
            uint2 pk;
            pk.x = pack_bf16x2(c[u][0], c[u][1]);
            pk.y = pack_bf16x2(c[u][2], c[u][3]);
            *reinterpret_cast<uint2*>(Gt + (long)n * GSLOT + d2 * GLD + d1) = pk;
        }
    };
    constexpr int NB = FE / 16 / NW / UN;
    uint4 bufA[UN][2], bufB[UN][2];
    load_batch(bufA, wave * UN);
#pragma unroll 1
    for (int bt = 0; bt < NB; bt += 2) {
        load_batch(bufB, (wave + NW * (bt + 1)) * UN);
        do_batch(bufA, (wave + NW * bt) * UN);
        if (bt + 2 < NB) load_batch(bufA, (wave + NW * (bt + 2)) * UN);
        do_batch(bufB, (wave + NW * (bt + 1)) * UN);
    }
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

__global__ __launch_bounds__(FT8, 2) void k_t16_out(const FsOutArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Gt = reinterpret_cast<u16*>(smem_raw);   // [8][64 d2][72]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kg = lane >> 4;
    const int L = xcd_swizzle(blockIdx.x, gridDim.x);
    const int ntt = (a.njg + 1) / 2, bh = L / ntt, it = L - bh * ntt, b = bh / a.H, h = bh - b * a.H;
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
        u16* Gb = Gt + bi * GSLOT;
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
        const int iA = it * TT + wave, iB = iA + 8;
        bf16x8 avA[4][2], avB[4][2];
        float ninvA = 1.f, ninvB = 1.f;
        trace_mark(a.trace, 0);
        if (iA < M) load_blk(avA, ninvA, iA, 0, S);
        trace_mark(a.trace, 1);
        mix16_tile_to_lds<0>(Gt, state_bh, a.njg, a.W, a.ldw, M, it * TT, tid);
        trace_mark(a.trace, 2);
        __syncthreads();
        trace_mark(a.trace, 3);
        if (iB < M) load_blk(avB, ninvB, iB, 0, S);
        if (iA < M) compute_store(avA, ninvA, wave, iA, 0, S);
        trace_mark(a.trace, 4);
        if (iB < M) compute_store(avB, ninvB, wave + 8, iB, 0, S);
        trace_mark(a.trace, 5);
        if (a.trace) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); trace_mark(a.trace, 6); }
        return;
    }

    mix16_tile_to_lds<0>(Gt, state_bh, a.njg, a.W, a.ldw, M, it * TT, tid);
    __syncthreads();
    for (int bi = wave; bi < TT; bi += 8) {
        const int i = it * TT + bi;
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

__global__ __launch_bounds__(FT8, 2) void k_t16_bwd_dq(const FsTokArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Gt = reinterpret_cast<u16*>(smem_raw);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, kg = lane >> 4;
    const int L = xcd_swizzle(blockIdx.x, gridDim.x);
    const int ntt = (a.njg + 1) / 2, bh = L / ntt, jgx = L - bh * ntt, b = bh / a.H, h = bh - b * a.H;
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
        u16* Gb = Gt + bi * GSLOT;
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
        const int jA = jgx * TT + wave, jB = jA + 8;
        bf16x8 gvA[4][2], gvB[4][2];
        Side sA, sB;
        trace_mark(a.trace, 0);
        if (jA < M) load_blk(gvA, sA, jA, 0, S);
        trace_mark(a.trace, 1);
        mix16_tile_to_lds<0>(Gt, a.state + sofs, a.njg, a.W, a.ldw, M, jgx * TT, tid);
        trace_mark(a.trace, 2);
        __syncthreads();
        trace_mark(a.trace, 3);
        if (jB < M) load_blk(gvB, sB, jB, 0, S);
        float dks_acc[2][8];
        if (jA < M) { zero_dks(dks_acc); compute_store(gvA, sA, dks_acc, wave, jA, 0, S); if (a.normalize) finish_dks(dks_acc, jA); }
        trace_mark(a.trace, 4);
        if (jB < M) { zero_dks(dks_acc); compute_store(gvB, sB, dks_acc, wave + 8, jB, 0, S); if (a.normalize) finish_dks(dks_acc, jB); }
        trace_mark(a.trace, 5);
        if (a.trace) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); trace_mark(a.trace, 6); }
        return;
    }
    mix16_tile_to_lds<0>(Gt, a.state + sofs, a.njg, a.W, a.ldw, M, jgx * TT, tid);
    __syncthreads();
    for (int bi = wave; bi < TT; bi += 8) {
        const int j = jgx * TT + bi;
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

__global__ __launch_bounds__(FT8, 2) void k_t16_bwd_dkv(const FsTokArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Gt = reinterpret_cast<u16*>(smem_raw);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, kg = lane >> 4;
    const int L = xcd_swizzle(blockIdx.x, gridDim.x);
    const int ntt = (a.njg + 1) / 2, bh = L / ntt, jgx = L - bh * ntt, b = bh / a.H, h = bh - b * a.H;
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
        u16* Gb = Gt + bi * GSLOT;
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
        const int jA = jgx * TT + wave, jB = jA + 8;
        bf16x8 kvA[4][2], kvB[4][2];
        trace_mark(a.trace, 0);
        if (jA < M) load_k(kvA, jA, 0, S);
        trace_mark(a.trace, 1);
        mix16_tile_to_lds<1>(Gt, a.dstate + sofs, a.njg, a.W, a.ldw, M, jgx * TT, tid);
        trace_mark(a.trace, 2);
        __syncthreads();
        trace_mark(a.trace, 3);
        if (jB < M) load_k(kvB, jB, 0, S);
        if (jA < M) compute_store(kvA, wave, jA, 0, S);
        trace_mark(a.trace, 4);
        if (jB < M) compute_store(kvB, wave + 8, jB, 0, S);
        trace_mark(a.trace, 5);
        if (a.trace) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); trace_mark(a.trace, 6); }
        return;
    }
    mix16_tile_to_lds<1>(Gt, a.dstate + sofs, a.njg, a.W, a.ldw, M, jgx * TT, tid);
    __syncthreads();
    for (int bi = wave; bi < TT; bi += 8) {
        const int j = jgx * TT + bi;
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
