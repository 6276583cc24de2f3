// k_sp_mixh2: k_sp_mixh (mixh.hpp) for 129 .. 256 blocks, re-cut after its slice loop was traced (tools/trace_mixh.py, 256 blocks of 16
// tokens, 20 600 ticks per 64-element slice):
//   * sixteen waves on 128 registers each spilled their per-thread row addresses, and a reload from scratch is a `s_waitcnt vmcnt(0)`: the
//     next slice's request waited for the previous slice's stores and then for its own first load (4 000 ticks);
//   * the weights were rescaled into fp16 hi + lo pairs for EVERY slice -- ~70 VALU operations per reduction step and lane, eight steps,
//     four waves per SIMD: 9 200 ticks per SIMD and slice, against 4 100 for the slice's MFMAs;
//   * every wave read the whole staged slice for one 16-row output tile (16 x 32 KB of LDS reads per slice).
// Here the rescaled pairs are KEPT -- rebuilt, with the weights streamed through the tiles one reduction step at a time (8 temporaries), only
// when the (b, h) changes (the previous slice's stores are flushed first); the rows' multipliers are read where they are needed (the
// rebuild), not with every slice; memory is addressed through buffer descriptors (uniform base and slice offset in SGPRs, one 32-bit
// register per unit and lane); and the cut is chosen so that the pairs FIT: twelve waves x one 16-row output tile for 129 .. 192 blocks
// (48 registers of pairs under a 168-register budget), and for 193 .. 256 blocks the output rows in TWO workgroups of eight waves that each
// read the whole input slice (IH = 2: 64 registers of pairs under 256; the input rows are read twice, the second reader mostly from the
// L2 / MALL).  Sixteen waves x one tile (128-register budget) and eight waves x two tiles (128 registers of pairs) both spill under hipcc
// 7.2 -- and what it spills is the prefetched rows, i.e. the prefetch (DESIGN_APPENDIX.md, "Round 6: measured and not kept").
// Results are bit-identical to k_sp_mixh's: the same expressions in the same order (tests/test_gpu_blockmix.py
// test_recut_mixing_kernel_is_bit_identical_to_the_one_it_replaced).  Used when a workgroup has at least eight slices to pay for the
// rebuilds (capi_common.hpp sp_mixh2_applies).
#pragma once
#include "mixh.hpp"

namespace mhla {
namespace sp {

template <int NW, int RT, int TE, int IH = 1>
__host__ __device__ constexpr int sp_mixh2_smem() {
    constexpr int RO = 16 * NW * RT, RI = RO * IH, tiles = (RI + RO) * (TE + 8) * 2, chunk = 64 * (RO + 4) > RO * 68 ? 64 * (RO + 4) * 4 : RO * 68 * 4;
    return (tiles > chunk ? tiles : chunk) + (RI + RO) * 4;   // (the staged weight chunk of the rebuild lies over the tiles)
}

// mixh_weights, one reduction step at a time: f(ks, w) is called with the lane's eight fp32 weights of step ks (input blocks 32 ks + 8 kg ..,
// output block o0 + obase + nl; ROWS output blocks from o0 are staged, NK steps of input blocks) while the chunk that holds them is staged -- the caller's temporaries are 8 registers, not 8 NK.  Ends with a barrier.
template <int TRANS, int NTH, int ROWS, int NK, typename F>
__device__ __forceinline__ void mixh_weights_ks(float* __restrict__ Tf, const float* __restrict__ W, int ldw, int M, int o0, int obase, int tid, F f) {
    constexpr int TR = TRANS ? 64 : ROWS, TC = TRANS ? ROWS : 64, LDT = TC + 4, PPRW = TC / 4, NCH = (NK + 1) / 2;
    const int lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    const bool vec = ((reinterpret_cast<uintptr_t>(W) & 15) == 0) && (ldw & 3) == 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        __syncthreads();
        for (int v = tid; v < TR * PPRW; v += NTH) {
            const int row = v / PPRW, c4 = (v - row * PPRW) * 4;
            const int gr = TRANS ? c * 64 + row : o0 + row, gc = TRANS ? o0 + c4 : c * 64 + c4;   // (output blocks o0 .. o0 + ROWS - 1 of the matrix)
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (gr < M) {
                const float* src = W + (long)gr * ldw + gc;
                if (vec && gc + 4 <= M) {
                    x = gld<f32x4>(src);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (gc + i < M) x[i] = gld<float>(src + i);
                }
            }
            *reinterpret_cast<f32x4*>(Tf + row * LDT + c4) = x;
        }
        __syncthreads();
        const int o = obase + nl;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            if (2 * c + k2 < NK) {
                float w[8];
                if (TRANS) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) w[t] = Tf[(k2 * 32 + kg * 8 + t) * LDT + o];
                } else {
                    const f32x4 lo4 = *reinterpret_cast<const f32x4*>(Tf + o * LDT + k2 * 32 + kg * 8);
                    const f32x4 hi4 = *reinterpret_cast<const f32x4*>(Tf + o * LDT + k2 * 32 + kg * 8 + 4);
#pragma unroll
                    for (int t = 0; t < 4; ++t) { w[t] = lo4[t]; w[4 + t] = hi4[t]; }
                }
                f(2 * c + k2, w);
            }
        }
    }
    __syncthreads();
}

// NW waves with RT 16-row output tiles each, over IH times as many input rows: a workgroup forms output blocks blockIdx.y * RO .. + RO - 1 (RO =
// 16 NW RT) from RI = IH RO input blocks.  IH = 2: the two halves of the output rows are two workgroups that each read the whole slice --
// the weights a wave keeps are those of ITS output tiles only (8 RI / 32 x 2 registers per tile), which is what lets 256 blocks keep
// hi + lo pairs without spilling: 8 waves x 1 tile x 256 input blocks = 64 registers under a 256-register budget.
template <int NW, int RT, int TRANS, int TE = 64, int IH = 1>
__global__ __launch_bounds__(64 * NW, (NW + 3) / 4) void k_sp_mixh2(const MixrArgs a) {
    constexpr int TEZ = TE / 2, RO = 16 * NW * RT, ROWS = RO * IH, LD = TE + 8, LDZ = LD / 2, NK = ROWS / 32, NT = TE / 16, NTH = 64 * NW;
    constexpr int UPR = TE / 8, NP = ROWS * UPR / NTH, NPO = RO * UPR / NTH;   // 16-byte units per row of a slice (8 payload elements; normaliser: 4 floats), per thread: loads, stores
    static_assert(NP * NTH == ROWS * UPR && NPO * NTH == RO * UPR, "units must tile the slice");
    static_assert(NTH >= ROWS, "one thread per row multiplier");
    const int o0 = blockIdx.y * RO;   // the workgroup's first output block
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Th = reinterpret_cast<u16*>(smem_raw);                  // the slice's input rows [ROWS][LD] (payload; normaliser: bf16 hi | lo halves)
    u16* Os = Th + ROWS * LD;                                    // output payload [RO][LD] (normaliser: fp32 [RO][LDZ])
    float* ms = reinterpret_cast<float*>(smem_raw + sp_mixh2_smem<NW, RT, TE, IH>() - (ROWS + RO) * 4);   // multipliers of the input rows (rows past M: 0), filled by the rebuild
    float* mo = ms + ROWS;                                       // of the workgroup's output rows [RO]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    const int M = a.M;
    const int nsl = (int)((a.E + TE - 1) / TE);
    const long s0 = (long)blockIdx.x * a.spw;
    const int cnt_s = (int)max(0L, min((long)a.spw, a.total - s0));
    const int nzs = (a.S + TEZ - 1) / TEZ;
    if (cnt_s <= 0 && (long)blockIdx.x >= a.ztotal) return;
    const int kend = (M + 31) / 32;   // (uniform) reduction steps that hold a block
    // the operand weights of the slices in hand: the (b, h)'s rescaled fp16 hi + lo pairs (summary slices), then the bf16 hi + lo pairs of
    // the plain weights (normaliser slices)
    bf16x8 wa[RT][NK], wb[RT][NK];
    int wbh = -1;   // the (b, h) the pairs were built for
    // the thread's units: unit v = tid + p NTH -> row v / UPR, piece v % UPR (rows past M: the last row, zeroed at the commit)
    // Memory goes through buffer descriptors (uniform base + uniform slice offset in SGPRs, ONE 32-bit register per unit and lane): with
    // 64-bit per-lane addresses hipcc parked the address pairs -- and then the prefetched rows themselves -- in scratch, and a reload from
    // scratch is a `s_waitcnt vmcnt(0)` in the middle of the prefetch.
    const int nbh = (int)(a.total / nsl);
    const auto rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), (short)0, (int)((long)nbh * M * a.es * 4), 0x00020000);
    const auto rout = __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, (int)((long)nbh * M * a.es * 4), 0x00020000);
    const auto rzin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.zin), (short)0, a.zin ? (int)((long)nbh * M * a.S * 4) : 0, 0x00020000);
    const auto rzout = __builtin_amdgcn_make_buffer_rsrc(a.zout, (short)0, a.zout ? (int)((long)nbh * M * a.S * 4) : 0, 0x00020000);
    unsigned goff[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int v = tid + p * NTH, row = v / UPR;
        goff[p] = (unsigned)(row < M ? row : M - 1) * ((unsigned)a.es * 4u) + (unsigned)(v % UPR) * 16u;   // the unit's place in a slice of its (b, h)
    }
    unsigned ooff[NPO];   // the workgroup's output units: local row (tid + p NTH) / UPR of its RO rows
#pragma unroll
    for (int p = 0; p < NPO; ++p) {
        const int v = tid + p * NTH, row = o0 + v / UPR;
        ooff[p] = (unsigned)(row < M ? row : M - 1) * ((unsigned)a.es * 4u) + (unsigned)(v % UPR) * 16u;
    }
    auto ucol = [&](int p) { return (tid + p * NTH) % UPR; };
    auto urow = [&](int p) { return (tid + p * NTH) / UPR; };
    uint4 pre[NP];
    auto bh_off = [&](int bh) { return (int)((long)bh * M * a.es * 4); };   // (uniform byte offsets: the descriptors' soffset)
    auto zslice_off = [&](int bh, int es) { return (int)(((long)bh * M * a.S + (long)es * TEZ) * 4); };
    auto advance = [&](int& bh, int& es) { if (++es == nsl) { es = 0; ++bh; } };
    // a normaliser slice: rows of S floats in 16-byte units (8-byte halves when S is only even); floats past the row's end are zeroed at the commit
    const bool zwide = (a.S & 3) == 0;   // (uniform)
    auto zoff = [&](int p) { const int row = urow(p); return (unsigned)((row < M ? row : M - 1) * a.S * 4 + ucol(p) * 16); };
    auto zf0 = [&](int p, int es) { return ucol(p) * 4 + es * TEZ; };
    auto zld = [&](int soff, int p, int es) __attribute__((always_inline)) {   // (units past the row's end read the slice's first bytes: dropped at the commit)
        const int f0 = zf0(p, es);
        const unsigned o = zoff(p);
        if (zwide) {
            const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rzin, f0 < a.S ? o : 0u, soff, 2);
            return make_uint4(v[0], v[1], v[2], v[3]);
        }
        const u32x2_t lo = __builtin_amdgcn_raw_buffer_load_b64(rzin, f0 < a.S ? o : 0u, soff, 0), hi = __builtin_amdgcn_raw_buffer_load_b64(rzin, f0 + 2 < a.S ? o + 8u : 0u, soff, 0);
        return make_uint4(lo[0], lo[1], hi[0], hi[1]);
    };
    auto zmask = [&](const uint4& x, bool ok, int p, int es) {
        const int f0 = zf0(p, es);
        return make_uint4((ok && f0 < a.S) ? x.x : 0u, (ok && f0 + 1 < a.S) ? x.y : 0u, (ok && f0 + 2 < a.S) ? x.z : 0u, (ok && f0 + 3 < a.S) ? x.w : 0u);
    };
    // (every load unconditional, from clamped addresses: no branch around a load -- see k_sp_mixr; the two kinds of slices are two loops)
    auto issue = [&]<bool ZS>(std::bool_constant<ZS>, int bh, int es) __attribute__((always_inline)) {
        if constexpr (ZS) {
            const int soff = zslice_off(bh, es);
#pragma unroll
            for (int p = 0; p < NP; ++p) pre[p] = zld(soff, p, es);
            return;
        }
        const int soff = bh_off(bh) + es * (TE * 2);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rin, goff[p], soff, 2);
            pre[p] = make_uint4(v[0], v[1], v[2], v[3]);
        }
    };
    // four floats -> bf16 hi at columns 4 c .., bf16 lo at columns TEZ + 4 c .. of the tile row
    auto commit_hl = [&](u16* th, const uint4& x, int row, int c) {
        const float f[4] = {__uint_as_float(x.x), __uint_as_float(x.y), __uint_as_float(x.z), __uint_as_float(x.w)};
        float l[4];
        unsigned short hs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            hs[i] = cvt_bf16(f[i]);
            l[i] = f[i] - __uint_as_float((unsigned)hs[i] << 16);
        }
        *reinterpret_cast<uint2*>(th + row * LD + c * 4) = make_uint2(hs[0] | ((unsigned)hs[1] << 16), hs[2] | ((unsigned)hs[3] << 16));
        *reinterpret_cast<uint2*>(th + row * LD + TEZ + c * 4) = make_uint2(pack_bf16x2(l[0], l[1]), pack_bf16x2(l[2], l[3]));
    };
    // The stores of a slice are issued at the top of the NEXT iteration, after that slice's loads have been committed (k_sp_mixr).
    int pbh = 0, pes = 0;
    bool pz = false;
    auto store_slice = [&](int bh, int es, bool zslice) __attribute__((always_inline)) {
        if (zslice) {   // the normaliser's rows: 1 / (eps + .) in the forward, as they are in the backward
            const int soff = zslice_off(bh, es);
#pragma unroll
            for (int p = 0; p < NPO; ++p) {
                const int lr = urow(p), row = o0 + lr, c = ucol(p);
                const unsigned zo = (unsigned)(min(row, M - 1) * a.S * 4 + c * 16);
                if (row < M && zf0(p, es) < a.S) {
                    f32x4 x = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(Os) + lr * LDZ + c * 4);
                    if (TRANS == 0)
#pragma unroll
                        for (int i = 0; i < 4; ++i) x[i] = 1.f / (a.eps + x[i]);
                    const u32x4_t xv = __builtin_bit_cast(u32x4_t, x);
                    if (zwide) {
                        __builtin_amdgcn_raw_buffer_store_b128(xv, rzout, zo, soff, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{xv[0], xv[1]}, rzout, zo, soff, 0);
                        if (zf0(p, es) + 2 < a.S) __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{xv[2], xv[3]}, rzout, zo + 8u, soff, 0);
                    }
                }
            }
            return;
        }
        const int soff = bh_off(bh) + es * (TE * 2);
#pragma unroll
        for (int p = 0; p < NPO; ++p) {
            const int lr = urow(p), c = ucol(p);
            if (o0 + lr < M && (long)es * TE + c * 8 < a.E) {
                const uint4 x = *reinterpret_cast<const uint4*>(Os + lr * LD + c * 8);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{x.x, x.y, x.z, x.w}, rout, ooff[p], soff, 0);
                // (the workgroup with a (b, h)'s first slice writes its rows' multipliers: c == 0, so ooff[p] is the row's start)
                if (es == 0 && c == 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mo[lr]), rout, ooff[p] + (unsigned)(2 * a.E), soff, 0);
            }
        }
    };
    // The (b, h)'s operand weights: w'(o, r) = w(o, r) m_r / m_o as fp16 hi + lo, m_o = the power of two >= beta_o = sum_r |w(o, r)| m_r (the
    // expressions of k_sp_mixh, in its order).  The weights come through the tiles (mixh_weights, one output tile at a time: 8 NK
    // temporaries), the input rows' multipliers straight from their rows into `ms`.  The caller has flushed the previous slice's stores and
    // passed a barrier; the first slice's commit follows the staging's last barrier.
    auto rebuild = [&](int bh) __attribute__((always_inline)) {
        if (tid < ROWS) {
            const unsigned mv = __builtin_amdgcn_raw_buffer_load_b32(rin, (unsigned)min(tid, M - 1) * ((unsigned)a.es * 4u) + (unsigned)(2 * a.E), bh_off(bh), 0);
            ms[tid] = tid < M ? __uint_as_float(mv) : 0.f;
        }   // (published by the staging's barriers; `ms`, `mo` lie above the staged chunk)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float beta = 0.f;
            mixh_weights_ks<TRANS, NTH, RO, NK>(reinterpret_cast<float*>(smem_raw), a.W, a.ldw, M, o0, (wave * RT + rt) * 16, tid, [&](int ks, const float (&w)[8]) {
                const f32x4 m0 = *reinterpret_cast<const f32x4*>(ms + ks * 32 + kg * 8), m1 = *reinterpret_cast<const f32x4*>(ms + ks * 32 + kg * 8 + 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) beta += fabsf(w[t] * m0[t]) + fabsf(w[4 + t] * m1[t]);
            });
            beta += __shfl_xor(beta, 16, 64);
            beta += __shfl_xor(beta, 32, 64);
            const float om = h16_mult_from_bound(beta), oinv = h16_inv(om);
            if (kg == 0) mo[(wave * RT + rt) * 16 + nl] = om;
            mixh_weights_ks<TRANS, NTH, RO, NK>(reinterpret_cast<float*>(smem_raw), a.W, a.ldw, M, o0, (wave * RT + rt) * 16, tid, [&](int ks, const float (&w)[8]) {
                const f32x4 m0 = *reinterpret_cast<const f32x4*>(ms + ks * 32 + kg * 8), m1 = *reinterpret_cast<const f32x4*>(ms + ks * 32 + kg * 8 + 4);
                f16x8 wh, wl;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const float x = w[t] * (t < 4 ? m0[t & 3] : m1[t & 3]) * oinv;
                    const _Float16 h = (_Float16)x;
                    wh[t] = h;
                    wl[t] = (_Float16)(x - (float)h);
                }
                wa[rt][ks] = __builtin_bit_cast(bf16x8, wh);
                wb[rt][ks] = __builtin_bit_cast(bf16x8, wl);
            });
        }
    };
    // the normaliser slices' operand weights: the plain weights as bf16 hi + lo
    auto rebuild_z = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
            mixh_weights_ks<TRANS, NTH, RO, NK>(reinterpret_cast<float*>(smem_raw), a.W, a.ldw, M, o0, (wave * RT + rt) * 16, tid, [&](int ks, const float (&w)[8]) {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const __bf16 h = (__bf16)w[t];
                    wa[rt][ks][t] = h;
                    wb[rt][ks][t] = (__bf16)(w[t] - (float)h);
                }
            });
        wbh = -1;
    };
    auto body = [&]<bool ZS>(std::bool_constant<ZS> zs, bool first, bool more, int bh, int es, int nbh, int nes) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int row = urow(p), c = ucol(p);
            const bool ok = row < M;   // rows past the last block: zeros (their weights are zero too, but 0 x NaN is not)
            if constexpr (ZS) {
                commit_hl(Th, zmask(pre[p], ok, p, es), row, c);
            } else {
                const bool live = ok && (long)es * TE + c * 8 < a.E;
                *reinterpret_cast<uint4*>(Th + row * LD + c * 8) = make_uint4(live ? pre[p].x : 0u, live ? pre[p].y : 0u, live ? pre[p].z : 0u, live ? pre[p].w : 0u);
            }
        }
        if (!first) store_slice(pbh, pes, pz);
        __syncthreads();
        if (more) issue(zs, nbh, nes);
        f32x4 acc[RT][NT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (ZS) {
            // plain fp32 rows as bf16 hi | lo halves, unscaled weights as bf16 hi + lo: out = wh hi + wh lo + wl hi (tiles 0 .. NT / 2 - 1: hi, then lo)
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                if (ks < kend) {
#pragma unroll
                    for (int t = 0; t < NT / 2; ++t) {
                        const bf16x8 sh = tr_read8(Th, LD, ks * 32, t * 16, lane), sl = tr_read8(Th, LD, ks * 32, TEZ + t * 16, lane);
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) {
                            acc[rt][t] = mfma_bf16(sh, wa[rt][ks], acc[rt][t]);
                            acc[rt][t] = mfma_bf16(sl, wa[rt][ks], acc[rt][t]);
                            acc[rt][t] = mfma_bf16(sh, wb[rt][ks], acc[rt][t]);
                        }
                    }
                }
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int t = 0; t < NT / 2; ++t)
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(Os) + ((wave * RT + rt) * 16 + nl) * LDZ + t * 16 + kg * 4) = acc[rt][t];
        } else {
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                if (ks < kend) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const f16x8 sv = as_f16x8(tr_read8(Th, LD, ks * 32, t * 16, lane));
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) {
                            acc[rt][t] = mfma_f16(sv, as_f16x8(wa[rt][ks]), acc[rt][t]);
                            acc[rt][t] = mfma_f16(sv, as_f16x8(wb[rt][ks]), acc[rt][t]);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);   // (one reduction step's transpose reads at a time: hoisted across steps they cost the registers the next slice's rows travel in)
            }
            // lane: payload elements 16 t + 4 kg .. + 3 of output block 16 (wave RT + rt) + nl -> staging tile [block][element]
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    *reinterpret_cast<uint2*>(Os + ((wave * RT + rt) * 16 + nl) * LD + t * 16 + kg * 4) =
                        make_uint2(h16_pack2(acc[rt][t][0], acc[rt][t][1]), h16_pack2(acc[rt][t][2], acc[rt][t][3]));
        }
        __syncthreads();
        pbh = bh;
        pes = es;
        pz = ZS;
    };
    // The workgroup's summary slices (a consecutive range), then its share of the normaliser slices (slice zi -> workgroup zi % gridDim.x)
    if (cnt_s > 0) {
        int bh = (int)(s0 / nsl), es = (int)(s0 - (long)bh * nsl);
        issue(std::false_type{}, bh, es);
        for (int it = 0; it < cnt_s; ++it) {
            bool flushed = it == 0;
            if (bh != wbh) {   // (uniform)
                if (it > 0) store_slice(pbh, pes, pz);
                __syncthreads();
                rebuild(bh);
                wbh = bh;
                flushed = true;
            }
            int nb = bh, ne = es;
            advance(nb, ne);
            body(std::false_type{}, flushed, it + 1 < cnt_s, bh, es, nb, ne);
            bh = nb;
            es = ne;
        }
    }
    if ((long)blockIdx.x < a.ztotal) {
        issue(std::true_type{}, (int)(blockIdx.x / nzs), (int)(blockIdx.x % nzs));
        if (cnt_s > 0) store_slice(pbh, pes, pz);   // (the weights go through the tiles)
        __syncthreads();
        rebuild_z();
        bool firstz = true;
        for (long zi = blockIdx.x; zi < a.ztotal; zi += gridDim.x) {
            const long nzi = zi + gridDim.x;
            body(std::true_type{}, firstz, nzi < a.ztotal, (int)(zi / nzs), (int)(zi % nzs), (int)(nzi / nzs), (int)(nzi % nzs));
            firstz = false;
        }
    }
    store_slice(pbh, pes, pz);
}

}  // namespace sp
}  // namespace mhla
