// k_sp_mixh: the resident-sequence mixing of split.hpp (k_sp_mixr) for block summaries in the h16 format -- an fp16 payload x one
// power-of-two multiplier per block row (split.hpp, "h16") -- computed ON THE PAYLOAD with the fp16 MFMA:
//
//     out_o[e] = sum_r Wm(o, r) m_r pay_r[e] = m_o * sum_r w'(o, r) pay_r[e],     w'(o, r) = Wm(o, r) m_r / m_o,  |w'| <= 1,
//
// where m_o is the power of two >= beta_o = sum_r |Wm(o, r)| m_r (the same bound in every workgroup, so the output row's multiplier
// needs no reduction over its slices).  The payload slice goes from memory to LDS as it is (no decode, no lo tile: half the LDS of the
// hi + lo form, so a 128-element slice -- 256-byte row pieces, 16 KB per workgroup and slice at 64 blocks -- fits four workgroups per
// CU: the bytes in flight that k_sp_mixr had with 3-byte summaries); the weights are rescaled per slice into fp16 hi + lo pairs
// (22 significand bits; 8 NK values per lane), two MFMAs per product instead of three; the accumulators ARE the output payload.
// DW (backward, M <= 128): dW[i][j] += m_i m'_j sum_e pay_i[e] pay'_j[e] from the staged dG and KV payload rows -- one fp16 MFMA per
// product, the products of two fp16 values are exact in the fp32 accumulator.
// The normaliser's rows (plain fp32: z / dn) ride along as extra slices of 64 values, staged as bf16 [hi (64) | lo (64)] halves of
// the same tile and multiplied on the bf16 MFMA with the unscaled weights (hi + lo), as in k_sp_mixr.
// Summary rows whose length is not a multiple of 128 (D = 72, 56, ...: E % 128 = 64) end in a half slice: its upper units are
// zeroed at the commit and not stored.
#pragma once
#include "split.hpp"

namespace mhla {
namespace sp {

using fast::f16x8;
using fast::mfma_f16;
using fast::as_f16x8;

// slice width in summary elements (256-byte row pieces; 128-byte ones above eight waves: 129 .. 256 blocks, where 8 NK fp32 weights per
// lane -- 48 / 64 registers -- leave room for four accumulator tiles, not eight) and in normaliser values (half: bf16 hi | lo halves)
template <int NW> __host__ __device__ constexpr int mixh_te() { return NW > 8 ? 64 : 128; }
template <int NW> __host__ __device__ constexpr int mixh_tez() { return mixh_te<NW>() / 2; }
template <int NW, bool DW>
__host__ __device__ constexpr int sp_mixh_smem() { return (DW ? 3 : 2) * 16 * NW * (mixh_te<NW>() + 8) * 2 + 3 * 16 * NW * 4; }

// The workgroup's mixing weights as fp32 B-operand fragments: wf[ks][t] = weight of input block r = 32 ks + 8 kg + t in output block
// o = obase + nl (TRANS: W[r][o], else W[o][r]); blocks past M: 0.  Fetched through LDS like mixr_weights.  Ends with a barrier.
template <int TRANS, int NTH, int ROWS, int NK>
__device__ __forceinline__ void mixh_weights(float (&wf)[NK][8], float* __restrict__ Tf, const float* __restrict__ W, int ldw, int M, int obase, int tid) {
    constexpr int TR = TRANS ? 64 : ROWS, TC = TRANS ? ROWS : 64, LDT = TC + 4, PPRW = TC / 4, NCH = (NK + 1) / 2;
    const int lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    const bool vec = ((reinterpret_cast<uintptr_t>(W) & 15) == 0) && (ldw & 3) == 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        __syncthreads();
        for (int v = tid; v < TR * PPRW; v += NTH) {
            const int row = v / PPRW, c4 = (v - row * PPRW) * 4;
            const int gr = TRANS ? c * 64 + row : row, gc = TRANS ? c4 : c * 64 + c4;
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
                if (TRANS) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) wf[2 * c + k2][t] = Tf[(k2 * 32 + kg * 8 + t) * LDT + o];
                } else {
                    const f32x4 lo4 = *reinterpret_cast<const f32x4*>(Tf + o * LDT + k2 * 32 + kg * 8);
                    const f32x4 hi4 = *reinterpret_cast<const f32x4*>(Tf + o * LDT + k2 * 32 + kg * 8 + 4);
#pragma unroll
                    for (int t = 0; t < 4; ++t) { wf[2 * c + k2][t] = lo4[t]; wf[2 * c + k2][4 + t] = hi4[t]; }
                }
            }
        }
    }
    __syncthreads();
}

// MixrArgs as for k_sp_mixr: total = bh * ceil(E / TE) summary slices, ztotal = bh * ceil(S / TEZ) normaliser slices.
template <int NW, int TRANS, bool DW>
// (launch bounds: four waves per SIMD -- 128 VGPRs -- up to four waves per workgroup, i.e. 8 / 4 workgroups per CU; eight waves would spill
// 30 registers there and run one workgroup per CU; the dW variants hold NW more accumulator tiles and two staged sets: 2 / 1 per CU)
__global__ __launch_bounds__(64 * NW, DW ? (NW + 3) / 4 : (NW <= 4 ? 4 : (NW + 3) / 4)) void k_sp_mixh(const MixrArgs a) {
    static_assert(!DW || (TRANS == 1 && NW <= 8), "dW rides in the backward's mixing kernel, M <= 128");
    constexpr int TE = mixh_te<NW>(), TEZ = mixh_tez<NW>(), ROWS = 16 * NW, LD = TE + 8, LDZ = LD / 2, NK = (NW + 1) / 2, NT = TE / 16, NTH = 64 * NW;
    constexpr int UPR = TE / 8, NP = ROWS * UPR / NTH;   // 16-byte units per row of a slice (8 payload elements; normaliser: 4 floats), per thread
    static_assert(NP * NTH == ROWS * UPR, "units must tile the slice");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Th = reinterpret_cast<u16*>(smem_raw);                  // the slice's input rows [ROWS][LD] (payload; normaliser: bf16 hi | lo halves)
    u16* Kh = Th + ROWS * LD;                                    // (DW: the other summary set's rows)
    u16* Os = Th + (DW ? 2 : 1) * ROWS * LD;                     // output payload [ROWS][LD] (normaliser: fp32 [ROWS][LDZ])
    float* ms = reinterpret_cast<float*>(Os + ROWS * LD);        // multipliers of the input rows (rows past M: 0)
    float* ms2 = ms + ROWS;                                      // (DW: of the other set's rows)
    float* mo = ms2 + ROWS;                                      // of the output rows
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nl = lane & 15, kg = lane >> 4;
    const int M = a.M;
    const int nsl = (int)((a.E + TE - 1) / TE);
    const long s0 = (long)blockIdx.x * a.spw;
    const int cnt_s = (int)max(0L, min((long)a.spw, a.total - s0));
    const int nzs = (a.S + TEZ - 1) / TEZ;
    if (cnt_s <= 0 && (long)blockIdx.x >= a.ztotal) return;
    float wf[NK][8];
    static_assert((TRANS ? 64 * (ROWS + 4) : ROWS * 68) * 4 <= sp_mixh_smem<NW, DW>(), "weight chunk must fit in the tiles");
    const int kend = (M + 31) / 32;   // (uniform) reduction steps that hold a block
    // the thread's units: unit v = tid + p NTH -> row v / 16, piece v % 16 (rows past M: the last row, zeroed at the commit)
    unsigned goff[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int v = tid + p * NTH, row = v / UPR;
        goff[p] = (unsigned)((long)(row < M ? row : M - 1) * a.es * 4);   // the row's start in its (b, h)
    }
    auto ucol = [&](int p) { return (tid + p * NTH) % UPR; };
    auto urow = [&](int p) { return (tid + p * NTH) / UPR; };
    struct Stage {
        uint4 pre[NP], pre2[DW ? NP : 1];
        float prs[NP], prs2[DW ? NP : 1];
    };
    Stage sga;
    auto bh_off = [&](int bh) { return (long)bh * M * a.es * 4; };
    auto zslice_off = [&](int bh, int es) { return ((long)bh * M * a.S + (long)es * TEZ) * 4; };
    auto advance = [&](int& bh, int& es) { if (++es == nsl) { es = 0; ++bh; } };
    // a normaliser slice: rows of S floats in 16-byte units (8-byte halves when S is only even); floats past the row's end are zeroed at the commit
    const bool zwide = (a.S & 3) == 0;   // (uniform)
    auto zoff = [&](int p) { const int row = urow(p); return (unsigned)((row < M ? row : M - 1) * a.S * 4 + ucol(p) * 16); };
    auto zf0 = [&](int p, int es) { return ucol(p) * 4 + es * TEZ; };
    auto zld = [&](const char* base, int p, int es) __attribute__((always_inline)) {
        const int f0 = zf0(p, es);
        const char* src = base + zoff(p);
        if (zwide) return gld_stream16(f0 < a.S ? src : base);
        const uint2 lo = gld<uint2>(f0 < a.S ? src : base), hi = gld<uint2>(f0 + 2 < a.S ? src + 8 : base);
        return make_uint4(lo.x, lo.y, hi.x, hi.y);
    };
    auto zmask = [&](const uint4& x, bool ok, int p, int es) {
        const int f0 = zf0(p, es);
        return make_uint4((ok && f0 < a.S) ? x.x : 0u, (ok && f0 + 1 < a.S) ? x.y : 0u, (ok && f0 + 2 < a.S) ? x.z : 0u, (ok && f0 + 3 < a.S) ? x.w : 0u);
    };
    // (every load unconditional, from clamped addresses: no branch around a load -- see k_sp_mixr; the two kinds of slices are two loops)
    auto issue = [&]<bool ZS>(std::bool_constant<ZS>, Stage& g, int bh, int es) __attribute__((always_inline)) {
        if constexpr (ZS) {
            const long boff = zslice_off(bh, es);
            const char* base = reinterpret_cast<const char*>(a.zin) + boff;
#pragma unroll
            for (int p = 0; p < NP; ++p) g.pre[p] = zld(base, p, es);
            if constexpr (DW) {
                const char* base2 = reinterpret_cast<const char*>(a.zin2) + boff;
#pragma unroll
                for (int p = 0; p < NP; ++p) g.pre2[p] = zld(base2, p, es);
            }
            return;
        }
        const char* rb = reinterpret_cast<const char*>(a.in) + bh_off(bh);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            // (a half slice's upper units read the row's multiplier and padding: dropped at the commit)
            g.pre[p] = gld_stream16(rb + goff[p] + (long)es * (TE * 2) + ucol(p) * 16);
            g.prs[p] = gld<float>(rb + goff[p] + 2 * a.E);
        }
        if constexpr (DW) {
            const char* rb2 = reinterpret_cast<const char*>(a.in2) + bh_off(bh);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                g.pre2[p] = gld_stream16(rb2 + goff[p] + (long)es * (TE * 2) + ucol(p) * 16);
                g.prs2[p] = gld<float>(rb2 + goff[p] + 2 * a.E);
            }
        }
    };
    f32x4 dwacc[DW ? NW : 1];
#pragma unroll
    for (int t = 0; t < (DW ? NW : 1); ++t) dwacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // four floats -> bf16 hi at columns 4 c .., bf16 lo at columns 64 + 4 c .. of the tile row
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
            char* zb = reinterpret_cast<char*>(a.zout) + zslice_off(bh, es);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int row = urow(p), c = ucol(p);
                if (row < M && zf0(p, es) < a.S) {
                    f32x4 x = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(Os) + row * LDZ + c * 4);
                    if (TRANS == 0)
#pragma unroll
                        for (int i = 0; i < 4; ++i) x[i] = 1.f / (a.eps + x[i]);
                    if (zwide) {
                        *reinterpret_cast<f32x4*>(zb + zoff(p)) = x;
                    } else {
                        *reinterpret_cast<f32x2*>(zb + zoff(p)) = f32x2{x[0], x[1]};
                        if (zf0(p, es) + 2 < a.S) *reinterpret_cast<f32x2*>(zb + zoff(p) + 8) = f32x2{x[2], x[3]};
                    }
                }
            }
            return;
        }
        char* ob = reinterpret_cast<char*>(a.out) + bh_off(bh);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int row = urow(p), c = ucol(p);
            if (row < M && (long)es * TE + c * 8 < a.E) {
                gst<uint4>(ob + goff[p] + (long)es * (TE * 2) + c * 16, *reinterpret_cast<const uint4*>(Os + row * LD + c * 8));
                if (es == 0 && c == 0) gst<float>(ob + goff[p] + 2 * a.E, mo[row]);   // (the workgroup with a (b, h)'s first slice writes its rows' multipliers)
            }
        }
    };
    // MIXH_TRACE builds (tools/trace_mixh.py): s_memtime stamps of the first eight workgroups' summary slices -- 0 entry | 1 commit + previous
    // slice's stores issued | 2 barrier | 3 next slice requested | 4 weights rescaled | 5 products | 6 staged | 7 barrier
    int tk = 0;
    auto stamp = [&](int i) {
#ifdef MIXH_TRACE
        if (a.trace && tid == 0 && blockIdx.x < 8 && tk < 32) a.trace[(blockIdx.x * 32 + tk) * 8 + i] = __builtin_amdgcn_s_memtime();
#else
        (void)i;
        (void)tk;
#endif
    };
    auto body = [&]<bool ZS>(std::bool_constant<ZS> zs, Stage& g, bool first, bool more, int bh, int es, int nbh, int nes) __attribute__((always_inline)) {
        if constexpr (!ZS) stamp(0);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int row = urow(p), c = ucol(p);
            const bool ok = row < M;   // rows past the last block: zeros (their weights are zero too, but 0 x NaN is not)
            if constexpr (ZS) {
                commit_hl(Th, zmask(g.pre[p], ok, p, es), row, c);
                if constexpr (DW) commit_hl(Kh, zmask(g.pre2[p], ok, p, es), row, c);
            } else {
                const bool live = ok && (long)es * TE + c * 8 < a.E;
                *reinterpret_cast<uint4*>(Th + row * LD + c * 8) = make_uint4(live ? g.pre[p].x : 0u, live ? g.pre[p].y : 0u, live ? g.pre[p].z : 0u, live ? g.pre[p].w : 0u);
                if (c == 0) ms[row] = ok ? g.prs[p] : 0.f;
                if constexpr (DW) {
                    *reinterpret_cast<uint4*>(Kh + row * LD + c * 8) = make_uint4(live ? g.pre2[p].x : 0u, live ? g.pre2[p].y : 0u, live ? g.pre2[p].z : 0u, live ? g.pre2[p].w : 0u);
                    if (c == 0) ms2[row] = ok ? g.prs2[p] : 0.f;
                }
            }
        }
        if (!first) store_slice(pbh, pes, pz);
        if constexpr (!ZS) stamp(1);
        __syncthreads();
        if constexpr (!ZS) stamp(2);
        if (more) issue(zs, g, nbh, nes);
        if constexpr (!ZS) stamp(3);
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (ZS) {
            // plain fp32 rows as bf16 hi | lo halves, unscaled weights as bf16 hi + lo: out = wh hi + wh lo + wl hi (tiles 0 .. 3: hi, 4 .. 7: lo)
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                if (ks < kend) {
                    bf16x8 zwh, zwl;
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const __bf16 h = (__bf16)wf[ks][t];
                        zwh[t] = h;
                        zwl[t] = (__bf16)(wf[ks][t] - (float)h);
                    }
#pragma unroll
                    for (int t = 0; t < NT / 2; ++t) {
                        const bf16x8 sh = tr_read8(Th, LD, ks * 32, t * 16, lane), sl = tr_read8(Th, LD, ks * 32, TEZ + t * 16, lane);
                        acc[t] = mfma_bf16(sh, zwh, acc[t]);
                        acc[t] = mfma_bf16(sl, zwh, acc[t]);
                        acc[t] = mfma_bf16(sh, zwl, acc[t]);
                    }
                }
            }
            if constexpr (DW) {   // <dn_i, z_j>: (hi + lo) . (hi + lo) without the lo . lo term, over the 64 values of the slice
                if (wave * 16 < M) {
#pragma unroll
                    for (int k2 = 0; k2 < TEZ / 32; ++k2) {
                        const bf16x8 ah = row_read8(Th, LD, wave * 16, k2 * 32, lane), al = row_read8(Th, LD, wave * 16, TEZ + k2 * 32, lane);
#pragma unroll
                        for (int jt = 0; jt < NW; ++jt) {
                            if (jt * 16 < M) {
                                const bf16x8 bh_ = row_read8(Kh, LD, jt * 16, k2 * 32, lane), bl_ = row_read8(Kh, LD, jt * 16, TEZ + k2 * 32, lane);
                                dwacc[jt] = mfma_bf16(ah, bh_, dwacc[jt]);
                                dwacc[jt] = mfma_bf16(ah, bl_, dwacc[jt]);
                                dwacc[jt] = mfma_bf16(al, bh_, dwacc[jt]);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < NT / 2; ++t)
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(Os) + (wave * 16 + nl) * LDZ + t * 16 + kg * 4) = acc[t];
        } else {
            // this slice's weights: w'(o, r) = w(o, r) m_r / m_o as fp16 hi + lo, m_o = the power of two >= beta_o = sum_r |w(o, r)| m_r
            float beta = 0.f;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const f32x4 m0 = *reinterpret_cast<const f32x4*>(ms + ks * 32 + kg * 8), m1 = *reinterpret_cast<const f32x4*>(ms + ks * 32 + kg * 8 + 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) beta += fabsf(wf[ks][t] * m0[t]) + fabsf(wf[ks][4 + t] * m1[t]);
            }
            beta += __shfl_xor(beta, 16, 64);
            beta += __shfl_xor(beta, 32, 64);
            const float om = h16_mult_from_bound(beta), oinv = h16_inv(om);
            if (kg == 0) mo[wave * 16 + nl] = om;
            stamp(4);
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                if (ks < kend) {
                    // (the multipliers are read again rather than kept: 8 NK registers less)
                    const f32x4 m0 = *reinterpret_cast<const f32x4*>(ms + ks * 32 + kg * 8), m1 = *reinterpret_cast<const f32x4*>(ms + ks * 32 + kg * 8 + 4);
                    f16x8 wh, wl;
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const float w = wf[ks][t] * (t < 4 ? m0[t & 3] : m1[t & 3]) * oinv;
                        const _Float16 h = (_Float16)w;
                        wh[t] = h;
                        wl[t] = (_Float16)(w - (float)h);
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const f16x8 sv = as_f16x8(tr_read8(Th, LD, ks * 32, t * 16, lane));
                        acc[t] = mfma_f16(sv, wh, acc[t]);
                        acc[t] = mfma_f16(sv, wl, acc[t]);
                    }
                }
            }
            stamp(5);
            if constexpr (DW) {   // dW[i][j] += m_i m'_j sum_e pay_i[e] pay'_j[e]: rows i of this wave, every column tile that holds a block
                if (wave * 16 < M) {   // (uniform)
                    f32x4 tmp[NW];
#pragma unroll
                    for (int jt = 0; jt < NW; ++jt) tmp[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k2 = 0; k2 < TE / 32; ++k2) {
                        const f16x8 av = as_f16x8(row_read8(Th, LD, wave * 16, k2 * 32, lane));   // A[m = i][k = e]
#pragma unroll
                        for (int jt = 0; jt < NW; ++jt)
                            if (jt * 16 < M) tmp[jt] = mfma_f16(av, as_f16x8(row_read8(Kh, LD, jt * 16, k2 * 32, lane)), tmp[jt]);   // B[k = e][n = j]
                    }
                    const f32x4 mi = *reinterpret_cast<const f32x4*>(ms + wave * 16 + kg * 4);
#pragma unroll
                    for (int jt = 0; jt < NW; ++jt) {
                        if (jt * 16 < M) {
                            const float mj = ms2[jt * 16 + nl];
#pragma unroll
                            for (int r = 0; r < 4; ++r) dwacc[jt][r] += tmp[jt][r] * (mi[r] * mj);
                        }
                    }
                }
            }
            // lane: payload elements 16 t + 4 kg .. + 3 of output block 16 wave + nl -> staging tile [block][element]
#pragma unroll
            for (int t = 0; t < NT; ++t)
                *reinterpret_cast<uint2*>(Os + (wave * 16 + nl) * LD + t * 16 + kg * 4) = make_uint2(h16_pack2(acc[t][0], acc[t][1]), h16_pack2(acc[t][2], acc[t][3]));
        }
        if constexpr (!ZS) stamp(6);
        __syncthreads();
        if constexpr (!ZS) { stamp(7); ++tk; }
        pbh = bh;
        pes = es;
        pz = ZS;
    };
    // The workgroup's summary slices (a consecutive range), then its share of the normaliser slices (slice zi -> workgroup zi % gridDim.x)
    // (the first slice's rows are requested before the weights are fetched through LDS: with four slices per workgroup at the C2 shape the
    // prologue's two round trips were a fifth of the kernel)
    if (cnt_s > 0) issue(std::false_type{}, sga, (int)(s0 / nsl), (int)(s0 - (long)(s0 / nsl) * nsl));
    mixh_weights<TRANS, NTH, ROWS, NK>(wf, reinterpret_cast<float*>(smem_raw), a.W, a.ldw, M, wave * 16, tid);
    if (cnt_s > 0) {
        int bh = (int)(s0 / nsl), es = (int)(s0 - (long)bh * nsl);
        for (int it = 0; it < cnt_s; ++it) {
            int nb = bh, ne = es;
            advance(nb, ne);
            body(std::false_type{}, sga, it == 0, it + 1 < cnt_s, bh, es, nb, ne);
            bh = nb;
            es = ne;
        }
    }
    {
        bool firstz = cnt_s <= 0;
        for (long zi = blockIdx.x; zi < a.ztotal; zi += gridDim.x) {
            const long nzi = zi + gridDim.x;
            if (zi == (long)blockIdx.x) issue(std::true_type{}, sga, (int)(zi / nzs), (int)(zi % nzs));
            body(std::true_type{}, sga, firstz, nzi < a.ztotal, (int)(zi / nzs), (int)(zi % nzs), (int)(nzi / nzs), (int)(nzi % nzs));
            firstz = false;
        }
    }
    store_slice(pbh, pes, pz);
    if constexpr (DW) {   // C layout: rows i = 16 wave + 4 kg + r, column j = 16 jt + nl
        float* dp = a.dwp + (long)blockIdx.x * M * M;
#pragma unroll
        for (int jt = 0; jt < NW; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = wave * 16 + kg * 4 + r, j = jt * 16 + nl;
                if (i < M && j < M) dp[(long)i * M + j] = dwacc[jt][r];
            }
    }
}

}  // namespace sp
}  // namespace mhla
