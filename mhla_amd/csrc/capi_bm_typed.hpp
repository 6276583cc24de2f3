// Generic (exact fp32 MFMA) and split-operand block-mix launches for one element type: the body of the (dtype, head-dim tile)
// dispatch of mhla_blockmix_fwd / _bwd.  Included by capi_bm_f32.hip, capi_bm_bf16.hip, capi_bm_f16.hip, each of which
// instantiates the two functions for its type -- the three are compiled side by side.
#pragma once
#include "capi_common.hpp"
#include "blockmix.hpp"
#include "split.hpp"
#include "split16.hpp"
#include "mixh2.hpp"

namespace mhla {
namespace capi {

// Resident-sequence mixing (sp::k_sp_mixr): 33 <= M <= 256 blocks, summaries in whole 256-byte row pieces; other M: the tiled
// kernel sp::k_sp_mix.
template <int DT> constexpr int sp_state_threads() {
#ifdef MHLA_SP_STATE_4WAVES
    return NTHREADS;
#else
    return DT == 8 ? 512 : NTHREADS;
#endif
}
template <bool S16>
inline bool sp_mixr_ok(int M, long E) {
    return (M > 32 || !S16) && M <= 256 && E % sp::mixr_te<4, S16>() == 0;   // (fp32-grade summaries: two waves for up to 32 blocks)
}
// the normaliser's product (k_wz) rides along in the LDS-DMA mixing kernel: same weights, at most 16 values per block
// ... and in the register-staged kernel at fp32 summaries, as extra slices (blocks of an even number of tokens: 16- or 8-byte pieces)
template <bool S16>
inline bool sp_mixr_takes_wz(int M, int S) { return S16 ? (M > 192 && M <= 256 && S <= 16) : (S % 2 == 0); }
template <int TRANS, bool S16, int P24 = 0>   // P24: 0 fp32 words (bf16 when S16), 1 24-bit floats, 2 h16
inline int sp_mixr(const float* W, int ldw, const void* in, void* out, int M, long E, long es, int BH, hipStream_t st,
                   const float* zin = nullptr, float* zout = nullptr, int S = 0, float eps = 0.f) {
#define MIXR(NW) do { \
        constexpr int TE = sp::mixr_te<NW, S16>(); \
        const long total = (long)BH * (E / TE); \
        const bool wz = !S16 && zin && sp_mixr_takes_wz<S16>(M, S); \
        const long zt = wz ? (long)BH * ((S + TE - 1) / TE) : 0;   /* normaliser slices: dealt round-robin over the workgroups */ \
        /* persistent workgroups: as many as fit a CU beside each other (35 KB of LDS at four waves, 70 KB at eight) */ \
        const int wgs = (int)std::min<long>(total, 256 * (NW <= 2 ? 8 : NW <= 4 ? 4 : NW <= 8 ? 2 : 1)); \
        sp::MixrArgs a{W, ldw, in, out, M, E, es, total, (int)((total + wgs - 1) / wgs), wz ? zin : nullptr, wz ? zout : nullptr, wz ? S : 0, eps, g_trace.load(), nullptr, nullptr, zt, nullptr}; \
        const int gw = (int)((total + a.spw - 1) / a.spw); \
        return launch(sp::k_sp_mixr<NW, TRANS, S16, false, P24>, dim3(gw), dim3(64 * NW), sp::sp_mixr_smem<NW, S16>(), st, TRANS ? "k_sp_mixr<1>" : "k_sp_mixr<0>", a); \
    } while (0)
    if constexpr (S16) {
        if (M > 192) {   // eight waves x 32 output blocks, LDS-DMA staging with three slices in flight
            const long total = (long)BH * (E / 64);
            const int wgs = (int)std::min<long>(total, 256);
            const bool wz = zin && sp_mixr_takes_wz<S16>(M, S);
            sp::MixrArgs a{W, ldw, in, out, M, E, es, total, (int)((total + wgs - 1) / wgs), wz ? zin : nullptr, wz ? zout : nullptr, S, eps, g_trace.load()};
            const int gw = (int)((total + a.spw - 1) / a.spw);
            return launch(sp::k_sp_mixr_dma<TRANS>, dim3(gw), dim3(sp::MIXR_DMA_T), sp::sp_mixr_dma_smem(), st, TRANS ? "k_sp_mixr_dma<1>" : "k_sp_mixr_dma<0>", a);
        }
    }
    if constexpr (!S16) {
        if (M <= 32) MIXR(2);
    }
    if (M <= 64) MIXR(4);
    if (M <= 128) MIXR(8);
    if (M <= 192) MIXR(12);
    if constexpr (P24) return fail(MHLA_EINVAL, "sp_mixr: p24 summaries at M=%d", M);   // (capi_common.hpp bm_sumfmt admits up to 192 blocks)
    else if constexpr (!S16) MIXR(16);   // (16-bit summaries: taken by the DMA kernel above)
    return fail(MHLA_EINVAL, "sp_mixr: M=%d out of range", M);
#undef MIXR
}

// The backward's mixing dKV = W^T dG with the dW products riding along (sp::k_sp_mixr<.., DW>): fp32 summaries, 33 <= M <= 128.
// Returns the number of [M][M] partials written to `dwp` (one per workgroup) through `nparts`.
inline bool sp_mixr_dw_ok(int M, long E) { return sp_mixr_ok<false>(M, E) && M <= 128; }
// dn / z / dz (null: no normaliser): dz = W^T dn and the <dn_i, z_j> term of dW ride along as extra slices when the block length allows
// (sp_mixr_takes_wz); `*wz_done` tells the caller whether they did.
template <int P24 = 0>
inline int sp_mixr_dw(const float* W, int ldw, const void* dg, const void* kv, void* dkv, float* dwp, int M, long E, long es, int BH,
                      hipStream_t st, int* nparts, const float* dn, const float* z, float* dz, int S, bool* wz_done) {
    const bool wz = dn && z && dz && sp_mixr_takes_wz<false>(M, S);
    *wz_done = wz;
#define MIXRDW(NW) do { \
        constexpr int TE = sp::mixr_te<NW, false>(); \
        const long total = (long)BH * (E / TE); \
        const long zt = wz ? (long)BH * ((S + TE - 1) / TE) : 0; \
        const int wgs = (int)std::min<long>(total, 256 * (NW <= 2 ? 4 : NW <= 4 ? 2 : 1));   /* 54 KB of LDS at four waves: two per CU; 27 KB at two waves */ \
        sp::MixrArgs a{W, ldw, dg, dkv, M, E, es, total, (int)((total + wgs - 1) / wgs), wz ? dn : nullptr, wz ? dz : nullptr, wz ? S : 0, 0.f, nullptr, kv, dwp, zt, wz ? z : nullptr}; \
        const int gw = (int)((total + a.spw - 1) / a.spw); \
        *nparts = gw; \
        return launch(sp::k_sp_mixr<NW, 1, false, true, P24>, dim3(gw), dim3(64 * NW), sp::sp_mixr_smem<NW, false, true>(), st, "k_sp_mixr<1,dw>", a); \
    } while (0)
    if (M <= 32) MIXRDW(2);
    if (M <= 64) MIXRDW(4);
    MIXRDW(8);
#undef MIXRDW
}

// h16 summaries: the mixing on the fp16 payload (mixh.hpp k_sp_mixh).  Slices of 128 elements (the last one of a row may be half), the
// normaliser's rows as extra slices of 64 values when the block length is even; persistent workgroups, as many as fit a CU.
#ifndef SP_MIXH2_TE
#define SP_MIXH2_TE 64   // slice width of k_sp_mixh2 in summary elements
#endif
#ifndef SP_MIXH2_IH
#define SP_MIXH2_IH 2   // 193 .. 256 blocks: the output rows of k_sp_mixh2 in this many workgroups (each reads the whole slice)
#endif
#ifndef SP_MIXH2_RT
#define SP_MIXH2_RT 1   // 16-row output tiles per wave of k_sp_mixh2 (1: twelve / sixteen waves, 2: six / eight)
#endif
template <int TRANS>
inline int sp_mixh(const float* W, int ldw, const void* in, void* out, int M, long E, long es, int BH, hipStream_t st,
                   const float* zin, float* zout, int S, float eps) {
#define MIXH(NW, PERCU) do { \
        const long total = (long)BH * ((E + sp::mixh_te<NW>() - 1) / sp::mixh_te<NW>()); \
        const bool wz = zin && sp_mixr_takes_wz<false>(M, S); \
        const long zt = wz ? (long)BH * ((S + sp::mixh_tez<NW>() - 1) / sp::mixh_tez<NW>()) : 0; \
        const int wgs = (int)std::min<long>(total, 256 * (PERCU)); \
        sp::MixrArgs a{W, ldw, in, out, M, E, es, total, (int)((total + wgs - 1) / wgs), wz ? zin : nullptr, wz ? zout : nullptr, wz ? S : 0, eps, g_trace.load(), nullptr, nullptr, zt, nullptr}; \
        const int gw = (int)((total + a.spw - 1) / a.spw); \
        return launch(sp::k_sp_mixh<NW, TRANS, false>, dim3(gw), dim3(64 * NW), sp::sp_mixh_smem<NW, false>(), st, TRANS ? "k_sp_mixh<1>" : "k_sp_mixh<0>", a); \
    } while (0)
    if (M <= 32) MIXH(2, 8);
    if (M <= 64) MIXH(4, 4);
    if (M <= 128) MIXH(8, 1);
    // 129 .. 256 blocks: 64-element slices, one workgroup per CU; dW by k_sp_dwr<.., h16>.  With eight or more slices per workgroup the
    // re-cut kernel (mixh2.hpp: half the waves, two output tiles each, the rescaled weights kept per (b, h))
#define MIXH2(NW, RT, IH) do { \
        const long total = (long)BH * ((E + SP_MIXH2_TE - 1) / SP_MIXH2_TE); \
        const bool wz = zin && sp_mixr_takes_wz<false>(M, S); \
        const long zt = wz ? (long)BH * ((S + SP_MIXH2_TE / 2 - 1) / (SP_MIXH2_TE / 2)) : 0; \
        const int wgs = (int)std::min<long>(total, 256); \
        const int spw = (int)((total + wgs - 1) / wgs); \
        if (SP_MIXH2_TE == 64 ? sp_mixh2_applies(M, E, BH, S) : spw >= SP_MIXH2_MIN_SLICES) { \
            sp::MixrArgs a{W, ldw, in, out, M, E, es, total, spw, wz ? zin : nullptr, wz ? zout : nullptr, wz ? S : 0, eps, nullptr, nullptr, nullptr, zt, nullptr}; \
            const int gw = (int)((total + a.spw - 1) / a.spw); \
            return launch(sp::k_sp_mixh2<NW, RT, TRANS, SP_MIXH2_TE, IH>, dim3(gw, IH), dim3(64 * NW), sp::sp_mixh2_smem<NW, RT, SP_MIXH2_TE, IH>(), st, TRANS ? "k_sp_mixh2<1>" : "k_sp_mixh2<0>", a); \
        } \
    } while (0)
    if (M <= 192) MIXH2(12 / SP_MIXH2_RT, SP_MIXH2_RT, 1);
    if (M <= 192) MIXH(12, 1);
    if (M <= 256) MIXH2(16 / SP_MIXH2_RT / SP_MIXH2_IH, SP_MIXH2_RT, SP_MIXH2_IH);
    if (M <= 256) MIXH(16, 1);
#undef MIXH2
    return fail(MHLA_EINVAL, "sp_mixh: M=%d out of range", M);
#undef MIXH
}
// ... the backward's, with the dW products riding along: one [M][M] partial per workgroup (`*nparts`); `*wz_done`: dz = W^T dn and the
// <dn_i, z_j> term of dW rode along too
inline int sp_mixh_dw(const float* W, int ldw, const void* dg, const void* kv, void* dkv, float* dwp, int M, long E, long es, int BH,
                      hipStream_t st, int* nparts, const float* dn, const float* z, float* dz, int S, bool* wz_done) {
    const bool wz = dn && z && dz && sp_mixr_takes_wz<false>(M, S);
    *wz_done = wz;
#define MIXHDW(NW, PERCU) do { \
        const long total = (long)BH * ((E + sp::mixh_te<NW>() - 1) / sp::mixh_te<NW>()); \
        const long zt = wz ? (long)BH * ((S + sp::mixh_tez<NW>() - 1) / sp::mixh_tez<NW>()) : 0; \
        const int wgs = (int)std::min<long>(total, 256 * (PERCU));   /* (bm_carve: at most 1024 / 512 / 256 partials at two / four / eight waves) */ \
        sp::MixrArgs a{W, ldw, dg, dkv, M, E, es, total, (int)((total + wgs - 1) / wgs), wz ? dn : nullptr, wz ? dz : nullptr, wz ? S : 0, 0.f, nullptr, kv, dwp, zt, wz ? z : nullptr}; \
        const int gw = (int)((total + a.spw - 1) / a.spw); \
        *nparts = gw; \
        return launch(sp::k_sp_mixh<NW, 1, true>, dim3(gw), dim3(64 * NW), sp::sp_mixh_smem<NW, true>(), st, "k_sp_mixh<1,dw>", a); \
    } while (0)
    if (M <= 32) MIXHDW(2, 4);
    if (M <= 64) MIXHDW(4, 2);
    if (M <= 128) MIXHDW(8, 1);
    return fail(MHLA_EINVAL, "sp_mixh_dw: M=%d out of range", M);
#undef MIXHDW
}

// p24 summaries (split.hpp: 24-bit floats, 3 / 4 of the bytes of every summary transfer) on the resident-mixing pipeline: 16-bit tensors
// with head dims up to 96 and up to 128 blocks (the fused dW), and fp32 tensors at head dims 113 .. 128 with 33 .. 192 blocks (the Wan
// shape, rotary tables and fused epilogue included: the operands are bf16 hi + lo pairs there too, 16 significand bits either way).  A
// function of the call's shape, dtype and flags only: a forward and the backward that reuses its state agree.
template <typename ET, int DT, bool S16>
constexpr bool bm_p24_built() { return !S16 && ((sizeof(ET) == 2 && DT <= 6) || (std::is_same<ET, float>::value && DT == 8)); }
// h16 (2-byte summaries, the default on 16-bit tensors): the same pipeline, instantiated for 16-bit element types (capi_common.hpp bm_sumfmt decides per call)
template <typename ET, int DT, bool S16>
constexpr bool bm_h16_built() { return !S16 && sizeof(ET) == 2 && DT <= 6; }
// Run `...` with PF = the kernels' summary-format template value of this call (1: p24, 2: h16); `fmt` is SF_P24 or SF_H16 here
#define WITH_PF(fmt, ...) do { \
        if ((fmt) == SF_H16) { if constexpr (H16OK) { constexpr int PF = 2; __VA_ARGS__; } else return fail(MHLA_EINVAL, "h16 summaries are not built for this type / head dim"); } \
        else { constexpr int PF = 1; __VA_ARGS__; } \
    } while (0)

// Blocks of exactly 16 tokens, bf16, D = 64, every row a whole number of 16-byte pieces: the wave-per-block kernels of split16.hpp
// replace the token kernels (same workspace formats).  `bwd`: the call's gradient views must qualify too.
template <typename ET, int DT>
inline bool s16_ok(const BmCall& c, bool bwd) {
    if constexpr (!std::is_same<ET, bf16_t>::value || DT != 4) {
        return false;
    } else {
        if (c.S != 16 || c.D != 64 || c.split || c.rcos || c.epi || !sp_shape_ok(c.D, c.flags)) return false;
        if (!(view_ok16(c.q_num) && view_ok16(c.k_num) && view_ok16(c.v))) return false;
        if (!bwd) return view_ok16m(c.out);
        return view_ok16(c.dout) && view_ok16(c.outv) && view_ok16m(c.dq_num) && view_ok16m(c.dk_num) && view_ok16m(c.dv);
    }
}
// dW partials with the whole M x M matrix in one workgroup (s16::k_sp_dwr): 64 < M <= 256, 16-bit summaries
inline bool sp_dwr_ok(int M, long E) { return M > 64 && M <= 256 && E % 64 == 0; }
inline int sp_dwr_splits(int BH, long E) {
    int ns = (256 + BH - 1) / BH;
    if (ns > DW_MAX_SPLIT) ns = DW_MAX_SPLIT;
    while (ns > 1 && E / ns < 256) --ns;
    return ns < 1 ? 1 : ns;
}
// (`es`: row stride in 16-bit elements.  h16: the rows are fp16 payload with their multiplier behind the E elements -- products on the fp16 MFMA)
inline int sp_dwr(const void* x, const void* y, long E, long es, const float* x2, const float* y2, int S2, float* out, int M, int BH, int nsplit, hipStream_t st,
                  bool h16 = false) {
    s16::DwrArgs d{(const sp::u16*)x, (const sp::u16*)y, E, es, x2, y2, S2, out, M, nsplit};
    if (h16) {
        if (M <= 128) return launch(s16::k_sp_dwr<2, true>, dim3(nsplit, BH), dim3(256), s16::dwr_smem<2>(), st, "k_sp_dwr<2,h16>", d);
        if (M <= 192) return launch(s16::k_sp_dwr<3, true>, dim3(nsplit, BH), dim3(576), s16::dwr_smem<3>(), st, "k_sp_dwr<3,h16>", d);
        return launch(s16::k_sp_dwr<4, true>, dim3(nsplit, BH), dim3(1024), s16::dwr_smem<4>(), st, "k_sp_dwr<4,h16>", d);
    }
    if (M <= 128) return launch(s16::k_sp_dwr<2>, dim3(nsplit, BH), dim3(256), s16::dwr_smem<2>(), st, "k_sp_dwr<2>", d);
    if (M <= 192) return launch(s16::k_sp_dwr<3>, dim3(nsplit, BH), dim3(576), s16::dwr_smem<3>(), st, "k_sp_dwr<3>", d);
    return launch(s16::k_sp_dwr<4>, dim3(nsplit, BH), dim3(1024), s16::dwr_smem<4>(), st, "k_sp_dwr<4>", d);
}

// KV/ksum/z, G for the forward and the recompute leg of the backward.
template <typename T, int DT, bool S16>
int bm_state_and_mix(const mhla_view& q_num, const mhla_view& k_num, const mhla_view& v, const mhla_view& q_den,
                     const mhla_view& k_den, const float* W, int ldw, const int32_t* idx, const BmWs& w, int B, int H,
                     int M, int S, int D, float eps, unsigned flags, bool normalize, bool split, hipStream_t st,
                     const float* rcos = nullptr, const float* rsin = nullptr, long ldr = 0, bool s16 = false) {
    (void)q_num;
    constexpr bool P24OK = bm_p24_built<T, DT, S16>(), H16OK = bm_h16_built<T, DT, S16>();
    const bool p24 = w.fmt == SF_P24 || w.fmt == SF_H16;   // (capi_common.hpp bm_sumfmt; the row stride w.es is the format's)
    const long es = w.es;
    (void)H16OK;
    StateArgs a{};
    a.rcos = rcos; a.rsin = rsin; a.ldr = ldr;
    a.x = cv(k_num); a.y = cv(v); a.kd = cv(k_den); a.qd = cv(q_den); a.idx = idx;
    a.out = w.kv; a.ksum = w.ksum; a.zo = w.z; a.es = es;
    a.H = H; a.M = M; a.S = S; a.D = D; a.eps = eps;
    a.relu = (flags & MHLA_FLAG_RELU_EPS) ? 1 : 0; a.normalize = normalize; a.split = split;
    MixArgs m{W, ldw, w.kv, w.g, M, (long)D * D, es};
    if (sp_shape_ok(D, flags)) {   // split-bf16 MFMA kernels (split.hpp)
        constexpr int SNT = sp_state_threads<DT>();   // eight waves at D = 128 (split.hpp)
        if constexpr (P24OK) {
            if (p24) {
                if (a.rcos) {
                    if constexpr (std::is_same<T, float>::value)
                        RC(launch(sp::k_sp_state<T, DT, 0, true, SNT, false, 1>, dim3(M, B * H), dim3(SNT), sp::sp_state_smem<DT>(), st, "k_sp_state<rope>", a));
                } else
                    WITH_PF(w.fmt, RC(launch(sp::k_sp_state<T, DT, 0, false, SNT, false, PF>, dim3(M, B * H), dim3(SNT), sp::sp_state_smem<DT>(), st, "k_sp_state", a)));
                if (w.fmt == SF_H16) RC(sp_mixh<0>(W, ldw, w.kv, w.g, M, m.E, es, B * H, st, normalize ? (const float*)w.z : nullptr, w.ninv, S, eps));
                else RC((sp_mixr<0, false, 1>(W, ldw, w.kv, w.g, M, m.E, es, B * H, st, normalize ? (const float*)w.z : nullptr, w.ninv, S, eps)));
                if (normalize && !sp_mixr_takes_wz<false>(M, S))
                    RC(launch(k_wz<0>, dim3((S + 63) / 64, (M + 63) / 64, B * H), dim3(NTHREADS), 0, st, "k_wz<0>", W, ldw, (const float*)w.z, w.ninv, M, S, eps));
                return MHLA_OK;
            }
        }
        if (s16)    RC(launch(s16::k_s16_state<0>, dim3((M + s16::WPB - 1) / s16::WPB, B * H), dim3(64 * s16::WPB), s16::state_smem(), st, "k_s16_state<0>", a));
        else if (a.rcos) RC(launch(sp::k_sp_state<T, DT, 0, true, SNT, S16>, dim3(M, B * H), dim3(SNT), sp::sp_state_smem<DT>(), st, "k_sp_state<rope>", a));
        else        RC(launch(sp::k_sp_state<T, DT, 0, false, SNT, S16>, dim3(M, B * H), dim3(SNT), sp::sp_state_smem<DT>(), st, "k_sp_state", a));
        const bool mixr = sp_mixr_ok<S16>(M, m.E), wz_fused = mixr && normalize && sp_mixr_takes_wz<S16>(M, S);
        if (mixr) RC((sp_mixr<0, S16>(W, ldw, w.kv, w.g, M, m.E, w.es, B * H, st, normalize ? (const float*)w.z : nullptr, w.ninv, S, eps)));
        else RC(launch(sp::k_sp_mix<0, S16>, dim3((unsigned)((m.E + sp::SPM_TE - 1) / sp::SPM_TE), (M + 63) / 64, B * H), dim3(NTHREADS), sp::sp_mix_smem<S16>(), st, "k_sp_mix<0>", m));
        if (normalize && !wz_fused)
            RC(launch(k_wz<0>, dim3((S + 63) / 64, (M + 63) / 64, B * H), dim3(NTHREADS), 0, st, "k_wz<0>", W, ldw, (const float*)w.z, w.ninv, M, S, eps));
        return MHLA_OK;
    }
    RC(launch(k_bm_state<T, DT, 0>, dim3(M, B * H), dim3(NTHREADS), state_smem_floats<DT>() * 4, st, "k_bm_state<0>", a));
    dim3 grid((unsigned)((m.E + MIX_TE - 1) / MIX_TE), (M + MIX_TI - 1) / MIX_TI, B * H);
    RC(launch(k_mix<0, 0>, grid, dim3(NTHREADS), MIX_SMEM_FLOATS * 4, st, "k_mix<0,0>", m));
    if (normalize)
        RC(launch(k_wz<0>, dim3((S + 63) / 64, (M + 63) / 64, B * H), dim3(NTHREADS), 0, st, "k_wz<0>", W, ldw, (const float*)w.z, w.ninv, M, S, eps));
    return MHLA_OK;
}

// S16: the block summaries are stored as bf16 (bf16 tensors with MHLA_FLAG_BF16_SUMMARIES); false: fp32 summaries, hi + lo operands
template <typename ET, bool S16>
int bm_fwd_typed(const BmCall& c) {
    const mhla_view &q_num = c.q_num, &k_num = c.k_num, &v = c.v, &q_den = c.q_den, &k_den = c.k_den, &dout = c.dout, &gate = c.gate;
    const mhla_view& out_view = c.outv;
    const mhla_mview &dq_num = c.dq_num, &dk_num = c.dk_num, &dv = c.dv, &dq_den = c.dq_den, &dk_den = c.dk_den;
    const float* W = c.W; const int ldw = c.ldw; float* dW = c.dW; const int32_t* block_index = c.block_index;
    const BmWs& w = c.w;
    const int B = c.B, H = c.H, M = c.M, S = c.S, D = c.D; const float eps = c.eps; const unsigned flags = c.flags;
    const bool normalize = c.normalize, split = c.split, reuse = c.reuse, epi = c.epi;
    hipStream_t st = c.st;
    const float *rcos = c.rcos, *rsin = c.rsin; const long ldr = c.ldr;
    const float* nw = c.nw; const float neps = c.neps; const int out_dtype = c.out_dtype;
    const int dt = dt_for(D), relu = (flags & MHLA_FLAG_RELU_EPS) ? 1 : 0;
    (void)dout; (void)gate; (void)out_view; (void)dq_num; (void)dk_num; (void)dv; (void)dq_den; (void)dk_den; (void)dW; (void)reuse; (void)epi;
    (void)rcos; (void)rsin; (void)ldr; (void)nw; (void)neps; (void)out_dtype; (void)relu;
    DISPATCH_DT(dt, {
        const bool s16 = S16 && s16_ok<ET, DT>(c, false);
        RC((bm_state_and_mix<ET, DT, S16>(q_num, k_num, v, q_den, k_den, W, ldw, block_index, w, B, H, M, S, D, eps, flags, normalize, split, st, rcos, rsin, ldr, s16)));
        OutArgs o{};
        o.rcos = rcos; o.rsin = rsin; o.ldr = ldr;
        o.q = cv(q_num); o.o = cmv(c.out); o.idx = block_index; o.W = W; o.ldw = ldw; o.g = w.g; o.ninv = w.ninv;
        constexpr bool P24OK = bm_p24_built<ET, DT, S16>(), H16OK = bm_h16_built<ET, DT, S16>();
        const bool p24 = w.fmt == SF_P24 || w.fmt == SF_H16;
        (void)H16OK;
        o.H = H; o.M = M; o.S = S; o.D = D; o.eps = eps; o.es = w.es;
        o.relu = (flags & MHLA_FLAG_RELU_EPS) ? 1 : 0; o.normalize = normalize;
        o.olo = (normalize && !epi && !(flags & MHLA_FLAG_NO_BWD_STATE)) ? w.olo : nullptr;   // (16-bit tensors, default arithmetic: BmWs::olo)
        if (epi) {
            if constexpr (std::is_same<ET, float>::value) {
                o.nw = nw; o.neps = neps; o.gate = cv(gate);
                const dim3 g(M, B * H), blk(sp::SP_OUT_T);
                if (p24) {
                    if constexpr (P24OK) {
                        if (out_dtype == MHLA_BF16)     RC(launch(sp::k_sp_out<float, DT, bf16_t, true, false, 1>, g, blk, sp::sp_out_smem<DT>(), st, "k_sp_out<norm>", o));
                        else if (out_dtype == MHLA_F16) RC(launch(sp::k_sp_out<float, DT, f16_t, true, false, 1>, g, blk, sp::sp_out_smem<DT>(), st, "k_sp_out<norm>", o));
                        else                            RC(launch(sp::k_sp_out<float, DT, float, true, false, 1>, g, blk, sp::sp_out_smem<DT>(), st, "k_sp_out<norm>", o));
                    }
                } else
                if (out_dtype == MHLA_BF16)     RC(launch(sp::k_sp_out<float, DT, bf16_t, true>, g, blk, sp::sp_out_smem<DT>(), st, "k_sp_out<norm>", o));
                else if (out_dtype == MHLA_F16) RC(launch(sp::k_sp_out<float, DT, f16_t, true>, g, blk, sp::sp_out_smem<DT>(), st, "k_sp_out<norm>", o));
                else                            RC(launch(sp::k_sp_out<float, DT, float, true>, g, blk, sp::sp_out_smem<DT>(), st, "k_sp_out<norm>", o));
            }
        } else if (s16)
            RC(launch(s16::k_s16_out<0>, dim3((M + s16::WPB - 1) / s16::WPB, B * H), dim3(64 * s16::WPB), s16::out_smem(), st, "k_s16_out", o));
        else if (p24) {
            if constexpr (P24OK) WITH_PF(w.fmt, RC(launch(sp::k_sp_out<ET, DT, ET, false, false, PF>, dim3(M, B * H), dim3(sp::SP_OUT_T), sp::sp_out_smem<DT, false>(), st, "k_sp_out", o)));
        } else if (sp_shape_ok(D, flags))
            RC(launch(sp::k_sp_out<ET, DT, ET, false, S16>, dim3(M, B * H), dim3(sp::SP_OUT_T), sp::sp_out_smem<DT, S16>(), st, "k_sp_out", o));
        else
            RC(launch(k_bm_out<ET, DT>, dim3(M, B * H), dim3(NTHREADS), out_smem_floats<DT>() * 4, st, "k_bm_out", o));
    });
    return MHLA_OK;
}

template <typename ET, bool S16>
int bm_bwd_typed(const BmCall& c) {
    const mhla_view &q_num = c.q_num, &k_num = c.k_num, &v = c.v, &q_den = c.q_den, &k_den = c.k_den, &dout = c.dout, &gate = c.gate;
    const mhla_view& out_view = c.outv;
    const mhla_mview &dq_num = c.dq_num, &dk_num = c.dk_num, &dv = c.dv, &dq_den = c.dq_den, &dk_den = c.dk_den;
    const float* W = c.W; const int ldw = c.ldw; float* dW = c.dW; const int32_t* block_index = c.block_index;
    const BmWs& w = c.w;
    const int B = c.B, H = c.H, M = c.M, S = c.S, D = c.D; const float eps = c.eps; const unsigned flags = c.flags;
    const bool normalize = c.normalize, split = c.split, reuse = c.reuse, epi = c.epi;
    hipStream_t st = c.st;
    const float *rcos = c.rcos, *rsin = c.rsin; const long ldr = c.ldr;
    const float* nw = c.nw; const float neps = c.neps; const int out_dtype = c.out_dtype;
    const int dt = dt_for(D), relu = (flags & MHLA_FLAG_RELU_EPS) ? 1 : 0;
    (void)dout; (void)gate; (void)out_view; (void)dq_num; (void)dk_num; (void)dv; (void)dq_den; (void)dk_den; (void)dW; (void)reuse; (void)epi;
    (void)rcos; (void)rsin; (void)ldr; (void)nw; (void)neps; (void)out_dtype; (void)relu;
    DISPATCH_DT(dt, {
        const bool s16 = S16 && s16_ok<ET, DT>(c, true);
        if (!reuse)
            RC((bm_state_and_mix<ET, DT, S16>(q_num, k_num, v, q_den, k_den, W, ldw, block_index, w, B, H, M, S, D, eps, flags, normalize, split, st, rcos, rsin, ldr, s16)));
        constexpr bool P24OK = bm_p24_built<ET, DT, S16>(), H16OK = bm_h16_built<ET, DT, S16>();
        const bool p24 = w.fmt == SF_P24 || w.fmt == SF_H16;
        const long es = w.es;
        (void)H16OK;
        const bool want_olo = normalize && w.olo != nullptr;
        if (want_olo && (!reuse || (flags & MHLA_FLAG_NO_BWD_STATE))) {
            // what the forward's 16-bit store of O rounded away (BmWs::olo), recomputed: the output kernel without its output
            OutArgs o{};
            o.q = cv(q_num); o.o = MView{nullptr, 0, 0, 0}; o.idx = block_index; o.W = W; o.ldw = ldw; o.g = w.g; o.ninv = w.ninv;
            o.H = H; o.M = M; o.S = S; o.D = D; o.eps = eps; o.es = es; o.relu = relu; o.normalize = normalize;
            o.olo = c.olo_own; o.skip_out = 1;
            if (p24) {
                if constexpr (P24OK) WITH_PF(w.fmt, RC(launch(sp::k_sp_out<ET, DT, ET, false, false, PF>, dim3(M, B * H), dim3(sp::SP_OUT_T), sp::sp_out_smem<DT, false>(), st, "k_sp_out<olo>", o)));
            } else if (sp_shape_ok(D, flags))
                RC(launch(sp::k_sp_out<ET, DT, ET, false, S16>, dim3(M, B * H), dim3(sp::SP_OUT_T), sp::sp_out_smem<DT, S16>(), st, "k_sp_out<olo>", o));
            else
                RC(launch(k_bm_out<ET, DT>, dim3(M, B * H), dim3(NTHREADS), out_smem_floats<DT>() * 4, st, "k_bm_out<olo>", o));
        }
        // dG_i = Q_i^T (dO_i / n_i), dn_i
        StateArgs a{};
        a.x = cv(q_num); a.y = cv(dout); a.o = cv(out_view); a.idx = block_index; a.W = W; a.ldw = ldw; a.ninv = w.ninv;
        a.out = w.dg; a.dn = w.dn; a.es = es; a.H = H; a.M = M; a.S = S; a.D = D; a.eps = eps;
        a.relu = relu; a.normalize = normalize; a.split = split;
        a.olo = !want_olo ? nullptr : ((!reuse || (flags & MHLA_FLAG_NO_BWD_STATE)) ? c.olo_own : w.olo);
        const int tiles = (M + 63) / 64;
        TokArgs t{};
        t.q = cv(q_num); t.k = cv(k_num); t.v = cv(v); t.qd = cv(q_den); t.kd = cv(k_den); t.dout = cv(dout);
        t.dq = cmv(dq_num); t.dk = cmv(dk_num); t.dv = cmv(dv); t.dqd = cmv(dq_den); t.dkd = cmv(dk_den);
        t.idx = block_index; t.W = W; t.ldw = ldw; t.g = w.g; t.dkv = w.dkv; t.ninv = w.ninv; t.dz = w.dz; t.ksum = w.ksum;
        t.dks = w.dks; t.es = es;
        t.H = H; t.M = M; t.S = S; t.D = D; t.eps = eps; t.relu = relu; t.normalize = normalize; t.split = split;
        if (sp_shape_ok(D, flags)) {   // split-bf16 MFMA kernels (split.hpp)
            const long E = (long)D * D;
            a.rcos = rcos; a.rsin = rsin; a.ldr = ldr;
            t.rcos = rcos; t.rsin = rsin; t.ldr = ldr;
            constexpr int SNT = sp_state_threads<DT>();
            if constexpr (P24OK) {
                if (p24) {   // the same kernels on 24-bit summaries: dG, then dKV = W^T dG with dz (and, up to 128 blocks, dW) riding along, the token gradients
                    constexpr bool F32 = std::is_same<ET, float>::value;
                    const bool rope = F32 && rcos != nullptr;
                    a.g = w.g;   // (16-bit tensors, D <= 64: the row dots come from G_i; two more LDS tiles)
                    constexpr int SM1 = (sizeof(ET) == 2 && DT <= 4) ? sp::sp_state_rd_smem<DT>() : sp::sp_state_smem<DT>();
                    if (rope) {
                        if constexpr (F32) RC(launch(sp::k_sp_state<ET, DT, 1, true, SNT, false, 1>, dim3(M, B * H), dim3(SNT), sp::sp_state_smem<DT>(), st, "k_sp_state<1,rope>", a));
                    } else
                        WITH_PF(w.fmt, RC(launch(sp::k_sp_state<ET, DT, 1, false, SNT, false, PF>, dim3(M, B * H), dim3(SNT), SM1, st, "k_sp_state<1>", a)));
                    int parts = 0;
                    bool wz_done = false, dwz_done = false;   // dz = W^T dn formed / the <dn_i, z_j> term of dW included
                    if (sp_mixr_dw_ok(M, E)) {
                        if (w.fmt == SF_H16) RC(sp_mixh_dw(W, ldw, w.dg, w.kv, w.dkv, w.dwp, M, E, es, B * H, st, &parts, normalize ? (const float*)w.dn : nullptr,
                                                           normalize ? (const float*)w.z : nullptr, w.dz, S, &wz_done));
                        else RC(sp_mixr_dw<1>(W, ldw, w.dg, w.kv, w.dkv, w.dwp, M, E, es, B * H, st, &parts, normalize ? (const float*)w.dn : nullptr,
                                            normalize ? (const float*)w.z : nullptr, w.dz, S, &wz_done));
                        dwz_done = wz_done;
                    } else if (w.fmt == SF_H16) {   // 129 .. 256 blocks on h16: the mixing without the dW tiles, dW (and its <dn, z> term) by the
                        wz_done = normalize && sp_mixr_takes_wz<false>(M, S);   // whole-matrix kernel on the two payload sets
                        RC(sp_mixh<1>(W, ldw, w.dg, w.dkv, M, E, es, B * H, st, normalize ? (const float*)w.dn : nullptr, w.dz, S, 0.f));
                        const int nsplit = sp_dwr_splits(B * H, E);
                        RC(sp_dwr(w.dg, w.kv, E, 2 * es, normalize ? w.dn : nullptr, normalize ? w.z : nullptr, S, w.dwp, M, B * H, nsplit, st, true));
                        parts = B * H * nsplit;
                        dwz_done = true;
                    } else {   // 129 .. 192 blocks: twelve waves have no registers for the dW tiles -- k_sp_dw reads dG and KV again
                        wz_done = normalize && sp_mixr_takes_wz<false>(M, S);
                        RC((sp_mixr<1, false, 1>(W, ldw, w.dg, w.dkv, M, E, es, B * H, st, normalize ? (const float*)w.dn : nullptr, w.dz, S, 0.f)));
                        int nsplit = dw_splits(tiles * tiles * B * H, E);
                        if (nsplit > DW_MAX_SPLIT - 1) nsplit = DW_MAX_SPLIT - 1;
                        DwArgs d{w.dg, w.kv, E, nullptr, nullptr, 0, w.dwp, M, tiles, nsplit, es};
                        RC(launch(sp::k_sp_dwt<true>, dim3(tiles * tiles, B * H, nsplit), dim3(NTHREADS), sp::SP_DWT_SMEM, st, "k_sp_dwt", d));
                        parts = B * H * nsplit;
                    }
                    if (normalize && !wz_done)
                        RC(launch(k_wz<1>, dim3((S + 63) / 64, (M + 63) / 64, B * H), dim3(NTHREADS), 0, st, "k_wz<1>", W, ldw, (const float*)w.dn, w.dz, M, S, 0.f));
                    if (normalize && !dwz_done) {
                        DwArgs dzz{w.dn, w.z, (long)S, nullptr, nullptr, 0, w.dwp + (size_t)parts * M * M, M, tiles, 1};
                        RC(launch(k_dw<0>, dim3(tiles * tiles, B * H, 1), dim3(NTHREADS), DW_SMEM_FLOATS * 4, st, "k_dw", dzz));
                        parts += B * H;
                    }
                    RC(launch(k_dw_reduce<0, 16>, dim3((M * M + 15) / 16), dim3(256), 0, st, "k_dw_reduce", (const float*)w.dwp, (const float*)nullptr, dW, M, M, parts, B * H));
                    if (rope) {
                        if constexpr (F32) {
                            RC(launch(sp::k_sp_bwd_dq<ET, DT, true, false, false, 1>, dim3(M, B * H), dim3(NTHREADS), sp::sp_tok_smem<DT, false>(), st, "k_sp_bwd_dq<rope>", t));
                            RC(launch(sp::k_sp_bwd_dkv<ET, DT, true, false, 1>, dim3(M, B * H), dim3(NTHREADS), sp::sp_tok_smem<DT, false>(), st, "k_sp_bwd_dkv<rope>", t));
                        }
                        break;
                    }
                    if (normalize && !relu && sizeof(ET) == 2 && DT != 5) WITH_PF(w.fmt, RC(launch(sp::k_sp_bwd_dq<ET, DT, false, false, true, PF>, dim3(M, B * H), dim3(NTHREADS), sp::sp_tok_smem<DT, false>(), st, "k_sp_bwd_dq", t)));
                    else WITH_PF(w.fmt, RC(launch(sp::k_sp_bwd_dq<ET, DT, false, false, false, PF>, dim3(M, B * H), dim3(NTHREADS), sp::sp_tok_smem<DT, false>(), st, "k_sp_bwd_dq", t)));
                    WITH_PF(w.fmt, RC(launch(sp::k_sp_bwd_dkv<ET, DT, false, false, PF>, dim3(M, B * H), dim3(NTHREADS), sp::sp_tok_smem<DT, false>(), st, "k_sp_bwd_dkv", t)));
                    break;
                }
            }
            if constexpr (std::is_same<ET, float>::value) {
                if (rcos) RC(launch(sp::k_sp_state<ET, DT, 1, true, SNT, S16>, dim3(M, B * H), dim3(SNT), sp::sp_state_smem<DT>(), st, "k_sp_state<1,rope>", a));
            }
            if (s16) RC(launch(s16::k_s16_state<1>, dim3((M + s16::WPB - 1) / s16::WPB, B * H), dim3(64 * s16::WPB), s16::state_smem(), st, "k_s16_state<1>", a));
            else if (!rcos) RC(launch(sp::k_sp_state<ET, DT, 1, false, SNT, S16>, dim3(M, B * H), dim3(SNT), sp::sp_state_smem<DT>(), st, "k_sp_state<1>", a));
            const bool mixr = sp_mixr_ok<S16>(M, E), wz_fused = mixr && normalize && sp_mixr_takes_wz<S16>(M, S);
            if (normalize && !wz_fused)
                RC(launch(k_wz<1>, dim3((S + 63) / 64, (M + 63) / 64, B * H), dim3(NTHREADS), 0, st, "k_wz<1>", W, ldw, (const float*)w.dn, w.dz, M, S, 0.f));
            bool dwz_done = false;   // the <dn_i, z_j> term of dW is in the fused kernel's partials
            MixArgs m{W, ldw, w.dg, w.dkv, M, E, w.es};
            // fp32 summaries, 33 .. 128 blocks: dW's products ride in the mixing kernel (its staged dG slices + the KV slices), one
            // partial per workgroup -- dG and KV are not read a second time by k_sp_dw
            const bool dwfused = !S16 && mixr && sp_mixr_dw_ok(M, E);
            int fused_parts = 0;
            if (dwfused) RC(sp_mixr_dw(W, ldw, w.dg, w.kv, w.dkv, w.dwp, M, E, w.es, B * H, st, &fused_parts, normalize ? (const float*)w.dn : nullptr,
                                       normalize ? (const float*)w.z : nullptr, w.dz, S, &dwz_done));
            else if (mixr) RC((sp_mixr<1, S16>(W, ldw, w.dg, w.dkv, M, E, w.es, B * H, st, normalize ? (const float*)w.dn : nullptr, w.dz, S, 0.f)));
            else RC(launch(sp::k_sp_mix<1, S16>, dim3((unsigned)((E + sp::SPM_TE - 1) / sp::SPM_TE), tiles, B * H), dim3(NTHREADS), sp::sp_mix_smem<S16>(), st, "k_sp_mix<1>", m));
            int nsplit = dw_splits(tiles * tiles * B * H, E);
            if (nsplit > DW_MAX_SPLIT - 1) nsplit = DW_MAX_SPLIT - 1;   // one more part per (b, h) holds the <dn_i, z_j> term
            DwArgs d{w.dg, w.kv, E, nullptr, nullptr, 0, w.dwp, M, tiles, nsplit, w.es};
            const bool dwr = S16 && sp_dwr_ok(M, E);   // whole-matrix workgroups: the <dn_i, z_j> term is one of their stages
            if (dwfused) {
                // (nothing to launch: the partials are in w.dwp)
            } else if (dwr) {
                nsplit = sp_dwr_splits(B * H, E);
                RC(sp_dwr(w.dg, w.kv, E, w.es, normalize ? w.dn : nullptr, normalize ? w.z : nullptr, S, w.dwp, M, B * H, nsplit, st));
            } else if (M <= 16)      RC(launch(sp::k_sp_dw<S16, 1>, dim3(1, B * H, nsplit), dim3(NTHREADS), sp::SP_DW_SMEM, st, "k_sp_dw<16>", d));
            else if (M <= 32) RC(launch(sp::k_sp_dw<S16, 2>, dim3(1, B * H, nsplit), dim3(NTHREADS), sp::SP_DW_SMEM, st, "k_sp_dw<32>", d));
            else              RC(launch(sp::k_sp_dw<S16>, dim3(tiles * tiles, B * H, nsplit), dim3(NTHREADS), sp::SP_DW_SMEM, st, "k_sp_dw", d));
            int nparts = dwfused ? fused_parts : B * H * nsplit;
            if (normalize && !dwr && !dwz_done) {
                DwArgs dzz{w.dn, w.z, (long)S, nullptr, nullptr, 0, w.dwp + (size_t)nparts * M * M, M, tiles, 1};
                RC(launch(k_dw<0>, dim3(tiles * tiles, B * H, 1), dim3(NTHREADS), DW_SMEM_FLOATS * 4, st, "k_dw", dzz));
                nparts += B * H;
            }
            // (16 elements x 16 part-lanes per workgroup when the matrix is small or the parts are many: a thread's chain of dependent
            // load batches is what the kernel takes -- 512 parts of 64 x 64 at 64 x 4: 10 us)
            if (M * M <= 1024 || nparts >= 128) RC(launch(k_dw_reduce<0, 16>, dim3((M * M + 15) / 16), dim3(256), 0, st, "k_dw_reduce", (const float*)w.dwp,
                      (const float*)nullptr, dW, M, M, nparts, B * H));
            else RC(launch(k_dw_reduce<0>, dim3((M * M + 63) / 64), dim3(256), 0, st, "k_dw_reduce", (const float*)w.dwp,
                      (const float*)nullptr, dW, M, M, nparts, B * H));
            if constexpr (std::is_same<ET, float>::value) {
                if (rcos) {
                    RC(launch(sp::k_sp_bwd_dq<ET, DT, true>, dim3(M, B * H), dim3(NTHREADS), sp::sp_tok_smem<DT, false>(), st, "k_sp_bwd_dq<rope>", t));
                    RC(launch(sp::k_sp_bwd_dkv<ET, DT, true>, dim3(M, B * H), dim3(NTHREADS), sp::sp_tok_smem<DT, false>(), st, "k_sp_bwd_dkv<rope>", t));
                    break;
                }
            }
            if (s16) {   // (the dK kernel forms dksum itself: no order between the two)
                const dim3 g16((M + s16::WPB - 1) / s16::WPB, B * H), b16(64 * s16::WPB);
                RC(launch(s16::k_s16_bwd_dq<0>, g16, b16, 0, st, "k_s16_bwd_dq", t));
                RC(launch(s16::k_s16_bwd_dkv<0>, g16, b16, s16::dkv_smem(), st, "k_s16_bwd_dkv", t));
                break;
            }
            // (16-bit tensors: q_den in 16-byte pieces when it only feeds dksum; not at D = 72 / 80, where the wider rows cost a wave of occupancy)
            if (normalize && !relu && sizeof(ET) == 2 && DT != 5) RC(launch(sp::k_sp_bwd_dq<ET, DT, false, S16, true>, dim3(M, B * H), dim3(NTHREADS), sp::sp_tok_smem<DT, S16>(), st, "k_sp_bwd_dq", t));
            else RC(launch(sp::k_sp_bwd_dq<ET, DT, false, S16>, dim3(M, B * H), dim3(NTHREADS), sp::sp_tok_smem<DT, S16>(), st, "k_sp_bwd_dq", t));
            RC(launch(sp::k_sp_bwd_dkv<ET, DT, false, S16>, dim3(M, B * H), dim3(NTHREADS), sp::sp_tok_smem<DT, S16>(), st, "k_sp_bwd_dkv", t));
            break;
        }
        RC(launch(k_bm_state<ET, DT, 1>, dim3(M, B * H), dim3(NTHREADS), state_smem_floats<DT>() * 4, st, "k_bm_state<1>", a));
        if (normalize)
            RC(launch(k_wz<1>, dim3((S + 63) / 64, (M + 63) / 64, B * H), dim3(NTHREADS), 0, st, "k_wz<1>", W, ldw, (const float*)w.dn, w.dz, M, S, 0.f));
        // dKV = W^T dG
        MixArgs m{W, ldw, w.dg, w.dkv, M, (long)D * D};
        dim3 mgrid((unsigned)((m.E + MIX_TE - 1) / MIX_TE), (M + MIX_TI - 1) / MIX_TI, B * H);
        RC(launch(k_mix<1, 0>, mgrid, dim3(NTHREADS), MIX_SMEM_FLOATS * 4, st, "k_mix<1,0>", m));
        // dW = sum_bh (<dG_i, KV_j> + <dn_i, z_j>)
        const int nsplit = dw_splits(tiles * tiles * B * H, (long)D * D);
        DwArgs d{w.dg, w.kv, (long)D * D, normalize ? w.dn : nullptr, normalize ? w.z : nullptr, (long)S, w.dwp, M, tiles, nsplit};
        RC(launch(k_dw<0>, dim3(tiles * tiles, B * H, nsplit), dim3(NTHREADS), DW_SMEM_FLOATS * 4, st, "k_dw", d));
        if (M * M <= 1024 || B * H * nsplit >= 128) RC(launch(k_dw_reduce<0, 16>, dim3((M * M + 15) / 16), dim3(256), 0, st, "k_dw_reduce", (const float*)w.dwp,
                  (const float*)nullptr, dW, M, M, B * H * nsplit, B * H));
        else RC(launch(k_dw_reduce<0>, dim3((M * M + 63) / 64), dim3(256), 0, st, "k_dw_reduce", (const float*)w.dwp,
                  (const float*)nullptr, dW, M, M, B * H * nsplit, B * H));
        // dQ, dK, dV
        RC(launch(k_bm_bwd_tok<ET, DT>, dim3(M, B * H), dim3(NTHREADS), tok_smem_floats<DT>() * 4, st, "k_bm_bwd_tok", t));
    });
    return MHLA_OK;
}

}  // namespace capi
}  // namespace mhla
