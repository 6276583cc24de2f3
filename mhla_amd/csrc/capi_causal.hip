// C ABI, causal chunk-mixing operator (mhla_causal_*): generic fp32-MFMA kernels (causal.hpp) and the bf16 pipeline
// (causal_bf16.hpp, causal_mix.hpp).
#include "capi_common.hpp"
#include "blockmix.hpp"
#include "causal.hpp"
#include "causal_bf16.hpp"
#include "causal_mix.hpp"

using namespace mhla;
using namespace mhla::capi;

namespace {

struct CsWs {
    float *S, *P, *dP, *dS, *dwp, *diag;
    size_t total_fwd, total_bwd;
};
// esz: bytes per summary element (2 on the bf16 pipeline, 4 on the generic one)
// Launch plan of the resident-sequence mixing kernels (causal_mix.hpp): waves per workgroup (16 chunks each), workgroups and
// slices per workgroup.  One workgroup per CU at 8 waves (132 KB of LDS), two / four at 4 / 2 waves.
struct Mix2Plan { int nw, te, wgs, spw; long total; };
Mix2Plan mix2_plan(size_t bh, int n, long E, bool bwd) {
    static const char* const knob = getenv("MHLA_CAUSAL_MIX_TE");   // tuning knob, read once: forward slice width 128 (default) or 256
    Mix2Plan p{};
    p.nw = n <= 32 ? 2 : n <= 64 ? 4 : 8;
    p.te = (!bwd && knob && knob[0] == '2' && E % 256 == 0) ? 256 : 128;
    p.total = (long)bh * (E / p.te);
    // workgroups the chip holds at once: forward 2 tiles of 16 nw rows and 64 nw threads, backward 3 tiles and 128 nw threads
    const long slots = 256L * (8 / p.nw) * (bwd ? 1 : 256 / p.te);
    if (p.total <= 0) return p;   // (summaries smaller than a slice: the dispatcher does not take this path)
    const long wgs = std::min(p.total, slots);
    p.spw = (int)((p.total + wgs - 1) / wgs);
    p.wgs = (int)((p.total + p.spw - 1) / p.spw);
    return p;
}
bool cs_mix2_ok(int n, long E) {
    static const char* const knob = getenv("MHLA_CAUSAL_MIX");   // tuning knob, read once: "old" keeps k_csf_mix / k_csf_dw
    return n <= 128 && E % fast::MF_TE == 0 && !(knob && knob[0] == 'o');
}
CsWs cs_carve(void* ws, int B, int T, int H, int K, int V, int chunk, int esz) {
    const size_t bh = (size_t)B * H, n = (size_t)(T + chunk - 1) / chunk, st = al4(bh * n * K * V) * esz / 4;
    const size_t parts = std::max(bh * DW_MAX_SPLIT, (size_t)mix2_plan(bh, (int)n, (long)K * V, true).wgs);
    float* p = (float*)ws;
    CsWs w;
    w.S = p; p += st;
    w.P = p; p += st;
    w.total_fwd = (size_t)(p - (float*)ws) * 4;
    w.dP = p; p += st;
    w.dS = p; p += st;
    w.dwp = p; p += al4(parts * n * n);
    w.diag = p; p += al4(bh * n);
    w.total_bwd = (size_t)(p - (float*)ws) * 4;
    return w;
}
int cs_check(int B, int T, int H, int K, int V, int chunk, int dtype) {
    if (B <= 0 || T <= 0 || H <= 0 || K <= 0 || V <= 0) return fail(MHLA_EINVAL, "non-positive dimension B=%d T=%d H=%d K=%d V=%d", B, T, H, K, V);
    if (chunk != 64) return fail(MHLA_ENOTSUP, "chunk=%d: only 64 is supported", chunk);
    if ((K | V) & 3) return fail(MHLA_EINVAL, "K=%d and V=%d must be multiples of 4", K, V);
    if (dtype < 0 || dtype > 2) return fail(MHLA_EINVAL, "unknown dtype %d", dtype);
    if ((size_t)B * H > 65535) return fail(MHLA_ENOTSUP, "B*H=%zu exceeds grid limit 65535", (size_t)B * H);
    return MHLA_OK;
}

// S_j (or dP_i) = alpha X_j^T Y_j with 64x64 strips
template <typename T>
int cs_xty(const mhla_view& x, const mhla_view& y, float* out, float alpha, int B, int T_, int H, int n, int DX,
                  int DY, hipStream_t st) {
    StateArgs a{};
    a.x = cv(x); a.y = cv(y); a.out = out; a.H = H; a.M = n; a.S = CS; a.D = 64; a.DX = DX; a.DY = DY; a.T = T_;
    a.alpha = alpha;
    const int strips = ((DX + 63) / 64) * ((DY + 63) / 64);
    return launch(k_bm_state<T, 4, 2>, dim3(n, B * H, strips), dim3(NTHREADS), state_smem_floats<4>() * 4, st, "k_bm_state<2>", a);
}

}  // namespace

extern "C" {

// ---------------------------------------------------------------------------------------------
// causal
// ---------------------------------------------------------------------------------------------
size_t mhla_causal_fwd_ws_bytes(int B, int T, int H, int K, int V, int chunk, int dtype) {
    return cs_carve(nullptr, B, T, H, K, V, chunk, cs_bf16_ok(K, V, dtype) ? 2 : 4).total_fwd;
}
size_t mhla_causal_bwd_ws_bytes(int B, int T, int H, int K, int V, int chunk, int dtype) {
    return cs_carve(nullptr, B, T, H, K, V, chunk, cs_bf16_ok(K, V, dtype) ? 2 : 4).total_bwd;
}

// chunk summaries X^T Y of the bf16 pipeline (S = K^T V, dP = scale Q^T dO)
static int cs_state16(const mhla_view& x, const mhla_view& y, uint16_t* out, float mul, int B, int T, int H, int n, int K, int V, hipStream_t st) {
    static const char* const knob = getenv("MHLA_CAUSAL_STATE");   // tuning knob, read once: "old" keeps the per-K-slice kernel
    fast::CsfStateArgs s{cv(x), cv(y), out, H, n, K, V, (long)T, mul};
    if (knob && knob[0] == 'o')
        return launch(fast::k_csf_state, dim3(n, B * H, K / 64), dim3(NTHREADS), fast::CSF_STATE_SMEM, st, "k_csf_state", s);
    const int blocks = ((K + fast::ST2_KW - 1) / fast::ST2_KW) * ((V + fast::ST2_VW - 1) / fast::ST2_VW);
    return launch(fast::k_csf_state2, dim3((n + fast::ST2_CPW - 1) / fast::ST2_CPW, B * H, blocks), dim3(NTHREADS), fast::CSF_STATE2_SMEM, st, "k_csf_state", s);
}

// P = strictly-lower mix of S (bf16 pipeline)
static int cs_mix_fwd(const float* mix, int ldmix, const uint16_t* S, uint16_t* P, int BH, int n, long E, hipStream_t st) {
    if (cs_mix2_ok(n, E)) {
        const Mix2Plan pl = mix2_plan((size_t)BH, n, E, false);
        fast::CsfMix2Args mf{mix, ldmix, S, nullptr, P, nullptr, n, E, pl.total, pl.spw};
#define MIXF(NW, TE) launch(fast::k_csf_mixf<NW, TE>, dim3(pl.wgs), dim3(64 * NW), fast::mixf_smem<NW, TE>(), st, "k_csf_mixf", mf)
        if (pl.te == 256) return pl.nw == 2 ? MIXF(2, 256) : pl.nw == 4 ? MIXF(4, 256) : MIXF(8, 256);
        return pl.nw == 2 ? MIXF(2, 128) : pl.nw == 4 ? MIXF(4, 128) : MIXF(8, 128);
#undef MIXF
    }
    fast::CsfMixArgs m{mix, ldmix, S, P, n, E};
    return launch(fast::k_csf_mix<0>, dim3((unsigned)(E / fast::MX_TE), (n + 63) / 64, BH), dim3(NTHREADS), fast::CSF_MIX_SMEM, st, "k_csf_mix<0>", m);
}

static int cs_fwd_impl(mhla_view q, mhla_view k, mhla_view v, const float* mix, int ldmix, mhla_mview out, void* ws,
                       size_t ws_bytes, int B, int T, int H, int K, int V, int chunk, float scale, int dtype, void* stream,
                       bool epi, const float* nw, float neps, mhla_view gate, mhla_mview y) {
    RC(cs_check(B, T, H, K, V, chunk, dtype));
    CHECK_VIEW(q); CHECK_VIEW(k); CHECK_VIEW(v);
    if (!epi || out.ptr) CHECK_VIEW(out);
    if (epi) {
        if (!cs_bf16_ok(K, V, dtype) || V > 64 * fast::CSF_OUT_VS)
            return fail(MHLA_ENOTSUP, "fused norm x gate epilogue needs bf16 tensors, K %% 64 == 0 and V %% 64 == 0, V <= %d (K=%d V=%d dtype=%d)",
                        64 * fast::CSF_OUT_VS, K, V, dtype);
        const mhla_view yv{y.ptr, y.sb, y.sn, y.sh};
        if (!view_ok16(yv) || (gate.ptr && !view_ok16(gate)) || (out.ptr && !view_ok16m(out)))
            return fail(MHLA_EINVAL, "fused norm x gate epilogue: y, gate and out must be 16-byte aligned views (strides multiples of 8)");
    }
    const int n = (T + chunk - 1) / chunk;
    if (!mix || ldmix < n) return fail(MHLA_EINVAL, "mix null or ldmix=%d < n=%d chunks (T=%d)", ldmix, n, T);
    const bool pipe16 = cs_bf16_ok(K, V, dtype);
    if (pipe16 && !(view_ok16(q) && view_ok16(k) && view_ok16(v) && (epi || view_ok16m(out))))
        return fail(MHLA_EINVAL, "bf16 tensors with K, V multiples of 64 must be 16-byte aligned views (strides multiples of 8)");
    const CsWs w = cs_carve(ws, B, T, H, K, V, chunk, pipe16 ? 2 : 4);
    if (!ws || ws_bytes < w.total_fwd) return fail(MHLA_EINVAL, "workspace too small: %zu < %zu bytes", ws_bytes, w.total_fwd);
    if (((uintptr_t)ws) % 16) return fail(MHLA_EINVAL, "workspace not 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const long E = (long)K * V;
    if (pipe16) {
        // bf16 pipeline (causal_bf16.hpp): bf16 chunk summaries, bf16 MFMA everywhere
        RC(cs_state16(k, v, (uint16_t*)w.S, 1.f, B, T, H, n, K, V, st));
        RC(cs_mix_fwd(mix, ldmix, (const uint16_t*)w.S, (uint16_t*)w.P, B * H, n, E, st));
        CsOutArgs o{cv(q), cv(k), cv(v), cmv(out), mix, ldmix, w.P, H, n, K, V, (long)T, scale, cmv(y), cv(gate), nw, neps};
        static const char* const outv = getenv("MHLA_CAUSAL_OUT");   // tuning knob, read once: "old" keeps the four-wave kernel
        if (outv && outv[0] == 'o') {
            if (epi) RC(launch(fast::k_csf_out<uint16_t, true>, dim3(n, B * H, 1), dim3(NTHREADS), fast::CSF_OUT_SMEM, st, "k_csf_out<norm>", o));
            else     RC(launch(fast::k_csf_out<uint16_t>, dim3(n, B * H, (V / 64 + fast::CSF_OUT_VS - 1) / fast::CSF_OUT_VS), dim3(NTHREADS), fast::CSF_OUT_SMEM, st, "k_csf_out", o));
            return MHLA_OK;
        }
        // V slices per workgroup: the largest of 4, 3, 2, 1 that divides V / 64 (the fused epilogue owns the head: V / 64 <= 4)
        const int nvs = V / 64, nv = nvs % 4 == 0 ? 4 : nvs % 3 == 0 ? 3 : nvs % 2 == 0 ? 2 : 1;
#define OUT4(NV, EPI) launch(fast::k_csf_out4<NV, EPI>, dim3(EPI ? n : (n + fast::CSF_OUT4_CPW - 1) / fast::CSF_OUT4_CPW, B * H, nvs / NV), dim3(fast::NT4), fast::csf_out4_smem<NV, EPI>(), st, EPI ? "k_csf_out4<norm>" : "k_csf_out4", o)
        if (epi) RC(nvs == 1 ? OUT4(1, true) : nvs == 2 ? OUT4(2, true) : nvs == 3 ? OUT4(3, true) : OUT4(4, true));
        else     RC(nv == 1 ? OUT4(1, false) : nv == 2 ? OUT4(2, false) : nv == 3 ? OUT4(3, false) : OUT4(4, false));
#undef OUT4
        return MHLA_OK;
    }
    DISPATCH_T(dtype, {
        RC(cs_xty<ET>(k, v, w.S, 1.f, B, T, H, n, K, V, st));
        MixArgs m{mix, ldmix, w.S, w.P, n, E};
        dim3 mgrid((unsigned)((m.E + MIX_TE - 1) / MIX_TE), (n + MIX_TI - 1) / MIX_TI, B * H);
        RC(launch(k_mix<0, 1>, mgrid, dim3(NTHREADS), MIX_SMEM_FLOATS * 4, st, "k_mix<0,1>", m));
        CsOutArgs o{cv(q), cv(k), cv(v), cmv(out), mix, ldmix, w.P, H, n, K, V, (long)T, scale, cmv(out), cv(mhla_view{nullptr, 0, 0, 0}), nullptr, 0.f};
        RC(launch(k_cs_out<ET>, dim3(n, B * H, (V + 63) / 64), dim3(NTHREADS), CS_OUT_SMEM_FLOATS * 4, st, "k_cs_out", o));
    });
    return MHLA_OK;
}

int mhla_causal_fwd(mhla_view q, mhla_view k, mhla_view v, const float* mix, int ldmix, mhla_mview out, void* ws,
                    size_t ws_bytes, int B, int T, int H, int K, int V, int chunk, float scale, int dtype, void* stream) {
    return cs_fwd_impl(q, k, v, mix, ldmix, out, ws, ws_bytes, B, T, H, K, V, chunk, scale, dtype, stream, false, nullptr, 0.f,
                       mhla_view{nullptr, 0, 0, 0}, mhla_mview{nullptr, 0, 0, 0});
}

int mhla_causal_normgate_fwd(mhla_view q, mhla_view k, mhla_view v, const float* mix, int ldmix, mhla_mview out, mhla_view gate,
                             const float* norm_w, float norm_eps, mhla_mview y, void* ws, size_t ws_bytes, int B, int T, int H,
                             int K, int V, int chunk, float scale, int dtype, void* stream) {
    if (!y.ptr) return fail(MHLA_EINVAL, "y null");
    return cs_fwd_impl(q, k, v, mix, ldmix, out, ws, ws_bytes, B, T, H, K, V, chunk, scale, dtype, stream, true, norm_w, norm_eps, gate, y);
}

int mhla_causal_bwd(mhla_view q, mhla_view k, mhla_view v, const float* mix, int ldmix, mhla_view dout, mhla_mview dq,
                    mhla_mview dk, mhla_mview dv, float* dmix, int lddmix, void* ws, size_t ws_bytes, const void* fwd_ws,
                    int B, int T, int H, int K, int V, int chunk, float scale, int dtype, void* stream) {
    RC(cs_check(B, T, H, K, V, chunk, dtype));
    CHECK_VIEW(q); CHECK_VIEW(k); CHECK_VIEW(v); CHECK_VIEW(dout); CHECK_VIEW(dq); CHECK_VIEW(dk); CHECK_VIEW(dv);
    const int n = (T + chunk - 1) / chunk;
    if (!mix || ldmix < n || !dmix || lddmix < n) return fail(MHLA_EINVAL, "mix/dmix null or leading dim < n=%d chunks", n);
    const bool pipe16 = cs_bf16_ok(K, V, dtype);
    if (pipe16 && !(view_ok16(q) && view_ok16(k) && view_ok16(v) && view_ok16(dout) && view_ok16m(dq) && view_ok16m(dk) && view_ok16m(dv)))
        return fail(MHLA_EINVAL, "bf16 tensors with K, V multiples of 64 must be 16-byte aligned views (strides multiples of 8)");
    CsWs w = cs_carve(ws, B, T, H, K, V, chunk, pipe16 ? 2 : 4);
    if (!ws || ws_bytes < w.total_bwd) return fail(MHLA_EINVAL, "workspace too small: %zu < %zu bytes", ws_bytes, w.total_bwd);
    if (((uintptr_t)ws) % 16 || ((uintptr_t)fwd_ws) % 16) return fail(MHLA_EINVAL, "workspace not 16-byte aligned");
    if (fwd_ws) {   // chunk summaries S, P left by mhla_causal_fwd with the same arguments: skip their recomputation
        const CsWs f = cs_carve(const_cast<void*>(fwd_ws), B, T, H, K, V, chunk, pipe16 ? 2 : 4);
        w.S = f.S;
        w.P = f.P;
    }
    hipStream_t st = (hipStream_t)stream;
    const long E = (long)K * V;
    const int tiles = (n + 63) / 64;
    const int nsplit = dw_splits(tiles * tiles * B * H, E);
    if (pipe16) {
        uint16_t *S = (uint16_t*)w.S, *P = (uint16_t*)w.P, *dP = (uint16_t*)w.dP, *dS = (uint16_t*)w.dS;
        const dim3 mgrid((unsigned)(E / fast::MX_TE), tiles, B * H);
        if (!fwd_ws) {
            RC(cs_state16(k, v, S, 1.f, B, T, H, n, K, V, st));
            RC(cs_mix_fwd(mix, ldmix, S, P, B * H, n, E, st));
        }
        RC(cs_state16(q, dout, dP, scale, B, T, H, n, K, V, st));
        const bool mix2 = cs_mix2_ok(n, E);
        int nparts = B * H * nsplit;
        if (mix2) {   // dS and the dmix partials from one pass over dP and S
            const Mix2Plan pl = mix2_plan((size_t)B * H, n, E, true);
            fast::CsfMix2Args mb{mix, ldmix, dP, S, dS, w.dwp, n, E, pl.total, pl.spw};
#define MIXB(NW) launch(fast::k_csf_mixb<NW>, dim3(pl.wgs), dim3(128 * NW), fast::mixb_smem<NW>(), st, "k_csf_mixb", mb)
            RC(pl.nw == 2 ? MIXB(2) : pl.nw == 4 ? MIXB(4) : MIXB(8));
#undef MIXB
            nparts = pl.wgs;
        } else {
            fast::CsfMixArgs mt{mix, ldmix, dP, dS, n, E};
            RC(launch(fast::k_csf_mix<1>, mgrid, dim3(NTHREADS), fast::CSF_MIX_SMEM, st, "k_csf_mix<1>", mt));
        }
        CsTokArgs t{cv(q), cv(k), cv(v), cv(dout), cmv(dq), cmv(dk), cmv(dv), mix, ldmix, w.P, w.dS, w.diag, H, n, K, V, (long)T, scale};
        static const char* const tokv = getenv("MHLA_CAUSAL_TOK");   // tuning knob, read once: "2" forces the K-slice-outer kernels
        const bool tok3 = !(tokv && tokv[0] == '2');
        const bool tok4 = !(tokv && tokv[0] == '3');                 // "3": the four-wave kernels
#define TOK4(NK) launch(fast::k_csf_bwd_tok4<uint16_t, NK>, dim3(n, B * H), dim3(fast::NT4), fast::csf_tok4_smem<NK>(), st, "k_csf_bwd_tok4", t)
        if (tok3 && tok4 && K <= 256) RC(K == 64 ? TOK4(1) : K == 128 ? TOK4(2) : K == 192 ? TOK4(3) : TOK4(4));
#undef TOK4
        else if (tok3 && K <= 128)      RC(launch(fast::k_csf_bwd_tok3<uint16_t, 2>, dim3(n, B * H), dim3(NTHREADS), fast::csf_tok3_smem<2>(), st, "k_csf_bwd_tok3", t));
        else if (tok3 && K <= 256) RC(launch(fast::k_csf_bwd_tok3<uint16_t, 4>, dim3(n, B * H), dim3(NTHREADS), fast::csf_tok3_smem<4>(), st, "k_csf_bwd_tok3", t));
        else if (V <= 128) RC(launch(fast::k_csf_bwd_tok2<uint16_t, 2>, dim3(n, B * H), dim3(NTHREADS), fast::CSF_TOK2_SMEM, st, "k_csf_bwd_tok", t));
        else if (V <= 256) RC(launch(fast::k_csf_bwd_tok2<uint16_t, 4>, dim3(n, B * H), dim3(NTHREADS), fast::CSF_TOK2_SMEM, st, "k_csf_bwd_tok", t));
        else if (V <= 512) RC(launch(fast::k_csf_bwd_tok2<uint16_t, 8>, dim3(n, B * H), dim3(NTHREADS), fast::CSF_TOK2_SMEM, st, "k_csf_bwd_tok", t));
        else               RC(launch(fast::k_csf_bwd_tok<uint16_t>, dim3(n, B * H), dim3(NTHREADS), fast::CSF_TOK_SMEM, st, "k_csf_bwd_tok", t));
        if (!mix2) {
            fast::CsfDwArgs d{dP, S, E, w.dwp, n, tiles, nsplit};
            if (n <= 16)      RC(launch(fast::k_csf_dw<1>, dim3(1, B * H, nsplit), dim3(NTHREADS), fast::CSF_DW_SMEM, st, "k_csf_dw<16>", d));
            else if (n <= 32) RC(launch(fast::k_csf_dw<2>, dim3(1, B * H, nsplit), dim3(NTHREADS), fast::CSF_DW_SMEM, st, "k_csf_dw<32>", d));
            else              RC(launch(fast::k_csf_dw<4>, dim3(tiles * tiles, B * H, nsplit), dim3(NTHREADS), fast::CSF_DW_SMEM, st, "k_csf_dw", d));
        }
        // few elements, many partials (short sequences): 16 part-lanes per element instead of 4
        if (n <= 64) RC(launch(k_dw_reduce<1, 16>, dim3((n * n + 15) / 16), dim3(256), 0, st, "k_dw_reduce<1>", (const float*)w.dwp,
                               (const float*)w.diag, dmix, lddmix, n, nparts, B * H));
        else         RC(launch(k_dw_reduce<1>, dim3((n * n + 63) / 64), dim3(256), 0, st, "k_dw_reduce<1>", (const float*)w.dwp,
                               (const float*)w.diag, dmix, lddmix, n, nparts, B * H));
        return MHLA_OK;
    }
    DISPATCH_T(dtype, {
        MixArgs m{mix, ldmix, w.S, w.P, n, E};
        dim3 mgrid((unsigned)((m.E + MIX_TE - 1) / MIX_TE), (n + MIX_TI - 1) / MIX_TI, B * H);
        if (!fwd_ws) {
            RC(cs_xty<ET>(k, v, w.S, 1.f, B, T, H, n, K, V, st));
            RC(launch(k_mix<0, 1>, mgrid, dim3(NTHREADS), MIX_SMEM_FLOATS * 4, st, "k_mix<0,1>", m));
        }
        RC(cs_xty<ET>(q, dout, w.dP, scale, B, T, H, n, K, V, st));
        MixArgs mt{mix, ldmix, w.dP, w.dS, n, E};
        RC(launch(k_mix<1, 1>, mgrid, dim3(NTHREADS), MIX_SMEM_FLOATS * 4, st, "k_mix<1,1>", mt));
        CsTokArgs t{cv(q), cv(k), cv(v), cv(dout), cmv(dq), cmv(dk), cmv(dv), mix, ldmix, w.P, w.dS, w.diag, H, n, K, V, (long)T, scale};
        RC(launch(k_cs_bwd_tok<ET>, dim3(n, B * H), dim3(NTHREADS), CS_TOK_SMEM_FLOATS * 4, st, "k_cs_bwd_tok", t));
        DwArgs d{w.dP, w.S, E, nullptr, nullptr, 0, w.dwp, n, tiles, nsplit};
        RC(launch(k_dw<1>, dim3(tiles * tiles, B * H, nsplit), dim3(NTHREADS), DW_SMEM_FLOATS * 4, st, "k_dw<1>", d));
        RC(launch(k_dw_reduce<1>, dim3((n * n + 63) / 64), dim3(256), 0, st, "k_dw_reduce<1>", (const float*)w.dwp,
                  (const float*)w.diag, dmix, lddmix, n, B * H * nsplit, B * H));
    });
    return MHLA_OK;
}

}  // extern "C"
