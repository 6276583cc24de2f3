// C ABI, causal chunk-mixing operator (mhla_causal_*): generic fp32-MFMA kernels (causal.hpp) and the 16-bit-MFMA pipeline
// (causal_bf16.hpp, causal_mix.hpp).  The kernel family is a function of the problem and of the caller's flags only -- no
// environment variables.
#include "capi_common.hpp"
#include "blockmix.hpp"
#include "causal.hpp"
#include "causal_bf16.hpp"
#include "causal_mix.hpp"

#include <string>

using namespace mhla;
using namespace mhla::capi;

namespace {

struct CsWs {
    float *S, *P, *dP, *dS, *dwp, *diag;
    size_t total_fwd, total_bwd;
};
// Which kernel family serves a causal problem:
//   16-bit pipeline (causal_bf16.hpp, causal_mix.hpp): bf16 tensors, K and V multiples of 64, K <= 256, at most 256 chunks
//   (the mixing kernels keep every chunk of a sequence resident: 16 waves of 16 chunks); hl: summaries and score tiles as bf16 hi + lo pairs (the
//   reference's fp32 arithmetic, default), !hl: single bf16 (MHLA_CAUSAL_BF16_SUMMARIES);
//   everything else: the generic kernels (causal.hpp: exact fp32 MFMA, fp32 summaries).
struct CsPath { bool pipe16; int hl, esz; };   // hl: the summaries' format on the 16-bit pipeline (causal_bf16.hpp: 0 single bf16, 1 bf16 hi + lo, 2 h16); esz: bytes per logical summary element
CsPath cs_path(int T, int K, int V, int chunk, int dtype, unsigned flags) {
    CsPath p{};
    p.esz = 4;
    if (chunk <= 0) return p;   // (the size / capability queries reach this before cs_check: the generic path, no division by zero)
    const int n = (T + chunk - 1) / chunk;
    p.pipe16 = dtype == MHLA_BF16 && (K & 63) == 0 && (V & 63) == 0 && K <= 256 && n <= 256 && !(flags & MHLA_CAUSAL_FORCE_GENERIC);
    // default: h16 (fp16 payload x a multiplier per 16-row strip of a chunk tile: 11 significand bits, 2 bytes); opt-in: hi + lo pairs
    // (>= 16 bits, 4 bytes: MHLA_CAUSAL_FP32_GRADE_SUMMARIES) or single bf16 (reduced precision: MHLA_CAUSAL_BF16_SUMMARIES)
    p.hl = !p.pipe16 ? 0 : (flags & MHLA_CAUSAL_BF16_SUMMARIES) ? 0 : (flags & MHLA_CAUSAL_FP32_GRADE_SUMMARIES) ? 1 : 2;
    p.esz = p.pipe16 && p.hl != 1 ? 2 : 4;
    return p;
}
// Launch plan of the resident-sequence mixing kernels (causal_mix.hpp): waves per workgroup (16 chunks each), workgroups and
// slices per workgroup.  Forward: two workgroups per CU at 8 waves (70-74 KB of LDS each), four / eight at 4 / 2 waves;
// backward: one workgroup of 16 waves per CU (104-111 KB), two / four at 8 / 4.
struct Mix2Plan { int nw, te, wgs, spw; long total; };
Mix2Plan mix2_plan(size_t bh, int n, long E, bool bwd, bool hl) {   // hl: two planes per slice row in LDS (format 1)
    Mix2Plan p{};
    p.nw = n <= 32 ? 2 : n <= 64 ? 4 : n <= 128 ? 8 : 16;
    p.te = (hl ? 64 : 128) / (p.nw > 8 ? 2 : 1);
    p.total = (long)bh * (E / p.te);
    const long slots = p.nw > 8 ? 256L : 256L * (8 / p.nw) * (bwd ? 1 : 2);   // (16 waves: one workgroup per CU either way)
    if (p.total <= 0) return p;
    const long wgs = std::min(p.total, slots);
    p.spw = (int)((p.total + wgs - 1) / wgs);
    p.wgs = (int)((p.total + p.spw - 1) / p.spw);
    return p;
}
// head widths the fused norm x gate epilogue covers: one workgroup owns a head's V channels -- up to four 64-wide slices at once,
// or two halves of three / four slices (V = 384, 512) with the first half's outputs parked in LDS
bool cs_epi_ok(const CsPath& path, int V) { return path.pipe16 && (V <= 256 || V == 384 || V == 512); }
CsWs cs_carve(void* ws, int B, int T, int H, int K, int V, int chunk, const CsPath& path) {
    // (16-bit pipeline: bf16 planes in the tile-major layout of fast::cs_layout, CS_CHUNK_PAD = 2 176 bytes of padding per chunk tile)
    if (chunk <= 0) { CsWs z{}; return z; }
    const size_t bh = (size_t)B * H, n = (size_t)(T + chunk - 1) / chunk;
    const size_t st = path.pipe16 ? al4((bh * (size_t)fast::cs_layout((int)n, (long)K * V, path.esz / 2).bhs + 1) / 2) : al4(bh * n * K * V);
    const size_t parts = std::max(bh * DW_MAX_SPLIT, path.pipe16 ? (size_t)mix2_plan(bh, (int)n, (long)K * V, true, path.hl == 1).wgs : (size_t)0);
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
int cs_check(int B, int T, int H, int K, int V, int chunk, int dtype, unsigned flags) {
    if (B <= 0 || T <= 0 || H <= 0 || K <= 0 || V <= 0) return fail(MHLA_EINVAL, "non-positive dimension B=%d T=%d H=%d K=%d V=%d", B, T, H, K, V);
    if (chunk != 64) return fail(MHLA_ENOTSUP, "chunk=%d: only 64 is supported", chunk);
    if ((K | V) & 3) return fail(MHLA_EINVAL, "K=%d and V=%d must be multiples of 4", K, V);
    if (dtype < 0 || dtype > 2) return fail(MHLA_EINVAL, "unknown dtype %d", dtype);
    if (flags & ~(MHLA_CAUSAL_FORCE_GENERIC | MHLA_CAUSAL_BF16_SUMMARIES | MHLA_CAUSAL_FP32_GRADE_SUMMARIES)) return fail(MHLA_EINVAL, "unknown causal flags 0x%x", flags);
    if ((flags & MHLA_CAUSAL_BF16_SUMMARIES) && (flags & MHLA_CAUSAL_FP32_GRADE_SUMMARIES)) return fail(MHLA_EINVAL, "MHLA_CAUSAL_BF16_SUMMARIES and MHLA_CAUSAL_FP32_GRADE_SUMMARIES exclude each other");
    if ((size_t)B * H > 65535) return fail(MHLA_ENOTSUP, "B*H=%zu exceeds grid limit 65535", (size_t)B * H);
    return MHLA_OK;
}

// S_j (or dP_i) = alpha X_j^T Y_j with 64x64 strips (generic path)
template <typename T>
int cs_xty(const mhla_view& x, const mhla_view& y, float* out, float alpha, int B, int T_, int H, int n, int DX,
                  int DY, hipStream_t st) {
    StateArgs a{};
    a.x = cv(x); a.y = cv(y); a.out = out; a.H = H; a.M = n; a.S = CS; a.D = 64; a.DX = DX; a.DY = DY; a.T = T_;
    a.alpha = alpha;
    const int strips = ((DX + 63) / 64) * ((DY + 63) / 64);
    return launch(k_bm_state<T, 4, 2>, dim3(n, B * H, strips), dim3(NTHREADS), state_smem_floats<4>() * 4, st, "k_bm_state<2>", a);
}

// chunk summaries X^T Y of the 16-bit pipeline (S = K^T V, dP = scale Q^T dO)
template <int HL>
int cs_state16(const mhla_view& x, const mhla_view& y, uint16_t* out, float mul, int B, int T, int H, int n, int K, int V, hipStream_t st) {
    fast::CsfStateArgs s{cv(x), cv(y), out, H, n, K, V, (long)T, mul};
    const int blocks = ((K + fast::ST2_KW - 1) / fast::ST2_KW) * ((V + fast::ST2_VW - 1) / fast::ST2_VW);
    return launch(fast::k_csf_state2<HL>, dim3((n + fast::ST2_CPW - 1) / fast::ST2_CPW, B * H, blocks), dim3(ST2_T),
                  fast::csf_state2_smem<HL>(), st, "k_csf_state", s);
}

// P = strictly-lower mix of S
template <int HL>
int cs_mix_fwd(const float* mix, int ldmix, const uint16_t* S, uint16_t* P, int BH, int n, long E, hipStream_t st) {
    const Mix2Plan pl = mix2_plan((size_t)BH, n, E, false, HL == 1);
    fast::CsfMix2Args mf{mix, ldmix, S, nullptr, P, nullptr, n, E, pl.total, pl.spw};
#define MIXF(NW) launch(fast::k_csf_mixf<NW, HL>, dim3(pl.wgs), dim3(64 * NW), fast::mixf_smem<NW, HL>(), st, "k_csf_mixf", mf)
    return pl.nw == 2 ? MIXF(2) : pl.nw == 4 ? MIXF(4) : pl.nw == 8 ? MIXF(8) : MIXF(16);
#undef MIXF
}

template <int HL>
int cs_fwd16(const mhla_view& q, const mhla_view& k, const mhla_view& v, const float* mix, int ldmix, const mhla_mview& out,
             const CsWs& w, int B, int T, int H, int K, int V, int n, float scale, hipStream_t st, bool epi, const float* nw,
             float neps, const mhla_view& gate, const mhla_mview& y) {
    const long E = (long)K * V;
    RC(cs_state16<HL>(k, v, (uint16_t*)w.S, 1.f, B, T, H, n, K, V, st));
    RC(cs_mix_fwd<HL>(mix, ldmix, (const uint16_t*)w.S, (uint16_t*)w.P, B * H, n, E, st));
    CsOutArgs o{cv(q), cv(k), cv(v), cmv(out), mix, ldmix, w.P, H, n, K, V, (long)T, scale, cmv(y), cv(gate), nw, neps};
    // V slices per workgroup: the largest of 4, 3, 2, 1 that divides V / 64 (the fused epilogue owns the head: V / 64 <= 4); the
    // more slices, the fewer times a chunk's Q and K rows and its score tile are fetched / formed
    const int nvs = V / 64, nv = nvs % 4 == 0 ? 4 : nvs % 3 == 0 ? 3 : nvs % 2 == 0 ? 2 : 1;
#define OUT4(NV, EPI) launch(fast::k_csf_out4<NV, EPI, HL>, dim3(EPI ? n : (n + fast::CSF_OUT4_CPW - 1) / fast::CSF_OUT4_CPW, B * H, nvs / NV), dim3(fast::NT4), fast::csf_out4_smem<NV, EPI, HL>(), st, EPI ? "k_csf_out4<norm>" : "k_csf_out4", o)
#define OUT4H2(NV) launch(fast::k_csf_out4<NV, true, HL, 2>, dim3(n, B * H, 1), dim3(fast::NT4), fast::csf_out4_smem<NV, true, HL, 2>(), st, "k_csf_out4<norm,2>", o)
    if (epi && nvs > 4) RC(nvs == 6 ? OUT4H2(3) : OUT4H2(4));   // V = 384, 512: the head in two halves (cs_epi_ok)
    else if (epi) RC(nvs == 1 ? OUT4(1, true) : nvs == 2 ? OUT4(2, true) : nvs == 3 ? OUT4(3, true) : OUT4(4, true));
    else     RC(nv == 1 ? OUT4(1, false) : nv == 2 ? OUT4(2, false) : nv == 3 ? OUT4(3, false) : OUT4(4, false));
#undef OUT4
#undef OUT4H2
    return MHLA_OK;
}

inline int csf_tok4_walk(int most, int n, int bh) {
    int cpw = 1;
    while (cpw * 2 <= most && (long)((n + cpw * 2 - 1) / (cpw * 2)) * bh >= 256) cpw *= 2;
    return cpw;
}
template <int HL>
int cs_bwd16(const mhla_view& q, const mhla_view& k, const mhla_view& v, const float* mix, int ldmix, const mhla_view& dout,
             const mhla_mview& dq, const mhla_mview& dk, const mhla_mview& dv, float* dmix, int lddmix, const CsWs& w, bool have_fwd,
             int B, int T, int H, int K, int V, int n, float scale, hipStream_t st) {
    const long E = (long)K * V;
    uint16_t *S = (uint16_t*)w.S, *P = (uint16_t*)w.P, *dP = (uint16_t*)w.dP, *dS = (uint16_t*)w.dS;
    if (!have_fwd) {
        RC(cs_state16<HL>(k, v, S, 1.f, B, T, H, n, K, V, st));
        RC(cs_mix_fwd<HL>(mix, ldmix, S, P, B * H, n, E, st));
    }
    RC(cs_state16<HL>(q, dout, dP, scale, B, T, H, n, K, V, st));
    // dS and the dmix partials from one pass over dP and S
    const Mix2Plan pl = mix2_plan((size_t)B * H, n, E, true, HL == 1);
    fast::CsfMix2Args mb{mix, ldmix, dP, S, dS, w.dwp, n, E, pl.total, pl.spw};
#define MIXB(NW) launch(fast::k_csf_mixb<NW, HL>, dim3(pl.wgs), dim3(128 * NW), fast::mixb_smem<NW, HL>(), st, "k_csf_mixb", mb)
    if (pl.nw <= 8) {
        RC(pl.nw == 2 ? MIXB(2) : pl.nw == 4 ? MIXB(4) : MIXB(8));
    } else {   // 129..256 chunks: the two roles as two launches of 16 waves
        RC(launch(fast::k_csf_mixb<16, HL, 1>, dim3(pl.wgs), dim3(1024), fast::mixb_smem<16, HL>(), st, "k_csf_mixb<dS>", mb));
        RC(launch(fast::k_csf_mixb<16, HL, 2>, dim3(pl.wgs), dim3(1024), fast::mixb_smem<16, HL>(), st, "k_csf_mixb<dmix>", mb));
    }
#undef MIXB
    CsTokArgs t{cv(q), cv(k), cv(v), cv(dout), cmv(dq), cmv(dk), cmv(dv), mix, ldmix, w.P, w.dS, w.diag, H, n, K, V, (long)T, scale};
    // chunks per workgroup: the largest power of two (<= the variant's limit) that still leaves every CU a workgroup
#define TOK4(NK) (t.cpw = csf_tok4_walk(fast::csf_tok4_cpw<NK, HL>(), n, B * H), \
                  launch(fast::k_csf_bwd_tok4<NK, HL>, dim3((n + t.cpw - 1) / t.cpw, B * H), dim3(fast::NT4), fast::csf_tok4_smem<NK, HL>(), st, "k_csf_bwd_tok4", t))
    RC(K == 64 ? TOK4(1) : K == 128 ? TOK4(2) : K == 192 ? TOK4(3) : TOK4(4));
#undef TOK4
    // up to 128 chunks: 16 part-lanes per element instead of 4 -- a thread's chain of dependent load batches is what the kernel takes
    // (128 chunks, 256 partials: 64 loads per thread 10.8 us, 16 loads 8.3 us)
#ifndef CS_RED16_MAX_N
#define CS_RED16_MAX_N 128
#endif
    if (n <= CS_RED16_MAX_N) RC(launch(k_dw_reduce<1, 16>, dim3((n * n + 15) / 16), dim3(256), 0, st, "k_dw_reduce<1>", (const float*)w.dwp,
                           (const float*)w.diag, dmix, lddmix, n, pl.wgs, B * H));
    else         RC(launch(k_dw_reduce<1>, dim3((n * n + 63) / 64), dim3(256), 0, st, "k_dw_reduce<1>", (const float*)w.dwp,
                           (const float*)w.diag, dmix, lddmix, n, pl.wgs, B * H));
    return MHLA_OK;
}

int cs_fwd_impl(mhla_view q, mhla_view k, mhla_view v, const float* mix, int ldmix, mhla_mview out, void* ws,
                size_t ws_bytes, int B, int T, int H, int K, int V, int chunk, float scale, int dtype, unsigned flags, void* stream,
                bool epi, const float* nw, float neps, mhla_view gate, mhla_mview y) {
    RC(cs_check(B, T, H, K, V, chunk, dtype, flags));
    CHECK_VIEW(q); CHECK_VIEW(k); CHECK_VIEW(v);
    if (!epi || out.ptr) CHECK_VIEW(out);
    const CsPath path = cs_path(T, K, V, chunk, dtype, flags);
    if (epi) {
        if (!cs_epi_ok(path, V))
            return fail(MHLA_ENOTSUP, "fused norm x gate epilogue needs bf16 tensors, K %% 64 == 0, K <= 256, V %% 64 == 0, V <= 256 or "
                        "V = 384 / 512, and at most 256 chunks (T=%d K=%d V=%d dtype=%d flags=0x%x)", T, K, V, dtype, flags);
        const mhla_view yv{y.ptr, y.sb, y.sn, y.sh};
        if (!view_ok16(yv) || (gate.ptr && !view_ok16(gate)) || (out.ptr && !view_ok16m(out)))
            return fail(MHLA_EINVAL, "fused norm x gate epilogue: y, gate and out must be 16-byte aligned views (strides multiples of 8)");
    }
    const int n = (T + chunk - 1) / chunk;
    if (!mix || ldmix < n) return fail(MHLA_EINVAL, "mix null or ldmix=%d < n=%d chunks (T=%d)", ldmix, n, T);
    if (path.pipe16 && !(view_ok16(q) && view_ok16(k) && view_ok16(v) && (epi || view_ok16m(out))))
        return fail(MHLA_EINVAL, "bf16 tensors with K, V multiples of 64 must be 16-byte aligned views (strides multiples of 8)");
    const CsWs w = cs_carve(ws, B, T, H, K, V, chunk, path);
    if (!ws || ws_bytes < w.total_fwd) return fail(MHLA_EINVAL, "workspace too small: %zu < %zu bytes", ws_bytes, w.total_fwd);
    if (((uintptr_t)ws) % 16) return fail(MHLA_EINVAL, "workspace not 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (path.pipe16)
        return path.hl == 2 ? cs_fwd16<2>(q, k, v, mix, ldmix, out, w, B, T, H, K, V, n, scale, st, epi, nw, neps, gate, y)
             : path.hl == 1 ? cs_fwd16<1>(q, k, v, mix, ldmix, out, w, B, T, H, K, V, n, scale, st, epi, nw, neps, gate, y)
                            : cs_fwd16<0>(q, k, v, mix, ldmix, out, w, B, T, H, K, V, n, scale, st, epi, nw, neps, gate, y);
    const long E = (long)K * V;
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

}  // namespace

extern "C" {

// ---------------------------------------------------------------------------------------------
// causal
// ---------------------------------------------------------------------------------------------
size_t mhla_causal_fwd_ws_bytes(int B, int T, int H, int K, int V, int chunk, int dtype, unsigned flags) {
    return cs_carve(nullptr, B, T, H, K, V, chunk, cs_path(T, K, V, chunk, dtype, flags)).total_fwd;
}
size_t mhla_causal_bwd_ws_bytes(int B, int T, int H, int K, int V, int chunk, int dtype, unsigned flags) {
    return cs_carve(nullptr, B, T, H, K, V, chunk, cs_path(T, K, V, chunk, dtype, flags)).total_bwd;
}
int mhla_causal_normgate_fusable(int T, int K, int V, int chunk, int dtype, unsigned flags) {
    return cs_epi_ok(cs_path(T, K, V, chunk, dtype, flags), V) ? 1 : 0;
}

// The causal operator's kernel family and summary format as text (see mhla_describe_dispatch).
int mhla_causal_describe_dispatch(int T, int K, int V, int chunk, int dtype, unsigned flags, char* buf, size_t cap) {
    RC(cs_check(1, T, 1, K, V, chunk, dtype, flags));
    const CsPath p = cs_path(T, K, V, chunk, dtype, flags);
    const int n = (T + chunk - 1) / chunk;
    std::string txt;
    if (p.pipe16) {
        txt = std::string("family=16-bit pipeline (chunk summaries tile-major, resident-sequence mixing); summaries=") +
              (p.hl == 2 ? "h16 (fp16 payload x strip multiplier: 11 significand bits, 2 bytes)" : p.hl == 1 ? "bf16 hi + lo pairs (16 significand bits, 4 bytes)"
                                                                                                            : "bf16 (single bf16 values: reduced precision, opt-in)") +
              "; fwd=k_csf_state k_csf_mixf k_csf_out4; bwd=k_csf_state " + (n > 128 ? "k_csf_mixb<dS> k_csf_mixb<dmix>" : "k_csf_mixb") + " k_csf_bwd_tok4 k_dw_reduce<1>";
    } else {
        txt = "family=generic (exact fp32 MFMA); summaries=fp32 words; fwd=k_bm_state<2> k_mix<0,1> k_cs_out; bwd=k_bm_state<2> k_mix<1,1> k_cs_bwd_tok k_dw<1> k_dw_reduce<1>";
    }
    if (buf && cap) {
        const size_t nn = txt.size() < cap - 1 ? txt.size() : cap - 1;
        memcpy(buf, txt.data(), nn);
        buf[nn] = 0;
    }
    return (int)txt.size();
}

int mhla_causal_fwd(mhla_view q, mhla_view k, mhla_view v, const float* mix, int ldmix, mhla_mview out, void* ws,
                    size_t ws_bytes, int B, int T, int H, int K, int V, int chunk, float scale, int dtype, unsigned flags, void* stream) {
    return cs_fwd_impl(q, k, v, mix, ldmix, out, ws, ws_bytes, B, T, H, K, V, chunk, scale, dtype, flags, stream, false, nullptr, 0.f,
                       mhla_view{nullptr, 0, 0, 0}, mhla_mview{nullptr, 0, 0, 0});
}

int mhla_causal_normgate_fwd(mhla_view q, mhla_view k, mhla_view v, const float* mix, int ldmix, mhla_mview out, mhla_view gate,
                             const float* norm_w, float norm_eps, mhla_mview y, void* ws, size_t ws_bytes, int B, int T, int H,
                             int K, int V, int chunk, float scale, int dtype, unsigned flags, void* stream) {
    if (!y.ptr) return fail(MHLA_EINVAL, "y null");
    return cs_fwd_impl(q, k, v, mix, ldmix, out, ws, ws_bytes, B, T, H, K, V, chunk, scale, dtype, flags, stream, true, norm_w, norm_eps, gate, y);
}

int mhla_causal_bwd(mhla_view q, mhla_view k, mhla_view v, const float* mix, int ldmix, mhla_view dout, mhla_mview dq,
                    mhla_mview dk, mhla_mview dv, float* dmix, int lddmix, void* ws, size_t ws_bytes, const void* fwd_ws,
                    int B, int T, int H, int K, int V, int chunk, float scale, int dtype, unsigned flags, void* stream) {
    RC(cs_check(B, T, H, K, V, chunk, dtype, flags));
    CHECK_VIEW(q); CHECK_VIEW(k); CHECK_VIEW(v); CHECK_VIEW(dout); CHECK_VIEW(dq); CHECK_VIEW(dk); CHECK_VIEW(dv);
    const int n = (T + chunk - 1) / chunk;
    if (!mix || ldmix < n || !dmix || lddmix < n) return fail(MHLA_EINVAL, "mix/dmix null or leading dim < n=%d chunks", n);
    const CsPath path = cs_path(T, K, V, chunk, dtype, flags);
    if (path.pipe16 && !(view_ok16(q) && view_ok16(k) && view_ok16(v) && view_ok16(dout) && view_ok16m(dq) && view_ok16m(dk) && view_ok16m(dv)))
        return fail(MHLA_EINVAL, "bf16 tensors with K, V multiples of 64 must be 16-byte aligned views (strides multiples of 8)");
    CsWs w = cs_carve(ws, B, T, H, K, V, chunk, path);
    if (!ws || ws_bytes < w.total_bwd) return fail(MHLA_EINVAL, "workspace too small: %zu < %zu bytes", ws_bytes, w.total_bwd);
    if (((uintptr_t)ws) % 16 || ((uintptr_t)fwd_ws) % 16) return fail(MHLA_EINVAL, "workspace not 16-byte aligned");
    if (fwd_ws) {   // chunk summaries S, P left by mhla_causal_fwd with the same arguments: skip their recomputation
        const CsWs f = cs_carve(const_cast<void*>(fwd_ws), B, T, H, K, V, chunk, path);
        w.S = f.S;
        w.P = f.P;
    }
    hipStream_t st = (hipStream_t)stream;
    if (path.pipe16)
        return path.hl == 2 ? cs_bwd16<2>(q, k, v, mix, ldmix, dout, dq, dk, dv, dmix, lddmix, w, fwd_ws != nullptr, B, T, H, K, V, n, scale, st)
             : path.hl == 1 ? cs_bwd16<1>(q, k, v, mix, ldmix, dout, dq, dk, dv, dmix, lddmix, w, fwd_ws != nullptr, B, T, H, K, V, n, scale, st)
                            : cs_bwd16<0>(q, k, v, mix, ldmix, dout, dq, dk, dv, dmix, lddmix, w, fwd_ws != nullptr, B, T, H, K, V, n, scale, st);
    const long E = (long)K * V;
    const int tiles = (n + 63) / 64;
    const int nsplit = dw_splits(tiles * tiles * B * H, E);
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
