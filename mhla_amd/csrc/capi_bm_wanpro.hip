// mhla_blockmix_wan_pro_fwd: the Wan2.1 inference operator (rotary prologue, per-head norm x gate epilogue: mhla_blockmix_wan_fwd) with
// the q / k prologue of wan/mhla_utils.py:268-272 folded into the operator's loads.  q, k, v are the 16-bit projection outputs, read in
// place; the summary and output kernels apply relu(x rstd[token] w[channel]) + eps in fp32 while loading (split.hpp PRO) -- the same
// numbers k_qk_prologue used to write as fp32 tensors (2 x 194 MB per layer at the 1.3B shape) and the operator to read back at 4 bytes
// per element.  Products as for fp32 tensors (bf16 hi + lo operands), summaries as 24-bit floats.
#include "capi_bm_typed.hpp"

using namespace mhla;
using namespace mhla::capi;

namespace {

template <typename ET>
int wan_pro_fwd(const mhla_view& q, const mhla_view& k, const mhla_view& v, const float* rstd_q, const float* rstd_k, const float* wq,
                const float* wk, bool normalize, const float* W, int ldw, const float* rcos, const float* rsin, long ldr, const float* nw,
                float neps, const mhla_view& gate, const mhla_mview& out, int out_dtype, const int32_t* idx, const BmWs& w, int B, int H,
                int M, int S, int D, float eps, hipStream_t st) {
    constexpr int DT = 8, SNT = 512;
    const long es = w.es, E = (long)D * D;
    StateArgs a{};
    a.rcos = rcos; a.rsin = rsin; a.ldr = ldr;
    a.x = cv(k); a.y = cv(v); a.kd = cv(k); a.qd = cv(q); a.idx = idx;
    a.out = w.kv; a.ksum = w.ksum; a.zo = w.z; a.es = es;
    a.H = H; a.M = M; a.S = S; a.D = D; a.eps = eps; a.relu = 1; a.normalize = normalize; a.split = 0;
    a.pro_rk = rstd_k; a.pro_wk = wk; a.pro_rq = rstd_q; a.pro_wq = wq; a.pro_n = (long)M * S;
    if (rcos) RC(launch(sp::k_sp_state<ET, DT, 0, true, SNT, false, 1, true>, dim3(M, B * H), dim3(SNT), sp::sp_state_smem<DT>(), st, "k_sp_state<rope,pro>", a));
    else      RC(launch(sp::k_sp_state<ET, DT, 0, false, SNT, false, 1, true>, dim3(M, B * H), dim3(SNT), sp::sp_state_smem<DT>(), st, "k_sp_state<pro>", a));
    RC((sp_mixr<0, false, 1>(W, ldw, w.kv, w.g, M, E, es, B * H, st, normalize ? (const float*)w.z : nullptr, w.ninv, S, eps)));
    if (normalize && !sp_mixr_takes_wz<false>(M, S))
        RC(launch(k_wz<0>, dim3((S + 63) / 64, (M + 63) / 64, B * H), dim3(NTHREADS), 0, st, "k_wz<0>", W, ldw, (const float*)w.z, w.ninv, M, S, eps));
    OutArgs o{};
    o.rcos = rcos; o.rsin = rsin; o.ldr = ldr;
    o.q = cv(q); o.o = cmv(out); o.idx = idx; o.W = W; o.ldw = ldw; o.g = w.g; o.ninv = w.ninv;
    o.H = H; o.M = M; o.S = S; o.D = D; o.eps = eps; o.es = es; o.relu = 1; o.normalize = normalize;
    o.nw = nw; o.neps = neps; o.gate = cv(gate);
    o.pro_rq = rstd_q; o.pro_wq = wq; o.pro_n = (long)M * S;
    const dim3 g(M, B * H), blk(sp::SP_OUT_T);
    // the flat tile list (split.hpp k_sp_out<.., FLAT>): one persistent workgroup per CU, when the blocks are long enough for its rounds and
    // there are enough tiles to cut (mhla_set_option("recut_kernels", 0) / MHLA_WAN_FLAT=0: the block-per-workgroup launch, for A/B)
    o.nbh = B * H;
    const int tpi = (S + 15) / 16;
    if (g_recut.load() && tpi >= 8 && (long)B * H * M * tpi >= 256L * 32) {
        const dim3 gf(256);
        if (out_dtype == MHLA_BF16)     RC(launch(sp::k_sp_out<ET, DT, bf16_t, true, false, 1, true, true>, gf, blk, 2 * sp::sp_out_smem<DT>(), st, "k_sp_out<norm,pro,flat>", o));
        else if (out_dtype == MHLA_F16) RC(launch(sp::k_sp_out<ET, DT, f16_t, true, false, 1, true, true>, gf, blk, 2 * sp::sp_out_smem<DT>(), st, "k_sp_out<norm,pro,flat>", o));
        else                            RC(launch(sp::k_sp_out<ET, DT, float, true, false, 1, true, true>, gf, blk, 2 * sp::sp_out_smem<DT>(), st, "k_sp_out<norm,pro,flat>", o));
        return MHLA_OK;
    }
    if (out_dtype == MHLA_BF16)     RC(launch(sp::k_sp_out<ET, DT, bf16_t, true, false, 1, true>, g, blk, sp::sp_out_smem<DT>(), st, "k_sp_out<norm,pro>", o));
    else if (out_dtype == MHLA_F16) RC(launch(sp::k_sp_out<ET, DT, f16_t, true, false, 1, true>, g, blk, sp::sp_out_smem<DT>(), st, "k_sp_out<norm,pro>", o));
    else                            RC(launch(sp::k_sp_out<ET, DT, float, true, false, 1, true>, g, blk, sp::sp_out_smem<DT>(), st, "k_sp_out<norm,pro>", o));
    return MHLA_OK;
}

}  // namespace

extern "C" {

// 1 when mhla_blockmix_wan_pro_fwd serves this problem (otherwise: mhla_qk_prologue + mhla_blockmix_wan_fwd)
int mhla_blockmix_wan_pro_ok(int M, int S, int D, int dtype, unsigned flags) {
    return (dtype == MHLA_BF16 || dtype == MHLA_F16) && D > 96 && D <= 128 && sp_shape_ok(D, flags) && bm_sumfmt(M, S, D, MHLA_F32, flags) == SF_P24;
}

int mhla_blockmix_wan_pro_fwd(mhla_view q, mhla_view k, mhla_view v, const float* rstd_q, const float* rstd_k, const float* wq,
                              const float* wk, int normalize, const float* W, int ldw, const float* rope_cos, const float* rope_sin,
                              int64_t ld_rope, const float* norm_w, float norm_eps, mhla_view gate, mhla_mview out, int out_dtype,
                              const int32_t* block_index, void* ws, size_t ws_bytes, int B, int H, int M, int S, int D, int dtype,
                              float eps, unsigned flags, void* stream) {
    RC(bm_check(B, H, M, S, D, dtype, flags, normalize != 0, false));
    if (!mhla_blockmix_wan_pro_ok(M, S, D, dtype, flags))
        return fail(MHLA_ENOTSUP, "prologue on load needs 16-bit q, k, v, 96 < D <= 128, D %% 8 == 0 and at most 192 blocks (M=%d D=%d dtype=%d)", M, D, dtype);
    CHECK_VIEW(q); CHECK_VIEW(k); CHECK_VIEW(v); CHECK_VIEW(out);
    if (!(view_ok16(q) && view_ok16(k) && view_ok16(v))) return fail(MHLA_EINVAL, "q, k, v must be 16-byte aligned views (strides multiples of 8)");
    if (!W || ldw < M) return fail(MHLA_EINVAL, "W null or ldw=%d < M=%d", ldw, M);
    if ((rope_cos == nullptr) != (rope_sin == nullptr)) return fail(MHLA_EINVAL, "rope_cos and rope_sin must be given together");
    if (rope_cos && (ld_rope < D / 2 || (ld_rope & 3) || ((uintptr_t)rope_cos | (uintptr_t)rope_sin) % 16))
        return fail(MHLA_EINVAL, "rope tables: ld=%lld must be >= D/2, a multiple of 4, and the tables 16-byte aligned", (long long)ld_rope);
    if (((uintptr_t)wq | (uintptr_t)wk | (uintptr_t)norm_w) % 16) return fail(MHLA_EINVAL, "norm weights must be 16-byte aligned");
    if (out_dtype < 0 || out_dtype > 2) return fail(MHLA_EINVAL, "unknown out_dtype %d", out_dtype);
    if (gate.ptr && (((uintptr_t)gate.ptr) % 8 || ((gate.sb | gate.sn | gate.sh) & 3))) return fail(MHLA_EINVAL, "gate: pointer must be 8-byte aligned, strides multiples of 4");
    if (flags & MHLA_FLAG_RELU_EPS) return fail(MHLA_EINVAL, "the prologue (norm, relu, eps) is part of this entry point: no MHLA_FLAG_RELU_EPS");
    if (!ws || ((uintptr_t)ws) % 16) return fail(MHLA_EINVAL, "workspace null or not 16-byte aligned");
    // the workspace of the fp32-tensor operator (mhla_blockmix_fwd_ws_bytes(.., MHLA_F32, ..)): 24-bit summaries
    const BmWs w = bm_carve(ws, B, H, M, S, D, SF_P24, true, false);
    if (ws_bytes < w.total_fwd) return fail(MHLA_EINVAL, "workspace too small: %zu < %zu bytes", ws_bytes, w.total_fwd);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MHLA_BF16)
        return wan_pro_fwd<bf16_t>(q, k, v, rstd_q, rstd_k, wq, wk, normalize != 0, W, ldw, rope_cos, rope_sin, (long)ld_rope, norm_w, norm_eps, gate, out,
                                   out_dtype, block_index, w, B, H, M, S, D, eps, st);
    return wan_pro_fwd<f16_t>(q, k, v, rstd_q, rstd_k, wq, wk, normalize != 0, W, ldw, rope_cos, rope_sin, (long)ld_rope, norm_w, norm_eps, gate, out,
                              out_dtype, block_index, w, B, H, M, S, D, eps, st);
}

}  // extern "C"
