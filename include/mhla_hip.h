/*
 * mhla_hip.h -- C ABI of libmhla_hip.so: the MI355X (gfx950) MHLA operator.
 *
 * This is the drop-in boundary for the MHLA hot path.  The reference
 * (DAGroup-PKU/MHLA) has no FFI / plugin registry: its boundary is plain Python
 * class substitution, and the operator itself is a handful of eager
 * torch.matmul / 1x1-conv calls.  Each entry point below names the reference
 * lines it replaces (paths relative to the reference tree).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer; the library owns no memory, all
 *     buffers (inputs, outputs, workspaces) belong to the caller;
 *   - all work is enqueued on the caller's `stream` (a hipStream_t passed as
 *     void*), nothing synchronises, no global mutable state: re-entrant;
 *   - token tensors are token-major `[B, N, H, D]` with ELEMENT strides
 *     (sb, sn, sh) and unit stride along D, so heads can be read in place from
 *     a fused QKV projection output; D and every stride must be multiples of 4
 *     elements and base pointers 16-byte aligned;
 *   - tokens are in block-major order (token p = m*S + s is offset s of block
 *     m); `block_index` (int32[N], may be NULL) maps block-major position p to
 *     the token's row in memory, which realises the reference's
 *     `rearrange(... "(fb p1 hb p2 wb p3) -> (fb hb wb) (p1 p2 p3)")` gather
 *     without a copy;
 *   - returns 0 on success, a negative MHLA_E* code otherwise (outputs
 *     untouched); mhla_last_error() gives a thread-local message.
 */
#ifndef MHLA_HIP_H
#define MHLA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MHLA_ABI_VERSION 9

enum { MHLA_F32 = 0, MHLA_BF16 = 1, MHLA_F16 = 2 };

enum {
    MHLA_OK = 0,
    MHLA_EINVAL = -22,   /* bad shape / stride / alignment / flag combination */
    MHLA_ENOTSUP = -95,  /* shape outside what the kernels cover */
    MHLA_ELAUNCH = -5    /* HIP launch error */
};

/* flags */
#define MHLA_FLAG_RELU_EPS 1u      /* apply relu(x)+eps to q,k while loading (mhla_dit/mhla/mhla.py:229-230) */
#define MHLA_FLAG_FORCE_GENERIC 2u /* testing aid: take the generic fp32-MFMA path even where the bf16 fast path applies */
#define MHLA_FLAG_NO_SMALLN 4u     /* testing aid: skip the single-launch small-sequence path (S = 16, N <= 256) */
#define MHLA_FLAG_BF16_SUMMARIES 8u /* REDUCED PRECISION, opt-in, 16-bit tensors only: every intermediate that feeds a second
                                    * contraction -- the block summaries KV, G, dG, dKV, dP = dO / n, the score tiles of the
                                    * small-sequence path -- is kept as ONE bf16 value (what the reference's own matmuls store
                                    * under bf16 autocast; half the summary traffic; 2-3e-3 of the gradients' maximum).
                                    * bf16 tensors only (fp16 tensors: MHLA_EINVAL).  Default: bf16 hi + lo operand pairs and fp32
                                    * accumulation everywhere, summaries stored with >= 11 significand bits (see
                                    * MHLA_FLAG_FP32_GRADE_SUMMARIES), one rounding of the results at the store. */

#define MHLA_FLAG_FP32_GRADE_SUMMARIES 32u /* 16-bit tensors, opt-in: keep >= 16 significand bits in the block summaries (24-bit floats /
                                    * fp32 words in the workspace: round 5's default).  DEFAULT since ABI 9 for bf16 / fp16 tensors
                                    * with blocks of >= 16 tokens, D <= 96, M <= 128: KV, G, dG, dKV are stored in 2 bytes as an fp16
                                    * payload x one power-of-two multiplier per block row -- 11 significand bits, what the
                                    * reference's own matmul / 1x1 conv (mhla_dit/mhla/mhla.py:262-263) run at under
                                    * torch.backends.cuda.matmul.allow_tf32 = True (mhla_dit/train.py:12-13); operands stay bf16
                                    * hi + lo pairs, accumulation fp32; <= 4e-4 of a result's maximum from the fp32 result
                                    * (tools/sim_h16.py, tests).  fp32 tensors are never affected. */
#define MHLA_FLAG_NO_BWD_STATE 16u  /* mhla_blockmix_fwd: no backward will use this forward's workspace (inference) -- skip what only
                                    * the backward reads (16-bit tensors, default arithmetic: the bf16 residual O - fl(O) that
                                    * gives the backward's row dots dO . O an fp32-grade O).  A backward given this flag
                                    * recomputes it. */

/* flags of the causal entry points (mhla_causal_*) */
#define MHLA_CAUSAL_FORCE_GENERIC 1u   /* testing aid: the generic kernels (exact fp32 MFMA, fp32 summaries) for every shape */
#define MHLA_CAUSAL_BF16_SUMMARIES 2u  /* REDUCED PRECISION, opt-in: chunk summaries S, P, dP, dS and the score tiles as single
                                        * bf16 values (half the summary traffic; 2-3e-3 of the result's maximum).  Default: bf16
                                        * hi + lo pairs, >= 16 significand bits wherever the reference holds fp32
                                        * (mhla_nlp/fla/ops/mhla/naive.py:39, :60-78). */

#define MHLA_CAUSAL_FP32_GRADE_SUMMARIES 4u /* opt-in: chunk summaries as bf16 hi + lo pairs (>= 16 significand bits, 4 bytes per element: round
                                        * 5's default).  DEFAULT since ABI 9 on the 16-bit pipeline: S, P, dP, dS are stored as an fp16
                                        * payload x one power-of-two multiplier per 16-row strip of a 64 x 64 chunk tile -- 11
                                        * significand bits (TF32 grade), 2 bytes; score tiles and operands stay bf16 hi + lo pairs with
                                        * fp32 accumulation; <= 4.4e-4 of a result's maximum from the fp32 result
                                        * (tools/sim_h16_causal.py, tests). */

/* A token-major view [B, N, H, D]: element strides, D contiguous. */
typedef struct {
    const void* ptr;
    int64_t sb, sn, sh;
} mhla_view;

typedef struct {
    void* ptr;
    int64_t sb, sn, sh;
} mhla_mview;

int mhla_abi_version(void);
/* Device-code options the library was compiled with, e.g. "gfx950 -O3 no-packed-fp32" (mhla_amd/build.py); the Python loader
 * refuses a library built with another set. */
const char* mhla_build_flags(void);
const char* mhla_last_error(void);

/* Process-wide options.  "bwd_two_launches" (0 / 1): run the two tile roles of the bf16-summary fast path's backward as two
 * launches instead of one fused launch with an in-launch hand-over; the library sets it itself, asynchronously and without any
 * synchronisation, after a fused launch reported an expired hand-over (see mhla_blockmix_bwd_status).  "debug_drop_signal" (0 / 1):
 * testing aid for that bounded wait.  "fp32_summaries" (0 / 1): the block-mixing operator's default arithmetic keeps its block
 * summaries as fp32 words in the workspace instead of 24-bit floats (a measurement aid: same 16-significand-bit operands either
 * way; set it between calls, not between a forward and the backward that reuses its workspace).  "recut_kernels" (1 / 0): 0 runs the
 * kernels that two re-cut ones replaced -- the twelve / sixteen-wave payload mixing at 129 .. 256 blocks, the block-per-workgroup Wan
 * inference output kernel -- with bit-identical results (A/B timing, equality tests).  Initial values:
 * MHLA_BWD_TWO_LAUNCHES=1 / MHLA_DEBUG_DROP_SIGNAL=1 / MHLA_FP32_SUMMARIES=1 / MHLA_RECUT=0 in the environment when the library is loaded.
 * Returns the previous value, MHLA_EINVAL for an unknown name. */
int mhla_set_option(const char* name, int value);

/* Which kernel family, block-summary format and launch sequence serve a block-mix / causal problem, as text:
 * "family=...; summaries=...; fwd=k1 k2 ...; bwd=k1 k2 ..." (kernel names as mhla_prof_report prints them, separated by blanks; the
 * dispatcher's own predicates; 16-byte aligned views assumed).
 * Writes at most cap - 1 characters + NUL into buf (buf may be NULL), returns the full length or a negative error code.  The
 * table in DESIGN.md section 0a lists the same for every BASELINE.json configuration. */
int mhla_describe_dispatch(int B, int H, int M, int S, int D, int dtype, int split, unsigned flags, char* buf, size_t cap);
int mhla_causal_describe_dispatch(int T, int K, int V, int chunk, int dtype, unsigned flags, char* buf, size_t cap);

/* Profiling aid (used by bench.py): when enabled, every kernel launch is bracketed by hipEvents on
 * its own stream; mhla_prof_report waits for them, writes one "kernel_name count total_ms" line per
 * kernel into buf (NUL-terminated, truncated to cap) and clears the records.  Off by default. */
void mhla_prof_enable(int on);
int mhla_prof_report(char* buf, size_t cap);
/* Debugging aid (tools/trace_tiles.py): when buf is a device buffer of 3 * tile_workgroups * 16 uint64, the three tile
 * kernels of the bf16 fast path write per-workgroup phase timestamps (s_memtime ticks) into it; NULL switches it off. */
void mhla_debug_set_trace(void* buf);

/* ---- block-mixing (non-causal) MHLA: DiT / ViT / Wan ------------------- */

/* Bytes of workspace mhla_blockmix_fwd / _bwd need for this problem (dtype: MHLA_F32/BF16/F16;
 * split: 1 when q_den/k_den do not alias q_num/k_num; flags: the flags of the call). */
size_t mhla_blockmix_fwd_ws_bytes(int B, int H, int M, int S, int D, int dtype, int split, unsigned flags);
size_t mhla_blockmix_bwd_ws_bytes(int B, int H, int M, int S, int D, int dtype, int split, unsigned flags);
/* 1 when the forward leaves reusable block summaries in `ws` (pass that buffer as `fwd_ws` to the backward). */
int mhla_blockmix_fwd_keeps_state(int B, int H, int M, int S, int D, int dtype, int split, unsigned flags);

/*
 * Forward.  Replaces mhla_dit/mhla/mhla.py:262-268 (identical:
 * mhla_image_classification/models/modules/attention/mhla.py:275-282) and
 * mhla_videogen/diffusion/model/wan/mhla_utils.py:331-341:
 *   KV_j = K_j^T V_j ; G_i = sum_j W[i,j] KV_j ;
 *   z_j[s] = Qd_j[s] . sum_s' Kd_j[s'] ; n_i[s] = sum_j W[i,j] z_j[s] + eps ;
 *   O_i = (Q_i G_i) / n_i          (no division when q_den.ptr == NULL).
 * q_num/k_num feed KV and the numerator, q_den/k_den the normaliser (Wan passes
 * roped / un-roped pairs); q_den may alias q_num (same ptr) -- the DiT/ViT case.
 * W is [M, M] fp32 row-major (row = output block i, col = input block j), ldw
 * its row stride.
 */
int mhla_blockmix_fwd(mhla_view q_num, mhla_view k_num, mhla_view v,
                      mhla_view q_den, mhla_view k_den,
                      const float* W, int ldw, mhla_mview out,
                      const int32_t* block_index, void* ws, size_t ws_bytes,
                      int B, int H, int M, int S, int D,
                      int dtype, float eps, unsigned flags, void* stream);

/*
 * Forward with the rotary prologue of the Wan host fused in (inference): replaces
 * wan/mhla_utils.py:314 (rope_apply of q and k, :127-156) + :317-341.  q, k are the un-rotated
 * tensors; the kernels rotate consecutive channel pairs (2i, 2i+1) of token row n by
 * (rope_cos[n*ld_rope + i], rope_sin[n*ld_rope + i]) while loading them: rotated k feeds KV,
 * rotated q the numerator, the plain q, k the normaliser (normalize != 0).  The tables are
 * [rows of one batch item][D/2] fp32, indexed by the token's memory row (the same index
 * block_index maps to), shared by all heads and batch items.  No q_rope / k_rope tensors exist.
 * Workspace: mhla_blockmix_fwd_ws_bytes(..., split = 0, flags).  Needs D % 8 == 0.
 */
int mhla_blockmix_rope_fwd(mhla_view q, mhla_view k, mhla_view v, int normalize,
                           const float* W, int ldw,
                           const float* rope_cos, const float* rope_sin, int64_t ld_rope,
                           mhla_mview out, const int32_t* block_index,
                           void* ws, size_t ws_bytes,
                           int B, int H, int M, int S, int D,
                           int dtype, float eps, unsigned flags, void* stream);

/*
 * The Wan layer's operator with prologue and epilogue fused (inference): as mhla_blockmix_rope_fwd (rope tables
 * optional: NULL = no rotation), and the per-head RMSNorm (x SiLU gate) that follows the operator in the host
 * (wan/mhla_utils.py:356-362: out.to(dtype); g_norm(out) [* silu(g)]) applied to each token's D outputs before
 * they are stored:  y = rmsnorm_D(round_out_dtype(O)) * norm_w [* gate * sigmoid(gate)].  q, k, v are fp32 (`dtype`
 * must be MHLA_F32, the host's .float() at :308); `out` and `gate` ([B, N, H, D] views) are in `out_dtype`.
 * norm_w: fp32 [D] or NULL; gate.ptr NULL = no gate.  The fp32 O tensor is never written.
 */
int mhla_blockmix_wan_fwd(mhla_view q, mhla_view k, mhla_view v, int normalize,
                          const float* W, int ldw,
                          const float* rope_cos, const float* rope_sin, int64_t ld_rope,
                          const float* norm_w, float norm_eps, mhla_view gate,
                          mhla_mview out, int out_dtype, const int32_t* block_index,
                          void* ws, size_t ws_bytes,
                          int B, int H, int M, int S, int D,
                          int dtype, float eps, unsigned flags, void* stream);

/*
 * mhla_blockmix_wan_fwd with the q / k prologue of the Wan layer (wan/mhla_utils.py:268-272 after the .float() at :308:
 * relu(rmsnorm_C(x) * w) + eps over the full channel dim C = H * D) folded into the operator's loads ("read directly
 * from the QKV GEMM output").  q, k, v: the 16-bit projection outputs [B, N, H, D] (`dtype` MHLA_BF16 / MHLA_F16, 16-byte aligned
 * views), read in place; rstd_q / rstd_k: fp32 [B * N], 1 / sqrt(mean_C(x^2) + norm_eps) per token from mhla_rms_rstd (NULL: no
 * norm); wq / wk: fp32 [H * D] RMSNorm weights (NULL: 1).  The kernels form relu(x * rstd * w) + eps in fp32 while loading -- the
 * values mhla_qk_prologue writes as fp32 tensors -- then rotate (rope tables optional), multiply with bf16 hi + lo operands, and
 * apply the per-head norm x gate epilogue; the fp32 q / k / v tensors never exist.  Workspace: mhla_blockmix_fwd_ws_bytes(..,
 * MHLA_F32, 0, 0).  mhla_blockmix_wan_pro_ok: 1 when this entry point serves the problem (16-bit tensors, 96 < D <= 128, D % 8 == 0,
 * at most 192 blocks); otherwise MHLA_ENOTSUP and the caller composes mhla_qk_prologue + mhla_blockmix_wan_fwd.
 */
int mhla_blockmix_wan_pro_ok(int M, int S, int D, int dtype, unsigned flags);
int mhla_blockmix_wan_pro_fwd(mhla_view q, mhla_view k, mhla_view v,
                              const float* rstd_q, const float* rstd_k, const float* wq, const float* wk, int normalize,
                              const float* W, int ldw,
                              const float* rope_cos, const float* rope_sin, int64_t ld_rope,
                              const float* norm_w, float norm_eps, mhla_view gate,
                              mhla_mview out, int out_dtype, const int32_t* block_index,
                              void* ws, size_t ws_bytes,
                              int B, int H, int M, int S, int D,
                              int dtype, float eps, unsigned flags, void* stream);
/* rstd[row] = 1 / sqrt(mean(x[row][0 .. C)^2) + norm_eps): x [rows][ldx] in `dtype`, C % 8 == 0, rstd fp32 [rows]. */
int mhla_rms_rstd(const void* x, int64_t ldx, float* rstd, int64_t rows, int C, float norm_eps, int dtype, void* stream);

/*
 * Backward (autograd of the forward above; hand-derived).
 * Needs only the forward's inputs, its output `out` and the upstream gradient
 * `dout`: the block summaries are recomputed -- unless the caller kept the
 * forward's workspace and passes it as `fwd_ws` (same call arguments, contents
 * untouched since mhla_blockmix_fwd returned; NULL = recompute).  dq_den/dk_den are
 * written only when q_den does not alias q_num; otherwise both parts are summed
 * into dq_num/dk_num.  dW is [M, M] fp32 (ld = M), overwritten (not
 * accumulated), reduced over (b, h) in a fixed order: deterministic.
 */
int mhla_blockmix_bwd(mhla_view q_num, mhla_view k_num, mhla_view v,
                      mhla_view q_den, mhla_view k_den,
                      const float* W, int ldw, mhla_view out, mhla_view dout,
                      mhla_mview dq_num, mhla_mview dk_num, mhla_mview dv,
                      mhla_mview dq_den, mhla_mview dk_den, float* dW,
                      const int32_t* block_index, void* ws, size_t ws_bytes,
                      const void* fwd_ws,
                      int B, int H, int M, int S, int D,
                      int dtype, float eps, unsigned flags, void* stream);

/* Backward of mhla_blockmix_rope_fwd (the rope of wan/mhla_utils.py:314 inside the operator's backward).
 * q, k: the UN-rotated tensors the forward was given (numerator pair = their rotation, normaliser pair = themselves when
 * normalize != 0); out / dout as in mhla_blockmix_bwd; dq, dk: gradients w.r.t. the un-rotated q, k (the transposed rotation
 * of the numerator gradients and the normaliser's part are summed inside the kernels) -- replaces autograd through
 * rope_apply (mhla_utils.py:127-156) and the 5-tensor concat / rearrange (:317-326) in the backward direction.
 * fp32 tensors, D % 8 == 0.  fwd_ws: workspace of mhla_blockmix_rope_fwd for the same arguments, or NULL. */
int mhla_blockmix_rope_bwd(mhla_view q, mhla_view k, mhla_view v, int normalize, const float* W, int ldw,
                           const float* rope_cos, const float* rope_sin, int64_t ld_rope, mhla_view out, mhla_view dout,
                           mhla_mview dq, mhla_mview dk, mhla_mview dv, float* dW, const int32_t* block_index, void* ws,
                           size_t ws_bytes, const void* fwd_ws, int B, int H, int M, int S, int D, int dtype, float eps,
                           unsigned flags, void* stream);

/* Status of the last mhla_blockmix_bwd that used `ws` (same problem arguments).  The bf16 fast path hands a tile's dksum rows
 * from its dQ workgroup to its dK/dV workgroup inside one launch through a flag; the wait for that flag is bounded, and a
 * waiter that gives up raises an error word in the workspace.  This call synchronises `stream`, reads the word and returns
 * MHLA_ELAUNCH if it is set (dk is then invalid), MHLA_OK otherwise -- also for problems that take another path.
 * Replaces nothing in the reference (mhla_dit/mhla/mhla.py:262-268 is a sequence of separate kernels); it is the fail-safe
 * of this library's own fusion.  Without this call the failure is still loud (dk = NaN) and self-healing: the library reads the
 * word asynchronously after every fused launch and runs the two roles as two launches from the next backward on (mhla_set_option). */
int mhla_blockmix_bwd_status(const void* ws, size_t ws_bytes, int B, int H, int M, int S, int D, int dtype, int split,
                             unsigned flags, void* stream);

/* ---- causal chunk-mixing MHLA: fla ------------------------------------- */

/* Workspace bytes.  bf16 tensors with K, V multiples of 64, K <= 256 and at most 256 chunks run the 16-bit-MFMA pipeline,
 * whose chunk summaries are bf16 hi + lo pairs (4 bytes per element; 2 with MHLA_CAUSAL_BF16_SUMMARIES); everything else the
 * generic kernels with fp32 summaries.  `flags` must be the flags of the calls the workspace is for. */
size_t mhla_causal_fwd_ws_bytes(int B, int T, int H, int K, int V, int chunk, int dtype, unsigned flags);
size_t mhla_causal_bwd_ws_bytes(int B, int T, int H, int K, int V, int chunk, int dtype, unsigned flags);

/*
 * Forward.  Replaces naive_chunk_simple_mhla_fixed
 * (mhla_nlp/fla/ops/mhla/naive.py:10-83): fp32 compute, q scaled by `scale`
 * (the reference always uses K^-0.5), T right-padded with zeros to a multiple
 * of `chunk`, S_j = K_j^T V_j,
 *   O_i = Q_i (sum_{j<i} mix[i,j] S_j) + mix[i,i] tril(Q_i K_i^T) V_i.
 * q,k: [B,T,H,K]  v,out: [B,T,H,V]  mix: fp32 [n.., n..] row stride ldmix,
 * n = ceil(T/chunk) rows/cols are read.  chunk must be 64.
 * Arithmetic: every product accumulates in fp32 and every intermediate that the reference keeps in fp32 (naive.py:39,
 * :60-78: S_j, the prefix mix, tril(Q K^T)) keeps >= 16 significand bits (exact fp32 MFMA on the generic path, bf16 hi + lo
 * pairs on the 16-bit pipeline); one rounding to the tensor dtype at the store (naive.py:82).
 */
int mhla_causal_fwd(mhla_view q, mhla_view k, mhla_view v, const float* mix, int ldmix,
                    mhla_mview out, void* ws, size_t ws_bytes,
                    int B, int T, int H, int K, int V, int chunk,
                    float scale, int dtype, unsigned flags, void* stream);

/* N1: the causal operator with the fla layer's epilogue fused into its store --
 *   y = o * rsqrt(mean(o^2 over V) + norm_eps) * norm_w * gate * sigmoid(gate)
 * i.e. FusedRMSNormGated over each head's V channels (mhla_nlp/fla/layers/mhla.py:351-355; kernel math
 * mhla_nlp/fla/modules/fused_norm_gate.py:77-99).  `out` (the operator's own output o) is optional: a NULL ptr skips its
 * store (inference); training passes it so that the norm's backward (mhla_rmsnorm_gate_bwd) has its input.  `gate` ptr NULL:
 * no gate; `norm_w` NULL: no affine weight.  Covers what mhla_causal_normgate_fusable() reports (bf16 tensors, K % 64 == 0,
 * K <= 256, V % 64 == 0 with V <= 256 or V = 384 / 512 -- one workgroup owns a head's channels, the wide heads in two halves --,
 * at most 256 chunks); otherwise MHLA_ENOTSUP and
 * the caller runs mhla_causal_fwd + mhla_rmsnorm_gate_fwd.  Workspace: as mhla_causal_fwd (usable as `fwd_ws` of
 * mhla_causal_bwd with the same flags). */
int mhla_causal_normgate_fusable(int T, int K, int V, int chunk, int dtype, unsigned flags);
int mhla_causal_normgate_fwd(mhla_view q, mhla_view k, mhla_view v, const float* mix, int ldmix, mhla_mview out, mhla_view gate,
                             const float* norm_w, float norm_eps, mhla_mview y, void* ws, size_t ws_bytes, int B, int T, int H,
                             int K, int V, int chunk, float scale, int dtype, unsigned flags, void* stream);

/* Backward (autograd of naive.py:39-82).  dmix is [n, n] fp32 with row stride lddmix;
 * every entry of the leading n x n is written (zeros above the diagonal).
 * fwd_ws: the workspace mhla_causal_fwd was given for the same arguments and flags, contents untouched
 * (its chunk summaries S_j and the prefix mixes are reused), or NULL to recompute them. */
int mhla_causal_bwd(mhla_view q, mhla_view k, mhla_view v, const float* mix, int ldmix,
                    mhla_view dout, mhla_mview dq, mhla_mview dk, mhla_mview dv,
                    float* dmix, int lddmix, void* ws, size_t ws_bytes, const void* fwd_ws,
                    int B, int T, int H, int K, int V, int chunk,
                    float scale, int dtype, unsigned flags, void* stream);

/* ---- prologue: q / k of the Wan host -------------------------------------- */

/*
 * y = relu(rmsnorm_C(x) * w) + eps  per token row over the whole channel dim C = H*D.
 * Replaces wan/mhla_utils.py:268-272 (norm_q / norm_k = WanRMSNorm(dim), wan/model.py:181-196,
 * then relu + eps) after the .float() at :308: x is the projection output in `dtype`, y fp32.
 * norm == 0 (qk_norm=False): y = relu(x) + eps.  w: fp32 [C] or NULL.  C % 8 == 0, C <= 4096.
 */
int mhla_qk_prologue(const void* x, int64_t ldx, const float* w, float* y, int64_t ldy,
                     int64_t rows, int C, int norm, float norm_eps, float eps,
                     int dtype, void* stream);
/* The same with a second output y_rope = rope(y) (wan/mhla_utils.py:314, rope_apply :127-156): consecutive channel pairs
 * of every head (head dim D, C = H*D) rotated by (rope_cos, rope_sin)[row % ntok][pair], tables fp32 [ntok][D/2].  The
 * training path of the Wan host: y feeds the normaliser, y_rope KV and the numerator.  y_rope NULL = mhla_qk_prologue. */
int mhla_qk_prologue_rope(const void* x, int64_t ldx, const float* w, float* y, int64_t ldy,
                          float* y_rope, int64_t ldyr,
                          const float* rope_cos, const float* rope_sin, int64_t ld_tab, int ntok, int D,
                          int64_t rows, int C, int norm, float norm_eps, float eps,
                          int dtype, void* stream);
/* Backward of the two above: dx (dtype of x) from dy and / or dy_rope (fp32, either may be NULL), and per-workgroup
 * partial weight gradients dw_partial fp32 [mhla_qk_prologue_dw_rows(rows)][C] (sum the rows; NULL when w is NULL).
 * C <= 2048. */
int64_t mhla_qk_prologue_dw_rows(int64_t rows);
int mhla_qk_prologue_bwd(const void* x, int64_t ldx, const float* w,
                         const float* dy, int64_t lddy, const float* dy_rope, int64_t lddyr,
                         const float* rope_cos, const float* rope_sin, int64_t ld_tab, int ntok, int D,
                         void* dx, int64_t lddx, float* dw_partial,
                         int64_t rows, int C, int norm, float norm_eps, int dtype, void* stream);

/*
 * q / k prologue of the fla layer in one pass: feature map (0 identity, 1 relu, 2 elu+1;
 * mhla_nlp/fla/layers/mhla.py:297-299) then the NeoX-style rotary embedding (:311,
 * mhla_nlp/fla/modules/rotary.py:45-135): y[i] = f(x[i]) c - f(x[i+K/2]) s, y[i+K/2] = f(x[i+K/2]) c + f(x[i]) s with
 * (c, s) = (cos, sin)[t_offset + t][i], fp32 math, tables [rows][K/2] in the activation dtype (row stride ld_tab).
 * backward != 0: x is the upstream gradient, x_saved the forward's input; y receives dL/dx.  x, y: [B, T, H, K] views.
 */
int mhla_featmap_rotary(mhla_view x, mhla_view x_saved, const void* cos, const void* sin, int64_t ld_tab,
                        int64_t t_offset, mhla_mview y, int B, int T, int H, int K,
                        int feature_map, int backward, int dtype, void* stream);

/* ---- LePE: depthwise K x K convolution over V on the token layout --------- */

/*
 * y[b, n, c] = (bias[c]) + sum_taps w_taps[tap][c] * x[b, nbr(n, tap), c] (+ add[b, n, c]).
 * Replaces the NCHW round trip around nn.Conv2d(dim, dim, K, 1, K/2, groups=dim) at
 * mhla_dit/mhla/mhla.py:246-247 (+ the add at :271-273) and
 * mhla_image_classification/models/modules/attention/mhla.py:169 (K = 5).  Tokens are block-major:
 * n = m*S + s, block m = (py, px) on a pieces_len^2 grid, s = (by, bx) on a block_len^2 grid, pixel
 * (py*block_len + by, px*block_len + bx); zero padding at the image border.  x, add, y: [B, N, C] with batch /
 * token strides in elements (x may be the V slice of the fused QKV buffer).  w_taps: fp32 [K*K][C]
 * (the conv weight [C,1,K,K] transposed), bias: fp32 [C] or NULL, add: NULL or a tensor to add.
 * flip = 1 correlates with the flipped kernel: the gradient w.r.t. x when `x` is dout.
 */
int mhla_lepe2d(const void* x, int64_t x_sb, int64_t x_sn, const float* w_taps, const float* bias,
                const void* add, int64_t add_sb, int64_t add_sn,
                void* y, int64_t y_sb, int64_t y_sn,
                int B, int pieces_len, int block_len, int C, int K, int flip,
                int dtype, void* stream);
/* Weight and bias gradient: dwb fp32 [(K*K + 1)][C], rows 0..K*K-1 = dw_taps, row K*K = dbias;
 * overwritten, deterministic (fixed-order two-stage sum). */
size_t mhla_lepe2d_wgrad_ws_bytes(int C, int K);
int mhla_lepe2d_wgrad(const void* x, int64_t x_sb, int64_t x_sn, const void* dout, int64_t g_sb, int64_t g_sn,
                      float* dwb, void* ws, size_t ws_bytes,
                      int B, int pieces_len, int block_len, int C, int K, int dtype, void* stream);

/*
 * 3-D LePE of the Wan host: y = conv3d(x as video, w [C,1,3,3,3], bias, padding 1, groups = C) (+ add), replacing the
 * 'b (f h w) c -> b c f h w' round trip around nn.Conv3d at wan/mhla_utils.py:199-201, 349-352.  Tokens are in raster
 * order n = (f*H + h)*W + w.  w_taps: fp32 [27][C], tap = (df*3 + dh)*3 + dw; other arguments as mhla_lepe2d.
 */
int mhla_lepe3d(const void* x, int64_t x_sb, int64_t x_sn, const float* w_taps, const float* bias,
                const void* add, int64_t add_sb, int64_t add_sn,
                void* y, int64_t y_sb, int64_t y_sn,
                int B, int F, int H, int W, int C, int flip, int dtype, void* stream);
/* dwb fp32 [28][C]: rows 0..26 = dw_taps, row 27 = dbias; overwritten, deterministic. */
size_t mhla_lepe3d_wgrad_ws_bytes(int C);
int mhla_lepe3d_wgrad(const void* x, int64_t x_sb, int64_t x_sn, const void* dout, int64_t g_sb, int64_t g_sn,
                      float* dwb, void* ws, size_t ws_bytes,
                      int B, int F, int H, int W, int C, int dtype, void* stream);

/* ---- epilogue: per-head RMSNorm (x optional swish gate) ----------------- */

/*
 * y = x * rsqrt(mean_d(x^2) + eps) * w[d]  [ * g * sigmoid(g) ]   per (token, head).
 * Replaces FusedRMSNormGated (mhla_nlp/fla/modules/fused_norm_gate.py:77-99,
 * used at mhla_nlp/fla/layers/mhla.py:351-355) and Wan's per-head g_norm
 * [x SiLU gate] (mhla_videogen/diffusion/model/wan/mhla_utils.py:357-362).
 * x, g, y: [rows, D] with row strides (elements); g.ptr NULL => no gate.
 * rstd (fp32 [rows], may be NULL) is saved for the backward.
 */
int mhla_rmsnorm_gate_fwd(const void* x, int64_t ldx, const void* g, int64_t ldg,
                          const float* w, void* y, int64_t ldy, float* rstd,
                          int64_t rows, int D, float eps, int dtype, void* stream);

/* dx, dg (when gated) and a per-row-block partial of dw: dw_partial is fp32
 * [mhla_rmsnorm_gate_dw_rows(rows), D]; the caller sums it over dim 0. */
int64_t mhla_rmsnorm_gate_dw_rows(int64_t rows);
int mhla_rmsnorm_gate_bwd(const void* x, int64_t ldx, const void* g, int64_t ldg,
                          const float* w, const void* dy, int64_t lddy,
                          void* dx, int64_t lddx, void* dg, int64_t lddg,
                          float* dw_partial, int64_t rows, int D, float eps,
                          int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MHLA_HIP_H */
