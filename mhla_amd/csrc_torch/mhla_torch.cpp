// Autograd nodes of the two operators in C++ (torch::autograd::Function), calling the C ABI of libmhla_hip.so directly.
//
// Why: every reference host calls the operator eagerly (mhla_dit/mhla/mhla.py:262-268 inside MHLA4DiT.forward, naive.py:10 inside
// the fla layer).  At the DiT shape the kernels of a forward + backward take 0.126 ms, while the Python autograd.Function path
// (mhla_amd/ops.py: argument checks, seven torch.empty calls, ~20 ctypes struct conversions, two trips through the Python
// autograd machinery) costs ~180 us of host time per forward + backward (tools/host_overhead.py): the eager operator was bound
// by the host.  These nodes do the same work -- same checks, same C ABI calls, same saved state -- without the interpreter.
// PyTorch stays plumbing (tensors, streams, the autograd graph); no kernel lives here.  mhla_amd/ops.py uses these nodes when
// this library is built (mhla_amd/build.py builds it next to libmhla_hip.so) and its own Python nodes otherwise; both call the
// same C ABI, so there is no fallback in the numerical sense.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>   // (PyTorch-ROCm tensors carry the device type "cuda")
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <dlfcn.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/library.h>

#include <mutex>
#include <string>

#include "../../include/mhla_hip.h"

namespace {

using torch::autograd::AutogradContext;
using torch::autograd::variable_list;
using at::Tensor;

struct Abi {
    void* handle = nullptr;
    decltype(&mhla_blockmix_fwd) blockmix_fwd = nullptr;
    decltype(&mhla_blockmix_bwd) blockmix_bwd = nullptr;
    decltype(&mhla_blockmix_fwd_ws_bytes) blockmix_fwd_ws_bytes = nullptr;
    decltype(&mhla_blockmix_bwd_ws_bytes) blockmix_bwd_ws_bytes = nullptr;
    decltype(&mhla_blockmix_fwd_keeps_state) blockmix_fwd_keeps_state = nullptr;
    decltype(&mhla_causal_fwd) causal_fwd = nullptr;
    decltype(&mhla_causal_bwd) causal_bwd = nullptr;
    decltype(&mhla_causal_fwd_ws_bytes) causal_fwd_ws_bytes = nullptr;
    decltype(&mhla_causal_bwd_ws_bytes) causal_bwd_ws_bytes = nullptr;
    decltype(&mhla_last_error) last_error = nullptr;
    decltype(&mhla_abi_version) abi_version = nullptr;
};
Abi g_abi;
std::mutex g_mu;

template <typename F>
void bind(F& f, const char* name) {
    f = reinterpret_cast<F>(dlsym(g_abi.handle, name));
    TORCH_CHECK(f != nullptr, "mhla_torch: symbol ", name, " not found in libmhla_hip.so");
}

// binds the C ABI of the library at `path` (once); returns its ABI version
int64_t init(const std::string& path) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_abi.handle) {
        g_abi.handle = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
        TORCH_CHECK(g_abi.handle != nullptr, "mhla_torch: cannot load ", path, ": ", dlerror());
        bind(g_abi.blockmix_fwd, "mhla_blockmix_fwd");
        bind(g_abi.blockmix_bwd, "mhla_blockmix_bwd");
        bind(g_abi.blockmix_fwd_ws_bytes, "mhla_blockmix_fwd_ws_bytes");
        bind(g_abi.blockmix_bwd_ws_bytes, "mhla_blockmix_bwd_ws_bytes");
        bind(g_abi.blockmix_fwd_keeps_state, "mhla_blockmix_fwd_keeps_state");
        bind(g_abi.causal_fwd, "mhla_causal_fwd");
        bind(g_abi.causal_bwd, "mhla_causal_bwd");
        bind(g_abi.causal_fwd_ws_bytes, "mhla_causal_fwd_ws_bytes");
        bind(g_abi.causal_bwd_ws_bytes, "mhla_causal_bwd_ws_bytes");
        bind(g_abi.last_error, "mhla_last_error");
        bind(g_abi.abi_version, "mhla_abi_version");
    }
    return g_abi.abi_version();
}

void check_rc(int rc, const char* what) {
    TORCH_CHECK(rc == 0, what, " failed (code ", rc, "): ", g_abi.last_error());
}

int dtype_code(const Tensor& t) {
    switch (t.scalar_type()) {
        case at::kFloat: return MHLA_F32;
        case at::kBFloat16: return MHLA_BF16;
        case at::kHalf: return MHLA_F16;
        default: TORCH_CHECK_TYPE(false, "mhla_amd: unsupported dtype ", t.scalar_type(), " (float32 / bfloat16 / float16)");
    }
}

// addressable in place by every kernel family: 16-byte aligned base, strides that are multiples of 16 bytes (ops.py: _strided_ok)
bool strided_ok(const Tensor& t) {
    const int64_t mult = t.element_size() == 2 ? 8 : 4;
    return t.dim() == 4 && t.stride(3) == 1 && t.stride(0) % mult == 0 && t.stride(1) % mult == 0 && t.stride(2) % mult == 0 &&
           reinterpret_cast<uintptr_t>(t.data_ptr()) % 16 == 0;
}
Tensor prep(const Tensor& t) { return strided_ok(t) ? t : t.contiguous(); }
mhla_view view(const Tensor& t) { return mhla_view{t.data_ptr(), t.stride(0), t.stride(1), t.stride(2)}; }
mhla_mview mview(const Tensor& t) { return mhla_mview{t.data_ptr(), t.stride(0), t.stride(1), t.stride(2)}; }
const mhla_view NULL_VIEW{nullptr, 0, 0, 0};
const mhla_mview NULL_MVIEW{nullptr, 0, 0, 0};

Tensor workspace(size_t bytes, const Tensor& like) {
    return at::empty({(int64_t)(std::max<size_t>(bytes, 16) / 4 + 4)}, like.options().dtype(at::kFloat));
}
void* stream_of(const Tensor& t) { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }

void require_gpu(const Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda(), "mhla_amd operators run only on a ROCm GPU through libmhla_hip.so; ", name, " is on ", t.device(),
                " (there is no CPU fallback)");
}
void check_like(const Tensor& ref, const Tensor& t, const char* name, at::IntArrayRef shape) {
    TORCH_CHECK_VALUE(t.device() == ref.device(), name, " is on ", t.device(), ", expected ", ref.device());
    TORCH_CHECK_TYPE(t.scalar_type() == ref.scalar_type(), name, " has dtype ", t.scalar_type(), ", expected ", ref.scalar_type(),
                " (cast it: the kernels read every token tensor with one element type)");
    TORCH_CHECK_VALUE(t.sizes() == shape, name, " has shape ", t.sizes(), ", expected ", shape);
}

// ------------------------------------------------------------------------------------------------------------------------
// block-mixing operator (mhla_dit/mhla/mhla.py:262-268; wan/mhla_utils.py:331-341)
// ------------------------------------------------------------------------------------------------------------------------
struct BlockMixFn : public torch::autograd::Function<BlockMixFn> {
    static Tensor forward(AutogradContext* ctx, Tensor q, Tensor k, Tensor v, Tensor W, c10::optional<Tensor> q_den,
                          c10::optional<Tensor> k_den, c10::optional<Tensor> block_index, double eps, bool normalize, int64_t flags,
                          int64_t keep_limit) {
        at::AutoDispatchBelowADInplaceOrView guard;
        require_gpu(q, "q");
        const c10::hip::HIPGuardMasqueradingAsCUDA device_guard(q.device());   // launches go to the tensors' device, whatever is current
        TORCH_CHECK_VALUE(q.dim() == 4, "q: expected [B, N, H, D], got ", q.sizes());
        TORCH_CHECK_VALUE(W.dim() >= 2 && W.size(0) > 0, "W must be a non-empty [M, M] (or [M, M, 1, 1]) matrix, got ", W.sizes());
        const int64_t B = q.size(0), N = q.size(1), H = q.size(2), D = q.size(3), M = W.size(0);
        TORCH_CHECK_VALUE(N % M == 0, "N=", N, " tokens not divisible into M=", M, " blocks");
        const int64_t S = N / M;
        const bool split = q_den.has_value();
        TORCH_CHECK_VALUE(!split || normalize, "q_den/k_den given but normalize=False");
        TORCH_CHECK_VALUE(split == k_den.has_value(), "q_den and k_den must be given together");
        check_like(q, k, "k", q.sizes());
        check_like(q, v, "v", q.sizes());
        if (split) {
            check_like(q, *q_den, "q_den", q.sizes());
            check_like(q, *k_den, "k_den", q.sizes());
        }
        TORCH_CHECK_VALUE(W.device() == q.device() && W.dim() >= 2 && W.size(1) == M, "W must be a [M, M] (or [M, M, 1, 1]) matrix on ",
                    q.device(), ", got ", W.sizes(), " on ", W.device());
        if (block_index.has_value()) {
            const Tensor& bi = *block_index;
            TORCH_CHECK_TYPE(bi.scalar_type() == at::kInt && bi.is_contiguous(), "block_index must be a contiguous int32 tensor");
            TORCH_CHECK_VALUE(bi.device() == q.device() && bi.numel() == N, "block_index: ", N, " entries on ", q.device(), " expected");
        }
        q = prep(q); k = prep(k); v = prep(v);
        Tensor qd, kd;
        if (split) { qd = prep(*q_den); kd = prep(*k_den); }
        Tensor Wf = W.detach().reshape({M, M}).to(at::kFloat).contiguous();
        Tensor out = at::empty({B, N, H, D}, q.options());
        const int dt = dtype_code(q);
        const unsigned fl = (unsigned)flags;
        Tensor ws = workspace(g_abi.blockmix_fwd_ws_bytes(B, H, M, S, D, dt, split, fl), q);
        const mhla_view qv = view(q), kv = view(k);
        const mhla_view qdv = normalize ? (split ? view(qd) : qv) : NULL_VIEW, kdv = normalize ? (split ? view(kd) : kv) : NULL_VIEW;
        const int32_t* idx = block_index.has_value() ? block_index->data_ptr<int32_t>() : nullptr;
        check_rc(g_abi.blockmix_fwd(qv, kv, view(v), qdv, kdv, Wf.data_ptr<float>(), (int)M, mview(out), idx, ws.data_ptr(),
                                    (size_t)ws.numel() * 4, B, H, M, S, D, dt, (float)eps, fl, stream_of(q)),
                 "mhla_blockmix_fwd");
        const bool keep = g_abi.blockmix_fwd_keeps_state(B, H, M, S, D, dt, split, fl) == 1 && ws.numel() * 4 <= keep_limit;
        ctx->save_for_backward({q, k, v, Wf, out, split ? qd : Tensor(), split ? kd : Tensor(),
                                block_index.has_value() ? *block_index : Tensor(), keep ? ws : Tensor()});
        ctx->saved_data["eps"] = eps;
        ctx->saved_data["normalize"] = normalize;
        ctx->saved_data["flags"] = flags;
        ctx->saved_data["w_shape"] = W.sizes().vec();
        ctx->saved_data["w_dtype"] = (int64_t)W.scalar_type();
        return out;
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const auto saved = ctx->get_saved_variables();
        const Tensor &q = saved[0], &k = saved[1], &v = saved[2], &Wf = saved[3], &out = saved[4], &qd = saved[5], &kd = saved[6],
                     &bidx = saved[7], &fwd_ws = saved[8];
        const double eps = ctx->saved_data["eps"].toDouble();
        const bool normalize = ctx->saved_data["normalize"].toBool();
        const unsigned fl = (unsigned)ctx->saved_data["flags"].toInt();
        const bool split = qd.defined();
        const c10::hip::HIPGuardMasqueradingAsCUDA device_guard(q.device());
        const int64_t B = q.size(0), N = q.size(1), H = q.size(2), D = q.size(3), M = Wf.size(0), S = N / M;
        Tensor dout = grads[0];
        if (dout.scalar_type() != q.scalar_type()) dout = dout.to(q.scalar_type());
        check_like(q, dout, "dout", q.sizes());
        dout = prep(dout);
        Tensor dq = at::empty({B, N, H, D}, q.options()), dk = at::empty({B, N, H, D}, q.options()), dv = at::empty({B, N, H, D}, q.options());
        Tensor dW = at::empty({M, M}, q.options().dtype(at::kFloat));
        Tensor dqd, dkd;
        if (split) { dqd = at::empty_like(dq); dkd = at::empty_like(dq); }
        const int dt = dtype_code(q);
        Tensor ws = workspace(g_abi.blockmix_bwd_ws_bytes(B, H, M, S, D, dt, split, fl), q);
        const mhla_view qv = view(q), kv = view(k);
        const mhla_view qdv = normalize ? (split ? view(qd) : qv) : NULL_VIEW, kdv = normalize ? (split ? view(kd) : kv) : NULL_VIEW;
        check_rc(g_abi.blockmix_bwd(qv, kv, view(v), qdv, kdv, Wf.data_ptr<float>(), (int)M, view(out), view(dout), mview(dq), mview(dk),
                                    mview(dv), split ? mview(dqd) : NULL_MVIEW, split ? mview(dkd) : NULL_MVIEW, dW.data_ptr<float>(),
                                    bidx.defined() ? bidx.data_ptr<int32_t>() : nullptr, ws.data_ptr(), (size_t)ws.numel() * 4,
                                    fwd_ws.defined() ? fwd_ws.data_ptr() : nullptr, B, H, M, S, D, dt, (float)eps, fl, stream_of(q)),
                 "mhla_blockmix_bwd");
        const auto w_shape = ctx->saved_data["w_shape"].toIntVector();
        const auto w_dtype = (at::ScalarType)ctx->saved_data["w_dtype"].toInt();
        return {dq, dk, dv, dW.reshape(w_shape).to(w_dtype), dqd, dkd, Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

Tensor blockmix(const Tensor& q, const Tensor& k, const Tensor& v, const Tensor& W, const c10::optional<Tensor>& q_den,
                const c10::optional<Tensor>& k_den, const c10::optional<Tensor>& block_index, double eps, bool normalize, int64_t flags,
                int64_t keep_limit) {
    return BlockMixFn::apply(q, k, v, W, q_den, k_den, block_index, eps, normalize, flags, keep_limit);
}

// ------------------------------------------------------------------------------------------------------------------------
// causal chunk-mixing operator (naive_chunk_simple_mhla_fixed, mhla_nlp/fla/ops/mhla/naive.py:10-83)
// ------------------------------------------------------------------------------------------------------------------------
struct CausalFn : public torch::autograd::Function<CausalFn> {
    static Tensor forward(AutogradContext* ctx, Tensor q, Tensor k, Tensor v, Tensor mix, int64_t chunk, double scale, int64_t flags,
                          int64_t keep_limit, bool needs_grad) {
        at::AutoDispatchBelowADInplaceOrView guard;
        require_gpu(q, "q");
        const c10::hip::HIPGuardMasqueradingAsCUDA device_guard(q.device());
        TORCH_CHECK_VALUE(q.dim() == 4 && v.dim() == 4, "q, k: [B, T, H, K], v: [B, T, H, V]");
        const int64_t B = q.size(0), T = q.size(1), H = q.size(2), K = q.size(3), V = v.size(3);
        TORCH_CHECK_VALUE(chunk > 0, "chunk_size must be positive, got ", chunk);
        TORCH_CHECK_VALUE(mix.dim() >= 2, "mixing_matrix must be [L, L(, 1, 1, 1, 1)], got ", mix.sizes());
        const int64_t n = (T + chunk - 1) / chunk, L = mix.size(0);
        TORCH_CHECK_INDEX(n <= L, "sequence of ", T, " tokens needs ", n, " chunks but mixing_matrix has only ", L, " rows");
        check_like(q, k, "k", q.sizes());
        check_like(q, v, "v", {B, T, H, V});
        TORCH_CHECK_VALUE(mix.device() == q.device() && mix.dim() >= 2 && mix.size(1) >= n, "mixing_matrix must be [L, L(, 1, 1, 1, 1)] with L >= ",
                    n, " on ", q.device());
        q = prep(q); k = prep(k); v = prep(v);
        Tensor mixf = mix.detach().reshape({L, mix.size(1)}).to(at::kFloat).contiguous();
        Tensor out = at::empty({B, T, H, V}, q.options());
        const int dt = dtype_code(q);
        const unsigned fl = (unsigned)flags;
        Tensor ws = workspace(g_abi.causal_fwd_ws_bytes(B, T, H, K, V, chunk, dt, fl), q);
        check_rc(g_abi.causal_fwd(view(q), view(k), view(v), mixf.data_ptr<float>(), (int)mixf.size(1), mview(out), ws.data_ptr(),
                                  (size_t)ws.numel() * 4, B, T, H, K, V, chunk, (float)scale, dt, fl, stream_of(q)),
                 "mhla_causal_fwd");
        const bool keep = ws.numel() * 4 <= keep_limit && needs_grad;   // (no backward to come: the summaries are not kept alive)
        ctx->save_for_backward({q, k, v, mixf, keep ? ws : Tensor()});
        ctx->saved_data["chunk"] = chunk;
        ctx->saved_data["scale"] = scale;
        ctx->saved_data["flags"] = flags;
        ctx->saved_data["mix_shape"] = mix.sizes().vec();
        ctx->saved_data["mix_dtype"] = (int64_t)mix.scalar_type();
        return out;
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const auto saved = ctx->get_saved_variables();
        const Tensor &q = saved[0], &k = saved[1], &v = saved[2], &mixf = saved[3], &fwd_ws = saved[4];
        const int64_t chunk = ctx->saved_data["chunk"].toInt();
        const double scale = ctx->saved_data["scale"].toDouble();
        const unsigned fl = (unsigned)ctx->saved_data["flags"].toInt();
        const c10::hip::HIPGuardMasqueradingAsCUDA device_guard(q.device());
        const int64_t B = q.size(0), T = q.size(1), H = q.size(2), K = q.size(3), V = v.size(3), n = (T + chunk - 1) / chunk;
        Tensor dout = prep(grads[0].to(q.scalar_type()));
        Tensor dq = at::empty({B, T, H, K}, q.options()), dk = at::empty({B, T, H, K}, q.options()), dv = at::empty({B, T, H, V}, q.options());
        // the library writes every entry of the leading [n, n] block (zeros above the diagonal)
        Tensor dmix = (mixf.size(0) == n && mixf.size(1) == n) ? at::empty(mixf.sizes(), mixf.options()) : at::zeros(mixf.sizes(), mixf.options());
        const int dt = dtype_code(q);
        Tensor ws = workspace(g_abi.causal_bwd_ws_bytes(B, T, H, K, V, chunk, dt, fl), q);
        check_rc(g_abi.causal_bwd(view(q), view(k), view(v), mixf.data_ptr<float>(), (int)mixf.size(1), view(dout), mview(dq), mview(dk),
                                  mview(dv), dmix.data_ptr<float>(), (int)dmix.size(1), ws.data_ptr(), (size_t)ws.numel() * 4,
                                  fwd_ws.defined() ? fwd_ws.data_ptr() : nullptr, B, T, H, K, V, chunk, (float)scale, dt, fl, stream_of(q)),
                 "mhla_causal_bwd");
        const auto mix_shape = ctx->saved_data["mix_shape"].toIntVector();
        const auto mix_dtype = (at::ScalarType)ctx->saved_data["mix_dtype"].toInt();
        return {dq, dk, dv, dmix.reshape(mix_shape).to(mix_dtype), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

Tensor causal(const Tensor& q, const Tensor& k, const Tensor& v, const Tensor& mix, int64_t chunk, double scale, int64_t flags,
              int64_t keep_limit) {
    const bool needs_grad = at::GradMode::is_enabled() && (q.requires_grad() || k.requires_grad() || v.requires_grad() || mix.requires_grad());
    return CausalFn::apply(q, k, v, mix, chunk, scale, flags, keep_limit, needs_grad);
}

}  // namespace

TORCH_LIBRARY(mhla_amd, m) {
    m.def("init(str path) -> int", &init);
    m.def("blockmix(Tensor q, Tensor k, Tensor v, Tensor W, Tensor? q_den, Tensor? k_den, Tensor? block_index, float eps, bool normalize, "
          "int flags, int keep_limit) -> Tensor", &blockmix);
    m.def("causal(Tensor q, Tensor k, Tensor v, Tensor mix, int chunk, float scale, int flags, int keep_limit) -> Tensor", &causal);
}
