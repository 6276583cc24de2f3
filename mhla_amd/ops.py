"""Operator-level API: torch.autograd Functions over the C ABI (libmhla_hip.so).

PyTorch is plumbing here (device memory, streams, autograd graph); every FLOP of the operator
runs in the hand-written HIP kernels.  There is no eager / CPU fallback: tensors must live on
a ROCm device and the library must be built, otherwise these functions raise.
"""
from __future__ import annotations

import functools
import os
import warnings
from typing import Optional

import torch

from . import _lib, _native
from ._lib import NULL_VIEW, View

# Autograd nodes in C++ (csrc_torch/mhla_torch.cpp) for the two plain operators when libmhla_torch.so is built: the eager host
# path of a forward + backward drops from ~180 us to the cost of the allocations and two C calls.  False forces the Python nodes
# below (same C ABI); MHLA_CHECK_HANDOVER=1 (a debugging aid of the Python nodes) does too.
USE_NATIVE_NODES = True


def _native_nodes() -> bool:
    return USE_NATIVE_NODES and _native.available() and os.environ.get("MHLA_CHECK_HANDOVER") != "1"

# (batch, head) pairs one launch of the library addresses (grid.y); larger batches are sliced by the operators below
_MAX_GRID_BH = 65535

# Forward workspaces up to these sizes are kept alive for the backward (block / chunk summaries); beyond the limit the backward
# recomputes them.  The memory is held per LAYER between its forward and its backward: the causal pipeline's hi + lo chunk
# summaries are 0.54 GB per layer at the 340M fla shape (B = 4, T = 8192) and 1.07 GB at the 1.3B-like one (B = 2) -- 13 / 26 GB
# over 24 layers -- so the causal operator has its own limit; `set_keep_state_limits` (or the fla layer's `keep_state_limit`
# argument) lowers it where memory matters more than the ~25 % of the backward the recomputation costs.
KEEP_STATE_LIMIT_BYTES = 4 << 30
CAUSAL_KEEP_STATE_LIMIT_BYTES = 4 << 30


def set_keep_state_limits(blockmix: Optional[int] = None, causal: Optional[int] = None):
    """Largest forward workspace (bytes) each operator keeps alive for its backward; 0 = always recompute.  Returns the pair in
    force.  Process-wide defaults; `mhla_causal(..., keep_state_limit=...)` overrides per call."""
    global KEEP_STATE_LIMIT_BYTES, CAUSAL_KEEP_STATE_LIMIT_BYTES
    if blockmix is not None:
        KEEP_STATE_LIMIT_BYTES = int(blockmix)
    if causal is not None:
        CAUSAL_KEEP_STATE_LIMIT_BYTES = int(causal)
    return KEEP_STATE_LIMIT_BYTES, CAUSAL_KEEP_STATE_LIMIT_BYTES

_DTYPES = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}


def _dtype_code(t: torch.Tensor) -> int:
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise TypeError(f"mhla_amd: unsupported dtype {t.dtype} (float32 / bfloat16 / float16)") from None


def _require_gpu(*ts: torch.Tensor):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "mhla_amd operators run only on a ROCm GPU through libmhla_hip.so; got a "
                f"{t.device} tensor (there is no CPU fallback)")


def _device_guard(fn):
    """Run `fn` with the device of its first GPU tensor argument current, so that `_stream()` is that device's stream and
    the library launches on it (a tensor on cuda:1 while cuda:0 is current would otherwise be launched on the wrong GPU)."""
    @functools.wraps(fn)
    def wrapped(*args, **kw):
        dev = next((a.device for a in args if isinstance(a, torch.Tensor) and a.is_cuda), None)
        # (the common case -- the tensor's device is already current -- costs one C call: an eager fwd+bwd of the DiT shape is
        # bound by the host, tools/host_overhead.py)
        if dev is None or dev.index == torch._C._cuda_getDevice():
            return fn(*args, **kw)
        with torch.cuda.device(dev):
            return fn(*args, **kw)
    return wrapped


def _check_like(ref: torch.Tensor, what: str, **tensors):
    """The C ABI receives raw pointers + strides: shape, dtype and device agreement is checked here, where it is cheap.
    Every named tensor must have `ref`'s dtype and device; a tuple value is (tensor, expected_shape)."""
    for name, t in tensors.items():
        shape = None
        if isinstance(t, tuple):
            t, shape = t
        if t is None:
            continue
        if t.device != ref.device:
            raise ValueError(f"{what}: {name} is on {t.device}, expected {ref.device}")
        if t.dtype != ref.dtype:
            raise TypeError(f"{what}: {name} has dtype {t.dtype}, expected {ref.dtype} (cast it: the kernels read every "
                            "token tensor with one element type)")
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f"{what}: {name} has shape {tuple(t.shape)}, expected {tuple(shape)}")


def _check_block_index(block_index: Optional[torch.Tensor], N: int, ref: torch.Tensor):
    if block_index is None:
        return
    if block_index.dtype != torch.int32 or not block_index.is_contiguous():
        raise TypeError("block_index must be a contiguous int32 tensor")
    if block_index.device != ref.device:
        raise ValueError(f"block_index is on {block_index.device}, expected {ref.device}")
    if block_index.numel() != N:
        raise ValueError(f"block_index has {block_index.numel()} entries, expected N={N}")


def _view(t: torch.Tensor) -> View:
    """[B, N, H, D] tensor -> mhla_view (element strides; D must be contiguous)."""
    if t.dim() != 4:
        raise ValueError(f"expected a [B, N, H, D] tensor, got shape {tuple(t.shape)}")
    if t.stride(3) != 1:
        raise ValueError("last dim must be contiguous")
    return View(t.data_ptr(), t.stride(0), t.stride(1), t.stride(2))


def _strided_ok(t: torch.Tensor) -> bool:
    """Addressable in place by every kernel family: 16-byte aligned base and 16-byte row pieces (strides that are multiples of
    8 elements for 16-bit types, 4 for fp32) -- what the bf16 fast paths need, so a view never lands on a slower path or on a
    workspace-size mismatch because of its alignment."""
    mult = 8 if t.element_size() == 2 else 4
    return (t.dim() == 4 and t.stride(3) == 1 and all(s % mult == 0 for s in t.stride()[:3])
            and t.data_ptr() % 16 == 0)


def _prep(t: torch.Tensor) -> torch.Tensor:
    """Use the tensor in place when the kernels can address it, otherwise make it contiguous."""
    return t if _strided_ok(t) else t.contiguous()


def _alloc_like_tokens(B, N, H, D, ref):
    return torch.empty((B, N, H, D), dtype=ref.dtype, device=ref.device)


def _stream() -> int:
    # raw handle of the current device's current stream (torch.cuda.current_stream() builds a Stream object through several
    # Python layers: 20 us per call)
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(nbytes, 16) // 4 + 4, dtype=torch.float32, device=device)


# Workspace sizes are pure functions of the problem: one ctypes round trip per distinct problem, not per call (the eager path of a
# small operator -- the DiT shape -- is bound by the host, tools/host_overhead.py).
@functools.lru_cache(maxsize=512)
def _bm_plan(B, H, M, S, D, dt, split, flags):
    lib = _lib.load()
    return (lib.mhla_blockmix_fwd_ws_bytes(B, H, M, S, D, dt, split, flags), lib.mhla_blockmix_bwd_ws_bytes(B, H, M, S, D, dt, split, flags),
            lib.mhla_blockmix_fwd_keeps_state(B, H, M, S, D, dt, split, flags) == 1)


@functools.lru_cache(maxsize=512)
def _cs_plan(B, T, H, K, V, chunk, dt, flags):
    lib = _lib.load()
    return (lib.mhla_causal_fwd_ws_bytes(B, T, H, K, V, chunk, dt, flags), lib.mhla_causal_bwd_ws_bytes(B, T, H, K, V, chunk, dt, flags))


# ------------------------------------------------------------------------------------------
# block-mixing MHLA (DiT / ViT / Wan)
# ------------------------------------------------------------------------------------------
def _check_handover(lib, ws, B, H, M, S, D, dt, split, flags):
    """MHLA_CHECK_HANDOVER=1: after every block-mix backward, synchronise and ask the library whether a dK/dV tile of the fused
    token-gradient launch gave up waiting for its dQ tile (`mhla_blockmix_bwd_status`); raises instead of returning an invalid
    dk.  Off by default: the check is a device synchronisation (not allowed while a HIP graph is being captured)."""
    if os.environ.get("MHLA_CHECK_HANDOVER") == "1":
        rc = lib.mhla_blockmix_bwd_status(ws.data_ptr(), ws.numel() * 4, B, H, M, S, D, dt, split, flags, _stream())
        _lib.check(rc, "mhla_blockmix_bwd_status")


SUMMARIES = ("tf32", "split", "bf16")


def _bm_flags(relu_eps: bool, force_generic: bool, no_smalln: bool, summaries: str) -> int:
    if summaries not in SUMMARIES:
        raise ValueError(f"summaries={summaries!r}: 'tf32' (default: 2-byte block summaries with 11 significand bits, the precision of the "
                         "reference's TF32 matmuls), 'split' (>= 16 significand bits: 24-bit / fp32 summaries) or 'bf16' (opt-in reduced precision)")
    return ((_lib.FLAG_RELU_EPS if relu_eps else 0) | (_lib.FLAG_FORCE_GENERIC if force_generic else 0)
            | (_lib.FLAG_NO_SMALLN if no_smalln else 0) | (_lib.FLAG_BF16_SUMMARIES if summaries == "bf16" else 0)
            | (_lib.FLAG_FP32_GRADE_SUMMARIES if summaries == "split" else 0))


def describe_dispatch(B: int, H: int, M: int, S: int, D: int, dtype, *, split: bool = False, summaries: str = "tf32", relu_eps: bool = False,
                      force_generic: bool = False, no_smalln: bool = False) -> dict:
    """Which kernel family, summary format and launches serve a block-mix problem (mhla_describe_dispatch; no GPU needed):
    {"family": ..., "summaries": ..., "fwd": [...], "bwd": [...]}.  dtype: a torch dtype."""
    import ctypes
    lib = _lib.load()
    buf = ctypes.create_string_buffer(1024)
    rc = lib.mhla_describe_dispatch(B, H, M, S, D, _DTYPES[dtype], int(split), _bm_flags(relu_eps, force_generic, no_smalln, summaries), buf, len(buf))
    if rc < 0:
        _lib.check(rc, "mhla_describe_dispatch")
    return _parse_dispatch(buf.value.decode())


def describe_causal_dispatch(T: int, K: int, V: int, dtype, *, chunk_size: int = 64, summaries: str = "tf32", force_generic: bool = False) -> dict:
    """The same for the causal operator (mhla_causal_describe_dispatch)."""
    import ctypes
    lib = _lib.load()
    buf = ctypes.create_string_buffer(1024)
    rc = lib.mhla_causal_describe_dispatch(T, K, V, chunk_size, _DTYPES[dtype], _causal_flags(summaries, force_generic), buf, len(buf))
    if rc < 0:
        _lib.check(rc, "mhla_causal_describe_dispatch")
    return _parse_dispatch(buf.value.decode())


def _parse_dispatch(txt: str) -> dict:
    d = dict(part.split("=", 1) for part in txt.split("; "))
    d["fwd"], d["bwd"] = d["fwd"].split(" "), d["bwd"].split(" ")
    d["text"] = txt
    return d


def set_option(name: str, value: int) -> int:
    """mhla_set_option through the package: the process-wide options change what the workspace-size queries return
    ("fp32_summaries"), so the cached plans are dropped with every change.  Returns the previous value."""
    rc = _lib.load().mhla_set_option(name.encode(), int(value))
    if rc < 0:
        _lib.check(rc, "mhla_set_option")
    _bm_plan.cache_clear()
    _cs_plan.cache_clear()
    return rc


class _BlockMix(torch.autograd.Function):
    @staticmethod
    @_device_guard
    def forward(ctx, q, k, v, W, q_den, k_den, block_index, eps, normalize, flags):
        lib = _lib.load()
        _require_gpu(q, k, v, W, q_den, k_den, block_index)
        B, N, H, D = q.shape
        M = W.shape[0]
        if N % M:
            raise ValueError(f"N={N} tokens not divisible into M={M} blocks")
        S = N // M
        split = q_den is not None
        if split and not normalize:
            raise ValueError("q_den/k_den given but normalize=False")
        _check_like(q, "mhla_blockmix", k=(k, q.shape), v=(v, q.shape), q_den=(q_den, q.shape), k_den=(k_den, q.shape))
        _check_block_index(block_index, N, q)
        if W.device != q.device or W.dim() < 2 or W.shape[1] != M:
            raise ValueError(f"W must be a [M, M] (or [M, M, 1, 1]) matrix on {q.device}, got {tuple(W.shape)} on {W.device}")
        q, k, v = _prep(q), _prep(k), _prep(v)
        if split:
            q_den, k_den = _prep(q_den), _prep(k_den)
        Wf = W.detach().reshape(M, M).to(torch.float32).contiguous()
        out = _alloc_like_tokens(B, N, H, D, q)
        dt = _dtype_code(q)
        fwd_bytes, _, keeps = _bm_plan(B, H, M, S, D, dt, int(split), flags)
        ws = _ws(fwd_bytes, q.device)
        qv, kv = _view(q), _view(k)
        if normalize:
            qd, kd = (_view(q_den), _view(k_den)) if split else (qv, kv)
        else:
            qd, kd = NULL_VIEW, NULL_VIEW
        idx_ptr = block_index.data_ptr() if block_index is not None else None
        rc = lib.mhla_blockmix_fwd(qv, kv, _view(v), qd, kd, Wf.data_ptr(), M, _view(out), idx_ptr,
                                   ws.data_ptr(), ws.numel() * 4, B, H, M, S, D, dt, float(eps), flags, _stream())
        if rc:
            _lib.check(rc, "mhla_blockmix_fwd")
        # keep the forward's block summaries for the backward when they are the compact bf16 ones (fast path)
        keep = keeps and ws.numel() * 4 <= KEEP_STATE_LIMIT_BYTES
        ctx.save_for_backward(q, k, v, Wf, out, q_den if split else None, k_den if split else None, block_index,
                              ws if keep else None)
        ctx.cfg = (float(eps), bool(normalize), split, W.shape, W.dtype, flags)
        return out

    @staticmethod
    @_device_guard
    def backward(ctx, dout):
        lib = _lib.load()
        q, k, v, Wf, out, q_den, k_den, block_index, fwd_ws = ctx.saved_tensors
        eps, normalize, split, w_shape, w_dtype, flags = ctx.cfg
        B, N, H, D = q.shape
        M = Wf.shape[0]
        S = N // M
        if dout.dtype != q.dtype:
            dout = dout.to(q.dtype)
        _check_like(q, "mhla_blockmix backward", dout=(dout, q.shape))
        dout = _prep(dout)
        dq = _alloc_like_tokens(B, N, H, D, q)
        dk = _alloc_like_tokens(B, N, H, D, q)
        dv = _alloc_like_tokens(B, N, H, D, q)
        dW = torch.empty((M, M), dtype=torch.float32, device=q.device)
        dqd = dkd = None
        if split:
            dqd, dkd = torch.empty_like(dq), torch.empty_like(dq)
        dt = _dtype_code(q)
        ws = _ws(_bm_plan(B, H, M, S, D, dt, int(split), flags)[1], q.device)
        qv, kv = _view(q), _view(k)
        if normalize:
            qd, kd = (_view(q_den), _view(k_den)) if split else (qv, kv)
        else:
            qd, kd = NULL_VIEW, NULL_VIEW
        idx_ptr = block_index.data_ptr() if block_index is not None else None
        rc = lib.mhla_blockmix_bwd(qv, kv, _view(v), qd, kd, Wf.data_ptr(), M, _view(out), _view(dout),
                                   _view(dq), _view(dk), _view(dv),
                                   _view(dqd) if split else NULL_VIEW, _view(dkd) if split else NULL_VIEW,
                                   dW.data_ptr(), idx_ptr, ws.data_ptr(), ws.numel() * 4,
                                   fwd_ws.data_ptr() if fwd_ws is not None else None, B, H, M, S, D,
                                   dt, eps, flags, _stream())
        _lib.check(rc, "mhla_blockmix_bwd")
        _check_handover(lib, ws, B, H, M, S, D, dt, int(split), flags)
        return dq, dk, dv, dW.reshape(w_shape).to(w_dtype), dqd, dkd, None, None, None, None


def mhla_blockmix(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, W: torch.Tensor, *, eps: float = 1e-6,
                  q_den: Optional[torch.Tensor] = None, k_den: Optional[torch.Tensor] = None,
                  normalize: bool = True, block_index: Optional[torch.Tensor] = None,
                  relu_eps: bool = False, force_generic: bool = False, no_smalln: bool = False,
                  summaries: str = "tf32") -> torch.Tensor:
    """Block-mixing MHLA operator (mhla_dit/mhla/mhla.py:262-268; wan/mhla_utils.py:331-341).

    q, k, v : [B, N, H, D] token-major (any batch/token/head strides, e.g. views into a fused QKV
              projection), tokens in block-major order -- or in any order with `block_index`
              (int32[N]: block-major position -> token row).
    W       : [M, M] (or the conv weight [M, M, 1, 1]); W[i, j] mixes block j's KV summary into block i.
    q_den, k_den : optional separate pair for the normaliser (Wan: un-roped q, k).
    normalize    : False skips the division (Wan `normalize_out=False`).
    relu_eps     : apply relu(x)+eps to q and k inside the kernels (mhla.py:229-230) -- q, k are then
                   the raw projections and receive the masked gradient.
    force_generic / no_smalln: testing aids -- take the generic fp32-MFMA kernels / skip the single-launch
                   small-sequence path where they would otherwise be chosen.
    summaries    : how 16-bit problems keep what feeds a SECOND contraction (the block summaries KV, G, dG, dKV, dP = dO / n,
                   the score tiles of the 256-token path).  Operands are bf16 hi + lo pairs with fp32 accumulation in every case but
                   "bf16".  "tf32" (default): the summaries are STORED with 11 significand bits (fp16 payload x one power-of-two
                   multiplier per block row, 2 bytes) -- the precision the reference's own matmul / 1x1 conv run at under
                   allow_tf32 (mhla_dit/train.py:12-13) -- where the kernels have the format (blocks of >= 16 tokens, D <= 96,
                   M <= 128), >= 16 bits elsewhere; results within one final rounding + 1e-3 of the fp32 result (observed 4e-4).
                   "split": >= 16 significand bits everywhere (24-bit / fp32 summaries; 1e-5).  "bf16" (bf16 tensors only):
                   single bf16 values -- REDUCED PRECISION (2-3e-3 of a gradient's maximum).
    Returns [B, N, H, D] contiguous, same dtype; differentiable w.r.t. q, k, v, W (and q_den, k_den).
    """
    if (q_den is None) != (k_den is None):
        raise ValueError("q_den and k_den must be given together")
    if q.dim() != 4:
        raise ValueError(f"q: expected [B, N, H, D], got {tuple(q.shape)}")
    _check_block_index(block_index, q.shape[1], q)
    if q.shape[-1] > 128:
        return _blockmix_wide_head(q, k, v, W, eps, q_den, k_den, normalize, block_index, relu_eps, force_generic, no_smalln, summaries)
    flags = _bm_flags(relu_eps, force_generic, no_smalln, summaries)
    if not (torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (q, k, v, W, q_den, k_den))):
        flags |= _lib.FLAG_NO_BWD_STATE   # inference: the forward skips what only a backward would read
    if q.shape[0] == 0:   # empty batch: nothing to launch; keep the autograd graph connected (all gradients are zero)
        return torch.zeros_like(v) + 0 * (q.sum() + k.sum() + W.sum()).to(v.dtype)
    nb = _MAX_GRID_BH // q.shape[2]
    if q.shape[0] > nb:
        # more (batch, head) pairs than one launch addresses (the kernels index them with grid.y <= 65535; the C ABI returns
        # MHLA_ENOTSUP): batch slices through the same autograd node, the gradient of W accumulates over the slices
        sl = lambda t, i: None if t is None else t[i:i + nb]
        return torch.cat([_BlockMix.apply(sl(q, i), sl(k, i), sl(v, i), W, sl(q_den, i), sl(k_den, i), block_index, eps, normalize,
                                          flags) for i in range(0, q.shape[0], nb)], dim=0)
    if _native_nodes():
        return torch.ops.mhla_amd.blockmix(q, k, v, W, q_den, k_den, block_index, float(eps), bool(normalize), flags, KEEP_STATE_LIMIT_BYTES)
    return _BlockMix.apply(q, k, v, W, q_den, k_den, block_index, eps, normalize, flags)


def _wide_head_chunk(D: int) -> int:
    """Slice width for a head dim above 128: the largest divisor of D in [32, 128] that is a multiple of 8 (the kernels' head dims);
    without one (D = 132, 136, 152, 184, ...) the width in [64, 128] that needs the fewest slices and then the least zero padding."""
    for c in range(128, 31, -8):
        if D % c == 0:
            return c
    return min(range(64, 129, 8), key=lambda c: (-(-D // c), -(-D // c) * c - D))


def _blockmix_wide_head(q, k, v, W, eps, q_den, k_den, normalize, block_index, relu_eps, force_generic, no_smalln, summaries):
    """Head dims above 128 (the reference module takes any dim_head, mhla_dit/mhla/mhla.py:155-158; the kernels stop at 128, where a
    block summary is 64 KB): the D x D summary is a (D / c)^2 grid of c x c blocks, and the operator is linear in them --
        O[:, b] = sum_a Q[:, a] G[a, b],   G[a, b] = W (K[:, a]^T V[:, b])
    -- so the numerator is (D / c)^2 un-normalised calls of the operator on c-wide slices, and the normaliser, which couples all D
    features of a token (n_i[s] = sum_j W_ij q_j[s] . ksum_j + eps), a few [B, M, S, H] tensor ops.  Everything stays differentiable
    (the slices' autograd nodes and eager PyTorch); no BASELINE shape comes here."""
    B, N, H, D = q.shape
    out_dtype = v.dtype
    if relu_eps and q_den is not None:
        raise ValueError("relu_eps needs q_den / k_den to alias q / k (as for head dims up to 128: MHLA_FLAG_RELU_EPS)")
    c = _wide_head_chunk(D)
    Wm = W.reshape(W.shape[0], W.shape[1]) if W.dim() == 4 else W
    M = Wm.shape[0]
    if N % M:
        raise ValueError(f"N={N} tokens not divisible into M={M} blocks")
    S = N // M
    # (16-bit tensors: everything below runs on ONE fp32 copy of each tensor -- the partial products O_ab and the gradient pieces of the
    # slices would otherwise each carry their own 16-bit rounding into their sums -- and results / gradients are rounded once)
    q, k, v = q.float(), k.float(), v.float()
    if q_den is not None:
        q_den, k_den = q_den.float(), k_den.float()
    if relu_eps:   # (in PyTorch, before any padding: a padded column must stay 0, not become eps)
        q, k = torch.relu(q) + eps, torch.relu(k) + eps
    nc = -(-D // c)
    pad = nc * c - D   # zero columns add nothing to either product (no divisor of D among the kernels' head dims)
    qp, kp, vp = (torch.nn.functional.pad(t, (0, pad)) if pad else t for t in (q, k, v))
    qs, ks_, vs = ([t[..., a * c:(a + 1) * c].contiguous() for a in range(nc)] for t in (qp, kp, vp))   # sliced once
    kw = dict(eps=eps, normalize=False, block_index=block_index, relu_eps=False, force_generic=force_generic, no_smalln=no_smalln,
              summaries=summaries)
    cols = []
    for b in range(nc):
        acc = None
        for a in range(nc):
            o = mhla_blockmix(qs[a], ks_[a], vs[b], W, **kw)
            acc = o if acc is None else acc + o
        cols.append(acc)
    out = torch.cat(cols, dim=-1)[..., :D]
    if normalize:
        qd, kd = (q, k) if q_den is None else (q_den, k_den)
        if block_index is not None:   # block-major position p lives at row block_index[p]
            rows = block_index.long()
            qd, kd = qd[:, rows], kd[:, rows]
        ksum = kd.reshape(B, M, S, H, D).sum(2)                                  # [B, M, H, D]
        z = (qd.reshape(B, M, S, H, D) * ksum[:, :, None]).sum(-1)               # [B, M, S, H]
        n = torch.einsum("ij,bjsh->bish", Wm.float(), z) + eps                   # the quirk normaliser: same offset s in every block j
        n = n.reshape(B, N, H)
        if block_index is not None:   # back to the tensors' row order
            n = torch.zeros_like(n).index_copy(1, block_index.long(), n)
        out = out / n[..., None]
    return out.to(out_dtype)


class _BlockMixRope(torch.autograd.Function):
    """mhla_blockmix_rope_fwd / mhla_blockmix_rope_bwd: the operator with the rotation of q, k inside its kernels, both ways."""

    @staticmethod
    @_device_guard
    def forward(ctx, q, k, v, W, cos, sin, eps, normalize, block_index):
        lib = _lib.load()
        B, N, H, D = q.shape
        M = W.shape[0]
        S = N // M
        q, k, v = _prep(q.detach()), _prep(k.detach()), _prep(v.detach())
        Wf = W.detach().reshape(M, M).to(torch.float32).contiguous()
        out = torch.empty((B, N, H, D), dtype=q.dtype, device=q.device)
        dt = _dtype_code(q)
        ws = _ws(_bm_plan(B, H, M, S, D, dt, 0, 0)[0], q.device)
        rc = lib.mhla_blockmix_rope_fwd(_view(q), _view(k), _view(v), int(bool(normalize)), Wf.data_ptr(), M, cos.data_ptr(),
                                        sin.data_ptr(), cos.stride(0), _view(out),
                                        block_index.data_ptr() if block_index is not None else None, ws.data_ptr(),
                                        ws.numel() * 4, B, H, M, S, D, dt, float(eps), 0, _stream())
        _lib.check(rc, "mhla_blockmix_rope_fwd")
        keep = any(ctx.needs_input_grad[:4]) and ws.numel() * 4 <= KEEP_STATE_LIMIT_BYTES   # KV, G, z, ksum, 1/n for the backward
        ctx.save_for_backward(q, k, v, Wf, out, cos, sin, block_index, ws if keep else None)
        ctx.cfg = (float(eps), bool(normalize), W.shape, W.dtype)
        return out

    @staticmethod
    @_device_guard
    def backward(ctx, dout):
        lib = _lib.load()
        q, k, v, Wf, out, cos, sin, block_index, fwd_ws = ctx.saved_tensors
        eps, normalize, w_shape, w_dtype = ctx.cfg
        B, N, H, D = q.shape
        M = Wf.shape[0]
        S = N // M
        dout = _prep(dout.to(q.dtype))
        dq = _alloc_like_tokens(B, N, H, D, q)
        dk = _alloc_like_tokens(B, N, H, D, q)
        dv = _alloc_like_tokens(B, N, H, D, q)
        dW = torch.empty((M, M), dtype=torch.float32, device=q.device)
        dt = _dtype_code(q)
        ws = _ws(_bm_plan(B, H, M, S, D, dt, 0, 0)[1], q.device)
        rc = lib.mhla_blockmix_rope_bwd(_view(q), _view(k), _view(v), int(normalize), Wf.data_ptr(), M, cos.data_ptr(),
                                        sin.data_ptr(), cos.stride(0), _view(out), _view(dout), _view(dq), _view(dk), _view(dv),
                                        dW.data_ptr(), block_index.data_ptr() if block_index is not None else None,
                                        ws.data_ptr(), ws.numel() * 4, fwd_ws.data_ptr() if fwd_ws is not None else None,
                                        B, H, M, S, D, dt, eps, 0, _stream())
        _lib.check(rc, "mhla_blockmix_rope_bwd")
        return dq, dk, dv, dW.reshape(w_shape).to(w_dtype), None, None, None, None, None


def mhla_blockmix_rope(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, W: torch.Tensor, rope_cos: torch.Tensor,
                       rope_sin: torch.Tensor, *, eps: float = 1e-6, normalize: bool = True,
                       block_index: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Block-mixing operator with Wan's rotary prologue fused in (wan/mhla_utils.py:314 + :317-341), forward and backward.

    q, k are the un-rotated tensors ([B, N, H, D]); rope_cos / rope_sin are fp32 [N, D/2] (the multiplier of `rope_apply`
    per token row).  Rotated k feeds KV, rotated q the numerator, plain q / k the normaliser -- no q_rope / k_rope
    tensors are materialised, in either direction: the backward rotates q and k again where the rotated ones are needed and
    returns the gradients w.r.t. the un-rotated tensors (fp32 tensors, D % 8 == 0 for the backward)."""
    _require_gpu(q, k, v, W, rope_cos, rope_sin, block_index)
    if block_index is not None and (block_index.dtype != torch.int32 or not block_index.is_contiguous()):
        raise TypeError("block_index must be a contiguous int32 tensor")
    B, N, H, D = q.shape
    M = W.shape[0]
    if N % M:
        raise ValueError(f"N={N} tokens not divisible into M={M} blocks")
    _check_like(q, "mhla_blockmix_rope", k=(k, q.shape), v=(v, q.shape))
    _check_block_index(block_index, N, q)
    if rope_cos.shape != (N, D // 2) or rope_sin.shape != (N, D // 2) or rope_cos.dtype != torch.float32 or rope_sin.dtype != torch.float32:
        raise ValueError(f"rope tables must be fp32 [N={N}, D/2={D // 2}]")
    if torch.is_grad_enabled() and any(t.requires_grad for t in (q, k, v, W)) and (q.dtype != torch.float32 or D % 8):
        raise RuntimeError("the backward of mhla_blockmix_rope needs fp32 tensors with D % 8 == 0; rotate in the host and call "
                           "mhla_blockmix(q_rope, k_rope, v, W, q_den=q, k_den=k) otherwise")
    return _BlockMixRope.apply(q, k, v, W, rope_cos.contiguous(), rope_sin.contiguous(), eps, normalize, block_index)


# ------------------------------------------------------------------------------------------
# LePE: depthwise conv over V on the block-major token layout (DiT / ViT)
# ------------------------------------------------------------------------------------------
def _tok3(t: torch.Tensor) -> torch.Tensor:
    """[B, N, C] view with contiguous channels (copy only if the channels are strided)."""
    return t if t.stride(-1) == 1 else t.contiguous()


class _Lepe2d(torch.autograd.Function):
    @staticmethod
    @_device_guard
    def forward(ctx, v, weight, bias, add, pieces_len, block_len):
        lib = _lib.load()
        _require_gpu(v, weight, bias, add)
        B, N, C = v.shape
        K = weight.shape[-1]
        v = _tok3(v)
        w_taps = weight.detach().reshape(C, K * K).t().to(torch.float32).contiguous()
        b32 = bias.detach().to(torch.float32).contiguous() if bias is not None else None
        addc = _tok3(add) if add is not None else None
        y = torch.empty((B, N, C), dtype=v.dtype, device=v.device)
        rc = lib.mhla_lepe2d(v.data_ptr(), v.stride(0), v.stride(1), w_taps.data_ptr(),
                             b32.data_ptr() if b32 is not None else None,
                             addc.data_ptr() if addc is not None else None,
                             addc.stride(0) if addc is not None else 0, addc.stride(1) if addc is not None else 0,
                             y.data_ptr(), y.stride(0), y.stride(1), B, pieces_len, block_len, C, K, 0, _dtype_code(v), _stream())
        _lib.check(rc, "mhla_lepe2d")
        ctx.save_for_backward(v, w_taps)
        ctx.cfg = (pieces_len, block_len, K, weight.shape, weight.dtype, bias is not None, bias.dtype if bias is not None else None,
                   add is not None)
        return y

    @staticmethod
    @_device_guard
    def backward(ctx, dy):
        lib = _lib.load()
        v, w_taps = ctx.saved_tensors
        pl, bl, K, w_shape, w_dtype, has_bias, b_dtype, has_add = ctx.cfg
        B, N, C = v.shape
        dy = _tok3(dy.to(v.dtype))
        dv = dw = db = None
        if ctx.needs_input_grad[0]:
            dv = torch.empty((B, N, C), dtype=v.dtype, device=v.device)
            rc = lib.mhla_lepe2d(dy.data_ptr(), dy.stride(0), dy.stride(1), w_taps.data_ptr(), None, None, 0, 0,
                                 dv.data_ptr(), dv.stride(0), dv.stride(1), B, pl, bl, C, K, 1, _dtype_code(v), _stream())
            _lib.check(rc, "mhla_lepe2d (input gradient)")
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            dwb = torch.empty((K * K + 1, C), dtype=torch.float32, device=v.device)
            ws = _ws(lib.mhla_lepe2d_wgrad_ws_bytes(C, K), v.device)
            rc = lib.mhla_lepe2d_wgrad(v.data_ptr(), v.stride(0), v.stride(1), dy.data_ptr(), dy.stride(0), dy.stride(1),
                                       dwb.data_ptr(), ws.data_ptr(), ws.numel() * 4, B, pl, bl, C, K, _dtype_code(v), _stream())
            _lib.check(rc, "mhla_lepe2d_wgrad")
            dw = dwb[:K * K].t().reshape(w_shape).to(w_dtype)
            if has_bias:
                db = dwb[K * K].to(b_dtype)
        return dv, dw, db, (dy if has_add else None), None, None


def lepe2d(v: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], pieces_len: int, block_len: int,
           add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Depthwise conv of the DiT / ViT hosts' LePE branch on the operator's own token layout:
    `conv2d(v as image, weight [C,1,K,K], bias, padding=K//2, groups=C)` (+ `add`), with v, add, result [B, N, C] in
    block-major token order (N = pieces_len^2 * block_len^2).  Replaces the rearranges + nn.Conv2d at
    mhla_dit/mhla/mhla.py:246-247 and the add at :271-273.  Differentiable w.r.t. v, weight, bias, add."""
    if v.dim() != 3 or weight.dim() != 4 or weight.shape[1] != 1 or weight.shape[2] != weight.shape[3]:
        raise ValueError("v: [B, N, C]; weight: [C, 1, K, K]")
    if v.shape[1] != (pieces_len * block_len) ** 2 or weight.shape[0] != v.shape[2]:
        raise ValueError(f"N={v.shape[1]} != (pieces_len*block_len)^2 or channel mismatch")
    return _Lepe2d.apply(v, weight, bias, add, int(pieces_len), int(block_len))


class _Lepe3d(torch.autograd.Function):
    @staticmethod
    @_device_guard
    def forward(ctx, v, weight, bias, add, grid):
        lib = _lib.load()
        _require_gpu(v, weight, bias, add)
        B, N, C = v.shape
        F_, H_, W_ = grid
        v = _tok3(v)
        w_taps = weight.detach().reshape(C, 27).t().to(torch.float32).contiguous()
        b32 = bias.detach().to(torch.float32).contiguous() if bias is not None else None
        addc = _tok3(add) if add is not None else None
        y = torch.empty((B, N, C), dtype=v.dtype, device=v.device)
        rc = lib.mhla_lepe3d(v.data_ptr(), v.stride(0), v.stride(1), w_taps.data_ptr(),
                             b32.data_ptr() if b32 is not None else None,
                             addc.data_ptr() if addc is not None else None,
                             addc.stride(0) if addc is not None else 0, addc.stride(1) if addc is not None else 0,
                             y.data_ptr(), y.stride(0), y.stride(1), B, F_, H_, W_, C, 0, _dtype_code(v), _stream())
        _lib.check(rc, "mhla_lepe3d")
        ctx.save_for_backward(v, w_taps)
        ctx.cfg = (grid, weight.shape, weight.dtype, bias is not None, bias.dtype if bias is not None else None, add is not None)
        return y

    @staticmethod
    @_device_guard
    def backward(ctx, dy):
        lib = _lib.load()
        v, w_taps = ctx.saved_tensors
        (F_, H_, W_), w_shape, w_dtype, has_bias, b_dtype, has_add = ctx.cfg
        B, N, C = v.shape
        dy = _tok3(dy.to(v.dtype))
        dv = dw = db = None
        if ctx.needs_input_grad[0]:
            dv = torch.empty((B, N, C), dtype=v.dtype, device=v.device)
            rc = lib.mhla_lepe3d(dy.data_ptr(), dy.stride(0), dy.stride(1), w_taps.data_ptr(), None, None, 0, 0,
                                 dv.data_ptr(), dv.stride(0), dv.stride(1), B, F_, H_, W_, C, 1, _dtype_code(v), _stream())
            _lib.check(rc, "mhla_lepe3d (input gradient)")
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            dwb = torch.empty((28, C), dtype=torch.float32, device=v.device)
            ws = _ws(lib.mhla_lepe3d_wgrad_ws_bytes(C), v.device)
            rc = lib.mhla_lepe3d_wgrad(v.data_ptr(), v.stride(0), v.stride(1), dy.data_ptr(), dy.stride(0), dy.stride(1),
                                       dwb.data_ptr(), ws.data_ptr(), ws.numel() * 4, B, F_, H_, W_, C, _dtype_code(v), _stream())
            _lib.check(rc, "mhla_lepe3d_wgrad")
            dw = dwb[:27].t().reshape(w_shape).to(w_dtype)
            if has_bias:
                db = dwb[27].to(b_dtype)
        return dv, dw, db, (dy if has_add else None), None


def lepe3d(v: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], grid, add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Depthwise 3 x 3 x 3 conv of the Wan host's LePE branch on its own token layout:
    `conv3d(v as video, weight [C,1,3,3,3], bias, padding=1, groups=C)` (+ `add`), with v, add, result [B, N, C] in raster
    token order n = (f*H + h)*W + w, grid = (F, H, W).  Replaces the rearranges + nn.Conv3d at wan/mhla_utils.py:199-201,
    349-352 and the add at :363-364.  Differentiable w.r.t. v, weight, bias, add."""
    F_, H_, W_ = (int(g) for g in grid)
    if v.dim() != 3 or weight.dim() != 5 or tuple(weight.shape[1:]) != (1, 3, 3, 3):
        raise ValueError("v: [B, N, C]; weight: [C, 1, 3, 3, 3]")
    if v.shape[1] != F_ * H_ * W_ or weight.shape[0] != v.shape[2]:
        raise ValueError(f"N={v.shape[1]} != F*H*W={F_ * H_ * W_} or channel mismatch")
    return _Lepe3d.apply(v, weight, bias, add, (F_, H_, W_))


@_device_guard
def mhla_blockmix_wan(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, W: torch.Tensor, rope_cos: Optional[torch.Tensor],
                      rope_sin: Optional[torch.Tensor], norm_weight: Optional[torch.Tensor], norm_eps: float,
                      gate: Optional[torch.Tensor], out_dtype: torch.dtype, *, eps: float = 1e-6, normalize: bool = True,
                      block_index: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The Wan layer's operator with prologue and epilogue fused (inference): rotary prologue as `mhla_blockmix_rope`
    (tables optional) and the per-head RMSNorm (x SiLU gate) of wan/mhla_utils.py:356-362 applied before the store.
    q, k, v: fp32 [B, N, H, D]; gate: [B, N, H, D] in `out_dtype` or None; returns [B, N, H, D] in `out_dtype`."""
    lib = _lib.load()
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (q, k, v, W, norm_weight, gate)):
        raise RuntimeError("mhla_blockmix_wan is forward-only")
    _require_gpu(q, k, v, W, rope_cos, rope_sin, norm_weight, gate, block_index)
    if q.dtype != torch.float32:
        raise TypeError("mhla_blockmix_wan takes fp32 q, k, v (the host's .float())")
    if block_index is not None and (block_index.dtype != torch.int32 or not block_index.is_contiguous()):
        raise TypeError("block_index must be a contiguous int32 tensor")
    B, N, H, D = q.shape
    M = W.shape[0]
    if N % M:
        raise ValueError(f"N={N} tokens not divisible into M={M} blocks")
    S = N // M
    _check_like(q, "mhla_blockmix_wan", k=(k, q.shape), v=(v, q.shape))
    _check_block_index(block_index, N, q)
    q, k, v = _prep(q.detach()), _prep(k.detach()), _prep(v.detach())
    cos = sin = None
    if rope_cos is not None:
        if rope_cos.shape != (N, D // 2) or rope_cos.dtype != torch.float32 or rope_sin.shape != (N, D // 2) or rope_sin.dtype != torch.float32:
            raise ValueError(f"rope tables must be fp32 [N={N}, D/2={D // 2}]")
        cos, sin = rope_cos.contiguous(), rope_sin.contiguous()
    if gate is not None:
        if gate.shape != (B, N, H, D) or gate.dtype != out_dtype:
            raise ValueError("gate: [B, N, H, D] in out_dtype")
        gate = _prep(gate.detach())
    nw = norm_weight.detach().to(torch.float32).contiguous() if norm_weight is not None else None
    Wf = W.detach().reshape(M, M).to(torch.float32).contiguous()
    out = torch.empty((B, N, H, D), dtype=out_dtype, device=q.device)
    ws = _ws(_bm_plan(B, H, M, S, D, _lib.F32, 0, 0)[0], q.device)
    rc = lib.mhla_blockmix_wan_fwd(_view(q), _view(k), _view(v), int(bool(normalize)), Wf.data_ptr(), M,
                                   cos.data_ptr() if cos is not None else None, sin.data_ptr() if sin is not None else None,
                                   cos.stride(0) if cos is not None else 0, nw.data_ptr() if nw is not None else None,
                                   float(norm_eps), _view(gate) if gate is not None else NULL_VIEW, _view(out),
                                   _DTYPES[out_dtype], block_index.data_ptr() if block_index is not None else None,
                                   ws.data_ptr(), ws.numel() * 4, B, H, M, S, D, _lib.F32, float(eps), 0, _stream())
    _lib.check(rc, "mhla_blockmix_wan_fwd")
    return out


def wan_pro_supported(q: torch.Tensor, M: int) -> bool:
    """True when `mhla_blockmix_wan_pro` serves q [B, N, H, D] with M blocks (16-bit tensors, 96 < D <= 128, at most 192 blocks)."""
    if q.dim() != 4 or q.dtype not in (torch.bfloat16, torch.float16) or not q.is_cuda or q.shape[1] % M:
        return False
    return _lib.load().mhla_blockmix_wan_pro_ok(M, q.shape[1] // M, q.shape[3], _dtype_code(q), 0) == 1


def mhla_blockmix_wan_pro(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, wq: Optional[torch.Tensor], wk: Optional[torch.Tensor],
                          qk_norm_eps: float, W: torch.Tensor, rope_cos: Optional[torch.Tensor], rope_sin: Optional[torch.Tensor],
                          norm_weight: Optional[torch.Tensor], norm_eps: float, gate: Optional[torch.Tensor], *, eps: float = 1e-6,
                          normalize: bool = True, block_index: Optional[torch.Tensor] = None, qk_norm: bool = True) -> torch.Tensor:
    """The Wan layer's inference operator with the q / k prologue folded into its loads (mhla_blockmix_wan_pro_fwd): q, k, v are the
    16-bit projection outputs [B, N, H, D] (views of [B, N, C] are fine), wq / wk the full-dim RMSNorm weights [H * D] (None: no affine),
    `qk_norm=False`: no norm at all (relu(x) + eps).  One small kernel per tensor computes the per-token rstd; the operator's kernels
    apply relu(x * rstd * w) + eps, the rotation, and the per-head norm x gate epilogue.  Same numbers as
    `mhla_blockmix_wan(qk_prologue(q), qk_prologue(k), v.float(), ...)` without the three fp32 tensors.  Forward only; returns
    [B, N, H, D] in the dtype of q."""
    lib = _lib.load()
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (q, k, v, W, wq, wk, norm_weight, gate)):
        raise RuntimeError("mhla_blockmix_wan_pro is forward-only")
    _require_gpu(q, k, v, W, wq, wk, rope_cos, rope_sin, norm_weight, gate, block_index)
    B, N, H, D = q.shape
    M = W.shape[0]
    if N % M:
        raise ValueError(f"N={N} tokens not divisible into M={M} blocks")
    S = N // M
    _check_like(q, "mhla_blockmix_wan_pro", k=(k, q.shape), v=(v, q.shape))
    _check_block_index(block_index, N, q)
    q, k, v = _prep(q.detach()), _prep(k.detach()), _prep(v.detach())
    dt = _dtype_code(q)
    C = H * D
    f32 = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
    wq, wk, nw = f32(wq), f32(wk), f32(norm_weight)
    rq = rk = None
    if qk_norm:
        if not (q.stride(2) == D and k.stride(2) == D):
            raise ValueError("q, k: the heads of a token must be contiguous (views of the [B, N, H * D] projection)")
        rq = torch.empty(B * N, dtype=torch.float32, device=q.device)
        rk = torch.empty(B * N, dtype=torch.float32, device=q.device)
        for x, r in ((q, rq), (k, rk)):
            if x.stride(0) != N * x.stride(1):
                raise ValueError("q, k: batch stride must be N * token stride")
            _lib.check(lib.mhla_rms_rstd(x.data_ptr(), x.stride(1), r.data_ptr(), B * N, C, float(qk_norm_eps), dt, _stream()), "mhla_rms_rstd")
    cos = sin = None
    if rope_cos is not None:
        if rope_cos.shape != (N, D // 2) or rope_cos.dtype != torch.float32 or rope_sin.shape != (N, D // 2) or rope_sin.dtype != torch.float32:
            raise ValueError(f"rope tables must be fp32 [N={N}, D/2={D // 2}]")
        cos, sin = rope_cos.contiguous(), rope_sin.contiguous()
    if gate is not None:
        if gate.shape != (B, N, H, D) or gate.dtype != q.dtype:
            raise ValueError("gate: [B, N, H, D] in the dtype of q")
        gate = _prep(gate.detach())
    Wf = W.detach().reshape(M, M).to(torch.float32).contiguous()
    out = torch.empty((B, N, H, D), dtype=q.dtype, device=q.device)
    ws = _ws(_bm_plan(B, H, M, S, D, _lib.F32, 0, 0)[0], q.device)
    p = lambda t: t.data_ptr() if t is not None else None
    rc = lib.mhla_blockmix_wan_pro_fwd(_view(q), _view(k), _view(v), p(rq), p(rk), p(wq), p(wk), int(bool(normalize)), Wf.data_ptr(), M,
                                       p(cos), p(sin), cos.stride(0) if cos is not None else 0, p(nw), float(norm_eps),
                                       _view(gate) if gate is not None else NULL_VIEW, _view(out), dt,
                                       block_index.data_ptr() if block_index is not None else None, ws.data_ptr(), ws.numel() * 4,
                                       B, H, M, S, D, dt, float(eps), 0, _stream())
    _lib.check(rc, "mhla_blockmix_wan_pro_fwd")
    return out


class _DitCore(torch.autograd.Function):
    """Operator + LePE of the DiT / ViT module as ONE autograd node on the packed QKV projection output
    (mhla_dit/mhla/mhla.py:245-273): q, k, v are the three slices of `qkv` [B, N, 3, H, D] read in place; the backward writes
    dq, dk and dv (operator part + LePE part, summed inside the LePE kernel) straight into one [B, N, 3, H, D] gradient --
    no per-slice gradient tensors, no zero-fill and slice-adds by autograd."""

    @staticmethod
    @_device_guard
    def forward(ctx, qkv, W, lepe_w, lepe_b, pieces_len, block_len, eps, flags):
        lib = _lib.load()
        _require_gpu(qkv, W, lepe_w, lepe_b)
        B, N, _, H, D = qkv.shape
        M, C, K = W.shape[0], H * D, lepe_w.shape[-1]
        S = N // M
        qkv = qkv if (qkv.is_contiguous() and _strided_ok(qkv[:, :, 0])) else qkv.contiguous()
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        Wf = W.detach().reshape(M, M).to(torch.float32).contiguous()
        dt = _dtype_code(qkv)
        attn = torch.empty((B, N, H, D), dtype=qkv.dtype, device=qkv.device)
        ws = _ws(_bm_plan(B, H, M, S, D, dt, 0, flags)[0], qkv.device)
        qv, kv = _view(q), _view(k)
        rc = lib.mhla_blockmix_fwd(qv, kv, _view(v), qv, kv, Wf.data_ptr(), M, _view(attn), None, ws.data_ptr(), ws.numel() * 4,
                                   B, H, M, S, D, dt, float(eps), flags, _stream())
        _lib.check(rc, "mhla_blockmix_fwd")
        w_taps = lepe_w.detach().reshape(C, K * K).t().to(torch.float32).contiguous()
        b32 = lepe_b.detach().to(torch.float32).contiguous() if lepe_b is not None else None
        y = torch.empty((B, N, C), dtype=qkv.dtype, device=qkv.device)
        v3 = v.reshape(B, N, C)          # view: H and D are adjacent in the packed buffer
        rc = lib.mhla_lepe2d(v3.data_ptr(), v3.stride(0), v3.stride(1), w_taps.data_ptr(), b32.data_ptr() if b32 is not None else None,
                             attn.data_ptr(), N * C, C, y.data_ptr(), N * C, C, B, pieces_len, block_len, C, K, 0, dt, _stream())
        _lib.check(rc, "mhla_lepe2d")
        keep = _bm_plan(B, H, M, S, D, dt, 0, flags)[2] and ws.numel() * 4 <= KEEP_STATE_LIMIT_BYTES
        ctx.save_for_backward(qkv, Wf, attn, w_taps, ws if keep else None)
        ctx.cfg = (pieces_len, block_len, float(eps), flags, W.shape, W.dtype, lepe_w.shape, lepe_w.dtype,
                   lepe_b.dtype if lepe_b is not None else None)
        return y

    @staticmethod
    @_device_guard
    def backward(ctx, dy):
        lib = _lib.load()
        qkv, Wf, attn, w_taps, fwd_ws = ctx.saved_tensors
        pl, bl, eps, flags, w_shape, w_dtype, lw_shape, lw_dtype, lb_dtype = ctx.cfg
        B, N, _, H, D = qkv.shape
        M, C, K = Wf.shape[0], H * D, lw_shape[-1]
        S = N // M
        dt = _dtype_code(qkv)
        dy = dy.to(qkv.dtype).contiguous()
        dy4 = dy.reshape(B, N, H, D)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        dqkv = torch.empty_like(qkv)
        dv_attn = torch.empty((B, N, H, D), dtype=qkv.dtype, device=qkv.device)
        dW = torch.empty((M, M), dtype=torch.float32, device=qkv.device)
        ws = _ws(_bm_plan(B, H, M, S, D, dt, 0, flags)[1], qkv.device)
        qv, kv = _view(q), _view(k)
        rc = lib.mhla_blockmix_bwd(qv, kv, _view(v), qv, kv, Wf.data_ptr(), M, _view(attn), _view(dy4),
                                   _view(dqkv[:, :, 0]), _view(dqkv[:, :, 1]), _view(dv_attn), NULL_VIEW, NULL_VIEW,
                                   dW.data_ptr(), None, ws.data_ptr(), ws.numel() * 4,
                                   fwd_ws.data_ptr() if fwd_ws is not None else None, B, H, M, S, D, dt, eps, flags, _stream())
        _lib.check(rc, "mhla_blockmix_bwd")
        _check_handover(lib, ws, B, H, M, S, D, dt, 0, flags)
        # dv = operator part + LePE part (flipped-kernel correlation of dy), written into the V slice of the packed gradient
        dv3 = dqkv[:, :, 2].reshape(B, N, C)
        rc = lib.mhla_lepe2d(dy.data_ptr(), N * C, C, w_taps.data_ptr(), None, dv_attn.data_ptr(), N * C, C,
                             dv3.data_ptr(), dv3.stride(0), dv3.stride(1), B, pl, bl, C, K, 1, dt, _stream())
        _lib.check(rc, "mhla_lepe2d (input gradient)")
        dwb = torch.empty((K * K + 1, C), dtype=torch.float32, device=qkv.device)
        ws2 = _ws(lib.mhla_lepe2d_wgrad_ws_bytes(C, K), qkv.device)
        v3 = v.reshape(B, N, C)
        rc = lib.mhla_lepe2d_wgrad(v3.data_ptr(), v3.stride(0), v3.stride(1), dy.data_ptr(), N * C, C, dwb.data_ptr(),
                                   ws2.data_ptr(), ws2.numel() * 4, B, pl, bl, C, K, dt, _stream())
        _lib.check(rc, "mhla_lepe2d_wgrad")
        dlw = dwb[:K * K].t().reshape(lw_shape).to(lw_dtype)
        dlb = dwb[K * K].to(lb_dtype) if lb_dtype is not None else None
        return dqkv, dW.reshape(w_shape).to(w_dtype), dlw, dlb, None, None, None, None


def mhla_dit_core(qkv: torch.Tensor, W: torch.Tensor, lepe_weight: torch.Tensor, lepe_bias: Optional[torch.Tensor],
                  pieces_len: int, block_len: int, *, eps: float = 1e-6, relu_eps: bool = True, summaries: str = "tf32") -> torch.Tensor:
    """`mhla_blockmix(q, k, v, W) + LePE(v)` of the DiT / ViT module on the packed projection output `qkv` [B, N, 3, H, D]
    (block-major tokens), returning [B, N, H*D]; one autograd node whose backward emits a single packed gradient.
    `summaries`: see mhla_blockmix."""
    if qkv.dim() != 5 or qkv.shape[2] != 3:
        raise ValueError("qkv: [B, N, 3, H, D]")
    if qkv.shape[1] % W.shape[0] or qkv.shape[1] != (pieces_len * block_len) ** 2:
        raise ValueError("token count does not match the block layout")
    flags = _bm_flags(relu_eps, False, False, summaries)
    if not (torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (qkv, W, lepe_weight, lepe_bias))):
        flags |= _lib.FLAG_NO_BWD_STATE
    return _DitCore.apply(qkv, W, lepe_weight, lepe_bias, int(pieces_len), int(block_len), eps, flags)


_FMAPS = {None: 0, "identity": 0, "relu": 1, "elu": 2}


class _FmapRotary(torch.autograd.Function):
    @staticmethod
    @_device_guard
    def forward(ctx, x, cos, sin, fmap, t_offset):
        lib = _lib.load()
        _require_gpu(x, cos, sin)
        B, T, H, K = x.shape
        x = _prep(x)
        y = torch.empty((B, T, H, K), dtype=x.dtype, device=x.device)
        rc = lib.mhla_featmap_rotary(_view(x), NULL_VIEW, cos.data_ptr(), sin.data_ptr(), cos.stride(0), t_offset, _view(y),
                                     B, T, H, K, fmap, 0, _dtype_code(x), _stream())
        _lib.check(rc, "mhla_featmap_rotary")
        ctx.save_for_backward(x, cos, sin)
        ctx.cfg = (fmap, t_offset)
        return y

    @staticmethod
    @_device_guard
    def backward(ctx, dy):
        lib = _lib.load()
        x, cos, sin = ctx.saved_tensors
        fmap, t_offset = ctx.cfg
        B, T, H, K = x.shape
        dy = _prep(dy.to(x.dtype))
        dx = torch.empty((B, T, H, K), dtype=x.dtype, device=x.device)
        rc = lib.mhla_featmap_rotary(_view(dy), _view(x), cos.data_ptr(), sin.data_ptr(), cos.stride(0), t_offset, _view(dx),
                                     B, T, H, K, fmap, 1, _dtype_code(x), _stream())
        _lib.check(rc, "mhla_featmap_rotary (backward)")
        return dx, None, None, None, None


def featmap_rotary(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, feature_map: Optional[str] = None,
                   t_offset: int = 0) -> torch.Tensor:
    """Feature map (None / "identity", "relu", "elu" = elu + 1) followed by the NeoX-style rotary embedding, one HIP kernel
    each way (mhla_nlp/fla/layers/mhla.py:297-299 + :311).  x: [B, T, H, K]; cos, sin: [>= t_offset + T, K/2] in x's dtype."""
    if x.dim() != 4 or x.shape[-1] % 8:
        raise ValueError("x: [B, T, H, K] with K % 8 == 0")
    if feature_map not in _FMAPS:
        raise ValueError(f"feature_map {feature_map!r}: one of {sorted(k for k in _FMAPS if k)} or None")
    if cos.dtype != x.dtype or sin.dtype != x.dtype or cos.shape[-1] != x.shape[-1] // 2 or cos.shape[0] < t_offset + x.shape[1]:
        raise ValueError("cos/sin: [>= t_offset + T, K/2] tables in the dtype of x")
    if cos.stride(-1) != 1 or sin.stride(-1) != 1 or cos.stride(0) != sin.stride(0):
        cos, sin = cos.contiguous(), sin.contiguous()
    return _FmapRotary.apply(x, cos, sin, _FMAPS[feature_map], int(t_offset))


class _QkPrologue(torch.autograd.Function):
    @staticmethod
    @_device_guard
    def forward(ctx, x, weight, cos, sin, norm_eps, eps, head_dim):
        lib = _lib.load()
        _require_gpu(x, weight, cos, sin)
        C = x.shape[-1]
        x2 = x.detach().reshape(-1, C)
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
        rows = x2.shape[0]
        y = torch.empty((rows, C), dtype=torch.float32, device=x.device)
        w = weight.detach().to(torch.float32).contiguous() if weight is not None else None
        rope = cos is not None
        yr = torch.empty_like(y) if rope else None
        ntok = cos.shape[0] if rope else 0
        rc = lib.mhla_qk_prologue_rope(x2.data_ptr(), x2.stride(0), w.data_ptr() if w is not None else None, y.data_ptr(), C,
                                       yr.data_ptr() if rope else None, C, cos.data_ptr() if rope else None,
                                       sin.data_ptr() if rope else None, cos.stride(0) if rope else 0, ntok,
                                       int(head_dim) if rope else 0, rows, C, int(weight is not None), float(norm_eps),
                                       float(eps), _dtype_code(x2), _stream())
        _lib.check(rc, "mhla_qk_prologue_rope")
        ctx.save_for_backward(x2, w, cos, sin)
        ctx.cfg = (x.shape, float(norm_eps), int(head_dim) if rope else 0, weight.dtype if weight is not None else None)
        if rope:
            return y.reshape(x.shape), yr.reshape(x.shape)
        return y.reshape(x.shape), None

    @staticmethod
    @_device_guard
    def backward(ctx, dy, dyr):
        lib = _lib.load()
        x2, w, cos, sin = ctx.saved_tensors
        shape, norm_eps, head_dim, w_dtype = ctx.cfg
        rows, C = x2.shape
        f32 = lambda t: None if t is None else t.reshape(rows, C).to(torch.float32).contiguous()
        dy, dyr = f32(dy), f32(dyr)
        if dy is None and dyr is None:
            return None, None, None, None, None, None, None
        dx = torch.empty((rows, C), dtype=x2.dtype, device=x2.device)
        dwp = None
        if w is not None:
            dwp = torch.empty((lib.mhla_qk_prologue_dw_rows(rows), C), dtype=torch.float32, device=x2.device)
        rope = dyr is not None
        rc = lib.mhla_qk_prologue_bwd(x2.data_ptr(), x2.stride(0), w.data_ptr() if w is not None else None,
                                      dy.data_ptr() if dy is not None else None, C, dyr.data_ptr() if rope else None, C,
                                      cos.data_ptr() if rope else None, sin.data_ptr() if rope else None,
                                      cos.stride(0) if rope else 0, cos.shape[0] if rope else 0, head_dim if rope else 0,
                                      dx.data_ptr(), C, dwp.data_ptr() if dwp is not None else None, rows, C,
                                      int(w is not None), norm_eps, _dtype_code(x2), _stream())
        _lib.check(rc, "mhla_qk_prologue_bwd")
        dw = dwp.sum(0).to(w_dtype) if dwp is not None else None
        return dx.reshape(shape), dw, None, None, None, None, None


def qk_prologue(x: torch.Tensor, weight: Optional[torch.Tensor], norm_eps: float = 1e-5, eps: float = 1e-6,
                rope=None, head_dim: Optional[int] = None):
    """relu(rmsnorm(x) * weight) + eps over the last dim, fp32 output -- the q / k prologue of Wan's MHLA_Video_Uni
    (wan/mhla_utils.py:268-272 after the .float() at :308) as one HIP kernel each way.  weight None: relu(x) + eps.
    With `rope=(cos, sin)` (fp32 [N, head_dim/2] tables, token = row % N) a second tensor, the output rotated as by
    `rope_apply` (:314), is produced in the same pass and `(y, y_rope)` is returned.  Differentiable w.r.t. x and weight."""
    if x.shape[-1] % 8:
        raise ValueError("channel dim must be a multiple of 8")
    if rope is not None:
        cos, sin = rope
        if head_dim is None or cos.dtype != torch.float32 or sin.dtype != torch.float32 or cos.shape != sin.shape or \
                cos.shape[1] != head_dim // 2 or x.shape[-1] % head_dim:
            raise ValueError("rope: fp32 [N, head_dim/2] cos/sin tables and head_dim dividing the channel dim")
        y, yr = _QkPrologue.apply(x, weight, cos.contiguous(), sin.contiguous(), norm_eps, eps, head_dim)
        return y, yr
    return _QkPrologue.apply(x, weight, None, None, norm_eps, eps, 0)[0]


# ------------------------------------------------------------------------------------------
# causal chunk-mixing MHLA (fla)
# ------------------------------------------------------------------------------------------
class _Causal(torch.autograd.Function):
    @staticmethod
    @_device_guard
    def forward(ctx, q, k, v, mix, chunk_size, scale, flags, keep_limit):
        lib = _lib.load()
        _require_gpu(q, k, v, mix)
        B, T, H, K = q.shape
        V = v.shape[-1]
        n = (T + chunk_size - 1) // chunk_size
        L = mix.shape[0]
        if n > L:
            raise IndexError(f"sequence of {T} tokens needs {n} chunks but mixing_matrix has only {L} rows")
        _check_like(q, "mhla_causal", k=(k, q.shape), v=(v, (B, T, H, V)))
        if mix.device != q.device or mix.dim() < 2 or mix.shape[1] < n:
            raise ValueError(f"mixing_matrix must be [L, L(, 1, 1, 1, 1)] with L >= {n} on {q.device}")
        q, k, v = _prep(q), _prep(k), _prep(v)
        mixf = mix.detach().reshape(L, mix.shape[1]).to(torch.float32).contiguous()
        out = torch.empty((B, T, H, V), dtype=q.dtype, device=q.device)
        ws = _ws(_cs_plan(B, T, H, K, V, chunk_size, _dtype_code(q), flags)[0], q.device)
        rc = lib.mhla_causal_fwd(_view(q), _view(k), _view(v), mixf.data_ptr(), mixf.shape[1], _view(out),
                                 ws.data_ptr(), ws.numel() * 4, B, T, H, K, V, chunk_size, float(scale),
                                 _dtype_code(q), flags, _stream())
        _lib.check(rc, "mhla_causal_fwd")
        # keep the chunk summaries (S_j and their prefix mixes) for the backward unless they are very large
        keep = ws.numel() * 4 <= keep_limit and any(ctx.needs_input_grad[:4])
        ctx.save_for_backward(q, k, v, mixf, ws if keep else None)
        ctx.cfg = (chunk_size, float(scale), mix.shape, mix.dtype, flags)
        return out

    @staticmethod
    @_device_guard
    def backward(ctx, dout):
        lib = _lib.load()
        q, k, v, mixf, fwd_ws = ctx.saved_tensors
        chunk_size, scale, mix_shape, mix_dtype, flags = ctx.cfg
        B, T, H, K = q.shape
        V = v.shape[-1]
        dout = _prep(dout.to(q.dtype))
        dk = torch.empty((B, T, H, K), dtype=q.dtype, device=q.device)
        dv = torch.empty((B, T, H, V), dtype=q.dtype, device=q.device)
        dq = torch.empty((B, T, H, K), dtype=q.dtype, device=q.device)
        # the library writes every entry of the leading [n, n] block (zeros above the diagonal)
        n_chunks = (T + chunk_size - 1) // chunk_size
        dmix = (torch.empty if tuple(mixf.shape) == (n_chunks, n_chunks) else torch.zeros)(mixf.shape, dtype=torch.float32, device=q.device)
        ws = _ws(_cs_plan(B, T, H, K, V, chunk_size, _dtype_code(q), flags)[1], q.device)
        rc = lib.mhla_causal_bwd(_view(q), _view(k), _view(v), mixf.data_ptr(), mixf.shape[1], _view(dout),
                                 _view(dq), _view(dk), _view(dv), dmix.data_ptr(), dmix.shape[1],
                                 ws.data_ptr(), ws.numel() * 4, fwd_ws.data_ptr() if fwd_ws is not None else None,
                                 B, T, H, K, V, chunk_size, scale, _dtype_code(q), flags, _stream())
        _lib.check(rc, "mhla_causal_bwd")
        return dq, dk, dv, dmix.reshape(mix_shape).to(mix_dtype), None, None, None, None


def _causal_flags(summaries: str, force_generic: bool) -> int:
    if summaries not in SUMMARIES:
        raise ValueError(f"summaries={summaries!r}: 'tf32' / 'split' (bf16 hi + lo pairs, the reference's fp32 arithmetic) or 'bf16'")
    return ((_lib.CAUSAL_BF16_SUMMARIES if summaries == "bf16" else 0) | (_lib.CAUSAL_FP32_GRADE_SUMMARIES if summaries == "split" else 0)
            | (_lib.CAUSAL_FORCE_GENERIC if force_generic else 0))


def mhla_causal(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, mixing_matrix: torch.Tensor,
                chunk_size: int = 64, scale: Optional[float] = None, *, summaries: str = "tf32",
                force_generic: bool = False, keep_state_limit: Optional[int] = None) -> torch.Tensor:
    """Causal chunk-mixing MHLA operator (naive_chunk_simple_mhla_fixed,
    mhla_nlp/fla/ops/mhla/naive.py:10-83).  q, k: [B, T, H, K]; v: [B, T, H, V];
    mixing_matrix: [L, L] or [L, L, 1, 1, 1, 1], L >= ceil(T / chunk_size).  fp32 compute, output in
    the dtype of q; `scale` defaults to K**-0.5 as in the reference (naive.py:42).
    summaries: how bf16 problems keep the chunk summaries S, P, dP, dS between their two contractions (score tiles and operands are
    bf16 hi + lo pairs with fp32 accumulation in every case but "bf16") -- "tf32" (default): STORED with 11 significand bits (fp16
    payload x one power-of-two multiplier per 16-row strip of a chunk tile, 2 bytes per element), the precision of the reference's
    matmuls under allow_tf32; within one final rounding + 1e-3 of the fp32 result (observed 4e-4); "split": bf16 hi + lo pairs,
    >= 16 significand bits (naive.py:39, :60-78; 4 bytes per element); "bf16": one bf16 value each, score tiles too -- REDUCED
    PRECISION (2-3e-3 of the result's maximum).
    force_generic: testing aid -- the generic fp32-MFMA kernels for every shape.
    keep_state_limit: largest forward workspace (bytes: the chunk summaries S, P -- 4 B T H K V / 64 bytes at the default, 8 with hi + lo pairs)
    kept alive for the backward; above it the backward recomputes them.  Default: ops.CAUSAL_KEEP_STATE_LIMIT_BYTES
    (set_keep_state_limits)."""
    if q.dim() != 4 or v.dim() != 4:
        raise ValueError("q, k: [B, T, H, K], v: [B, T, H, V]")
    if int(chunk_size) <= 0:
        raise ValueError(f"chunk_size must be positive, got {chunk_size}")
    keep_limit = CAUSAL_KEEP_STATE_LIMIT_BYTES if keep_state_limit is None else int(keep_state_limit)
    flags = _causal_flags(summaries, force_generic)
    if scale is None:
        scale = q.shape[-1] ** -0.5
    if q.shape[0] == 0 or q.shape[1] == 0:   # empty batch / sequence
        return torch.zeros_like(v) + 0 * (q.sum() + k.sum() + mixing_matrix.sum()).to(v.dtype)
    nb = _MAX_GRID_BH // q.shape[2]
    if q.shape[0] > nb:   # see mhla_blockmix
        return torch.cat([_Causal.apply(q[i:i + nb], k[i:i + nb], v[i:i + nb], mixing_matrix, int(chunk_size), scale, flags, keep_limit)
                          for i in range(0, q.shape[0], nb)], dim=0)
    if _native_nodes():
        return torch.ops.mhla_amd.causal(q, k, v, mixing_matrix, int(chunk_size), float(scale), flags, keep_limit)
    return _Causal.apply(q, k, v, mixing_matrix, int(chunk_size), scale, flags, keep_limit)


def naive_chunk_simple_mhla_fixed(q, k, v, mixing_matrix, output_final_state: bool = False, chunk_size: int = 64,
                                  *args, **kwargs):
    """Drop-in for the reference op function of the same name (naive.py:11): same arguments,
    returns only `o` (the reference discards the state, naive.py:80)."""
    return mhla_causal(q, k, v, mixing_matrix, chunk_size)


class _CausalNormGate(torch.autograd.Function):
    """Causal operator + per-head RMSNorm x swish gate as ONE node: the forward applies the epilogue inside the operator's
    output kernel (mhla_causal_normgate_fwd); the backward is the norm's backward kernel followed by the operator's backward."""

    @staticmethod
    @_device_guard
    def forward(ctx, q, k, v, mix, gate, weight, chunk_size, scale, norm_eps, flags, keep_limit):
        lib = _lib.load()
        _require_gpu(q, k, v, mix, gate, weight)
        B, T, H, K = q.shape
        V = v.shape[-1]
        n = (T + chunk_size - 1) // chunk_size
        L = mix.shape[0]
        if n > L:
            raise IndexError(f"sequence of {T} tokens needs {n} chunks but mixing_matrix has only {L} rows")
        _check_like(q, "mhla_causal_normgate", k=(k, q.shape), v=(v, (B, T, H, V)), gate=(gate, (B, T, H, V)))
        q, k, v = _prep(q), _prep(k), _prep(v)
        gate = _prep(gate) if gate is not None else None
        mixf = mix.detach().reshape(L, mix.shape[1]).to(torch.float32).contiguous()
        wf = weight.detach().to(torch.float32).contiguous() if weight is not None else None
        need_grad = any(ctx.needs_input_grad[:6])
        out = torch.empty((B, T, H, V), dtype=q.dtype, device=q.device) if need_grad else None
        y = torch.empty((B, T, H, V), dtype=q.dtype, device=q.device)
        ws = _ws(_cs_plan(B, T, H, K, V, chunk_size, _dtype_code(q), flags)[0], q.device)
        rc = lib.mhla_causal_normgate_fwd(_view(q), _view(k), _view(v), mixf.data_ptr(), mixf.shape[1],
                                          _view(out) if out is not None else NULL_VIEW,
                                          _view(gate) if gate is not None else NULL_VIEW,
                                          wf.data_ptr() if wf is not None else None, float(norm_eps), _view(y),
                                          ws.data_ptr(), ws.numel() * 4, B, T, H, K, V, chunk_size, float(scale),
                                          _dtype_code(q), flags, _stream())
        _lib.check(rc, "mhla_causal_normgate_fwd")
        keep = ws.numel() * 4 <= keep_limit and need_grad
        ctx.save_for_backward(q, k, v, mixf, out, gate, wf, ws if keep else None)
        ctx.cfg = (chunk_size, float(scale), float(norm_eps), mix.shape, mix.dtype, weight.dtype if weight is not None else None, flags)
        return y

    @staticmethod
    @_device_guard
    def backward(ctx, dy):
        lib = _lib.load()
        q, k, v, mixf, out, gate, wf, fwd_ws = ctx.saved_tensors
        chunk_size, scale, norm_eps, mix_shape, mix_dtype, w_dtype, flags = ctx.cfg
        B, T, H, K = q.shape
        V = v.shape[-1]
        rows = B * T * H
        dyc = dy.contiguous().to(q.dtype)
        do = torch.empty_like(out)
        dg = torch.empty_like(out) if gate is not None else None
        gc = gate.contiguous() if gate is not None else None
        dwp = torch.empty((lib.mhla_rmsnorm_gate_dw_rows(rows), V), dtype=torch.float32, device=q.device)
        rc = lib.mhla_rmsnorm_gate_bwd(out.data_ptr(), V, gc.data_ptr() if gc is not None else None, V,
                                       wf.data_ptr() if wf is not None else None, dyc.data_ptr(), V, do.data_ptr(), V,
                                       dg.data_ptr() if dg is not None else None, V, dwp.data_ptr(), rows, V, norm_eps,
                                       _dtype_code(q), _stream())
        _lib.check(rc, "mhla_rmsnorm_gate_bwd")
        dq = torch.empty((B, T, H, K), dtype=q.dtype, device=q.device)
        dk = torch.empty((B, T, H, K), dtype=q.dtype, device=q.device)
        dv = torch.empty((B, T, H, V), dtype=q.dtype, device=q.device)
        # the library writes every entry of the leading [n, n] block (zeros above the diagonal)
        n_chunks = (T + chunk_size - 1) // chunk_size
        dmix = (torch.empty if tuple(mixf.shape) == (n_chunks, n_chunks) else torch.zeros)(mixf.shape, dtype=torch.float32, device=q.device)
        ws = _ws(_cs_plan(B, T, H, K, V, chunk_size, _dtype_code(q), flags)[1], q.device)
        rc = lib.mhla_causal_bwd(_view(q), _view(k), _view(v), mixf.data_ptr(), mixf.shape[1], _view(do),
                                 _view(dq), _view(dk), _view(dv), dmix.data_ptr(), dmix.shape[1],
                                 ws.data_ptr(), ws.numel() * 4, fwd_ws.data_ptr() if fwd_ws is not None else None,
                                 B, T, H, K, V, chunk_size, scale, _dtype_code(q), flags, _stream())
        _lib.check(rc, "mhla_causal_bwd")
        dw = dwp.sum(0).to(w_dtype) if wf is not None else None
        return dq, dk, dv, dmix.reshape(mix_shape).to(mix_dtype), dg, dw, None, None, None, None, None


def causal_normgate_fusable(q: torch.Tensor, v: torch.Tensor, chunk_size: int = 64, flags: int = 0) -> bool:
    """Shapes the fused epilogue covers (the library's own answer, mhla_causal_normgate_fusable: bf16, K % 64 == 0, K <= 256,
    V % 64 == 0, V <= 256 or V = 384 / 512, at most 256 chunks) within one launch's (batch, head) range."""
    if q.dtype not in _DTYPES or not (q.shape[0] > 0 and q.shape[1] > 0 and q.shape[0] * q.shape[2] <= _MAX_GRID_BH):
        return False
    return _lib.load().mhla_causal_normgate_fusable(q.shape[1], q.shape[-1], v.shape[-1], chunk_size, _DTYPES[q.dtype], flags) == 1


def mhla_causal_normgate(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, mixing_matrix: torch.Tensor,
                         gate: Optional[torch.Tensor], weight: Optional[torch.Tensor], norm_eps: float = 1e-5,
                         chunk_size: int = 64, scale: Optional[float] = None, *, summaries: str = "tf32",
                         keep_state_limit: Optional[int] = None) -> torch.Tensor:
    """`rmsnorm_gate(mhla_causal(q, k, v, mix), gate, weight, norm_eps)` -- the fla layer's operator + FusedRMSNormGated
    (mhla_nlp/fla/layers/mhla.py:330-355).  Where the fused epilogue applies (bf16, K, V multiples of 64, K <= 256, V <= 256 or
    384 / 512, at most 256 chunks) the norm x gate runs inside the operator's output kernel; other shapes compose the two HIP operators.
    `summaries`, `keep_state_limit`: see mhla_causal."""
    if int(chunk_size) <= 0:
        raise ValueError(f"chunk_size must be positive, got {chunk_size}")
    flags = _causal_flags(summaries, False)
    if scale is None:
        scale = q.shape[-1] ** -0.5
    if not causal_normgate_fusable(q, v, chunk_size, flags):
        return rmsnorm_gate(mhla_causal(q, k, v, mixing_matrix, chunk_size, scale, summaries=summaries, keep_state_limit=keep_state_limit),
                            gate, weight, norm_eps)
    keep_limit = CAUSAL_KEEP_STATE_LIMIT_BYTES if keep_state_limit is None else int(keep_state_limit)
    return _CausalNormGate.apply(q, k, v, mixing_matrix, gate, weight, int(chunk_size), scale, norm_eps, flags, keep_limit)


def naive_recurrent_mhla(q, k, v, mixing_matrix, chunk_size: int = 64, scale: Optional[float] = None,
                         initial_state: Optional[torch.Tensor] = None, output_final_state: bool = True):
    """Drop-in for the reference's token-recurrent form (mhla_nlp/fla/ops/mhla/naive.py:88-142), which the fla layer calls
    when T <= 64 (layers/mhla.py:247): same arguments, returns `(o, S)`.

    For T <= chunk_size (the only case the layer uses it for) the recurrence is exactly the single-chunk case of the chunk
    operator, which is what runs here (one HIP launch chain instead of a T-step Python loop).  Documented deviations, both
    from defects of the reference rather than from its intent: (1) beyond the first chunk the reference prepends a zero state
    and so reads every earlier chunk's state shifted by one (naive.py:124-127, 133); this function computes the chunk operator
    `naive_chunk_simple_mhla_fixed` instead; (2) the reference ignores `scale` (naive.py:101 overwrites it with K**-0.5) and
    `initial_state` only seeds the returned tensor `S`, never the output (naive.py:113-116) -- both reproduced: `scale` is
    ignored, and `S` is `initial_state` (or zeros) [B, H, K, V] in fp32, `None` when `output_final_state` is False."""
    if scale is not None and abs(float(scale) - q.shape[-1] ** -0.5) > 1e-12:
        warnings.warn("naive_recurrent_mhla ignores `scale` (the reference overwrites it with K**-0.5, naive.py:101)", stacklevel=2)
    if q.shape[1] > chunk_size:
        warnings.warn(f"naive_recurrent_mhla on T={q.shape[1]} > chunk_size={chunk_size} tokens runs the chunk operator; the reference's "
                      "recurrent form reads every earlier chunk's state shifted by one there (naive.py:124-133, not replicated)", stacklevel=2)
    o = mhla_causal(q, k, v, mixing_matrix, chunk_size, None)
    S = None
    if output_final_state:
        B, _, H, K = q.shape
        S = torch.zeros(B, H, K, v.shape[-1], dtype=torch.float32, device=q.device)
        if initial_state is not None:
            S = S + initial_state
    return o, S


# ------------------------------------------------------------------------------------------
# per-head RMSNorm x swish gate
# ------------------------------------------------------------------------------------------
class _RmsNormGate(torch.autograd.Function):
    @staticmethod
    @_device_guard
    def forward(ctx, x, g, weight, eps):
        lib = _lib.load()
        _require_gpu(x, g, weight)
        D = x.shape[-1]
        xc = x.contiguous()
        gc = g.contiguous().to(x.dtype) if g is not None else None
        wf = weight.detach().to(torch.float32).contiguous() if weight is not None else None
        rows = xc.numel() // D
        y = torch.empty_like(xc)
        rc = lib.mhla_rmsnorm_gate_fwd(xc.data_ptr(), D, gc.data_ptr() if gc is not None else None, D,
                                       wf.data_ptr() if wf is not None else None, y.data_ptr(), D, None, rows, D,
                                       float(eps), _dtype_code(xc), _stream())
        _lib.check(rc, "mhla_rmsnorm_gate_fwd")
        ctx.save_for_backward(xc, gc, wf)
        ctx.cfg = (float(eps), weight.dtype if weight is not None else None)
        return y

    @staticmethod
    @_device_guard
    def backward(ctx, dy):
        lib = _lib.load()
        xc, gc, wf = ctx.saved_tensors
        eps, wdtype = ctx.cfg
        D = xc.shape[-1]
        rows = xc.numel() // D
        dyc = dy.contiguous().to(xc.dtype)
        dx = torch.empty_like(xc)
        dg = torch.empty_like(xc) if gc is not None else None
        nrows = lib.mhla_rmsnorm_gate_dw_rows(rows)
        dwp = torch.empty((nrows, D), dtype=torch.float32, device=xc.device)
        rc = lib.mhla_rmsnorm_gate_bwd(xc.data_ptr(), D, gc.data_ptr() if gc is not None else None, D,
                                       wf.data_ptr() if wf is not None else None, dyc.data_ptr(), D, dx.data_ptr(), D,
                                       dg.data_ptr() if dg is not None else None, D, dwp.data_ptr(), rows, D, eps,
                                       _dtype_code(xc), _stream())
        _lib.check(rc, "mhla_rmsnorm_gate_bwd")
        dw = dwp.sum(0).to(wdtype) if wf is not None else None
        return dx, dg, dw, None


def rmsnorm_gate(x: torch.Tensor, g: Optional[torch.Tensor], weight: Optional[torch.Tensor],
                 eps: float = 1e-5) -> torch.Tensor:
    """y = x * rsqrt(mean(x^2, -1) + eps) * weight [* g * sigmoid(g)] over the last dim (<= 512).
    FusedRMSNormGated math (mhla_nlp/fla/modules/fused_norm_gate.py:77-99); with g=None it is Wan's
    per-head g_norm (wan/model.py:181-196)."""
    return _RmsNormGate.apply(x, g, weight, eps)
