"""CPU oracle for the MHLA hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module, and only as the checker / the timed CPU baseline.
The product path (``mhla_amd``) never routes through it; it fails loudly when
the HIP library is missing.

This is an eager-PyTorch (CPU) restatement of what the reference computes on
the hot path, op for op, each function citing the reference lines it follows
(paths relative to ``/root/reference``).  The reference's arithmetic for this
path is all ``aten::matmul`` / ``aten::conv2d(1x1)``; no third-party kernel
carries it.

Parity pinning: the reference has no tests / golden vectors of its own
(SURVEY.md section 4).  This oracle is pinned against fixtures generated in the
build container by importing the reference's own files
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``), checked by
``tests/test_oracle_golden.py``.

Token layout convention used by every function here (and by the C ABI):
``q, k, v : [B, N, H, D]`` token-major, tokens in *block-major* order
(token ``n = m * S + s`` belongs to block ``m`` at intra-block offset ``s``).
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# A1  mixing-weight initialiser
# ----------------------------------------------------------------------------
def block_distance_weights(
    layout: Sequence[int],
    transform: str = "linear",
    local_thres: float = 1.5,
    exp_sigma: float = 3.0,
) -> torch.Tensor:
    """Initial mixing matrix ``W[M, M]`` (out block i, in block j), fp32.

    2-D grids follow ``mhla_dit/mhla/mhla.py:63-122`` (``BlockDistanceConv``),
    3-D grids ``mhla_videogen/diffusion/model/wan/mhla_utils.py:61-118``
    (``BlockDistanceConv3D``): block centres at ``idx + 0.5`` enumerated
    row-major over the grid, pairwise Euclidean distance, then the transform;
    every transform but ``gaussian`` is column-normalised (``mat / mat.sum(0)``).
    """
    grids = torch.meshgrid(*[torch.arange(n, dtype=torch.float32) + 0.5 for n in layout], indexing="ij")
    centres = torch.stack([g.reshape(-1) for g in grids], dim=-1)  # [M, ndim], row-major
    diff = centres[:, None, :] - centres[None, :, :]
    dist = torch.sqrt((diff * diff).sum(-1))
    if transform == "linear":
        mat = 1.0 - dist / dist.max()
        return mat / mat.sum(dim=0, keepdim=True)
    if transform == "cos":
        mat = torch.cos(dist / dist.max() * math.pi / 4)
        return mat / mat.sum(dim=0, keepdim=True)
    if transform == "exp":
        mat = torch.exp(-dist / exp_sigma)
        return mat / mat.sum(dim=0, keepdim=True)
    if transform == "gaussian":
        sigma = dist.max() / 3
        return torch.exp(-(dist ** 2) / (2 * sigma ** 2))
    if transform == "local":
        mat = (dist <= local_thres).float()
        return mat / mat.sum(dim=0, keepdim=True)
    raise ValueError(f"Unknown transform: {transform}")


def causal_mixing_init(L: int = 32) -> torch.Tensor:
    """``tril(ones(L, L)) / rowcount`` -- ``mhla_nlp/fla/layers/mhla.py:196-200``."""
    lower = torch.tril(torch.ones(L, L, dtype=torch.float32))
    return lower / (torch.arange(L, dtype=torch.float32).unsqueeze(1) + 1.0)


# ----------------------------------------------------------------------------
# A2  block-mix forward
# ----------------------------------------------------------------------------
def _to_blocks(t: torch.Tensor, M: int) -> torch.Tensor:
    """[B, N, H, D] (block-major tokens) -> [(B H), M, S, D]  (mhla.py:232-237)."""
    B, N, H, D = t.shape
    S = N // M
    return t.permute(0, 2, 1, 3).reshape(B * H, M, S, D)


def _from_blocks(t: torch.Tensor, B: int, H: int) -> torch.Tensor:
    """[(B H), M, S, D] -> [B, N, H, D]  (mhla.py:271)."""
    BH, M, S, D = t.shape
    return t.reshape(B, H, M * S, D).permute(0, 2, 1, 3)


def _mix(W: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """1x1 conv over the block axis: out[b, i, ...] = sum_j W[i, j] x[b, j, ...].

    This is exactly ``nn.Conv2d(M, M, 1, bias=False)`` applied to ``[bh, M, a, b]``
    (``mhla.py:46-57,124-134``): out-channel i, in-channel j.
    """
    return torch.einsum("ij,bjxy->bixy", W.to(x.dtype), x)


def blockmix_fwd(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    W: torch.Tensor,
    eps: float = 1e-6,
    q_den: Optional[torch.Tensor] = None,
    k_den: Optional[torch.Tensor] = None,
    normalize: bool = True,
    return_aux: bool = False,
):
    """Block-mixing MHLA operator, forward.

    Follows ``mhla_dit/mhla/mhla.py:262-268`` (identical in
    ``mhla_image_classification/models/modules/attention/mhla.py:275-282``) and
    the Wan variant ``mhla_videogen/diffusion/model/wan/mhla_utils.py:331-341``
    where the roped pair ``(q, k)`` feeds KV / the numerator and the un-roped
    pair ``(q_den, k_den)`` feeds the normaliser; ``normalize=False`` skips the
    division (``mhla_utils.py:340-341``).

    The normaliser quirk is reproduced as the reference has it: the 1x1 conv
    mixes ``z_j[s] = Q_j[s] . ksum_j`` over blocks *at the same intra-block
    offset s* (``mhla.py:265-266``).
    """
    B, N, H, D = q.shape
    M = W.shape[0]
    qb, kb, vb = (_to_blocks(t, M) for t in (q, k, v))
    kv = torch.matmul(kb.transpose(-2, -1), vb)          # [bh, M, D, D]    mhla.py:262
    g = _mix(W, kv)                                      #                  mhla.py:263
    num = torch.matmul(qb, g)                            # [bh, M, S, D]    mhla.py:268
    aux = {"kv": kv, "g": g}
    if normalize:
        qd = qb if q_den is None else _to_blocks(q_den, M)
        kd = kb if k_den is None else _to_blocks(k_den, M)
        k_sum = kd.transpose(-2, -1).sum(dim=-1, keepdim=True)   # [bh, M, D, 1]  mhla.py:265
        z = torch.matmul(qd, k_sum)                              # [bh, M, S, 1]
        normalizer = _mix(W, z) + eps                            #                mhla.py:266
        out = num / normalizer
        aux.update({"ksum": k_sum.squeeze(-1), "z": z.squeeze(-1), "n": normalizer.squeeze(-1)})
    else:
        out = num
    out = _from_blocks(out, B, H)
    return (out, aux) if return_aux else out


# ----------------------------------------------------------------------------
# A3  block-mix backward, closed form (validated against autograd in tests)
# ----------------------------------------------------------------------------
def blockmix_bwd(
    q, k, v, W, dout, eps: float = 1e-6, q_den=None, k_den=None, normalize: bool = True
) -> dict:
    """Hand-derived gradients of :func:`blockmix_fwd`.

    dP = dO/n; dn = -(dO.O)/n; dG_i = Q_i^T dP_i; dKV_j = sum_i W_ij dG_i;
    dz_j = sum_i W_ij dn_i; dQnum_i = dP_i G_i^T; dQden_j = dz_j (x) ksum_j;
    dksum_j = sum_s dz_j[s] Qden_j[s]; dKnum_j = V_j dKV_j^T; dKden_j = 1 dksum_j^T;
    dV_j = Knum_j dKV_j; dW_ij = sum_bh(<dG_i, KV_j> + sum_s dn_i[s] z_j[s]).
    When the denominator pair aliases the numerator pair the two parts add.
    """
    B, N, H, D = q.shape
    M = W.shape[0]
    split = q_den is not None
    out, aux = blockmix_fwd(q, k, v, W, eps, q_den, k_den, normalize, return_aux=True)
    qb, kb, vb, dob = (_to_blocks(t, M) for t in (q, k, v, dout))
    ob = _to_blocks(out, M)
    Wt = W.to(qb.dtype)
    if normalize:
        n = aux["n"].unsqueeze(-1)
        dP = dob / n
        dn = -(dob * ob).sum(-1, keepdim=True) / n           # [bh, M, S, 1]
    else:
        dP = dob
    dG = torch.matmul(qb.transpose(-2, -1), dP)               # [bh, M, D, D]
    dKV = torch.einsum("ij,bixy->bjxy", Wt, dG)
    dq_num = torch.matmul(dP, aux["g"].transpose(-2, -1))
    dk_num = torch.matmul(vb, dKV.transpose(-2, -1))
    dv = torch.matmul(kb, dKV)
    dW = torch.einsum("bixy,bjxy->ij", dG, aux["kv"])
    res = {}
    if normalize:
        qd = qb if not split else _to_blocks(q_den, M)
        dz = torch.einsum("ij,bixy->bjxy", Wt, dn)            # [bh, M, S, 1]
        ksum = aux["ksum"].unsqueeze(-2)                      # [bh, M, 1, D]
        dq_den = dz * ksum
        dksum = (dz * qd).sum(-2, keepdim=True)               # [bh, M, 1, D]
        dk_den = dksum.expand_as(kb)
        dW = dW + torch.einsum("bixy,bjxy->ij", dn, aux["z"].unsqueeze(-1))
        if split:
            res["dq_den"] = _from_blocks(dq_den, B, H)
            res["dk_den"] = _from_blocks(dk_den.contiguous(), B, H)
        else:
            dq_num = dq_num + dq_den
            dk_num = dk_num + dk_den
    res.update({
        "dq": _from_blocks(dq_num, B, H),
        "dk": _from_blocks(dk_num, B, H),
        "dv": _from_blocks(dv, B, H),
        "dW": dW,
    })
    return res


# ----------------------------------------------------------------------------
# A10  causal chunk-mixing forward
# ----------------------------------------------------------------------------
def causal_fwd(q, k, v, mix, chunk_size: int = 64, return_aux: bool = False):
    """Causal chunk-mixing MHLA operator, forward.

    Restates ``naive_chunk_simple_mhla_fixed``
    (``mhla_nlp/fla/ops/mhla/naive.py:39-82``): fp32 compute, ``q *= K**-0.5``,
    right-pad T to a multiple of ``chunk_size``, ``S_j = K_j^T V_j``,
    ``O_i = Q_i (sum_{j<i} m_ij S_j) + m_ii tril(Q_i K_i^T) V_i``; cast back.
    ``q, k: [B, T, H, K]``, ``v: [B, T, H, V]``, ``mix: [L, L]`` (or
    ``[L, L, 1, 1, 1, 1]``) with ``L >= ceil(T / chunk_size)``.
    """
    dtype = q.dtype
    mix = mix.reshape(mix.shape[0], mix.shape[1]).float()
    qf, kf, vf = (t.permute(0, 2, 1, 3).float() for t in (q, k, v))     # naive.py:39
    scale = qf.shape[-1] ** -0.5                                        # naive.py:42
    T = qf.shape[-2]
    C = chunk_size
    pad = (C - T % C) % C                                               # naive.py:46-51
    if pad:
        qf, kf, vf = (F.pad(t, (0, 0, 0, pad)) for t in (qf, kf, vf))
    B, H, T1, K = qf.shape
    n = T1 // C
    m = mix[:n, :n]                                                     # naive.py:55
    qc, kc, vc = (t.reshape(B, H, n, C, t.shape[-1]) for t in (qf, kf, vf))
    qc = qc * scale                                                     # naive.py:58
    S_all = torch.matmul(kc.transpose(-1, -2), vc)                      # [B,H,n,K,V]  naive.py:60-64
    tril = torch.tril(torch.ones(C, C, dtype=torch.float32))
    A = torch.matmul(qc, kc.transpose(-1, -2)) * tril                   # naive.py:71
    m_strict = torch.tril(m, diagonal=-1)
    P = torch.einsum("ij,bhjkv->bhikv", m_strict, S_all)                # naive.py:73-75
    o = torch.matmul(qc, P) + torch.diagonal(m).view(1, 1, n, 1, 1) * torch.matmul(A, vc)  # naive.py:77-78
    o = o.reshape(B, H, T1, -1).permute(0, 2, 1, 3)[:, :T].to(dtype)    # naive.py:82
    if return_aux:
        return o, {"S": S_all, "P": P, "A": A}
    return o


def causal_bwd(q, k, v, mix, dout, chunk_size: int = 64) -> dict:
    """Closed-form gradients of :func:`causal_fwd`."""
    mix2 = mix.reshape(mix.shape[0], mix.shape[1]).float()
    qf, kf, vf, dof = (t.permute(0, 2, 1, 3).float() for t in (q, k, v, dout))
    scale = qf.shape[-1] ** -0.5
    T = qf.shape[-2]
    C = chunk_size
    pad = (C - T % C) % C
    if pad:
        qf, kf, vf, dof = (F.pad(t, (0, 0, 0, pad)) for t in (qf, kf, vf, dof))
    B, H, T1, K = qf.shape
    n = T1 // C
    m = mix2[:n, :n]
    qc, kc, vc, doc = (t.reshape(B, H, n, C, t.shape[-1]) for t in (qf, kf, vf, dof))
    qs = qc * scale
    S_all = torch.matmul(kc.transpose(-1, -2), vc)
    tril = torch.tril(torch.ones(C, C, dtype=torch.float32))
    A = torch.matmul(qs, kc.transpose(-1, -2)) * tril
    ms = torch.tril(m, diagonal=-1)
    md = torch.diagonal(m).view(1, 1, n, 1, 1)
    P = torch.einsum("ij,bhjkv->bhikv", ms, S_all)
    dA = torch.matmul(doc, vc.transpose(-1, -2)) * tril
    dQs = torch.matmul(doc, P.transpose(-1, -2)) + md * torch.matmul(dA, kc)
    dP = torch.matmul(qs.transpose(-1, -2), doc)
    dS = torch.einsum("ij,bhikv->bhjkv", ms, dP)
    dK = torch.matmul(vc, dS.transpose(-1, -2)) + md * torch.matmul(dA.transpose(-1, -2), qs)
    dV = torch.matmul(kc, dS) + md * torch.matmul(A.transpose(-1, -2), doc)
    dm = torch.tril(torch.einsum("bhikv,bhjkv->ij", dP, S_all), diagonal=-1)
    dm = dm + torch.diag(torch.einsum("bhicv,bhicv->i", doc, torch.matmul(A, vc)))
    dmix = torch.zeros_like(mix2)
    dmix[:n, :n] = dm

    def back(t, ref):
        return t.reshape(B, H, T1, -1).permute(0, 2, 1, 3)[:, :T].to(ref.dtype)

    return {"dq": back(dQs * scale, q), "dk": back(dK, k), "dv": back(dV, v), "dmix": dmix.reshape(mix.shape)}


# ----------------------------------------------------------------------------
# A4 / A8 / A13  prologue pieces (feature map, norms, rotary)
# ----------------------------------------------------------------------------
def relu_eps(x: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """``relu(x) + eps`` -- ``mhla_dit/mhla/mhla.py:229-230``."""
    return torch.relu(x) + eps


def rms_norm(x: torch.Tensor, weight: Optional[torch.Tensor], eps: Optional[float]) -> torch.Tensor:
    """RMSNorm over the last dim in fp32.

    ``nn.RMSNorm(dim)`` semantics for ``q_norm/k_norm`` (``mhla.py:166-167``;
    ``eps=None`` -> ``torch.finfo(x.dtype).eps``) and ``WanRMSNorm``
    (``mhla_videogen/diffusion/model/wan/model.py:181-196``).
    """
    if eps is None:
        eps = torch.finfo(x.dtype).eps
    xf = x.float()
    y = xf * torch.rsqrt(xf.pow(2).mean(dim=-1, keepdim=True) + eps)
    y = y.type_as(x)
    return y if weight is None else y * weight


def rms_norm_swish_gate(x, g, weight, eps: float = 1e-5) -> torch.Tensor:
    """``FusedRMSNormGated`` forward math -- ``mhla_nlp/fla/modules/fused_norm_gate.py:77-99``:
    ``y = x * rsqrt(mean(x^2) + eps) * w * g * sigmoid(g)`` in fp32, cast to x.dtype."""
    xf, gf = x.float(), g.float()
    rstd = 1.0 / torch.sqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    y = xf * rstd
    if weight is not None:
        y = y * weight.float()
    return (y * gf * torch.sigmoid(gf)).to(x.dtype)


def wan_rope_params(max_seq_len: int, dim: int, theta: float = 10000.0) -> torch.Tensor:
    """``rope_params`` -- ``mhla_videogen/diffusion/model/wan/model.py:139-146`` (complex128)."""
    freqs = torch.outer(
        torch.arange(max_seq_len), 1.0 / torch.pow(theta, torch.arange(0, dim, 2).to(torch.float64).div(dim))
    )
    return torch.polar(torch.ones_like(freqs), freqs)


def wan_freqs(head_dim: int, max_seq_len: int = 1024) -> torch.Tensor:
    """The ``[1024, D/2]`` complex table -- ``wan/model.py:1932-1936``."""
    d = head_dim
    return torch.cat(
        [
            wan_rope_params(max_seq_len, d - 4 * (d // 6)),
            wan_rope_params(max_seq_len, 2 * (d // 6)),
            wan_rope_params(max_seq_len, 2 * (d // 6)),
        ],
        dim=1,
    )


def wan_rope_apply(x: torch.Tensor, grid: Tuple[int, int, int], freqs: torch.Tensor) -> torch.Tensor:
    """3-axis complex RoPE in fp64, shared grid -- ``wan/mhla_utils.py:127-156``.

    ``x: [B, N, H, D]`` raster token order ``(f h w)``; consecutive channel
    pairs ``(2i, 2i+1)`` form the complex number; the frequency table is split
    ``[c - 2(c//3), c//3, c//3]`` over (f, h, w), ``c = D/2``.
    """
    f, h, w = grid
    B, N, Hh, D = x.shape
    c = D // 2
    fr = freqs.split([c - 2 * (c // 3), c // 3, c // 3], dim=1)
    seq = f * h * w
    mult = torch.cat(
        [
            fr[0][:f].view(f, 1, 1, -1).expand(f, h, w, -1),
            fr[1][:h].view(1, h, 1, -1).expand(f, h, w, -1),
            fr[2][:w].view(1, 1, w, -1).expand(f, h, w, -1),
        ],
        dim=-1,
    ).reshape(seq, 1, -1)
    xc = torch.view_as_complex(x[:, :seq].to(torch.float64).reshape(B, seq, Hh, c, 2))
    y = torch.view_as_real(xc * mult).flatten(3)
    if seq < N:
        y = torch.cat([y, x[:, seq:].to(torch.float64)], dim=1)
    return y.float()


def neox_rotary(x: torch.Tensor, base: float = 10000.0, offset: int = 0, positions: Optional[torch.Tensor] = None) -> torch.Tensor:
    """NeoX half-rotation rotary, non-interleaved -- ``rotary_embedding_ref``
    (``mhla_nlp/fla/modules/rotary.py:20-32``) with the cos/sin table of
    ``RotaryEmbedding._update_cos_sin_cache`` (``rotary.py:415-431``): fp32
    ``inv_freq = base^(-2i/D)``, ``freqs = outer(t, inv_freq)``, cos/sin cast to
    the dtype of ``x``.  ``x: [B, T, H, D]``.  ``positions`` ([T] integer) replaces ``offset + arange(T)``: the per-token
    positions of a packed (varlen) batch, which restart at every sequence start when ``cu_seqlens`` is given
    (``rotary.py:68-72``)."""
    B, T, H, D = x.shape
    inv_freq = 1.0 / (base ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    t = torch.arange(offset, offset + T, dtype=torch.float32) if positions is None else positions.to(torch.float32)
    fr = torch.outer(t, inv_freq)
    cos = torch.cos(fr).to(x.dtype)[None, :, None, :]
    sin = torch.sin(fr).to(x.dtype)[None, :, None, :]
    cos = torch.cat([cos, cos], dim=-1)
    sin = torch.cat([sin, sin], dim=-1)
    x1, x2 = x.chunk(2, dim=-1)
    return x * cos + torch.cat((-x2, x1), dim=-1) * sin


# ----------------------------------------------------------------------------
# layout helpers (A7 / A8): raster <-> block-major token permutations
# ----------------------------------------------------------------------------
def block_index_2d(pieces: int, block_len: int) -> torch.Tensor:
    """Gather map raster -> block-major for a square image of
    ``(pieces*block_len)^2`` tokens: ``idx[m*S + s]`` = raster token index.
    Same permutation as ``rearrange_patches``
    (``mhla_dit/piecewise_patchembed.py:47-63``)."""
    side = pieces * block_len
    r = torch.arange(side * side).reshape(pieces, block_len, pieces, block_len)
    return r.permute(0, 2, 1, 3).reshape(-1)


def block_index_3d(grid: Tuple[int, int, int], layout: Tuple[int, int, int]) -> torch.Tensor:
    """Gather map raster ``(f h w)`` -> block-major
    ``(fb hb wb) (p1 p2 p3)`` -- the rearrange at ``wan/mhla_utils.py:317-326``."""
    f, h, w = grid
    fb, hb, wb = layout
    p1, p2, p3 = f // fb, h // hb, w // wb
    r = torch.arange(f * h * w).reshape(fb, p1, hb, p2, wb, p3)
    return r.permute(0, 2, 4, 1, 3, 5).reshape(-1)


# ----------------------------------------------------------------------------
# module-level restatements (A4/A5/A8/A13) -- used to pin the drop-in modules
# ----------------------------------------------------------------------------
def dit_module_forward(sd: dict, x: torch.Tensor, heads: int, block_size: int, embed_len: int,
                       eps: float = 1e-6, qk_norm: bool = False, lepe_k: int = 3) -> torch.Tensor:
    """``MHLA4DiT.forward`` in eval mode -- ``mhla_dit/mhla/mhla.py:251-275``
    (``MHLA_Normed_Torch``: same with ``lepe_k=5``, ``qk_norm=True``).
    ``x: [B, M, S, C]``; ``sd``: the module's state dict."""
    B, M, S, C = x.shape
    x = F.layer_norm(x, (C,), sd["norm.weight"], sd["norm.bias"])                  # :252
    qkv = F.linear(x, sd["to_qkv.weight"], sd.get("to_qkv.bias"))                  # :245
    q, k, v = qkv.chunk(3, dim=-1)
    inner = q.shape[-1]
    D = inner // heads
    pl = int((embed_len // block_size) ** 0.5)
    bl = int(block_size ** 0.5)
    img = v.reshape(B, pl, pl, bl, bl, inner).permute(0, 5, 1, 3, 2, 4).reshape(B, inner, pl * bl, pl * bl)
    lepe = F.conv2d(img, sd["lepe.weight"], sd["lepe.bias"], padding=lepe_k // 2, groups=inner)  # :246
    lepe = lepe.reshape(B, inner, pl, bl, pl, bl).permute(0, 2, 4, 3, 5, 1).reshape(B, M, S, inner)
    if qk_norm:                                                                    # :226-227
        q = rms_norm(q, sd["q_norm.weight"], None)
        k = rms_norm(k, sd["k_norm.weight"], None)
    q, k = relu_eps(q, eps), relu_eps(k, eps)                                      # :229-230
    W = sd["piece_attn.conv.weight"].reshape(M, M)
    o = blockmix_fwd(q.reshape(B, M * S, heads, D), k.reshape(B, M * S, heads, D),
                     v.reshape(B, M * S, heads, D), W, eps)                        # :262-268
    o = o.reshape(B, M, S, inner) + lepe                                           # :271-273
    return F.linear(o, sd["to_out.0.weight"], sd["to_out.0.bias"])                 # :275


def wan_module_forward(sd: dict, x: torch.Tensor, grid: Tuple[int, int, int], freqs: torch.Tensor,
                       heads: int, layout=(3, 5, 10), eps: float = 1e-6, normalize_out: bool = True,
                       is_gated: bool = False) -> torch.Tensor:
    """``MHLA_Video_Uni.forward`` -- ``wan/mhla_utils.py:292-365`` (no LePE)."""
    B, N, C = x.shape
    D = C // heads
    q = F.linear(x, sd["q.weight"], sd["q.bias"]).float()
    k = F.linear(x, sd["k.weight"], sd["k.bias"]).float()
    v = F.linear(x, sd["v.weight"], sd["v.bias"]).float()
    q = relu_eps(rms_norm(q, sd["norm_q.weight"], eps), eps)                       # :268-272
    k = relu_eps(rms_norm(k, sd["norm_k.weight"], eps), eps)
    q, k, v = (t.reshape(B, N, heads, D) for t in (q, k, v))
    q_rope, k_rope = wan_rope_apply(q, grid, freqs), wan_rope_apply(k, grid, freqs)  # :314
    idx = block_index_3d(grid, layout)
    M = layout[0] * layout[1] * layout[2]
    W = sd["block_attn.conv.weight"].reshape(M, M)
    gq, gk, gv, gqr, gkr = (t[:, idx] for t in (q, k, v, q_rope, k_rope))          # :317-326
    o = blockmix_fwd(gqr, gkr, gv, W, eps, q_den=gq, k_den=gk, normalize=normalize_out)  # :331-341
    out = torch.empty_like(o)
    out[:, idx] = o                                                                # :343-354
    out = out.to(x.dtype)
    out = rms_norm(out, sd["g_norm.weight"], eps).reshape(B, N, C)                 # :357-362
    if is_gated:
        out = out * F.silu(F.linear(x, sd["g.weight"], sd["g.bias"]))
    return F.linear(out, sd["o.weight"], sd["o.bias"])                             # :365


def wan_variant_forward(kind: str, sd: dict, x: torch.Tensor, grid: Tuple[int, int, int], freqs: torch.Tensor,
                        heads: int, layout=(3, 5, 10), eps: float = 1e-6, normalize_out: bool = True) -> torch.Tensor:
    """The five older Wan MHLA classes (``wan/model.py``): identical operator, different epilogues --
    ``mhla`` / ``mhla_nope``: ``out_rmsnorm(o(out))`` (:1389, :804); ``gated_mhla``: ``o(g_norm_fulldim(out) * silu(g(x)))``
    (:614-618); ``mhla_lepe``: ``out_rmsnorm(o(out + lepe))`` (:1203); ``gated_mhla_lepe``:
    ``o(g_norm_perhead(out) * silu(g(x)) + lepe)`` (:1003-1007).  ``out_rmsnorm`` is present in ``sd`` only when enabled."""
    B, N, C = x.shape
    D = C // heads
    q = F.linear(x, sd["q.weight"], sd["q.bias"])
    k = F.linear(x, sd["k.weight"], sd["k.bias"])
    v = F.linear(x, sd["v.weight"], sd["v.bias"])
    lepe = None
    if "lepe.weight" in sd:                                                        # :891-892 -- conv over v as a video
        f, h, w = grid
        vid = v.reshape(B, f, h, w, C).permute(0, 4, 1, 2, 3)
        lepe = F.conv3d(vid, sd["lepe.weight"], sd["lepe.bias"], padding=1, groups=C).permute(0, 2, 3, 4, 1).reshape(B, N, C)
    qf = relu_eps(rms_norm(q.float(), sd["norm_q.weight"], eps), eps).reshape(B, N, heads, D)
    kf = relu_eps(rms_norm(k.float(), sd["norm_k.weight"], eps), eps).reshape(B, N, heads, D)
    vf = v.float().reshape(B, N, heads, D)
    q_rope, k_rope = wan_rope_apply(qf, grid, freqs), wan_rope_apply(kf, grid, freqs)
    idx = block_index_3d(grid, layout)
    M = layout[0] * layout[1] * layout[2]
    W = sd["block_attn.conv.weight"].reshape(M, M)
    o = blockmix_fwd(q_rope[:, idx], k_rope[:, idx], vf[:, idx], W, eps, q_den=qf[:, idx], k_den=kf[:, idx], normalize=normalize_out)
    out = torch.empty_like(o)
    out[:, idx] = o
    out = out.to(x.dtype).reshape(B, N, C)
    lin_o = lambda t: F.linear(t, sd["o.weight"], sd["o.bias"])
    post = (lambda t: rms_norm(t, sd["out_rmsnorm.weight"], eps)) if "out_rmsnorm.weight" in sd else (lambda t: t)
    if kind in ("mhla", "mhla_nope"):
        return post(lin_o(out))
    if kind == "mhla_lepe":
        return post(lin_o(out + lepe))
    gate = F.silu(F.linear(x, sd["g.weight"], sd["g.bias"]))
    if kind == "gated_mhla":
        return lin_o(rms_norm(out, sd["g_norm.weight"], eps) * gate)
    if kind == "gated_mhla_lepe":
        normed = rms_norm(out.reshape(B, N, heads, D), sd["g_norm.weight"], eps).reshape(B, N, C)
        return lin_o(normed * gate + lepe)
    raise ValueError(kind)


def short_conv(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, activation: Optional[str] = "silu",
               cu_seqlens=None, initial_state: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``ShortConvolution.forward`` -- ``mhla_nlp/fla/modules/convolution.py:889-964`` (torch twin of the Triton kernel):
    depthwise causal conv, y[t] = act(sum_i w[d, i] x[t - (W - 1) + i] + b[d]); before a sequence start the inputs are the last
    W - 1 columns of ``initial_state [N, D, W]`` (zeros without one); sequences of a packed batch (``cu_seqlens``) do not see
    each other.  x: [B, T, D], weight: [D, 1, W] or [D, W]."""
    w = weight.reshape(weight.shape[0], -1)
    D, W = w.shape
    B, T, _ = x.shape
    bounds = [(b, 0, T) for b in range(B)] if cu_seqlens is None else [(0, int(cu_seqlens[i]), int(cu_seqlens[i + 1])) for i in range(len(cu_seqlens) - 1)]
    y = torch.zeros_like(x)
    for n, (b, t0, t1) in enumerate(bounds):
        for t in range(t0, t1):
            acc = torch.zeros(D, dtype=x.dtype)
            for i in range(W):
                src = t - (W - 1) + i
                if src >= t0:
                    acc = acc + w[:, i] * x[b, src]
                elif initial_state is not None:
                    acc = acc + w[:, i] * initial_state[n, :, W + (src - t0)]
            y[b, t] = acc + (bias if bias is not None else 0)
    return F.silu(y) if activation is not None else y


def fla_layer_forward(sd: dict, x: torch.Tensor, heads: int, head_k: int, head_v: int,
                      norm_eps: float = 1e-5, chunk_size: int = 64, attention_mask: Optional[torch.Tensor] = None,
                      num_kv_heads: Optional[int] = None, feature_map: str = "relu", use_output_gate: bool = True,
                      gate_fn: str = "swish", use_short_conv: bool = False,
                      position_offsets: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``MHLA.forward`` (fla layer, no short conv, no cache) -- ``mhla_nlp/fla/layers/mhla.py:226-365``.  Defaults: the shipped
    configuration (``feature_map='relu'``, fused swish gate).  Options restated from the same file: grouped k / v heads
    (``num_kv_heads``, :290-292), ``feature_map`` in relu / elu (= elu + 1, :130-134) / identity, ``use_output_gate=False``
    (plain per-head RMSNorm, :357-358), a ``gate_fn`` other than swish (RMSNorm, then ``o * gate_fn(g)``, :355-356),
    ``use_short_conv`` (q / k / v through ``short_conv`` after their projections, :258-279), ``position_offsets`` [B] (the
    per-sequence rotary offsets of padded decoding, ``prepare_lens_from_mask(mask) - q_len``, :305-309).
    With a 0/1 ``attention_mask`` [B, T] the batch is
    unpadded into ONE packed sequence (:253-256), rotary positions restart per sequence (cu_seqlens, :311), the operator runs
    over the whole packed sequence (it ignores cu_seqlens: cross-sequence leakage, as in the reference, :330-336) and the
    result is padded back with zeros (:362-363)."""
    opts = dict(num_kv_heads=num_kv_heads, feature_map=feature_map, use_output_gate=use_output_gate, gate_fn=gate_fn,
                use_short_conv=use_short_conv)
    if attention_mask is not None:
        Bm, Tm, C = x.shape
        keep = attention_mask.flatten().nonzero().flatten()
        lens = attention_mask.sum(-1)
        offs = position_offsets if position_offsets is not None else torch.zeros_like(lens)
        pos = torch.cat([torch.arange(int(n)) + int(o) for n, o in zip(lens, offs)])
        cu = F.pad(lens.cumsum(0), (1, 0))
        y = _fla_layer_core(sd, x.reshape(Bm * Tm, C)[keep].unsqueeze(0), heads, head_k, head_v, norm_eps, chunk_size, pos,
                            cu_seqlens=cu, **opts)
        out = y.new_zeros(Bm * Tm, y.shape[-1])
        out[keep] = y.squeeze(0)
        return out.reshape(Bm, Tm, -1)
    return _fla_layer_core(sd, x, heads, head_k, head_v, norm_eps, chunk_size, None, **opts)


def _fla_layer_core(sd, x, heads, head_k, head_v, norm_eps, chunk_size, positions, num_kv_heads=None, feature_map="relu",
                    use_output_gate=True, gate_fn="swish", use_short_conv=False, cu_seqlens=None):
    B, T, C = x.shape
    kvh = heads if num_kv_heads is None else num_kv_heads
    groups = heads // kvh
    # :237 -- clamp(...).tril() on the [L, L, 1, 1, 1, 1] parameter: tril acts on the trailing 1x1 dims, a no-op;
    # the op reads only j <= i anyway
    mix = torch.clamp(sd["mixing_matrix"], 1e-5, 1).tril().reshape(sd["mixing_matrix"].shape[0], -1)
    q, k, v = F.linear(x, sd["q_proj.weight"]), F.linear(x, sd["k_proj.weight"]), F.linear(x, sd["v_proj.weight"])   # :281-283
    if use_short_conv:                                                             # :258-279
        q = short_conv(q, sd["q_conv1d.weight"], sd.get("q_conv1d.bias"), "silu", cu_seqlens)
        k = short_conv(k, sd["k_conv1d.weight"], sd.get("k_conv1d.bias"), "silu", cu_seqlens)
        v = short_conv(v, sd["v_conv1d.weight"], sd.get("v_conv1d.bias"), "silu", cu_seqlens)
    q, k, v = q.reshape(B, T, heads, head_k), k.reshape(B, T, kvh, head_k), v.reshape(B, T, kvh, head_v)   # :289-295
    if groups > 1:                                                                 # :290-292  '(h d) -> (h g) d'
        k = k.repeat_interleave(groups, dim=2)
        v = v.repeat_interleave(groups, dim=2)
    fmap = {"relu": torch.relu, "elu": lambda t: F.elu(t) + 1, "identity": lambda t: t}[feature_map]   # :130-142
    q, k = fmap(q), fmap(k)                                                        # :297-299
    q, k = neox_rotary(q, positions=positions), neox_rotary(k, positions=positions)   # :311
    o = causal_fwd(q, k, v, mix, chunk_size)                                       # :330-336
    if use_output_gate and gate_fn == "swish":                                     # fused norm x swish gate, :351-354
        g = F.linear(x, sd["g_proj.weight"]).reshape(B, T, heads, head_v)
        o = rms_norm_swish_gate(o, g, sd["g_norm_swish_gate.weight"], norm_eps)
        o = o.reshape(B, T, heads * head_v)
    else:
        o = rms_norm(o, sd["g_norm.weight"], norm_eps).reshape(B, T, heads * head_v)   # :355-358
        if use_output_gate:
            act = {"sigmoid": torch.sigmoid, "silu": F.silu, "relu": F.relu, "gelu": F.gelu}[gate_fn]
            o = o * act(F.linear(x, sd["g_proj.weight"]))
    return F.linear(o, sd["o_proj.weight"])                                        # :361
