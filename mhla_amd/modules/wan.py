"""Drop-in for Wan2.1's MHLA_Video_Uni (mhla_videogen/diffusion/model/wan/mhla_utils.py:158-365).

Same constructor, `forward(x, seq_lens, grid_sizes, freqs)` and state-dict keys.  What changes:
the reference concatenates q, k, v, q_rope, k_rope and materialises the block-major gather
(mhla_utils.py:317-326, ~1.9 GB per layer at 31.5k tokens); here the HIP operator reads the five
tensors where they lie through an int32 block-index map, and the per-head g_norm (x SiLU gate)
runs as one fused HIP kernel.
"""
from typing import Dict, Tuple

import torch
from torch import nn

from ..ops import lepe3d, mhla_blockmix, mhla_blockmix_rope, mhla_blockmix_wan, mhla_blockmix_wan_pro, qk_prologue, rmsnorm_gate, wan_pro_supported
from ..weights import block_index_3d
from .blockconv import BlockDistanceConv3D


class WanRMSNorm(nn.Module):
    """wan/model.py:181-196."""

    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.dim = dim
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        xf = x.float()
        return (xf * torch.rsqrt(xf.pow(2).mean(dim=-1, keepdim=True) + self.eps)).type_as(x) * self.weight


def rope_params(max_seq_len, dim, theta=10000):
    """wan/model.py:139-146."""
    assert dim % 2 == 0
    freqs = torch.outer(torch.arange(max_seq_len),
                        1.0 / torch.pow(theta, torch.arange(0, dim, 2).to(torch.float64).div(dim)))
    return torch.polar(torch.ones_like(freqs), freqs)


def wan_freqs(head_dim: int, max_seq_len: int = 1024) -> torch.Tensor:
    """The complex128 [1024, D/2] table the Wan model builds (wan/model.py:1932-1936)."""
    d = head_dim
    return torch.cat([rope_params(max_seq_len, d - 4 * (d // 6)), rope_params(max_seq_len, 2 * (d // 6)),
                      rope_params(max_seq_len, 2 * (d // 6))], dim=1)


def _rope_table(freqs: torch.Tensor, grid: Tuple[int, int, int], device) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos/sin [N, D/2] for the (f, h, w) raster -- the multiplier of rope_apply (mhla_utils.py:127-156),
    evaluated in fp64 like the reference and rounded once to fp32."""
    f, h, w = grid
    c = freqs.shape[1]
    fr = freqs.split([c - 2 * (c // 3), c // 3, c // 3], dim=1)
    mult = torch.cat([
        fr[0][:f].view(f, 1, 1, -1).expand(f, h, w, -1),
        fr[1][:h].view(1, h, 1, -1).expand(f, h, w, -1),
        fr[2][:w].view(1, 1, w, -1).expand(f, h, w, -1),
    ], dim=-1).reshape(f * h * w, -1)
    return mult.real.float().to(device), mult.imag.float().to(device)


def rope_apply(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """x: [B, N, H, D] fp32; consecutive channel pairs rotate by the token's angle."""
    B, N, H, D = x.shape
    seq = cos.shape[0]
    xs = x[:, :seq].reshape(B, seq, H, D // 2, 2)
    x0, x1 = xs[..., 0], xs[..., 1]
    c, s = cos[None, :, None, :], sin[None, :, None, :]
    y = torch.stack((x0 * c - x1 * s, x0 * s + x1 * c), dim=-1).reshape(B, seq, H, D)
    return y if seq == N else torch.cat([y, x[:, seq:]], dim=1)


class MHLA_Video_Uni(nn.Module):
    def __init__(self, dim, num_heads=8, dim_head=None, dropout=0.1, fixed_weight_value=None, qk_norm=True,
                 block_layout=(3, 5, 10), transform="linear", qkv_bias=False, eps=1e-6, is_gated=False,
                 is_lepe=False, **kwargs):
        super().__init__()
        dim_head = dim // num_heads      # the positional `dim_head` is ignored, as in the reference (:190)
        self.dim = dim
        self.num_heads = num_heads
        self.head_dim = dim_head
        self.q = nn.Linear(dim, dim)
        self.k = nn.Linear(dim, dim)
        self.v = nn.Linear(dim, dim)
        self.g = nn.Linear(dim, dim) if is_gated else None
        self.g_fn = nn.SiLU() if is_gated else None
        self.g_norm = WanRMSNorm(dim_head, eps=eps)
        self.is_gated = is_gated
        self.is_lepe = is_lepe
        self.norm_q = WanRMSNorm(dim, eps=eps) if qk_norm else nn.Identity()
        self.norm_k = WanRMSNorm(dim, eps=eps) if qk_norm else nn.Identity()
        self.out_norm = kwargs.get("out_rmsnorm", False)
        self.normalize_out = kwargs.get("normalize_out", True)
        self.blocks_layout = tuple(block_layout)
        self.num_blocks = self.blocks_layout[0] * self.blocks_layout[1] * self.blocks_layout[2]
        self.block_attn = BlockDistanceConv3D(blocks_layout=self.blocks_layout, transform=transform)
        self.lepe = nn.Conv3d(dim, dim, kernel_size=(3, 3, 3), stride=1, padding=(1, 1, 1), groups=dim) if is_lepe else None
        self.eps = eps
        self.o = nn.Linear(dim, dim)
        self.rope_after = kwargs.get("rope_after", False)
        self.power = kwargs.get("power", 1.0)
        self.without_rope = kwargs.get("without_rope", False)
        self._cache: Dict = {}
        if fixed_weight_value is not None:
            self._init_weights_with_fixed_value(fixed_weight_value)

    def _init_weights_with_fixed_value(self, value):
        for name, param in self.named_parameters():
            if "weight" in name:
                nn.init.constant_(param, value)
            elif "bias" in name and param is not None:
                nn.init.zeros_(param)

    @staticmethod
    def init_to_value(model, value=1.0):
        for name, param in model.named_parameters():
            if "weight" in name:
                nn.init.constant_(param, value)
            elif "bias" in name and param is not None:
                nn.init.zeros_(param)
        return model

    def _tables(self, grid, freqs, device):
        key = (grid, str(device), freqs.data_ptr())
        hit = self._cache.get(key)
        if hit is None:
            cos, sin = _rope_table(freqs.cpu() if freqs.is_cuda else freqs, grid, device)
            idx = block_index_3d(grid, self.blocks_layout).to(device)
            hit = (cos, sin, idx)
            self._cache = {key: hit}
        return hit

    def forward(self, x: torch.Tensor, seq_lens, grid_sizes, freqs) -> torch.Tensor:
        B, N, C = x.shape
        H, D = self.num_heads, self.head_dim
        g0 = grid_sizes[0].tolist() if torch.is_tensor(grid_sizes) else list(grid_sizes[0])
        grid = (int(g0[0]), int(g0[1]), int(g0[2]))                  # shared by the batch (mhla_utils.py:298)
        cos, sin, idx = self._tables(grid, freqs, x.device)
        if idx.numel() != N:
            raise ValueError(f"sequence length {N} != F*H*W = {idx.numel()} (no padding path, as in the reference)")
        q, k, v = self.q(x), self.k(x), self.v(x)
        # LePE (mhla_utils.py:283-285, 363-364): depthwise 3x3x3 conv over V on the raster token layout, added to the output
        v_lepe = v
        dtype = q.dtype
        W = self.block_attn.conv.weight
        fused = (not (torch.is_grad_enabled() and (x.requires_grad or W.requires_grad or self.q.weight.requires_grad))
                 and D % 8 == 0 and C <= 4096)
        if fused:
            # inference: norm + relu + eps in one kernel per tensor, rotation inside the operator's loads
            wq = self.norm_q.weight if isinstance(self.norm_q, WanRMSNorm) else None
            wk = self.norm_k.weight if isinstance(self.norm_k, WanRMSNorm) else None
            gate = self.g(x).reshape(B, N, H, D) if self.is_gated else None
            if wan_pro_supported(q.reshape(B, N, H, D), self.num_blocks):
                # the prologue inside the operator's loads: q, k, v stay the 16-bit projection outputs, no fp32 copies
                out = mhla_blockmix_wan_pro(q.reshape(B, N, H, D), k.reshape(B, N, H, D), v.reshape(B, N, H, D), wq, wk,
                                            getattr(self.norm_q, "eps", 0.0), W, cos, sin, self.g_norm.weight, self.g_norm.eps, gate,
                                            eps=self.eps, normalize=self.normalize_out, block_index=idx,
                                            qk_norm=isinstance(self.norm_q, WanRMSNorm)).reshape(B, N, C)
                if self.is_lepe:
                    out = lepe3d(v_lepe, self.lepe.weight, self.lepe.bias, grid, add=out)
                return self.o(out)
            q = qk_prologue(q, wq, getattr(self.norm_q, "eps", 0.0), self.eps).reshape(B, N, H, D)
            k = qk_prologue(k, wk, getattr(self.norm_k, "eps", 0.0), self.eps).reshape(B, N, H, D)
            # ... and the per-head g_norm (x SiLU gate) applied before the operator stores its output (:356-362)
            out = mhla_blockmix_wan(q, k, v.float().reshape(B, N, H, D), W, cos, sin, self.g_norm.weight, self.g_norm.eps,
                                    gate, dtype, eps=self.eps, normalize=self.normalize_out, block_index=idx).reshape(B, N, C)
            if self.is_lepe:
                out = lepe3d(v_lepe, self.lepe.weight, self.lepe.bias, grid, add=out)
            return self.o(out)
        elif D % 8 == 0 and C <= 2048:
            # training: norm + relu + eps in one kernel per tensor (and one for its backward); the rotation happens inside the
            # operator's kernels in both directions (mhla_blockmix_rope_fwd / _bwd): no q_rope / k_rope tensors, and the
            # operator's backward returns ONE gradient per tensor (rotated part turned back + normaliser part)
            wq = self.norm_q.weight if isinstance(self.norm_q, WanRMSNorm) else None
            wk = self.norm_k.weight if isinstance(self.norm_k, WanRMSNorm) else None
            q = qk_prologue(q, wq, getattr(self.norm_q, "eps", 0.0), self.eps).reshape(B, N, H, D)
            k = qk_prologue(k, wk, getattr(self.norm_k, "eps", 0.0), self.eps).reshape(B, N, H, D)
            out = mhla_blockmix_rope(q, k, v.float().reshape(B, N, H, D), W, cos, sin, eps=self.eps, normalize=self.normalize_out,
                                     block_index=idx)
        else:
            q, k, v = q.float(), k.float(), v.float()                     # mhla_utils.py:308
            q = torch.relu(self.norm_q(q)) + self.eps                     # :268-272
            k = torch.relu(self.norm_k(k)) + self.eps
            q, k, v = (t.reshape(B, N, H, D) for t in (q, k, v))
            q_rope, k_rope = rope_apply(q, cos, sin), rope_apply(k, cos, sin)   # :314
            if self.normalize_out:
                out = mhla_blockmix(q_rope, k_rope, v, W, eps=self.eps, q_den=q, k_den=k, block_index=idx)
            else:
                out = mhla_blockmix(q_rope, k_rope, v, W, eps=self.eps, normalize=False, block_index=idx)
        out = out.to(dtype)                                           # :356
        gate = self.g(x).reshape(B, N, H, D) if self.is_gated else None
        out = rmsnorm_gate(out, gate, self.g_norm.weight, self.g_norm.eps).reshape(B, N, C)   # :357-362
        if self.is_lepe:
            out = lepe3d(v_lepe, self.lepe.weight, self.lepe.bias, grid, add=out)
        return self.o(out)
