"""Drop-ins for the five older Wan2.1 MHLA self-attention classes (mhla_videogen/diffusion/model/wan/model.py:428-618, 621-804,
808-1007, 1010-1203, 1205-1389) and the registry that maps the YAML's `attn_type` keys onto them (model.py:1592-1605).  The
shipped configuration selects `mhla_uni` (modules/wan.py); these are its predecessors and share its operator exactly -- the
block-mixing operator on the roped numerator pair / un-roped normaliser pair, tokens gathered through the block index -- and
differ only in their parameters and in what happens after the operator:

  key                 class                    after the operator                                        extra parameters
  "mhla"              MHLA_Video               out_rmsnorm(o(out))                          :1387-1389    out_rmsnorm [dim] if out_rmsnorm
  "mhla_nope"         MHLA_Video_Nope          out_rmsnorm(o(out))   (applies rope too)     :802-804      out_rmsnorm [dim] if out_rmsnorm
  "gated_mhla"        Gated_MHLA_Video         o(g_norm_fulldim(out) * SiLU(g(x)))          :614-618      g, g_norm [dim]
  "mhla_lepe"         MHLA_Video_LePE          out_rmsnorm(o(out + lepe(v)))                :1199-1203    lepe (Conv3d 3x3x3 depthwise), out_rmsnorm
  "gated_mhla_lepe"   Gated_MHLA_Video_LePE    o(g_norm_perhead(out) * SiLU(g(x)) + lepe(v)) :1003-1007   g, g_norm [dim_head], lepe

One base class holds the shared prologue + operator (HIP: `qk_prologue` with the rotated copy, `mhla_blockmix` /
`mhla_blockmix_rope` under no_grad, `lepe3d`, `rmsnorm_gate` for the per-head norm); the full-dim norms of the epilogues are
stock elementwise PyTorch (they run over `dim` = 1536 channels, outside the operator).  Constructors, `forward(x, seq_lens,
grid_sizes, freqs)` and state-dict keys are the reference's."""
from typing import Dict

import torch
import torch.nn.functional as F
from torch import nn

from ..ops import lepe3d, mhla_blockmix, mhla_blockmix_rope, qk_prologue, rmsnorm_gate
from ..weights import block_index_3d
from .blockconv import BlockDistanceConv3D
from .wan import MHLA_Video_Uni, WanRMSNorm, _rope_table, rope_apply


class _WanMHLAVariant(nn.Module):
    gated = False          # g / g_fn / g_norm present
    gnorm_per_head = False  # g_norm over dim_head (Gated_MHLA_Video_LePE) instead of dim (Gated_MHLA_Video)
    has_lepe = False
    has_out_rmsnorm = True  # the `out_rmsnorm` kwarg builds a module (the gated classes read the kwarg and ignore it)

    def __init__(self, dim, num_heads=8, dim_head=None, dropout=0.1, fixed_weight_value=None, qk_norm=True,
                 block_layout=(3, 5, 10), transform="linear", qkv_bias=False, eps=1e-6, **kwargs):
        super().__init__()
        if dim_head is None:
            dim_head = dim // num_heads
        self.dim = dim
        self.num_heads = num_heads
        self.head_dim = dim_head                   # kept as given (the host passes its window_size here: model.py:1719-1728)
        self._D = dim // num_heads                 # the head dim the tensors actually have (rearrange with h = num_heads)
        self.q = nn.Linear(dim, dim)
        self.k = nn.Linear(dim, dim)
        self.v = nn.Linear(dim, dim)
        if self.gated:
            self.g = nn.Linear(dim, dim)
            self.g_fn = nn.SiLU()
            self.g_norm = WanRMSNorm(dim_head if self.gnorm_per_head else dim, eps=eps)
        self.norm_q = WanRMSNorm(dim, eps=eps) if qk_norm else nn.Identity()
        self.norm_k = WanRMSNorm(dim, eps=eps) if qk_norm else nn.Identity()
        self.out_norm = kwargs.get("out_rmsnorm", False)
        if self.has_out_rmsnorm:
            self.out_rmsnorm = WanRMSNorm(dim, eps=eps) if self.out_norm else nn.Identity()
        self.normalize_out = kwargs.get("normalize_out", True)
        self.blocks_layout = tuple(block_layout)
        self.num_blocks = self.blocks_layout[0] * self.blocks_layout[1] * self.blocks_layout[2]
        self.block_attn = BlockDistanceConv3D(blocks_layout=self.blocks_layout, transform=transform)
        if self.has_lepe:
            self.lepe = nn.Conv3d(dim, dim, kernel_size=(3, 3, 3), stride=1, padding=(1, 1, 1), groups=dim)
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
            hit = (cos, sin, block_index_3d(grid, self.blocks_layout).to(device))
            self._cache = {key: hit}
        return hit

    def _operator(self, x, grid_sizes, freqs):
        """Shared by the five classes (e.g. model.py:558-612): projections, fp32, norm + relu + eps, rope, the block-mixing operator
        on raster tokens through the block index.  Returns (out [B, N, C] in the projection dtype, v [B, N, C], grid)."""
        B, N, C = x.shape
        H, D = self.num_heads, self._D
        g0 = grid_sizes[0].tolist() if torch.is_tensor(grid_sizes) else list(grid_sizes[0])
        grid = (int(g0[0]), int(g0[1]), int(g0[2]))
        cos, sin, idx = self._tables(grid, freqs, x.device)
        if idx.numel() != N:
            raise ValueError(f"sequence length {N} != F*H*W = {idx.numel()} (no padding path, as in the reference)")
        q, k, v = self.q(x), self.k(x), self.v(x)
        dtype = q.dtype
        W = self.block_attn.conv.weight
        wq = self.norm_q.weight if isinstance(self.norm_q, WanRMSNorm) else None
        wk = self.norm_k.weight if isinstance(self.norm_k, WanRMSNorm) else None
        eq, ek = getattr(self.norm_q, "eps", 0.0), getattr(self.norm_k, "eps", 0.0)
        v4 = v.float().reshape(B, N, H, D)
        no_grad = not (torch.is_grad_enabled() and (x.requires_grad or W.requires_grad or self.q.weight.requires_grad))
        if D % 8 == 0 and C <= 2048 and no_grad:
            qn = qk_prologue(q, wq, eq, self.eps).reshape(B, N, H, D)          # norm + relu + eps, one kernel per tensor
            kn = qk_prologue(k, wk, ek, self.eps).reshape(B, N, H, D)
            out = mhla_blockmix_rope(qn, kn, v4, W, cos, sin, eps=self.eps, normalize=self.normalize_out, block_index=idx)
        else:
            if D % 8 == 0 and C <= 2048:
                qn, q_rope = qk_prologue(q, wq, eq, self.eps, rope=(cos, sin), head_dim=D)
                kn, k_rope = qk_prologue(k, wk, ek, self.eps, rope=(cos, sin), head_dim=D)
                qn, kn, q_rope, k_rope = (t.reshape(B, N, H, D) for t in (qn, kn, q_rope, k_rope))
            else:
                qn = (torch.relu(self.norm_q(q.float())) + self.eps).reshape(B, N, H, D)
                kn = (torch.relu(self.norm_k(k.float())) + self.eps).reshape(B, N, H, D)
                q_rope, k_rope = rope_apply(qn, cos, sin), rope_apply(kn, cos, sin)
            if self.normalize_out:
                out = mhla_blockmix(q_rope, k_rope, v4, W, eps=self.eps, q_den=qn, k_den=kn, block_index=idx)
            else:
                out = mhla_blockmix(q_rope, k_rope, v4, W, eps=self.eps, normalize=False, block_index=idx)
        return out.to(dtype).reshape(B, N, C), v, grid

    def _lepe(self, v, grid, add):
        return lepe3d(v, self.lepe.weight, self.lepe.bias, grid, add=add)


class MHLA_Video(_WanMHLAVariant):
    """model.py:1205-1389 (`attn_type: mhla`)."""

    def forward(self, x, seq_lens, grid_sizes, freqs):
        out, _, _ = self._operator(x, grid_sizes, freqs)
        return self.out_rmsnorm(self.o(out))                                   # :1389


class MHLA_Video_Nope(MHLA_Video):
    """model.py:621-804 (`attn_type: mhla_nope`): despite the name its forward applies the rope (:740) -- identical to MHLA_Video."""


class Gated_MHLA_Video(_WanMHLAVariant):
    """model.py:428-618 (`attn_type: gated_mhla`): RMSNorm over the full channel dim, times SiLU(g(x))."""
    gated, has_out_rmsnorm = True, False

    def forward(self, x, seq_lens, grid_sizes, freqs):
        out, _, _ = self._operator(x, grid_sizes, freqs)
        return self.o(self.g_norm(out) * self.g_fn(self.g(x)))                  # :614-618


class MHLA_Video_LePE(_WanMHLAVariant):
    """model.py:1010-1203 (`attn_type: mhla_lepe`): depthwise 3x3x3 conv of v (as a video) added before the output projection."""
    has_lepe = True

    def forward(self, x, seq_lens, grid_sizes, freqs):
        out, v, grid = self._operator(x, grid_sizes, freqs)
        return self.out_rmsnorm(self.o(self._lepe(v, grid, out)))               # :1199-1203


class Gated_MHLA_Video_LePE(_WanMHLAVariant):
    """model.py:808-1007 (`attn_type: gated_mhla_lepe`): per-head RMSNorm x SiLU gate, then + LePE."""
    gated, gnorm_per_head, has_lepe, has_out_rmsnorm = True, True, True, False

    def forward(self, x, seq_lens, grid_sizes, freqs):
        out, v, grid = self._operator(x, grid_sizes, freqs)
        B, N, C = out.shape
        gate = self.g(x).reshape(B, N, self.num_heads, self._D)
        out = rmsnorm_gate(out.reshape(B, N, self.num_heads, self._D), gate, self.g_norm.weight, self.g_norm.eps)   # :1003-1005
        return self.o(self._lepe(v, grid, out.reshape(B, N, C)))                # :1007


# model.py:1592-1605 (the softmax / plain linear-attention entries of that table are not MHLA and not built here)
WAN_SELFATTENTION_CLASSES = {
    "mhla": MHLA_Video,
    "gated_mhla": Gated_MHLA_Video,
    "mhla_nope": MHLA_Video_Nope,
    "mhla_lepe": MHLA_Video_LePE,
    "gated_mhla_lepe": Gated_MHLA_Video_LePE,
    "mhla_uni": MHLA_Video_Uni,
}
