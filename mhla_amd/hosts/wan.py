"""A thin shell of the Wan2.1 transformer block around the `MHLA_Video_Uni` drop-in: the adaLN-style
modulation, the gated residuals, the text cross-attention and the FFN of `WanAttentionBlock`
(mhla_videogen/diffusion/model/wan/model.py:1605-1766) with the reference's parameter names (`norm1`, `self_attn.*`, `norm3`,
`cross_attn.{q,k,v,o,norm_q,norm_k}`, `norm2`, `ffn.0`, `ffn.2`, `modulation`), so that block-level numbers for the C4
configuration can be produced without any reference Python.  Stock PyTorch besides the self-attention module; the
cross-attention uses `scaled_dot_product_attention` where the reference calls flash-attn (:1541).  The optional
`skip_ffn` conv branches (`ffn_type != "mlp"`) are not built.  Not part of the hot path."""
import torch
import torch.nn.functional as F
from torch import nn

from ..modules import MHLA_Video_Uni
from ..modules.wan import WanRMSNorm


class WanLayerNorm(nn.LayerNorm):
    """model.py:199-208: LayerNorm computed in the input's dtype, no affine unless asked."""

    def __init__(self, dim, eps=1e-6, elementwise_affine=False):
        super().__init__(dim, elementwise_affine=elementwise_affine, eps=eps)

    def forward(self, x):
        return super().forward(x).type_as(x)


class WanT2VCrossAttention(nn.Module):
    """model.py:1525-1546: q from the video tokens, k / v from the text context, softmax attention, output projection."""

    def __init__(self, dim, num_heads, qk_norm=True, eps=1e-6):
        super().__init__()
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.q, self.k, self.v, self.o = (nn.Linear(dim, dim) for _ in range(4))
        self.norm_q = WanRMSNorm(dim, eps=eps) if qk_norm else nn.Identity()
        self.norm_k = WanRMSNorm(dim, eps=eps) if qk_norm else nn.Identity()

    def forward(self, x, context, context_lens=None):
        b, n, d = x.size(0), self.num_heads, self.head_dim
        q = self.norm_q(self.q(x)).view(b, -1, n, d).transpose(1, 2)
        k = self.norm_k(self.k(context)).view(b, -1, n, d).transpose(1, 2)
        v = self.v(context).view(b, -1, n, d).transpose(1, 2)
        mask = None
        if context_lens is not None:
            L2 = context.size(1)
            mask = (torch.arange(L2, device=x.device)[None, :] < context_lens.to(x.device)[:, None])[:, None, None, :]
        out = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
        return self.o(out.transpose(1, 2).flatten(2))


class WanAttentionBlock_MHLA(nn.Module):
    def __init__(self, dim=1536, ffn_dim=8960, num_heads=12, qk_norm=True, cross_attn_norm=True, eps=1e-6,
                 norm_output=False, is_gated=True, is_lepe=False, block_layout=(3, 5, 10)):
        super().__init__()
        self.norm1 = WanLayerNorm(dim, eps)
        self.self_attn = MHLA_Video_Uni(dim, num_heads=num_heads, qk_norm=qk_norm, eps=eps, normalize_out=norm_output,
                                        is_gated=is_gated, is_lepe=is_lepe, block_layout=block_layout)
        self.norm3 = WanLayerNorm(dim, eps, elementwise_affine=True) if cross_attn_norm else nn.Identity()
        self.cross_attn = WanT2VCrossAttention(dim, num_heads, qk_norm, eps)
        self.norm2 = WanLayerNorm(dim, eps)
        self.ffn = nn.Sequential(nn.Linear(dim, ffn_dim), nn.GELU(approximate="tanh"), nn.Linear(ffn_dim, dim))
        self.modulation = nn.Parameter(torch.randn(1, 6, dim) / dim ** 0.5)

    def forward(self, x, e, seq_lens, grid_sizes, freqs, context, context_lens=None):
        """x [B, L, C]; e [B, 6, C] fp32 (time embedding projections); context [B, L2, C].  model.py:1686-1766."""
        assert e.dtype == torch.float32
        e = (self.modulation.float() + e).chunk(6, dim=1)
        dt = x.dtype
        y = self.self_attn((self.norm1(x).float() * (1 + e[1]) + e[0]).to(dt), seq_lens, grid_sizes, freqs)
        x = (x.float() + y.float() * e[2]).to(dt)
        x = x + self.cross_attn(self.norm3(x), context, context_lens)
        y = self.ffn((self.norm2(x).float() * (1 + e[4]) + e[3]).to(dt))
        return (x.float() + y.float() * e[5]).to(dt)


class WanStack_MHLA(nn.Module):
    """`num_layers` Wan blocks in sequence (30 in Wan2.1-1.3B, wan/model.py:1824-2389 builds `self.blocks` the same way) with a
    shared time-modulation input -- the self-attention / cross-attention / FFN body of the denoiser without its patch embedding,
    text encoder and head, so that the per-step cost of BASELINE.json configs[3] (81 frames at 832 x 480 = 31 500 video tokens)
    can be timed on one GPU.  `attn_type` selects the self-attention class by the reference's registry key."""

    def __init__(self, num_layers=30, dim=1536, ffn_dim=8960, num_heads=12, block_layout=(3, 5, 10), is_gated=True, is_lepe=False,
                 norm_output=False):
        super().__init__()
        self.blocks = nn.ModuleList([WanAttentionBlock_MHLA(dim=dim, ffn_dim=ffn_dim, num_heads=num_heads, block_layout=block_layout,
                                                            is_gated=is_gated, is_lepe=is_lepe, norm_output=norm_output)
                                     for _ in range(num_layers)])

    def forward(self, x, e, seq_lens, grid_sizes, freqs, context, context_lens=None):
        for blk in self.blocks:
            x = blk(x, e, seq_lens, grid_sizes, freqs, context, context_lens)
        return x
