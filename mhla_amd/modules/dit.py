"""Drop-in MHLA attention modules for the DiT and timm-ViT hosts.

Same constructor arguments, `forward(x)` contract and state-dict keys as the reference's
`MHLA4DiT` (mhla_dit/mhla/mhla.py:141-275) and `MHLA_Normed_Torch`
(mhla_image_classification/models/modules/attention/mhla.py:141-289); the operator
(mhla.py:262-268) runs in the HIP kernels via mhla_amd.ops.mhla_blockmix, reading q, k, v in place
from the fused QKV projection output, with relu(x)+eps fused into the kernel loads when no
q/k RMSNorm sits in between.  No torch.compile / Inductor / Triton anywhere.
"""
import torch
import torch.nn.functional as F
from torch import nn

from ..ops import qk_prologue, lepe2d, mhla_blockmix, mhla_dit_core
from .blockconv import BlockDistanceConv


class MHLA4DiT(nn.Module):
    _LEPE_KERNEL = 3          # mhla_dit/mhla/mhla.py:169
    _SIZE_KW = "block_size"   # mhla.py:171
    _DEFAULT_TRANSFORM = "linear"

    def __init__(self, dim, heads=8, dim_head=None, dropout=0.1, fixed_weight_value=None, qk_norm=False,
                 transform=None, **kwargs):
        super().__init__()
        if transform is None:
            transform = self._DEFAULT_TRANSFORM
        if dim_head is None:
            dim_head = dim // heads
        inner_dim = dim_head * heads
        self.num_heads = heads
        self.head_dim = dim_head
        self.scale = dim_head ** -0.5

        self.norm = nn.LayerNorm(dim)
        is_bias = kwargs["qkv_bias"] if "qkv_bias" in kwargs else False
        self.to_qkv = nn.Linear(dim, inner_dim * 3, bias=is_bias)
        self.q_norm = nn.RMSNorm(dim) if qk_norm else nn.Identity()
        self.k_norm = nn.RMSNorm(dim) if qk_norm else nn.Identity()
        self.qk_norm = bool(qk_norm)
        kk = self._LEPE_KERNEL
        self.lepe = nn.Conv2d(dim, dim, kk, 1, kk // 2, groups=dim)

        self.block_size = kwargs[self._SIZE_KW] if self._SIZE_KW in kwargs else 49
        self.block_len = int(self.block_size ** 0.5)
        self.embed_len = kwargs["embed_len"] if "embed_len" in kwargs else 196
        self.num_pieces = self.embed_len // self.block_size
        self.pieces_len = int(self.num_pieces ** 0.5)
        self.piece_attn = BlockDistanceConv(
            num_patches_per_side=int(self.embed_len ** 0.5),
            patch_group_size=self.block_size,
            transform=transform,
            local_thres=kwargs.get("local_thres", 1.5),
            exp_sigma=kwargs.get("exp_sigma", 3),
        )
        self.eps = kwargs.get("eps", 1e-6)
        # not in the reference: "split" (default) keeps the operator's intermediates at fp32 grade on 16-bit tensors, "bf16" is the
        # opt-in reduced-precision arithmetic (see mhla_amd.mhla_blockmix)
        self.summaries = kwargs.get("summaries", "tf32")
        self.to_out = nn.Sequential(nn.Linear(inner_dim, dim), nn.Dropout(dropout))
        if fixed_weight_value is not None:
            self._init_weights_with_fixed_value(fixed_weight_value)

    # the reference's known-answer aid (mhla.py:193-221: every weight = value, every bias = 0), kept under its two public names
    def _init_weights_with_fixed_value(self, value):
        MHLA4DiT.init_to_value(self, value)

    @staticmethod
    def init_to_value(model, value=1.0):
        with torch.no_grad():
            for name, p in model.named_parameters():
                if "weight" in name:
                    p.fill_(value)
                elif "bias" in name:
                    p.zero_()
        return model

    def _lepe(self, v: torch.Tensor, attn_out: torch.Tensor) -> torch.Tensor:
        """attn_out + depthwise conv over V laid out as the raster image (mhla.py:246-247, 271-273), computed by the
        HIP LePE kernel directly on the block-major token layout (v may be the strided V slice of the fused QKV buffer)."""
        return lepe2d(v, self.lepe.weight, self.lepe.bias, self.pieces_len, self.block_len, add=attn_out)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: [B, M, S, C] (block-major tokens, as the ViT host passes them) or [B, N, C] (the DiT host,
        whose blocks see 3-D tensors -- SURVEY.md 3.1); returns the same shape."""
        three_d = x.dim() == 3
        if three_d:
            x = x.reshape(x.shape[0], self.num_pieces, self.block_size, x.shape[-1])
        B, M, S, C = x.shape
        H, D = self.num_heads, self.head_dim
        x = self.norm(x)
        qkv = self.to_qkv(x).reshape(B, M * S, 3, H, D)                       # mhla.py:245
        W = self.piece_attn.conv.weight
        if self.qk_norm:
            # q_norm / k_norm (nn.RMSNorm over the full channel dim, eps = finfo(dtype).eps as constructed at mhla.py:166-167)
            # -> relu -> + eps (mhla.py:226-230): one HIP kernel per tensor and direction (mhla_qk_prologue), reading the q / k
            # slices of the packed projection in place; its fp32 result goes back to the projection dtype, as the reference's
            # matmuls do under autocast -- the operator takes one element type for all token tensors
            if (H * D) % 8 == 0 and H * D <= 4096:
                eq = self.q_norm.eps if self.q_norm.eps is not None else torch.finfo(qkv.dtype).eps
                ek = self.k_norm.eps if self.k_norm.eps is not None else torch.finfo(qkv.dtype).eps
                q = qk_prologue(qkv[:, :, 0].reshape(B, M * S, H * D), self.q_norm.weight, eq, self.eps)
                k = qk_prologue(qkv[:, :, 1].reshape(B, M * S, H * D), self.k_norm.weight, ek, self.eps)
            else:
                if not getattr(self, "_warned_eager_prologue", False):
                    import warnings
                    warnings.warn(f"MHLA module: inner dim {H * D} (not a multiple of 8, or > 4096): q_norm / k_norm + relu + eps run as "
                                  "eager PyTorch ops instead of the fused HIP prologue", stacklevel=2)
                    self._warned_eager_prologue = True
                q = torch.relu(self.q_norm(qkv[:, :, 0].reshape(B, M * S, H * D))) + self.eps
                k = torch.relu(self.k_norm(qkv[:, :, 1].reshape(B, M * S, H * D))) + self.eps
            q, k = q.to(qkv.dtype), k.to(qkv.dtype)
            out = mhla_blockmix(q.reshape(B, M * S, H, D), k.reshape(B, M * S, H, D), qkv[:, :, 2], W, eps=self.eps,
                                summaries=self.summaries)
        elif D > 128:
            # head dims the kernels do not cover: the operator composes itself from slices (ops._blockmix_wide_head), LePE separately
            out = mhla_blockmix(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], W, eps=self.eps, relu_eps=True, summaries=self.summaries)
        else:
            # operator (relu + eps folded into its loads) and LePE as one autograd node on the packed projection output
            out = mhla_dit_core(qkv, W, self.lepe.weight, self.lepe.bias, self.pieces_len, self.block_len, eps=self.eps,
                                relu_eps=True, summaries=self.summaries).reshape(B, M, S, H * D)
            out = self.to_out(out)
            return out.reshape(B, M * S, -1) if three_d else out
        out = self._lepe(qkv[:, :, 2].reshape(B, M * S, H * D), out.reshape(B, M * S, H * D)).reshape(B, M, S, H * D)
        out = self.to_out(out)
        return out.reshape(B, M * S, -1) if three_d else out


class MHLA_Normed_Torch(MHLA4DiT):
    """timm-ViT variant (mhla_image_classification/models/modules/attention/mhla.py:141-289): cosine
    distance init, 5x5 LePE (:169), `window_size` keyword (:171); the timm block passes
    qk_norm/norm_layer (timm_block/mhla.py:44-51)."""
    _LEPE_KERNEL = 5
    _SIZE_KW = "window_size"
    _DEFAULT_TRANSFORM = "cos"

    def __init__(self, dim, heads=8, dim_head=None, dropout=0.1, fixed_weight_value=None, qk_norm=False,
                 transform="cos", **kwargs):
        kwargs.pop("norm_layer", None)
        super().__init__(dim, heads, dim_head, dropout, fixed_weight_value, qk_norm, transform, **kwargs)
        self.window_size = self.block_size
        self.window_len = self.block_len
