"""A thin DiT host for the MHLA4DiT drop-in: the class-conditional diffusion transformer of mhla_dit/models.py
(`DiT_MHLA`, :240-400; block :115-186; final layer :208-232; embedders :27-108) with the same constructor arguments,
forward signature and parameter names, so checkpoints of the reference's DiT load unchanged.  Plumbing only -- every
layer besides the attention module is stock PyTorch.

Tokens are put into block-major order once after the patch embedding (the reference's
PiecewisePatchEmbed.rearrange_patches, mhla_dit/piecewise_patchembed.py:47-63) and back once before unpatchify; here
both are a single index_select with the int32 map the operator's gather uses (weights.block_index_2d)."""
import math
from typing import Dict, Optional

import torch
from torch import nn

from ..modules import MHLA4DiT
from ..weights import block_index_2d


def modulate(x, shift, scale):
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


class TimestepEmbedder(nn.Module):
    def __init__(self, hidden_size, frequency_embedding_size=256):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(frequency_embedding_size, hidden_size, bias=True), nn.SiLU(),
                                 nn.Linear(hidden_size, hidden_size, bias=True))
        self.frequency_embedding_size = frequency_embedding_size

    @staticmethod
    def timestep_embedding(t, dim, max_period=10000):
        half = dim // 2
        freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
        args = t[:, None].float() * freqs[None]
        emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
        if dim % 2:
            emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
        return emb

    def forward(self, t):
        return self.mlp(self.timestep_embedding(t, self.frequency_embedding_size).to(self.mlp[0].weight.dtype))


class LabelEmbedder(nn.Module):
    def __init__(self, num_classes, hidden_size, dropout_prob):
        super().__init__()
        self.embedding_table = nn.Embedding(num_classes + int(dropout_prob > 0), hidden_size)
        self.num_classes = num_classes
        self.dropout_prob = dropout_prob

    def forward(self, labels, train, force_drop_ids=None):
        if (train and self.dropout_prob > 0) or force_drop_ids is not None:
            drop = (torch.rand(labels.shape[0], device=labels.device) < self.dropout_prob) if force_drop_ids is None \
                else force_drop_ids == 1
            labels = torch.where(drop, torch.full_like(labels, self.num_classes), labels)
        return self.embedding_table(labels)


class PatchEmbed(nn.Module):
    """Non-overlapping patches -> tokens (timm's PatchEmbed as the reference uses it: a strided conv named `proj`)."""

    def __init__(self, img_size, patch_size, in_chans, embed_dim, bias=True):
        super().__init__()
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=bias)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU(approximate="tanh")
        self.fc2 = nn.Linear(hidden_features, in_features)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class DiTBlock_MHLA(nn.Module):
    """adaLN-Zero block (models.py:115-186) around the MHLA4DiT drop-in."""

    def __init__(self, hidden_size, num_heads, mlp_ratio=4.0, **block_kwargs):
        super().__init__()
        self.norm1 = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.attn = MHLA4DiT(dim=hidden_size, heads=num_heads, qkv_bias=True, **block_kwargs)
        self.norm2 = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.mlp = Mlp(hidden_size, int(hidden_size * mlp_ratio))
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 6 * hidden_size, bias=True))

    def forward(self, x, c):
        s1, sc1, g1, s2, sc2, g2 = self.adaLN_modulation(c).chunk(6, dim=1)
        x = x + g1.unsqueeze(1) * self.attn(modulate(self.norm1(x), s1, sc1))
        return x + g2.unsqueeze(1) * self.mlp(modulate(self.norm2(x), s2, sc2))


class FinalLayer(nn.Module):
    def __init__(self, hidden_size, patch_size, out_channels):
        super().__init__()
        self.norm_final = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.linear = nn.Linear(hidden_size, patch_size * patch_size * out_channels, bias=True)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 2 * hidden_size, bias=True))

    def forward(self, x, c):
        shift, scale = self.adaLN_modulation(c).chunk(2, dim=1)
        return self.linear(modulate(self.norm_final(x), shift, scale))


def sincos_pos_embed_2d(dim: int, grid: int) -> torch.Tensor:
    """Fixed 2-D sin-cos table [grid*grid, dim] (MAE convention: first half encodes one axis, second half the other)."""
    def axis(d, pos):
        omega = 1.0 / 10000 ** (torch.arange(d // 2, dtype=torch.float64) / (d / 2.0))
        out = pos.reshape(-1, 1).double() * omega[None]
        return torch.cat([torch.sin(out), torch.cos(out)], dim=1)
    gh, gw = torch.meshgrid(torch.arange(grid), torch.arange(grid), indexing="ij")
    return torch.cat([axis(dim // 2, gw), axis(dim // 2, gh)], dim=1).float()


class DiT_MHLA(nn.Module):
    def __init__(self, input_size=32, patch_size=2, in_channels=4, hidden_size=1152, depth=28, num_heads=16, mlp_ratio=4.0,
                 class_dropout_prob=0.1, num_classes=1000, learn_sigma=True, block_kwargs: Optional[Dict] = None):
        super().__init__()
        block_kwargs = dict(block_kwargs or {})
        self.learn_sigma = learn_sigma
        self.in_channels = in_channels
        self.out_channels = in_channels * 2 if learn_sigma else in_channels
        self.patch_size = patch_size
        self.num_heads = num_heads
        self.x_embedder = PatchEmbed(input_size, patch_size, in_channels, hidden_size, bias=True)
        self.t_embedder = TimestepEmbedder(hidden_size)
        self.y_embedder = LabelEmbedder(num_classes, hidden_size, class_dropout_prob)
        n = self.x_embedder.num_patches
        self.pos_embed = nn.Parameter(torch.zeros(1, n, hidden_size), requires_grad=False)
        self.block_size = block_kwargs.setdefault("block_size", 16)
        block_kwargs["embed_len"] = n                                   # models.py:278-280
        self.piece_size = int(self.block_size ** 0.5)
        side = self.x_embedder.grid_size[0]
        if side % self.piece_size:
            raise ValueError(f"{side} patches per side not divisible into blocks of {self.piece_size}")
        idx = block_index_2d(side // self.piece_size, self.piece_size).long()      # block-major position -> raster token
        self.register_buffer("to_block_major", idx, persistent=False)
        self.register_buffer("to_raster", torch.argsort(idx), persistent=False)
        self.blocks = nn.ModuleList([DiTBlock_MHLA(hidden_size, num_heads, mlp_ratio=mlp_ratio, **block_kwargs)
                                     for _ in range(depth)])
        self.final_layer = FinalLayer(hidden_size, patch_size, self.out_channels)
        self.initialize_weights()

    def initialize_weights(self):
        """As models.py:298-345: xavier linears, identity-centred depthwise convs, sin-cos positions, zeroed adaLN / head."""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Conv2d) and m.weight.shape[-1] >= 3 and m.weight.shape[-1] % 2 == 1:
                with torch.no_grad():
                    m.weight.zero_()
                    c = m.weight.shape[-1] // 2
                    m.weight[:, :, c, c] = 1
                    if m.bias is not None:
                        m.bias.zero_()
        with torch.no_grad():
            self.pos_embed.copy_(sincos_pos_embed_2d(self.pos_embed.shape[-1], self.x_embedder.grid_size[0]).unsqueeze(0))
            w = self.x_embedder.proj.weight
            nn.init.xavier_uniform_(w.view(w.shape[0], -1))
            nn.init.zeros_(self.x_embedder.proj.bias)
            nn.init.normal_(self.y_embedder.embedding_table.weight, std=0.02)
            nn.init.normal_(self.t_embedder.mlp[0].weight, std=0.02)
            nn.init.normal_(self.t_embedder.mlp[2].weight, std=0.02)
            for blk in self.blocks:
                nn.init.zeros_(blk.adaLN_modulation[-1].weight)
                nn.init.zeros_(blk.adaLN_modulation[-1].bias)
            nn.init.zeros_(self.final_layer.adaLN_modulation[-1].weight)
            nn.init.zeros_(self.final_layer.adaLN_modulation[-1].bias)
            nn.init.zeros_(self.final_layer.linear.weight)
            nn.init.zeros_(self.final_layer.linear.bias)

    def unpatchify(self, x):
        c, p = self.out_channels, self.patch_size
        h = w = int(x.shape[1] ** 0.5)
        x = x.reshape(x.shape[0], h, w, p, p, c).permute(0, 5, 1, 3, 2, 4)
        return x.reshape(x.shape[0], c, h * p, w * p)

    def forward(self, x, t, y):
        """x: [N, C, H, W] latents, t: [N] timesteps, y: [N] labels -> [N, out_channels, H, W]."""
        x = self.x_embedder(x) + self.pos_embed
        x = x.index_select(1, self.to_block_major)                      # models.py:369 (rearrange_patches)
        c = self.t_embedder(t) + self.y_embedder(y, self.training)
        for blk in self.blocks:
            x = blk(x, c)
        x = self.final_layer(x, c)
        x = x.index_select(1, self.to_raster)                           # models.py:378-383
        return self.unpatchify(x)


def DiT_configs():
    """Name -> constructor kwargs of the reference's model zoo (models.py: DiT-{S,B,L,XL}/2)."""
    return {"DiT-S/2": dict(depth=12, hidden_size=384, patch_size=2, num_heads=6),
            "DiT-B/2": dict(depth=12, hidden_size=768, patch_size=2, num_heads=12),
            "DiT-L/2": dict(depth=24, hidden_size=1024, patch_size=2, num_heads=16),
            "DiT-XL/2": dict(depth=28, hidden_size=1152, patch_size=2, num_heads=16)}
