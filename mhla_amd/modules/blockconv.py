"""Parameter containers for the mixing matrix, state-dict compatible with the reference
(`piece_attn.conv.weight [M, M, 1, 1]`, `block_attn.conv.weight [M, M, 1, 1]`).

The reference applies this weight as a 1x1 nn.Conv2d over [bh, M, D, D]
(mhla_dit/mhla/mhla.py:46-57,124-134); here the module only OWNS the parameter -- the mixing
itself happens inside the HIP operator (mhla_amd.ops.mhla_blockmix)."""
from typing import Sequence

import torch
from torch import nn

from ..weights import block_distance_weights


class _MixWeight(nn.Module):
    """`.weight` is a [M, M, 1, 1] parameter, like nn.Conv2d(M, M, 1, bias=False).weight."""

    def __init__(self, w: torch.Tensor):
        super().__init__()
        self.weight = nn.Parameter(w.clone().unsqueeze(-1).unsqueeze(-1))


class BlockDistanceConv(nn.Module):
    """2-D block grid (mhla_dit/mhla/mhla.py:10-138).  Trainable, like the reference (its freeze is
    commented out, mhla.py:59-60)."""

    def __init__(self, num_patches_per_side=16, patch_group_size=16, transform="linear", local_thres=1.5,
                 exp_sigma=3):
        super().__init__()
        self.num_patches_per_side = num_patches_per_side
        self.patch_group_size = patch_group_size
        self.transform = transform
        self.local_thres = local_thres
        self.exp_sigma = exp_sigma
        patches_per_block_side = int(patch_group_size ** 0.5)
        self.blocks_per_side = num_patches_per_side // patches_per_block_side
        self.total_blocks = self.blocks_per_side ** 2
        w = block_distance_weights((self.blocks_per_side, self.blocks_per_side), transform, local_thres, exp_sigma)
        self.conv = _MixWeight(w)

    def get_weight_matrix(self) -> torch.Tensor:
        return self.conv.weight.data.squeeze(-1).squeeze(-1)

    def forward(self, x):
        raise RuntimeError("BlockDistanceConv only holds the mixing weights; the mixing runs inside "
                           "mhla_amd.ops.mhla_blockmix")


class BlockDistanceConv3D(nn.Module):
    """3-D block grid (mhla_videogen/diffusion/model/wan/mhla_utils.py:9-125)."""

    def __init__(self, blocks_layout: Sequence[int] = (4, 4, 4), transform="linear", local_thres=1.5, exp_sigma=3):
        super().__init__()
        self.blocks_layout = tuple(blocks_layout)
        self.transform = transform
        self.local_thres = local_thres
        self.exp_sigma = exp_sigma
        self.total_blocks = blocks_layout[0] * blocks_layout[1] * blocks_layout[2]
        self.conv = _MixWeight(block_distance_weights(self.blocks_layout, transform, local_thres, exp_sigma))

    def get_weight_matrix(self) -> torch.Tensor:
        return self.conv.weight.data.squeeze(-1).squeeze(-1)

    def forward(self, x):
        raise RuntimeError("BlockDistanceConv3D only holds the mixing weights; the mixing runs inside "
                           "mhla_amd.ops.mhla_blockmix")
