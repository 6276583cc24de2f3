"""Host logic: mixing-matrix initialisers and block layout maps (no GPU needed).

`block_distance_weights` mirrors BlockDistanceConv / BlockDistanceConv3D
(mhla_dit/mhla/mhla.py:63-122, mhla_videogen/diffusion/model/wan/mhla_utils.py:61-118):
Euclidean distance between block centres on the block grid, a transform, and column
normalisation (all transforms but "gaussian").  Vectorised instead of the reference's O(M^2)
Python loop.
"""
import math
from functools import lru_cache
from typing import Sequence, Tuple

import torch

TRANSFORMS = ("linear", "cos", "exp", "gaussian", "local")


def block_distance_weights(layout: Sequence[int], transform: str = "linear", local_thres: float = 1.5,
                           exp_sigma: float = 3.0) -> torch.Tensor:
    """W[M, M] fp32, W[i, j] = weight of input block j in output block i."""
    if transform not in TRANSFORMS:
        raise ValueError(f"Unknown transform: {transform}")
    axes = [torch.arange(int(n), dtype=torch.float32) + 0.5 for n in layout]
    centres = torch.cartesian_prod(*axes) if len(axes) > 1 else axes[0][:, None]
    centres = centres.reshape(-1, len(axes))
    dist = torch.cdist(centres.double(), centres.double()).float()
    dist.fill_diagonal_(0.0)
    if transform == "gaussian":
        sigma = dist.max() / 3
        return torch.exp(-(dist ** 2) / (2 * sigma ** 2))
    if transform == "linear":
        mat = 1.0 - dist / dist.max()
    elif transform == "cos":
        mat = torch.cos(dist / dist.max() * math.pi / 4)
    elif transform == "exp":
        mat = torch.exp(-dist / exp_sigma)
    else:  # local
        mat = (dist <= local_thres).float()
    return mat / mat.sum(dim=0, keepdim=True)


def causal_mixing_init(L: int = 32) -> torch.Tensor:
    """tril(ones(L, L)) / rowcount, shaped [L, L, 1, 1, 1, 1] (mhla_nlp/fla/layers/mhla.py:196-200)."""
    lower = torch.tril(torch.ones(L, L, dtype=torch.float32))
    lower = lower / (torch.arange(L, dtype=torch.float32).unsqueeze(1) + 1.0)
    return lower.view(L, L, 1, 1, 1, 1)


@lru_cache(maxsize=64)
def _block_index_2d_cpu(pieces: int, block_len: int) -> torch.Tensor:
    side = pieces * block_len
    r = torch.arange(side * side, dtype=torch.int32).reshape(pieces, block_len, pieces, block_len)
    return r.permute(0, 2, 1, 3).reshape(-1).contiguous()


def block_index_2d(pieces: int, block_len: int) -> torch.Tensor:
    """int32[N]: block-major position -> raster token index of a (pieces*block_len)^2 image
    (the permutation of rearrange_patches, mhla_dit/piecewise_patchembed.py:47-63)."""
    return _block_index_2d_cpu(int(pieces), int(block_len))


@lru_cache(maxsize=64)
def _block_index_3d_cpu(grid: Tuple[int, int, int], layout: Tuple[int, int, int]) -> torch.Tensor:
    f, h, w = grid
    fb, hb, wb = layout
    if f % fb or h % hb or w % wb:
        raise ValueError(f"grid {grid} is not divisible by block layout {layout}")
    r = torch.arange(f * h * w, dtype=torch.int32).reshape(fb, f // fb, hb, h // hb, wb, w // wb)
    return r.permute(0, 2, 4, 1, 3, 5).reshape(-1).contiguous()


def block_index_3d(grid: Sequence[int], layout: Sequence[int]) -> torch.Tensor:
    """int32[N]: block-major position -> raster (f h w) token index: the gather the reference does with
    rearrange "b (fb p1 hb p2 wb p3) h c -> (b h) (fb hb wb) (p1 p2 p3) c" (wan/mhla_utils.py:317-326)."""
    return _block_index_3d_cpu(tuple(int(x) for x in grid), tuple(int(x) for x in layout))
