from .blockconv import BlockDistanceConv, BlockDistanceConv3D
from .dit import MHLA4DiT, MHLA_Normed_Torch
from .fla import MHLA, FusedRMSNormGated, RotaryEmbedding
from .wan import MHLA_Video_Uni, WanRMSNorm, rope_params, wan_freqs

__all__ = ["BlockDistanceConv", "BlockDistanceConv3D", "MHLA4DiT", "MHLA_Normed_Torch", "MHLA",
           "FusedRMSNormGated", "RotaryEmbedding", "MHLA_Video_Uni", "WanRMSNorm", "rope_params", "wan_freqs"]
