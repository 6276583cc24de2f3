from .blockconv import BlockDistanceConv, BlockDistanceConv3D
from .dit import MHLA4DiT, MHLA_Normed_Torch
from .fla import MHLA, FusedRMSNormGated, RotaryEmbedding
from .wan import MHLA_Video_Uni, WanRMSNorm, rope_params, wan_freqs
from .wan_variants import (WAN_SELFATTENTION_CLASSES, Gated_MHLA_Video, Gated_MHLA_Video_LePE, MHLA_Video, MHLA_Video_LePE,
                           MHLA_Video_Nope)

__all__ = ["BlockDistanceConv", "BlockDistanceConv3D", "MHLA4DiT", "MHLA_Normed_Torch", "MHLA",
           "FusedRMSNormGated", "RotaryEmbedding", "MHLA_Video_Uni", "WanRMSNorm", "rope_params", "wan_freqs",
           "MHLA_Video", "MHLA_Video_Nope", "Gated_MHLA_Video", "MHLA_Video_LePE", "Gated_MHLA_Video_LePE", "WAN_SELFATTENTION_CLASSES"]
