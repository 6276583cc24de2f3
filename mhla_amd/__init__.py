"""mhla_amd -- the MHLA attention operator for AMD MI355X (gfx950 / CDNA4).

Hand-written HIP kernels behind a C ABI (include/mhla_hip.h, libmhla_hip.so), the autograd
operators over it (mhla_amd.ops) and drop-in attention modules for the DiT, timm-ViT, Wan2.1 and
flash-linear-attention hosts (mhla_amd.modules).  No CPU / eager fallback.
"""
from . import _lib
from .ops import (featmap_rotary, lepe2d, lepe3d, mhla_dit_core, mhla_blockmix, mhla_blockmix_rope, mhla_blockmix_wan, mhla_blockmix_wan_pro, mhla_causal, mhla_causal_normgate, naive_chunk_simple_mhla_fixed, naive_recurrent_mhla, qk_prologue,
                  rmsnorm_gate, set_option, describe_dispatch, describe_causal_dispatch)
from .weights import block_distance_weights, block_index_2d, block_index_3d, causal_mixing_init

__all__ = ["mhla_blockmix", "mhla_blockmix_rope", "mhla_blockmix_wan", "mhla_blockmix_wan_pro", "qk_prologue", "lepe2d", "lepe3d", "featmap_rotary", "mhla_dit_core", "mhla_causal", "mhla_causal_normgate", "naive_chunk_simple_mhla_fixed", "naive_recurrent_mhla", "rmsnorm_gate", "set_option", "describe_dispatch", "describe_causal_dispatch",
           "block_distance_weights", "block_index_2d", "block_index_3d", "causal_mixing_init", "_lib"]
__version__ = "0.1.0"
