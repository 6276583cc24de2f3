"""Loader of the C++ autograd nodes (mhla_amd/lib/libmhla_torch.so, built by mhla_amd.build.build_torch_ext from
csrc_torch/mhla_torch.cpp): `torch.ops.mhla_amd.blockmix / causal` -- the same checks, C ABI calls and saved state as the Python
autograd.Functions of ops.py, without the interpreter on the eager path.  Optional: when the library is not built (or was built
from other sources) ops.py runs its Python nodes, which call the same C ABI."""
import os

import torch

from . import _lib

_state = {"tried": False, "ok": False}


def available() -> bool:
    if _state["tried"]:
        return _state["ok"]
    _state["tried"] = True
    from . import build as b
    try:
        if not os.path.exists(b.TORCH_LIB) or b.torch_ext_stale():
            return False
        _lib.load()                                   # ABI / build-flag checks of the HIP library first
        torch.ops.load_library(b.TORCH_LIB)
        if torch.ops.mhla_amd.init(_lib.LIB_PATH) != _lib.ABI_VERSION:
            return False
        _state["ok"] = True
    except Exception:   # noqa: BLE001  (an unloadable optional library must not take the package down)
        _state["ok"] = False
    return _state["ok"]
