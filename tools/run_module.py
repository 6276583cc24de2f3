#!/usr/bin/env python
"""Run a drop-in module a few times (for rocprofv3 / timing):  run_module.py wan|dit [iters]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mhla_amd.modules import MHLA4DiT, MHLA_Video_Uni  # noqa: E402
from mhla_amd.modules.wan import wan_freqs  # noqa: E402

DEV = "cuda"
which = sys.argv[1] if len(sys.argv) > 1 else "wan"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.manual_seed(0)
if which == "wan":
    m = MHLA_Video_Uni(1536, num_heads=12, block_layout=(3, 5, 10), is_gated=True).to(DEV).to(torch.bfloat16).eval()
    x = torch.randn(1, 21 * 30 * 50, 1536, device=DEV, dtype=torch.bfloat16)
    grid = torch.tensor([[21, 30, 50]])
    freqs = wan_freqs(128)

    def step():
        with torch.no_grad():
            return m(x, None, grid, freqs)
elif which == "wanb":   # Wan layer in training mode (unfused prologue, backward through the operator)
    m = MHLA_Video_Uni(1536, num_heads=12, block_layout=(3, 5, 10), is_gated=True).to(DEV).to(torch.bfloat16)
    x = torch.randn(1, 21 * 30 * 50, 1536, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    grid = torch.tensor([[21, 30, 50]])
    freqs = wan_freqs(128)

    def step():
        y = m(x, None, grid, freqs)
        y.backward(torch.ones_like(y))
        return y
elif which == "fla":
    from mhla_amd.modules import MHLA
    m = MHLA(mode="chunk", hidden_size=1024, expand_k=0.5, expand_v=1.0, num_heads=4, feature_map="relu").to(DEV).to(torch.bfloat16)
    with torch.no_grad():
        m.mixing_matrix.copy_(torch.tril(torch.rand(32, 32)).clamp(1e-5, 1).view(32, 32, 1, 1, 1, 1))
    x = torch.randn(16, 2048, 1024, device=DEV, dtype=torch.bfloat16, requires_grad=True)

    def step():
        y = m(x)[0]
        y.backward(torch.ones_like(y))
        return y
else:
    m = MHLA4DiT(1152, 16, qkv_bias=True, block_size=16, embed_len=256).to(DEV).to(torch.bfloat16)
    x = torch.randn(32, 256, 1152, device=DEV, dtype=torch.bfloat16, requires_grad=True)

    def step():
        y = m(x)
        y.backward(torch.ones_like(y))
        return y
for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    step()
torch.cuda.synchronize()
print(f"{which}: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms per module call")
