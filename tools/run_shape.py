#!/usr/bin/env python
"""Run a few iterations of one BASELINE.json shape (for rocprofv3):  run_shape.py c3|c3f|c4|c4b|c5|c5b|xl|c2b|c2c [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mhla_amd  # noqa: E402
from mhla_amd import block_distance_weights, block_index_3d, causal_mixing_init  # noqa: E402

DEV = "cuda"
which = sys.argv[1] if len(sys.argv) > 1 else "c4b"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = torch.Generator().manual_seed(1)
if which in ("c4", "c4b"):
    B, N, H, D = 1, 31500, 12, 128
    mk = lambda: torch.randn(B, N, H, D, generator=g).to(DEV)
    ts = [mk().abs(), mk().abs(), mk(), mk().abs(), mk().abs()]
    W = block_distance_weights((3, 5, 10), "linear").to(DEV)
    idx = block_index_3d((21, 30, 50), (3, 5, 10)).to(DEV)
    do = mk()
    if which == "c4b":
        ts = [t.requires_grad_(True) for t in ts]
        W.requires_grad_(True)

    def step():
        o = mhla_amd.mhla_blockmix(ts[0], ts[1], ts[2], W, q_den=ts[3], k_den=ts[4], block_index=idx)
        if which == "c4b":
            o.backward(do)
            for t in ts + [W]:
                t.grad = None
elif which in ("c3", "c3f"):   # c3f: the same op in fp32 (the reference trains without autocast)
    B, N, H, D = 32, 256, 16, 72
    dt = torch.float32 if which == "c3f" else torch.bfloat16
    ts = [torch.randn(B, N, H, D, generator=g).abs().to(dt).to(DEV).requires_grad_(True) for _ in range(3)]
    do = torch.randn(B, N, H, D, generator=g).to(dt).to(DEV)
    W = block_distance_weights((4, 4), "linear").to(DEV).requires_grad_(True)

    def step():
        mhla_amd.mhla_blockmix(ts[0], ts[1], ts[2], W).backward(do)
        for t in ts + [W]:
            t.grad = None
elif which == "c2b":   # C2 variant: 256 blocks of 16 tokens
    B, N, H, D = 8, 4096, 16, 64
    ts = [torch.randn(B, N, H, D, generator=g).abs().bfloat16().to(DEV).requires_grad_(True) for _ in range(3)]
    do = torch.randn(B, N, H, D, generator=g).bfloat16().to(DEV)
    W = block_distance_weights((16, 16), "linear").to(DEV).requires_grad_(True)

    def step():
        mhla_amd.mhla_blockmix(ts[0], ts[1], ts[2], W).backward(do)
        for t in ts + [W]:
            t.grad = None
elif which == "c2c":   # C2 variant: 16 blocks of 256 tokens
    B, N, H, D = 8, 4096, 16, 64
    ts = [torch.randn(B, N, H, D, generator=g).abs().bfloat16().to(DEV).requires_grad_(True) for _ in range(3)]
    do = torch.randn(B, N, H, D, generator=g).bfloat16().to(DEV)
    W = block_distance_weights((4, 4), "linear").to(DEV).requires_grad_(True)

    def step():
        mhla_amd.mhla_blockmix(ts[0], ts[1], ts[2], W).backward(do)
        for t in ts + [W]:
            t.grad = None
elif which == "xl":
    B, N, H, D = 16, 1024, 16, 72
    ts = [torch.randn(B, N, H, D, generator=g).abs().bfloat16().to(DEV).requires_grad_(True) for _ in range(3)]
    do = torch.randn(B, N, H, D, generator=g).bfloat16().to(DEV)
    W = block_distance_weights((4, 4), "linear").to(DEV).requires_grad_(True)

    def step():
        mhla_amd.mhla_blockmix(ts[0], ts[1], ts[2], W).backward(do)
        for t in ts + [W]:
            t.grad = None
else:
    B, T, H, K, V = (2, 8192, 4, 256, 512) if which == "c5b" else (4, 8192, 4, 128, 256)
    q = torch.randn(B, T, H, K, generator=g).bfloat16().to(DEV).requires_grad_(True)
    k = torch.randn(B, T, H, K, generator=g).bfloat16().to(DEV).requires_grad_(True)
    v = torch.randn(B, T, H, V, generator=g).bfloat16().to(DEV).requires_grad_(True)
    do = torch.randn(B, T, H, V, generator=g).bfloat16().to(DEV)
    mix = causal_mixing_init(T // 64).reshape(T // 64, T // 64).to(DEV).requires_grad_(True)

    def step():
        mhla_amd.mhla_causal(q, k, v, mix).backward(do)
        q.grad = k.grad = v.grad = mix.grad = None
for _ in range(iters):
    step()
torch.cuda.synchronize()
print("done", which, iters)
