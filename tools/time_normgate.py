#!/usr/bin/env python
"""Causal operator + per-head RMSNorm x swish gate (the fla layer's epilogue, layers/mhla.py:330-355): fused into the operator's
output kernel vs the composition of the two HIP operators, forward only (the backward is shared).  python tools/time_normgate.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench_configs as bc  # noqa: E402
import mhla_amd  # noqa: E402
from mhla_amd import causal_mixing_init  # noqa: E402

for (B, T, H, K, V) in ((4, 8192, 4, 128, 256), (2, 8192, 4, 256, 512)):
    g = torch.Generator().manual_seed(1)
    mk = lambda d: torch.randn(B, T, H, d, generator=g).bfloat16().cuda()
    q, k, v, gate = mk(K), mk(K), mk(V), mk(V)
    w = torch.rand(V, generator=g).cuda() + 0.5
    mix = causal_mixing_init(T // 64).reshape(T // 64, T // 64).cuda()
    with torch.no_grad():
        fused = lambda: mhla_amd.mhla_causal_normgate(q, k, v, mix, gate, w, 1e-5)
        unfused = lambda: mhla_amd.rmsnorm_gate(mhla_amd.mhla_causal(q, k, v, mix), gate, w, 1e-5)
        for name, fn in (("fused", fused), ("unfused", unfused)):
            t = bc.timeit(fn, iters=20)
            print(f"B={B} T={T} H={H} K={K} V={V} {name:8s} {t * 1e3:.4f} ms  ", {k_: round(v_, 1) for k_, v_ in bc.kernel_times(fn).items()})
