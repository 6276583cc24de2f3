#!/usr/bin/env python
"""Where the host time of an eager operator call goes (cProfile over many un-synchronised fwd+bwd calls of the C3 shape):
python tools/host_overhead.py [iters]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mhla_amd  # noqa: E402
from mhla_amd.weights import block_distance_weights  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B, N, H, D, M = 32, 256, 16, 72, 16
g = torch.Generator().manual_seed(0)
mk = lambda: (torch.rand(B, N, H, D, generator=g) + 0.01).to(torch.bfloat16).cuda().requires_grad_(True)
q, k, v = mk(), mk(), mk()
W = block_distance_weights((4, 4), "linear").cuda().requires_grad_(True)
do = torch.randn(B, N, H, D, generator=g).to(torch.bfloat16).cuda()


def step():
    out = mhla_amd.mhla_blockmix(q, k, v, W)
    out.backward(do)


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host time per step (launch side) {1e6 * (t1 - t0) / iters:.1f} us; with final sync {1e6 * (t2 - t0) / iters:.1f} us")
pr = cProfile.Profile()
pr.enable()
for _ in range(iters):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(32)
