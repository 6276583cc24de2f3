#!/usr/bin/env python
"""Slice-loop timeline of k_sp_mixh (library built with -DMIXH_TRACE: tools/build_variant.sh trace -DMIXH_TRACE, run with
MHLA_LIB_PATH=mhla_amd/lib/variants/libmhla_trace.so): wave 0 of the first eight workgroups stamps s_memtime at
0 entry | 1 commit + previous slice's stores issued | 2 barrier | 3 next slice requested | 4 weights rescaled | 5 products | 6 staged | 7 barrier.
Prints the median length of each phase over the steady slices, in ticks (100 MHz) and as a share of the iteration.
  python tools/trace_mixh.py [M]     (C2's tensors with M blocks: 256 -> sixteen waves, 64 -> four)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mhla_amd  # noqa: E402
from mhla_amd import _lib  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B, N, H, D = 8, 4096, 16, 64
g = torch.Generator().manual_seed(1)
q, k, v = (torch.randn(B, N, H, D, generator=g).abs().bfloat16().cuda() for _ in range(3))
W = torch.rand(M, M, generator=g).cuda()
lib = _lib.load()
with torch.no_grad():
    for _ in range(3):
        mhla_amd.mhla_blockmix(q, k, v, W)
    torch.cuda.synchronize()
    buf = torch.zeros(8 * 32 * 8, dtype=torch.int64, device="cuda")
    lib.mhla_debug_set_trace(buf.data_ptr())
    mhla_amd.mhla_blockmix(q, k, v, W)
    torch.cuda.synchronize()
    lib.mhla_debug_set_trace(None)
t = buf.cpu().numpy().astype(np.int64).reshape(8, 32, 8)
names = ["commit+stores", "barrier", "request", "weights", "products", "staging", "barrier"]
for wg in range(8):
    x = t[wg]
    n = int((x[:, 7] > 0).sum())
    if n < 3:
        print(f"wg {wg}: {n} slices stamped")
        continue
    lo, hi = 1, n - 1
    it = np.diff(x[:n, 0])[lo:hi]
    ph = [np.median(x[lo:hi, i + 1] - x[lo:hi, i]) for i in range(7)]
    tot = sum(ph)
    print(f"wg {wg}: {n} slices, iteration {np.median(it):6.0f} ticks | " + " | ".join(f"{nm} {p:5.0f} ({p / tot:.0%})" for nm, p in zip(names, ph)) + f" | first slice {x[0, 7] - x[0, 0]} | whole loop {x[n - 1, 7] - x[0, 0]}")
