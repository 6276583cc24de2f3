#!/usr/bin/env python
"""Phase timeline of k_sn_bwd (small-sequence backward, DiT-XL/2 256^2 shape) from the library's trace hook: wave 0 of every
workgroup stamps s_memtime (shader cycles) at  0 start | 1 K,V staged | 2 ksum | 3 z, row dots | 4 1/n, dn, dz | 5 pass A done
(wave 0) | 6 barrier | 7 Q, dO' staged | 8 pass B done (wave 0) | 9 stores drained.  Prints median cycles per phase."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mhla_amd  # noqa: E402
from mhla_amd import _lib, block_distance_weights  # noqa: E402

B, N, H, D, M = int(sys.argv[2]) if len(sys.argv) > 2 else 32, 256, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 72, 16
dev = "cuda"
g = torch.Generator().manual_seed(1)
ts = [torch.randn(B, N, H, D, generator=g).abs().bfloat16().to(dev).requires_grad_(True) for _ in range(3)]
do = torch.randn(B, N, H, D, generator=g).bfloat16().to(dev)
W = block_distance_weights((4, 4), "linear").to(dev).requires_grad_(True)
lib = _lib.load()


def step():
    mhla_amd.mhla_blockmix(ts[0], ts[1], ts[2], W).backward(do)
    for t in ts + [W]:
        t.grad = None


for _ in range(3):
    step()
torch.cuda.synchronize()
nwg = B * H
buf = torch.zeros(nwg * 16, dtype=torch.int64, device=dev)
lib.mhla_debug_set_trace(buf.data_ptr())
step()
torch.cuda.synchronize()
lib.mhla_debug_set_trace(None)
x = buf.cpu().numpy().reshape(nwg, 16)[:, :10]
d = np.diff(x, axis=1)
names = ["stage K,V (HBM)", "ksum", "z + row dots (HBM)", "1/n, dn, dz", "pass A (wave 0)", "barrier", "stage Q, dO' (L2)", "pass B (wave 0)", "store drain"]
print("k_sn_bwd phases, median / p90 shader cycles per workgroup (2100 cycles ~ 1 us):")
for i, nm in enumerate(names):
    print(f"  {nm:22s} {int(np.median(d[:, i])):8d} {int(np.percentile(d[:, i], 90)):8d}")
print("  workgroup life         ", int(np.median(x[:, 9] - x[:, 0])))
if os.environ.get("TRACE_OUT"):
    os.makedirs(os.environ["TRACE_OUT"], exist_ok=True)
    np.save(os.path.join(os.environ["TRACE_OUT"], "trace_sn.npy"), x)
