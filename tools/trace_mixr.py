#!/usr/bin/env python
"""Slice-loop timeline of k_sp_mixr_dma at the 256 x 16 shape (library hook mhla_debug_set_trace): wave 0 of the first eight workgroups
stamps s_memtime at  0 loop top | 1 copy of slice k complete + barrier | 2 stores of slice k - 1 and request of slice k + 2 issued |
3 products + staging done.  Prints the median length of each phase over slices 4..27, in ticks and as a share of the iteration."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mhla_amd  # noqa: E402
from mhla_amd import _lib  # noqa: E402

B, N, H, D, M = 8, 4096, 16, 64, 256
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
for wg in range(8):
    x = t[wg]
    it = np.diff(x[:, 0])[4:27]
    ph = [x[4:28, 1] - x[4:28, 0], x[4:28, 2] - x[4:28, 1], x[4:28, 3] - x[4:28, 2]]
    print(f"wg {wg}: iteration {np.median(it):7.0f} ticks | vmcnt wait {np.median(x[4:28, 4] - x[4:28, 0]):6.0f} barrier {np.median(x[4:28, 1] - x[4:28, 4]):6.0f} | stores+request {np.median(ph[1]):6.0f} | products+staging {np.median(ph[2]):6.0f}"
          f" | whole loop {x[31, 3] - x[0, 0]} ticks")
