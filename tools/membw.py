#!/usr/bin/env python
"""Achievable HBM bandwidth on this box with plain torch kernels (copy, read-only reduction, fill) -- the practical ceiling
the operator's kernels are compared with in DESIGN.md."""
import torch

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

for mb in (64, 256, 1024, 4096):
    x = torch.empty(mb * 1024 * 1024 // 4, dtype=torch.float32, device="cuda").normal_()
    y = torch.empty_like(x)
    tc = t(lambda: y.copy_(x))
    tr = t(lambda: x.sum())
    tf = t(lambda: y.fill_(1.0))
    ta = t(lambda: torch.add(x, y, out=y))
    b = x.numel() * 4
    print(f"{mb:5d} MB: copy {2 * b / tc / 1e12:.2f} TB/s  read(sum) {b / tr / 1e12:.2f} TB/s  fill {b / tf / 1e12:.2f} TB/s  add(2r+1w) {3 * b / ta / 1e12:.2f} TB/s")
