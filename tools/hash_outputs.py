#!/usr/bin/env python
"""SHA-256 of the operator's outputs and gradients on seeded inputs, one line per shape: run it under two builds of the library
(MHLA_LIB_PATH=...) and compare the lines -- the check that a re-cut kernel is bit-identical to the one it replaces.
  python tools/hash_outputs.py [M ...]      (C2's tensors with M blocks; default 256 192 160 64)"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mhla_amd  # noqa: E402

Ms = [int(x) for x in sys.argv[1:]] or [256, 192, 160, 64]
for M in Ms:
    B, N, H, D = 8, 16 * M if M > 64 else 4096, 16, 64   # (128 (b, h): enough slices per workgroup for the resident mixing kernels' re-cut form)
    g = torch.Generator().manual_seed(M)
    q, k, v = (torch.randn(B, N, H, D, generator=g).bfloat16().cuda().requires_grad_(True) for _ in range(3))
    W = torch.rand(M, M, generator=g).cuda().requires_grad_(True)
    do = torch.randn(B, N, H, D, generator=g).bfloat16().cuda()
    out = mhla_amd.mhla_blockmix(q.abs(), k.abs(), v, W)
    out.backward(do)
    torch.cuda.synchronize()
    hs = [hashlib.sha256(t.detach().float().cpu().numpy().tobytes()).hexdigest()[:12] for t in (out, q.grad, k.grad, v.grad, W.grad)]
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(5):
        q.grad = k.grad = v.grad = W.grad = None
        mhla_amd.mhla_blockmix(q.abs(), k.abs(), v, W).backward(do)
    t1.record()
    torch.cuda.synchronize()
    print(f"M={M} N={N} ({t0.elapsed_time(t1) / 5:.3f} ms per fwd+bwd, autograd included): out {hs[0]} dq {hs[1]} dk {hs[2]} dv {hs[3]} dW {hs[4]} | {mhla_amd.describe_dispatch(B, H, M, N // M, D, torch.bfloat16)['family']}")
