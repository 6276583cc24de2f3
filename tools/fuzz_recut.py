#!/usr/bin/env python
"""k_sp_mixh2 against k_sp_mixh (mhla_set_option("recut_kernels", 0)) on odd block counts, head dims and block lengths, with and without the
normaliser, negative weights: every output and gradient bit for bit.  python tools/fuzz_recut.py"""
import itertools, sys, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import mhla_amd
bad = 0
cases = [(129, 64, 16), (150, 32, 16), (200, 40, 16), (255, 96, 16), (256, 72, 16), (131, 64, 30), (192, 56, 18), (256, 64, 17), (177, 80, 16), (140, 48, 64)]
for M, D, S in cases:
    for norm in (True, False):
        B, H = 4, 24
        N = M * S
        g = torch.Generator().manual_seed(M * 7 + D)
        q, k, v = (torch.randn(B, N, H, D, generator=g).bfloat16().cuda() for _ in range(3))
        W = (torch.rand(M, M, generator=g) - 0.3).cuda()
        do = torch.randn(B, N, H, D, generator=g).bfloat16().cuda()
        def run():
            ts = [t.clone().requires_grad_(True) for t in (q, k, v, W)]
            out = mhla_amd.mhla_blockmix(ts[0].abs(), ts[1].abs(), ts[2], ts[3], normalize=norm)
            out.backward(do)
            torch.cuda.synchronize()
            return [out.detach()] + [t.grad for t in ts]
        d = mhla_amd.describe_dispatch(B, H, M, S, D, torch.bfloat16)
        new = run()
        prev = mhla_amd.set_option("recut_kernels", 0)
        old = run()
        mhla_amd.set_option("recut_kernels", prev)
        eq = [torch.equal(a, b) for a, b in zip(new, old)]
        fin = all(torch.isfinite(a.float()).all().item() for a in new)
        print(M, D, S, norm, d["summaries"][:4], d["fwd"][1], eq, fin)
        bad += (not all(eq)) or (not fin)
print("BAD" if bad else "ALL EQUAL")
