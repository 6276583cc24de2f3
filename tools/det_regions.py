#!/usr/bin/env python
"""Which intermediate of the bf16 fast path's backward differs between two identical runs?  Repeats the C2-shaped backward and
compares every region of the library's workspace (layout of fast_carve in csrc/capi.hip) bit for bit with the first
repetition -- the tool that located the source of the packed-fp32 nondeterminism.
  MHLA_LIB_PATH=mhla_amd/lib/variants/libmhla_packed.so python tools/det_regions.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mhla_amd  # noqa: E402
from mhla_amd import ops, block_distance_weights  # noqa: E402

DEV = "cuda"
B, N, H, D, M = int(os.environ.get("B", "8")), 4096, 16, 64, 64
S = N // M
gen = torch.Generator(device=DEV).manual_seed(7)
mk = lambda relu: (lambda t: t.relu_().add_(1e-3) if relu else t)(torch.randn(B, N, H, D, device=DEV, dtype=torch.bfloat16, generator=gen))
q, k, v, do = mk(True), mk(True), mk(False), mk(False)
W = block_distance_weights((8, 8), "linear").to(DEV)

made = []
_orig = ops._ws
ops._ws = lambda nbytes, device: made.append(_orig(nbytes, device)) or made[-1]

al4 = lambda n: (n + 3) & ~3
bh, njg = B * H, (M + 7) // 8
st = bh * njg * 4096 * 8 * 2
sizes = [("state", st), ("z", al4(bh * M * S) * 4), ("ksum", al4(bh * M * 64) * 4), ("ninv", al4(bh * M * S) * 4),
         ("dstate", st), ("dn", al4(bh * M * S) * 4), ("dz", al4(bh * M * S) * 4), ("dksum", al4(bh * M * 64) * 4),
         ("dwp", bh * 8 * 4096 * 4)]
ref = None
for rep in range(int(os.environ.get("REPS", "4"))):
    made.clear()
    ts = [x.clone().requires_grad_(True) for x in (q, k, v)]
    Wg = W.clone().requires_grad_(True)
    out = mhla_amd.mhla_blockmix(*ts, Wg)
    out.backward(do)
    torch.cuda.synchronize()
    fwd_ws, bwd_ws = made[0], made[-1]
    raw = bwd_ws.view(torch.uint8)
    fraw = fwd_ws.view(torch.uint8)
    regs, off = {}, 0
    for name, sz in sizes:
        src = fraw if name in ("state", "z", "ksum", "ninv") else raw   # the backward reuses the forward's workspace for these
        regs[name] = src[off:off + sz].clone()
        off += sz
    regs.update(out=out.detach().view(torch.uint8).flatten().clone(), dq=ts[0].grad.view(torch.uint8).flatten().clone(),
                dk=ts[1].grad.view(torch.uint8).flatten().clone(), dv=ts[2].grad.view(torch.uint8).flatten().clone())
    if ref is None:
        ref = regs
        continue
    line = []
    for name in regs:
        nd = int((regs[name] != ref[name]).sum())
        line.append(f"{name}={'same' if nd == 0 else f'DIFF({nd}B)'}")
    print("rep", rep, " ".join(line))
    if os.environ.get("DZ_DETAIL") and rep == 1:
        a = ref["dz"].view(torch.float32).reshape(bh, M, S)
        b = regs["dz"].view(torch.float32).reshape(bh, M, S)
        d = (a != b).nonzero()
        print("differing dz elements:", d.shape[0], "of", a.numel())
        import collections
        print("by s % 4:", collections.Counter((d[:, 2] % 4).tolist()))
        print("by s // 16:", collections.Counter((d[:, 2] // 16).tolist()))
        print("by r % 8:", collections.Counter((d[:, 1] % 8).tolist()))
        rel = ((a - b).abs() / a.abs().clamp_min(1e-30))[a != b]
        print("relative differences: max %.3e median %.3e" % (rel.max().item(), rel.median().item()))
        for i in range(min(8, d.shape[0])):
            x, y, z = d[i].tolist()
            print((x, y, z), a[x, y, z].item(), b[x, y, z].item(), hex(a[x, y, z].view(torch.int32).item() & 0xffffffff), hex(b[x, y, z].view(torch.int32).item() & 0xffffffff))
