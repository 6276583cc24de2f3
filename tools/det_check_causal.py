"""Run-to-run determinism check of the causal operator: forward + backward REPS times on the C5 shape (B=4 T=8192 H=4 K=128 V=256 bf16,
default arithmetic), every output compared bit for bit with the first repetition.  Exits non-zero on any difference.
  python tools/det_check_causal.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mhla_amd  # noqa: E402

REPS = int(os.environ.get("REPS", "5"))
B, T, H, K, V = 4, 8192, 4, 128, 256
g = torch.Generator().manual_seed(3)
mk = lambda *s: torch.randn(*s, generator=g).bfloat16().cuda()  # noqa: E731
q, k, v, do = mk(B, T, H, K), mk(B, T, H, K), mk(B, T, H, V), mk(B, T, H, V)
n = T // 64
mix = mhla_amd.causal_mixing_init(n).reshape(n, n).cuda()
ref, bad = None, 0
for rep in range(REPS):
    qq, kk, vv, mm = (t.clone().requires_grad_(True) for t in (q, k, v, mix))
    out = mhla_amd.mhla_causal(qq, kk, vv, mm)
    out.backward(do)
    cur = [out.detach(), qq.grad, kk.grad, vv.grad, mm.grad]
    if ref is None:
        ref = [c.clone() for c in cur]
    same = [torch.equal(a, b) for a, b in zip(cur, ref)]
    bad += same.count(False)
    print("rep", rep + 1, " ".join(f"{name}={'same' if s else 'DIFF'}" for name, s in zip(("out", "dq", "dk", "dv", "dmix"), same)))
print("causal: deterministic" if not bad else "causal: NOT deterministic")
sys.exit(1 if bad else 0)
