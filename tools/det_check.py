"""Run-to-run determinism check of the block-mixing operator: forward + backward REPS times on the C2 shape, every
output compared bit for bit with the first repetition:
  python tools/det_check.py              # one stream
  STREAMS=2 python tools/det_check.py    # a second instance of the operator runs concurrently on another stream
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mhla_amd  # noqa: E402
from mhla_amd import block_distance_weights  # noqa: E402

DEV = "cuda"
B, N, H, D = 132, 4096, 16, 64
REPS = int(os.environ.get("REPS", "6"))
gen = torch.Generator(device=DEV).manual_seed(7)


def mk(relu):
    t = torch.randn(B, N, H, D, device=DEV, dtype=torch.bfloat16, generator=gen)
    return t.relu_().add_(1e-3) if relu else t


q, k, v, do = mk(True), mk(True), mk(False), mk(False)
W = block_distance_weights((8, 8), "linear").to(DEV).requires_grad_(True)
res = []
two = os.environ.get("STREAMS", "1") == "2"
side = torch.cuda.Stream()
if two:
    q2, k2, v2, do2 = mk(True)[:32], mk(True)[:32], mk(False)[:32], mk(False)[:32]
for rep in range(REPS):
    if two:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            t2 = [x.clone().requires_grad_(True) for x in (q2, k2, v2)]
            mhla_amd.mhla_blockmix(*t2, W.detach()).backward(do2)
    for t in (q, k, v):
        t.requires_grad_(True)
        t.grad = None
    W.grad = None
    out = mhla_amd.mhla_blockmix(q, k, v, W)
    out.backward(do)
    torch.cuda.synchronize()
    res.append([out.detach().clone(), q.grad.clone(), k.grad.clone(), v.grad.clone(), W.grad.clone()])
names = ["out", "dq", "dk", "dv", "dW"]
bad = 0
for r in range(1, REPS):
    eq = [bool(torch.equal(a, b)) for a, b in zip(res[0], res[r])]
    bad += sum(not e for e in eq)
    print("rep", r, " ".join(f"{n}={'same' if e else 'DIFF'}" for n, e in zip(names, eq)))
print("streams:", 2 if two else 1, "-> deterministic" if bad == 0 else f"-> {bad} mismatching tensors")
sys.exit(1 if bad else 0)
