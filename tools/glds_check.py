#!/usr/bin/env python
"""The hand-counted LDS-DMA copy of sp::k_sp_mixr_dma (common.hpp glds16: inline asm, outside hipcc's wait bookkeeping) against the
compiler-managed form: runs the 256 blocks x 16 tokens shape (bf16 summaries -- the one shape family that reaches that kernel) forward
+ backward with the shipped library and with the variant built by `tools/build_variant.sh glds -DMHLA_GLDS_BUILTIN=1`, each in its
own process, and demands bit-identical outputs and gradients.  `python tools/glds_check.py` (GPU)."""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import hashlib, sys, torch
sys.path.insert(0, %r)
import mhla_amd
from oracle import mhla_oracle as orc
g = torch.Generator().manual_seed(11)
B, H, M, S, D = 2, 16, 256, 16, 64
mk = lambda: torch.randn(B, M * S, H, D, generator=g).bfloat16().cuda()
q, k, v, do = torch.relu(mk()) + 1e-3, torch.relu(mk()) + 1e-3, mk(), mk()
W = orc.block_distance_weights((16, 16), "linear").cuda()
for rep in range(3):
    t = [x.clone().requires_grad_(True) for x in (q, k, v, W)]
    out = mhla_amd.mhla_blockmix(t[0], t[1], t[2], t[3], summaries="bf16")
    out.backward(do)
    torch.cuda.synchronize()
    print(rep, " ".join(hashlib.sha256(x.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()[:16] for x in [out] + [y.grad for y in t]))
"""


def run(lib):
    env = dict(os.environ)
    if lib:
        env["MHLA_LIB_PATH"] = lib
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, capture_output=True, text=True)
    if r.returncode:
        raise SystemExit(f"child failed ({lib or 'shipped'}):\n{r.stderr[-2000:]}")
    return [l for l in r.stdout.splitlines() if l[:1].isdigit()]


if __name__ == "__main__":
    var = os.path.join(ROOT, "mhla_amd", "lib", "variants", "libmhla_glds.so")
    if not os.path.exists(var):
        raise SystemExit("build the variant first: tools/build_variant.sh glds -DMHLA_GLDS_BUILTIN=1")
    a, b = run(None), run(var)
    for x, y in zip(a, b):
        print("asm    ", x)
        print("builtin", y)
    assert a == b and len(a) == 3, "the hand-counted LDS-DMA form and the compiler-managed form differ"
    print("bit-identical")
