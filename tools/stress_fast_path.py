"""Stress of the bf16 fast path's in-launch hand-over (k_t16_bwd: a tile's dK/dV workgroup waits for its dQ workgroup's flag):
many shapes -- tiles that are partly empty, a single tile per (b,h), few and many (b,h) pairs -- each run REPS times on two
streams at once.  Every repetition has fresh inputs and its own reference (the same backward as two launches,
mhla_set_option("bwd_two_launches", 1), alone on the device): a waiter that passed early cannot hide behind identical recycled workspace
contents.  The library's status call is made after every backward (MHLA_CHECK_HANDOVER=1).  Exits non-zero on any mismatch.
  timeout 300 python tools/stress_fast_path.py"""
import itertools
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mhla_amd  # noqa: E402

DEV = "cuda"
REPS = int(os.environ.get("REPS", "5"))
gen = torch.Generator(device=DEV).manual_seed(11)
side = torch.cuda.Stream()
bad = 0
cases = [(B, H, M, S) for (B, H), (M, S) in itertools.product([(1, 1), (1, 3), (2, 16), (8, 16), (33, 7)],
                                                               [(3, 5), (8, 64), (16, 16), (17, 33), (33, 64), (64, 64), (16, 256), (64, 128)])]
for B, H, M, S in cases:
    N = M * S
    if B * H * N > 8 * 16 * 4096 * 2:
        continue

    def mk(relu):
        t = torch.randn(B, N, H, 64, device=DEV, dtype=torch.bfloat16, generator=gen)
        return t.relu_().add_(1e-3) if relu else t

    for rep in range(REPS):
        q, k, v, do = mk(True), mk(True), mk(False), mk(False)
        W = torch.rand(M, M, device=DEV, generator=gen).add_(0.1)

        def run(beside):
            if beside:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):   # a second instance, other data, concurrently
                    t2 = [x.flip(0).clone().requires_grad_(True) for x in (q, k, v)]
                    mhla_amd.mhla_blockmix(*t2, W, summaries="bf16").backward(do)
            ts = [x.clone().requires_grad_(True) for x in (q, k, v)]
            Wg = W.clone().requires_grad_(True)
            out = mhla_amd.mhla_blockmix(*ts, Wg, summaries="bf16")
            out.backward(do)
            torch.cuda.synchronize()
            return [out.detach(), ts[0].grad, ts[1].grad, ts[2].grad, Wg.grad]

        mhla_amd._lib.load().mhla_set_option(b"bwd_two_launches", 1)
        ref = run(False)
        mhla_amd._lib.load().mhla_set_option(b"bwd_two_launches", 0)
        junk = torch.full((1 << 26,), float("nan"), device=DEV)   # recycled workspace blocks come back as NaN
        del junk
        os.environ["MHLA_CHECK_HANDOVER"] = "1"
        res = run(True)
        del os.environ["MHLA_CHECK_HANDOVER"]
        if not all(bool(torch.isfinite(r.float()).all()) for r in res):
            print("non-finite result", (B, H, M, S)); bad += 1
        if not all(bool(torch.equal(a, b)) for a, b in zip(ref, res)):
            print("MISMATCH", (B, H, M, S), "rep", rep); bad += 1
    print("ok" if bad == 0 else "..", (B, H, M, S), flush=True)
print("stress:", "all repetitions identical" if bad == 0 else f"{bad} problems")
sys.exit(1 if bad else 0)
