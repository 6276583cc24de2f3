#!/usr/bin/env python
"""Phase timeline of the tile workgroups of the bf16 fast path (k_t16_out; the dQ and dK/dV roles of k_t16_bwd) at C2.

Uses the library's debugging hook (mhla_debug_set_trace): wave 0 of every workgroup stamps s_memtime at
  0 start | 1 own-block loads issued | 2 mixing done | 3 barrier passed | 4 first block done | 5 second block done | 6 stores drained
Prints, per kernel, the median / p10 / p90 duration of each phase (microseconds at TICKS_PER_US = 2100 shader clocks) and the
workgroup start-time distribution (dispatch rounds; indicative only, the counters of different CUs are not synchronised)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mhla_amd  # noqa: E402
from mhla_amd import _lib, block_distance_weights  # noqa: E402

B, N, H, D, M = 8, 4096, 16, 64, 64
if len(sys.argv) > 1:
    B = int(sys.argv[1])
dev = "cuda"
g = torch.Generator().manual_seed(1)
ts = [torch.randn(B, N, H, D, generator=g).abs().bfloat16().to(dev).requires_grad_(True) for _ in range(3)]
do = torch.randn(B, N, H, D, generator=g).bfloat16().to(dev)
W = block_distance_weights((8, 8), "linear").to(dev).requires_grad_(True)
lib = _lib.load()
nwg = ((M + 7) // 8 + 1) // 2 * B * H
SL = 16


def step():
    mhla_amd.mhla_blockmix(ts[0], ts[1], ts[2], W).backward(do)
    for t in ts + [W]:
        t.grad = None


for _ in range(3):
    step()
torch.cuda.synchronize()
buf = torch.zeros(3 * nwg * SL, dtype=torch.int64, device=dev)
lib.mhla_debug_set_trace(buf.data_ptr())
lib.mhla_prof_enable(1)
step()
torch.cuda.synchronize()
lib.mhla_prof_enable(0)
lib.mhla_debug_set_trace(None)
import ctypes  # noqa: E402
rep = ctypes.create_string_buffer(1 << 16)
lib.mhla_prof_report(rep, len(rep))
kus = {ln.split()[0]: float(ln.split()[2]) * 1e3 / int(ln.split()[1]) for ln in rep.value.decode().splitlines() if ln.strip()}
print("kernel durations (HIP events, us):", {k: round(v, 1) for k, v in kus.items()})
t = buf.cpu().numpy().astype(np.uint64).reshape(3, nwg, SL)
out_dir = os.environ.get("TRACE_OUT")
if out_dir:
    os.makedirs(out_dir, exist_ok=True)
    np.save(os.path.join(out_dir, "trace.npy"), t)
names = ["k_t16_out", "k_t16_bwd: dQ tiles", "k_t16_bwd: dK/dV tiles"]   # the two backward roles share one launch
phases = ["issue own loads", "mix", "barrier wait", "block A", "block B", "store drain"]
for kI, nm in enumerate(names):
    x = t[kI].astype(np.int64)
    xcc = (t[kI][:, 15] >> np.uint64(32)).astype(np.int64)
    # s_memtime counters are per XCD: spans and start offsets are taken within each XCD
    spans = [float(x[xcc == c, 6].max() - x[xcc == c, 0].min()) for c in range(8) if (xcc == c).any()]
    span = float(np.median(spans))
    # s_memtime counts shader clocks (about 2100 per us under this load); the counters of different CUs are not synchronised,
    # so only differences within one workgroup are meaningful
    tick_us = 1.0 / float(os.environ.get("TICKS_PER_US", "2100"))
    print(f"== {nm}: {nwg} workgroups, {kus.get(nm.split(':')[0], float('nan')):.1f} us by HIP events (whole launch)")
    d = np.diff(x[:, :7], axis=1) * tick_us
    life = (x[:, 6] - x[:, 0]) * tick_us
    print(f"   workgroup life median {np.median(life):.2f} p10 {np.percentile(life, 10):.2f} p90 {np.percentile(life, 90):.2f} us")
    for p, pn in enumerate(phases):
        print(f"   {pn:16s} median {np.median(d[:, p]):7.2f}  p10 {np.percentile(d[:, p], 10):7.2f}  p90 {np.percentile(d[:, p], 90):7.2f} us")
    rel = np.concatenate([(x[xcc == c, 0] - x[xcc == c, 0].min()) * tick_us for c in range(8) if (xcc == c).any()])
    hist, edges = np.histogram(rel, bins=12)
    print("   start-time histogram (us since the XCD's first workgroup):", " ".join(f"{edges[i]:.0f}:{hist[i]}" for i in range(len(hist))))
    print("   workgroups per XCC:", np.bincount(xcc, minlength=8).tolist(), " blockIdx%8==xcc for", int((xcc == (np.arange(nwg) % 8)).sum()), "of", nwg)
