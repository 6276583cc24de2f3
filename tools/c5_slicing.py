#!/usr/bin/env python
"""Verdict r4 item 2: does slicing the causal pipeline over (b, h) groups -- so that a group's chunk summaries (S, P; dP, dS) fit the
256 MB memory-side cache between their producer and consumer kernels -- shorten the step?  Every variant is captured in ONE HIP graph
(launch count is free) and replayed; the slices are views of the same tensors (no copies), dmix accumulates over the slices.
usage: python tools/c5_slicing.py [340m|1p3b]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mhla_amd  # noqa: E402
from mhla_amd import causal_mixing_init  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def graph_of(step):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    return g


def run(name, B, T, H, K, V):
    g = torch.Generator().manual_seed(1)
    bf = torch.bfloat16
    q = torch.randn(B, T, H, K, generator=g).to(bf).to(DEV).requires_grad_(True)
    k = torch.randn(B, T, H, K, generator=g).to(bf).to(DEV).requires_grad_(True)
    v = torch.randn(B, T, H, V, generator=g).to(bf).to(DEV).requires_grad_(True)
    do = torch.randn(B, T, H, V, generator=g).to(bf).to(DEV)
    n = (T + 63) // 64
    mix = causal_mixing_init(n).reshape(n, n).to(DEV).requires_grad_(True)
    set_bytes = B * H * n * K * V * 4   # one summary set (hi + lo planes)

    def make(bs, hs, interleave):
        """slices of bs batch elements x hs heads; interleave: forward and backward of a slice back to back (the loss between
        them is a plain sum here), otherwise all forwards first, then all backwards (a training step's order)"""
        sl = [(b0, h0) for b0 in range(0, B, bs) for h0 in range(0, H, hs)]

        def step():
            outs = []
            for b0, h0 in sl:
                o = mhla_amd.mhla_causal(q[b0:b0 + bs, :, h0:h0 + hs], k[b0:b0 + bs, :, h0:h0 + hs], v[b0:b0 + bs, :, h0:h0 + hs], mix)
                if interleave:
                    o.backward(do[b0:b0 + bs, :, h0:h0 + hs])
                else:
                    outs.append(o)
            for (b0, h0), o in zip(sl, outs):
                o.backward(do[b0:b0 + bs, :, h0:h0 + hs])
            q.grad = k.grad = v.grad = mix.grad = None
        return step, len(sl)

    rows = []
    for bs, hs in [(B, H), (max(B // 2, 1), H), (1, H), (1, max(H // 2, 1)), (1, 1)]:
        for inter in (False, True):
            step, ns = make(bs, hs, inter)
            try:
                ms = timeit(graph_of(step).replay)
            except Exception as e:   # noqa: BLE001
                ms = None
                print("capture failed", bs, hs, inter, repr(e)[:200], file=sys.stderr)
            rows.append({"shape": name, "slice_bh": bs * hs, "slices": ns, "fwd_bwd_back_to_back": inter,
                         "live_S_plus_P_MB": 2 * set_bytes * bs * hs / (B * H) / 1e6, "ms_graph_replay": ms})
            print(json.dumps(rows[-1]), flush=True)
    return rows


if __name__ == "__main__":
    which = sys.argv[1:] or ["340m", "1p3b"]
    if "340m" in which:
        run("C5 340M B=4 T=8192 H=4 K=128 V=256", 4, 8192, 4, 128, 256)
    if "1p3b" in which:
        run("C5 1.3B-like B=2 T=8192 H=4 K=256 V=512", 2, 8192, 4, 256, 512)
