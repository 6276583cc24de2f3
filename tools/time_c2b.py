#!/usr/bin/env python
"""Per-kernel times (library HIP-event hook) and the replayed-graph step time of the block-mixing operator at the C2 variant
(B=8 N=4096 H=16 D=64 bf16, 256 blocks of 16) or any `B,N,H,D,M` given on the command line -- the A/B tool for the
blocks-of-16 kernels (split16.hpp) and the resident mixing / dW kernels; MHLA_LIB_PATH picks a tools/build_variant.sh library."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench_configs import kernel_times  # noqa: E402
import mhla_amd  # noqa: E402


def case(B, N, H, D, M):
    g = torch.Generator().manual_seed(0)
    ts = [torch.randn(B, N, H, D, generator=g).abs().bfloat16().cuda().requires_grad_(True) for _ in range(3)]
    do = torch.randn(B, N, H, D, generator=g).bfloat16().cuda()
    W = torch.rand(M, M, generator=g).cuda().requires_grad_(True)

    def step():
        mhla_amd.mhla_blockmix(ts[0], ts[1], ts[2], W).backward(do)
        for t in ts + [W]:
            t.grad = None
    for _ in range(3):
        step()
    ks = kernel_times(step)
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        step()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            step()
        for _ in range(3):
            gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(20):
            gr.replay()
        e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    alg = 12 * B * N * H * D * 2
    print(json.dumps({"shape": f"blockmix B={B} N={N} H={H} D={D} M={M} bf16", "ms_graph_replay": round(ms, 4),
                      "hbm_frac": round(alg / (ms * 1e-3) / 8e12, 4), "kernels_us_sum": round(sum(ks.values()), 1),
                      "kernels_us": {k_: round(v_, 1) for k_, v_ in sorted(ks.items(), key=lambda x: -x[1])}}))


if __name__ == "__main__":
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(8, 4096, 16, 64, 256)]
    for s in shapes:
        case(*s)
