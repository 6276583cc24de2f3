#!/usr/bin/env python
"""Per-kernel times (library HIP-event hook) and the step time of the causal operator at the two C5 shapes (and any
`B,T,H,K,V` given on the command line) -- the A/B tool for the causal pipeline (`split` / `bf16` on the command line pick the summary format; default both)."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench_configs import causal_case, kernel_times  # noqa: E402,F401
import mhla_amd  # noqa: E402
from mhla_amd import causal_mixing_init  # noqa: E402


def case(B, T, H, K, V, summaries="tf32"):
    r = causal_case(f"causal B={B} T={T} H={H} K={K} V={V} bf16 {summaries}", B, T, H, K, V, torch.bfloat16, summaries=summaries)
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B, T, H, K, generator=g).to(torch.bfloat16).cuda().requires_grad_(True)
    k = torch.randn(B, T, H, K, generator=g).to(torch.bfloat16).cuda().requires_grad_(True)
    v = torch.randn(B, T, H, V, generator=g).to(torch.bfloat16).cuda().requires_grad_(True)
    do = torch.randn(B, T, H, V, generator=g).to(torch.bfloat16).cuda()
    n = (T + 63) // 64
    mix = causal_mixing_init(n).reshape(n, n).cuda().requires_grad_(True)

    def step():
        out = mhla_amd.mhla_causal(q, k, v, mix, summaries=summaries)
        out.backward(do)
    ks = kernel_times(step)
    print(json.dumps({"shape": r["shape"], "ms": round(r["ms"], 4), "ms_graph_replay": r.get("ms_graph_replay"), "hbm_frac": round(r["hbm_frac"], 4),
                      "kernels_us": {k_: round(v_, 1) for k_, v_ in sorted(ks.items(), key=lambda x: -x[1])}}))


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a not in ("tf32", "split", "bf16")]
    modes = [a for a in sys.argv[1:] if a in ("tf32", "split", "bf16")] or ["tf32", "split", "bf16"]
    shapes = [tuple(int(x) for x in a.split(",")) for a in args] or [(4, 8192, 4, 128, 256), (2, 8192, 4, 256, 512), (16, 2048, 4, 128, 256)]
    for s in shapes:
        for m in modes:
            case(*s, summaries=m)
