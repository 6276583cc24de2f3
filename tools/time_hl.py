#!/usr/bin/env python
"""Block-mixing shapes on bf16 tensors at both arithmetics (default fp32-grade intermediates / opt-in summaries="bf16"), with the
per-kernel breakdown: `python tools/time_hl.py [c2 c3 c2b c2c xl512]`."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_configs import blockmix_case  # noqa: E402

SHAPES = {
    "c2": ("C2 B=8 N=4096 H=16 D=64 M=64", 8, 4096, 16, 64, 64, (8, 8)),
    "c3": ("C3 B=32 N=256 H=16 D=72 M=16", 32, 256, 16, 72, 16, (4, 4)),
    "c2b": ("C2b M=256 S=16", 8, 4096, 16, 64, 256, (16, 16)),
    "c2c": ("C2c M=16 S=256", 8, 4096, 16, 64, 16, (4, 4)),
    "xl512": ("DiT-XL/2 512^2 B=16 N=1024 H=16 D=72 M=16", 16, 1024, 16, 72, 16, (4, 4)),
}

if __name__ == "__main__":
    from mhla_amd import _lib
    ab = "--ab" in sys.argv   # also time the default arithmetic with fp32 summaries in the workspace (mhla_set_option "fp32_summaries")
    keys = [k for k in sys.argv[1:] if not k.startswith("--")] or ["c2", "c3"]
    for key in keys:
        name, B, N, H, D, M, layout = SHAPES[key]
        for summ in ("split", "split-fp32", "bf16") if ab else ("split", "bf16"):
            _lib.load().mhla_set_option(b"fp32_summaries", 1 if summ == "split-fp32" else 0)
            r = blockmix_case(f"{name} bf16 [{summ}]", B, N, H, D, M, torch.bfloat16, layout, graph=True, summaries=summ.split("-")[0])
            print(json.dumps({k: r[k] for k in ("shape", "ms", "ms_graph_replay", "hbm_frac", "kernel_us_per_step", "kernel_us")}), flush=True)
