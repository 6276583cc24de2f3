#!/usr/bin/env python
"""The dispatch table of DESIGN.md section 0a, printed from the library's own predicates (mhla_describe_dispatch /
mhla_causal_describe_dispatch: pure host logic, no GPU needed):  python tools/dispatch_table.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mhla_amd  # noqa: E402

bf, f32 = torch.bfloat16, torch.float32
ROWS = [
    ("C2 micro-bench (configs[1]): B=8 N=4096 H=16 D=64, M=64 S=64", "bf16 / fp16", "default", (8, 16, 64, 64, 64, bf), {}),
    ("same", "bf16 / fp16", '`summaries="split"` (`MHLA_FLAG_FP32_GRADE_SUMMARIES`)', (8, 16, 64, 64, 64, bf), {"summaries": "split"}),
    ("same", "bf16", '`summaries="bf16"` (`MHLA_FLAG_BF16_SUMMARIES`)', (8, 16, 64, 64, 64, bf), {"summaries": "bf16"}),
    ("same", "fp32", "any", (8, 16, 64, 64, 64, f32), {}),
    ("C2 variant M=16 S=256", "bf16", "default", (8, 16, 16, 256, 64, bf), {}),
    ("C2 variant M=256 S=16 (more than 128 blocks)", "bf16", "default", (8, 16, 256, 16, 64, bf), {}),
    ("same", "bf16", '`summaries="bf16"`', (8, 16, 256, 16, 64, bf), {"summaries": "bf16"}),
    ("C3 DiT-XL/2 256² (configs[2]): B=32 N=256 H=16 D=72, M=16 S=16", "bf16", "default", (32, 16, 16, 16, 72, bf), {}),
    ("same", "fp32", "any", (32, 16, 16, 16, 72, f32), {}),
    ("DiT-XL/2 512²: N=1024, M=16 S=64, D=72", "bf16", "default", (16, 16, 16, 64, 72, bf), {}),
    ("C4 Wan2.1-1.3B (configs[3]): B=1 N=31500 H=12 D=128, M=150 S=210 (training path; inference: `mhla_blockmix_wan_pro_fwd`, same kernels with the prologue on load)",
     "fp32", "any", (1, 12, 150, 210, 128, f32), {"split": True}),
    ("blocks of < 16 tokens, or < 4 blocks, or D < 32", "bf16 / fp16", "default", (2, 4, 64, 8, 64, bf), {}),
    ("D % 8 != 0 (e.g. D = 36), or `force_generic`", "any", "any", (2, 2, 16, 16, 36, f32), {}),
]
CAUSAL = [
    ("C5 fla 340M (configs[4]): T=8192 H=4 K=128 V=256 (128 chunks); 1.3B-like K=256 V=512", "default", (8192, 128, 256, bf), {}),
    ("same", '`summaries="split"` (`MHLA_CAUSAL_FP32_GRADE_SUMMARIES`)', (8192, 128, 256, bf), {"summaries": "split"}),
    ("same", '`summaries="bf16"` (`MHLA_CAUSAL_BF16_SUMMARIES`)', (8192, 128, 256, bf), {"summaries": "bf16"}),
    ("129 .. 256 chunks", "default", (16384, 64, 64, bf), {}),
    ("fp32 / fp16 tensors; K or V not multiples of 64; K > 256; more than 256 chunks", "any", (8192, 128, 256, f32), {}),
]
arrow = lambda ks: " → ".join("`" + k + "`" for k in ks)
print("| configuration | tensors | flags | kernel family | block / chunk summaries in HBM | forward launches | backward launches |\n|---|---|---|---|---|---|---|")
for name, dt, fl, a, kw in ROWS:
    d = mhla_amd.describe_dispatch(*a, **kw)
    print(f"| {name} | {dt} | {fl} | {d['family']} | {d['summaries']} | {arrow(d['fwd'])} | {arrow(d['bwd'])} |")
for name, fl, a, kw in CAUSAL:
    d = mhla_amd.describe_causal_dispatch(*a, **kw)
    print(f"| causal: {name} | {'bf16' if a[3] is bf else 'fp32'} | {fl} | {d['family']} | {d['summaries']} | {arrow(d['fwd'])} | {arrow(d['bwd'])} |")
