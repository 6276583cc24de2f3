#!/usr/bin/env python
"""Block-level number for the C4 configuration: one Wan2.1-1.3B transformer block (thin in-repo shell,
mhla_amd/hosts/wan.py) forward under no_grad, bf16 weights, B x 31500 video tokens (21 x 30 x 50), 512 context tokens.
  python tools/bench_wan_block.py [--B 1] [--iters 20] [--lepe] [--layers 30]
`--layers 30` times the whole 30-block stack of Wan2.1-1.3B (one denoising step's transformer body; `--B 2` = with classifier-free
guidance, as the sampler batches it)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mhla_amd import modules  # noqa: E402
from mhla_amd.hosts import WanAttentionBlock_MHLA, WanStack_MHLA  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=1)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--lepe", action="store_true")
ap.add_argument("--layers", type=int, default=1)
a = ap.parse_args()
DEV = "cuda"
torch.manual_seed(0)
dim, heads, grid = 1536, 12, (21, 30, 50)
N = grid[0] * grid[1] * grid[2]
blk = (WanAttentionBlock_MHLA(dim=dim, ffn_dim=8960, num_heads=heads, is_lepe=a.lepe) if a.layers == 1 else
       WanStack_MHLA(num_layers=a.layers, dim=dim, ffn_dim=8960, num_heads=heads, is_lepe=a.lepe)).to(DEV).to(torch.bfloat16).eval()
x = torch.randn(a.B, N, dim, device=DEV, dtype=torch.bfloat16)
e = torch.randn(a.B, 6, dim, device=DEV, dtype=torch.float32) * 0.1
ctx = torch.randn(a.B, 512, dim, device=DEV, dtype=torch.bfloat16)
grid_sizes = torch.tensor([list(grid)] * a.B, dtype=torch.long)
seq_lens = torch.tensor([N] * a.B)
freqs = modules.wan_freqs(dim // heads)


def step():
    with torch.no_grad():
        return blk(x, e, seq_lens, grid_sizes, freqs, ctx)


for _ in range(3):
    step()
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(a.iters):
    step()
t1.record()
torch.cuda.synchronize()
ms = t0.elapsed_time(t1) / a.iters
print(json.dumps({"what": f"Wan2.1-1.3B {a.layers}-block forward (thin host), bf16, no_grad", "B": a.B, "tokens": N, "lepe": a.lepe,
                  "layers": a.layers, "ms_total": ms, "ms_per_block": ms / a.layers, "tokens_per_s": a.B * N / ms * 1e3}))
