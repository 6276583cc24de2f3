#!/usr/bin/env python
"""Step time and per-kernel times of the split path at the Wan fwd+bwd shape (c4b) and the C2 256 x 16 variant (c2b); MHLA_SP_MIX=old
keeps the tiled mixing kernel (A/B):  python tools/time_split.py [c4b] [c2b]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench_configs import blockmix_case  # noqa: E402
from mhla_amd import block_index_3d  # noqa: E402

which = sys.argv[1:] or ["c4b", "c2b"]
for w in which:
    if w == "c4b":
        idx = block_index_3d((21, 30, 50), (3, 5, 10)).cuda()
        r = blockmix_case("C4 Wan fwd+bwd", 1, 31500, 12, 128, 150, torch.float32, (3, 5, 10), bwd=True, split=True, idx=idx, iters=10)
    else:
        r = blockmix_case("C2 variant M=256 S=16 bf16", 8, 4096, 16, 64, 256, torch.bfloat16, (16, 16), iters=10)
    print(json.dumps({"shape": r["shape"], "ms": round(r["ms"], 4), "hbm_frac": round(r["hbm_frac"], 4),
                      "kernels_us": {k: round(v, 1) for k, v in sorted(r["kernel_us"].items(), key=lambda x: -x[1])}}))
