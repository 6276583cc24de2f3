#!/usr/bin/env python
"""Wan2.1-1.3B 30-block forward (BASELINE.json configs[3]) on one GPU: ms per forward and the MHLA kernels' share, per kernel
(tools/bench_steps.wan_forward; what bench.py reports as `wan_1p3b_forward`)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench_steps  # noqa: E402

r = bench_steps.wan_forward(torch.device("cuda", 0), layers=30, iters=2, warm=1, batches=(1,))
print(json.dumps(r))
