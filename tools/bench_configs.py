#!/usr/bin/env python
"""Time the operator on the other BASELINE.json shapes (C3 DiT-XL/2, C4 Wan forward, C5 fla causal) with HIP events.
Informational (bench.py stays on C2); prints one JSON line per shape."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mhla_amd  # noqa: E402
from mhla_amd import block_distance_weights, block_index_3d, causal_mixing_init  # noqa: E402

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
    return e0.elapsed_time(e1) / iters * 1e-3


def blockmix_case(name, B, N, H, D, M, dtype, layout, bwd=True, split=False, idx=None, normalize=True):
    g = torch.Generator().manual_seed(1)
    mk = lambda relu: ((torch.relu(torch.randn(B, N, H, D, generator=g)) + 1e-6) if relu else torch.randn(B, N, H, D, generator=g)).to(dtype).to(DEV)
    q, k, v, do = mk(True), mk(True), mk(False), mk(False)
    qd = kd = None
    if split:
        qd, kd = mk(True), mk(True)
    W = block_distance_weights(layout, "linear").to(DEV)
    if bwd:
        for t in (q, k, v, W):
            t.requires_grad_(True)

    def step():
        out = mhla_amd.mhla_blockmix(q, k, v, W, q_den=qd, k_den=kd, normalize=normalize, block_index=idx)
        if bwd:
            out.backward(do)
            q.grad = k.grad = v.grad = W.grad = None

    t = timeit(step)
    esz = q.element_size()
    nde = B * H * N * D * esz
    alg = (12 if bwd else (6 if split else 4)) * nde
    print(json.dumps({"shape": name, "what": "fwd+bwd" if bwd else "fwd", "ms": t * 1e3, "tokens_per_s": B * N / t,
                      "algorithmic_GBps": alg / t / 1e9, "hbm_frac": alg / t / 8e12}))


def causal_case(name, B, T, H, K, V, dtype):
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B, T, H, K, generator=g).to(dtype).to(DEV).requires_grad_(True)
    k = torch.randn(B, T, H, K, generator=g).to(dtype).to(DEV).requires_grad_(True)
    v = torch.randn(B, T, H, V, generator=g).to(dtype).to(DEV).requires_grad_(True)
    do = torch.randn(B, T, H, V, generator=g).to(dtype).to(DEV)
    n = (T + 63) // 64
    mix = causal_mixing_init(n).reshape(n, n).to(DEV).requires_grad_(True)

    def step():
        out = mhla_amd.mhla_causal(q, k, v, mix)
        out.backward(do)
        q.grad = k.grad = v.grad = mix.grad = None

    t = timeit(step, iters=10)
    esz = q.element_size()
    alg = 3 * B * H * T * (2 * K + 2 * V) * esz
    print(json.dumps({"shape": name, "what": "fwd+bwd", "ms": t * 1e3, "tokens_per_s": B * T / t,
                      "algorithmic_GBps": alg / t / 1e9, "hbm_frac": alg / t / 8e12}))


if __name__ == "__main__":
    blockmix_case("C3 DiT-XL/2 op B=32 N=256 H=16 D=72 M=16 bf16", 32, 256, 16, 72, 16, torch.bfloat16, (4, 4))
    blockmix_case("C3 DiT-XL/2 op B=32 N=256 H=16 D=72 M=16 fp32", 32, 256, 16, 72, 16, torch.float32, (4, 4))
    blockmix_case("C1/DiT-S/2-shaped op B=32 N=256 H=6 D=64 M=16 bf16 (fast path)", 32, 256, 6, 64, 16, torch.bfloat16, (4, 4))
    idx = block_index_3d((21, 30, 50), (3, 5, 10)).to(DEV)
    blockmix_case("C4 Wan fwd B=1 N=31500 H=12 D=128 M=150 fp32 split, un-normalised (shipped YAML)", 1, 31500, 12, 128, 150,
                  torch.float32, (3, 5, 10), bwd=False, split=False, idx=idx, normalize=False)
    blockmix_case("C4 Wan fwd B=1 N=31500 H=12 D=128 M=150 fp32 split, normalised", 1, 31500, 12, 128, 150,
                  torch.float32, (3, 5, 10), bwd=False, split=True, idx=idx)
    blockmix_case("C4 Wan fwd+bwd B=1 N=31500 H=12 D=128 M=150 fp32 split, normalised", 1, 31500, 12, 128, 150,
                  torch.float32, (3, 5, 10), bwd=True, split=True, idx=idx)
    blockmix_case("DiT-XL/2 512x512 op B=16 N=1024 H=16 D=72 M=16 bf16 (generic)", 16, 1024, 16, 72, 16, torch.bfloat16, (4, 4))
    blockmix_case("DiT-S/2 2048x2048 op B=4 N=16384 H=6 D=64 M=64 S=256 bf16 (fast path)", 4, 16384, 6, 64, 64, torch.bfloat16, (8, 8))
    blockmix_case("DiT-S/2 4096x4096 op B=1 N=65536 H=6 D=64 M=256 S=256 bf16 (split path: M > 64)", 1, 65536, 6, 64, 256, torch.bfloat16, (16, 16))
    blockmix_case("C2 variant M=16 S=256 bf16 (fast path, multi-chunk blocks)", 8, 4096, 16, 64, 16, torch.bfloat16, (4, 4))
    blockmix_case("C2 variant M=256 S=16 bf16 (split path: M > 64)", 8, 4096, 16, 64, 256, torch.bfloat16, (16, 16))
    causal_case("C5 1.3B-like causal B=2 T=8192 H=4 K=256 V=512 bf16", 2, 8192, 4, 256, 512, torch.bfloat16)
    causal_case("C5 fla 340M causal B=4 T=8192 H=4 K=128 V=256 bf16", 4, 8192, 4, 128, 256, torch.bfloat16)
