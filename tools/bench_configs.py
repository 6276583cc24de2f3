#!/usr/bin/env python
"""Time the operator on the other BASELINE.json shapes (C3 DiT-XL/2, C4 Wan, C5 fla causal, the C2 variants) with HIP events
and name each shape's dominant kernel (library per-launch event hook).  `python tools/bench_configs.py` prints one JSON line
per shape; bench.py imports `run_extra_configs()` and attaches the list to its JSON line as `extra_configs` (after its timed
region), so the driver's BENCH record carries these numbers too."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mhla_amd  # noqa: E402
from mhla_amd import block_distance_weights, block_index_3d, causal_mixing_init  # noqa: E402

DEV = "cuda"
HBM_PEAK = 8e12


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


def kernel_times(fn, iters=5):
    """{kernel: us per step} from the library's per-launch HIP-event hook."""
    lib = mhla_amd._lib.load()
    torch.cuda.synchronize()
    lib.mhla_prof_enable(1)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    lib.mhla_prof_enable(0)
    buf = ctypes.create_string_buffer(1 << 16)
    lib.mhla_prof_report(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, tot = line.rsplit(" ", 2)
        out[name] = float(tot) / iters * 1e3
    return out


_TRAFFIC = None


def pmc_traffic(shape_key):
    """HBM bytes per step of a shape from the committed rocprofv3 PMC summary (profiles/r*_pmc_traffic.json, written by
    tools/prof_all.sh + tools/collect_profiles.py) -- only when that file was made from the kernel sources of this build."""
    global _TRAFFIC
    if _TRAFFIC is None:
        import glob
        import hashlib
        root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
        h = hashlib.sha256()
        for fn in sorted(glob.glob(os.path.join(root, "mhla_amd", "csrc", "*"))):
            h.update(open(fn, "rb").read())
        _TRAFFIC = {}
        for pj in sorted(glob.glob(os.path.join(root, "profiles", "r*_pmc_traffic.json")), reverse=True):
            try:
                rec = json.load(open(pj))
            except Exception:   # noqa: BLE001
                continue
            if rec.get("csrc_sha16") == h.hexdigest()[:16] and "shapes" in rec:
                _TRAFFIC = {k: (v["hbm_bytes_per_step"], os.path.basename(pj)) for k, v in rec["shapes"].items()}
                break
    return _TRAFFIC.get(shape_key, (None, None))


def _result(name, what, t, tokens, alg, step, graph_t=None, key=None):
    ks = kernel_times(step)
    dom = max(ks, key=ks.get) if ks else None
    r = {"shape": name, "what": what, "ms": t * 1e3, "tokens_per_s": tokens / t, "algorithmic_GBps": alg / t / 1e9,
         "hbm_frac": alg / t / HBM_PEAK, "kernel_us_per_step": sum(ks.values()),
         "dominant_kernel": dom, "dominant_kernel_us": ks.get(dom) if dom else None, "n_kernels": len(ks), "kernel_us": ks}
    if graph_t is not None:
        r["ms_graph_replay"] = graph_t * 1e3
        r["hbm_frac_graph_replay"] = alg / graph_t / HBM_PEAK
    traffic, src = pmc_traffic(key) if key else (None, None)
    r["algorithmic_bytes_per_step"] = alg
    r["traffic"] = traffic                      # HBM bytes per step by PMC counters (null without a same-source profile)
    r["traffic_over_algorithmic"] = traffic / alg if traffic else None
    r["traffic_source"] = src
    return r


def _graph_time(step, iters=20):
    """The same step captured once in a HIP graph and replayed (no Python on the launch path); None if capture is refused."""
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                step()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        return timeit(g.replay, iters=iters)
    except Exception:   # noqa: BLE001
        return None


def blockmix_case(name, B, N, H, D, M, dtype, layout, bwd=True, split=False, idx=None, normalize=True, iters=20, graph=False, key=None,
                  summaries="tf32"):
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
        out = mhla_amd.mhla_blockmix(q, k, v, W, q_den=qd, k_den=kd, normalize=normalize, block_index=idx, summaries=summaries)
        if bwd:
            out.backward(do)
            q.grad = k.grad = v.grad = W.grad = None

    t = timeit(step, iters=iters)
    nde = B * H * N * D * q.element_size()
    alg = (12 if bwd else (6 if split else 4)) * nde      # fwd 4 NDe (6 with split q/k), bwd 8 NDe
    r = _result(name, "fwd+bwd" if bwd else "fwd", t, B * N, alg, step, _graph_time(step, iters) if graph else None, key)
    if dtype != torch.float32:
        r["arithmetic"] = ("fp32-grade intermediates (fp32 block summaries / bf16 hi + lo operands and score tiles: the reference's fp32 arithmetic)"
                           if summaries == "split" else "REDUCED PRECISION: single-bf16 block summaries, dP and score tiles (opt-in)")
    return r


def causal_case(name, B, T, H, K, V, dtype, iters=10, key=None, summaries="tf32", graph=True):
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B, T, H, K, generator=g).to(dtype).to(DEV).requires_grad_(True)
    k = torch.randn(B, T, H, K, generator=g).to(dtype).to(DEV).requires_grad_(True)
    v = torch.randn(B, T, H, V, generator=g).to(dtype).to(DEV).requires_grad_(True)
    do = torch.randn(B, T, H, V, generator=g).to(dtype).to(DEV)
    n = (T + 63) // 64
    mix = causal_mixing_init(n).reshape(n, n).to(DEV).requires_grad_(True)

    def step():
        out = mhla_amd.mhla_causal(q, k, v, mix, summaries=summaries)
        out.backward(do)
        q.grad = k.grad = v.grad = mix.grad = None

    t = timeit(step, iters=iters)
    alg = 3 * B * H * T * (2 * K + 2 * V) * q.element_size()   # fwd reads q, k, v, writes o; bwd twice that
    r = _result(name, "fwd+bwd", t, B * T, alg, step, _graph_time(step, iters) if graph else None, key)
    r["arithmetic"] = ("bf16 hi + lo chunk summaries and score tiles (>= 16 significand bits: the reference's fp32 arithmetic)"
                       if summaries == "split" else "REDUCED PRECISION: single-bf16 chunk summaries and score tiles (opt-in)")
    return r


def run_extra_configs(full=False):
    """The BASELINE.json shapes besides C2 (bench.py's main line).  `full` adds the informational shapes of DESIGN.md 3f."""
    out = []
    bf, f32 = torch.bfloat16, torch.float32
    out.append(blockmix_case("C3 DiT-XL/2 256^2 op B=32 N=256 H=16 D=72 M=16 bf16", 32, 256, 16, 72, 16, bf, (4, 4), graph=True, key="c3"))
    out.append(blockmix_case("C3 DiT-XL/2 256^2 op B=32 N=256 H=16 D=72 M=16 fp32", 32, 256, 16, 72, 16, f32, (4, 4), graph=True, key="c3f"))
    idx = block_index_3d((21, 30, 50), (3, 5, 10)).to(DEV)
    out.append(blockmix_case("C4 Wan2.1-1.3B fwd B=1 N=31500 H=12 D=128 M=150 fp32, un-normalised (shipped YAML)", 1, 31500, 12, 128, 150,
                             f32, (3, 5, 10), bwd=False, split=False, idx=idx, normalize=False, graph=True))
    out.append(blockmix_case("C4 Wan2.1-1.3B fwd B=1 N=31500 H=12 D=128 M=150 fp32, normalised split q/k", 1, 31500, 12, 128, 150,
                             f32, (3, 5, 10), bwd=False, split=True, idx=idx, graph=True))
    out.append(blockmix_case("C4 Wan2.1-1.3B fwd+bwd B=1 N=31500 H=12 D=128 M=150 fp32, normalised split q/k", 1, 31500, 12, 128, 150,
                             f32, (3, 5, 10), bwd=True, split=True, idx=idx, iters=10, key="c4b", graph=True))
    out.append(causal_case("C5 fla 340M causal B=4 T=8192 H=4 K=128 V=256 bf16", 4, 8192, 4, 128, 256, bf, key="c5"))
    out.append(causal_case("C5 1.3B-like causal B=2 T=8192 H=4 K=256 V=512 bf16", 2, 8192, 4, 256, 512, bf, key="c5b"))
    # the opt-in reduced-precision variant of the causal operator (single-bf16 chunk summaries: round 3's arithmetic), for
    # comparison only -- the lines above are the numbers of record
    out.append(causal_case("C5 fla 340M causal B=4 T=8192 H=4 K=128 V=256 bf16 [reduced_precision: summaries=bf16]", 4, 8192, 4, 128, 256, bf,
                           summaries="bf16"))
    out.append(blockmix_case("C2 variant M=16 S=256 bf16", 8, 4096, 16, 64, 16, bf, (4, 4), graph=True, key="c2c"))
    out.append(blockmix_case("C2 variant M=256 S=16 bf16", 8, 4096, 16, 64, 256, bf, (16, 16), iters=10, key="c2b", graph=True))
    if full:
        out.append(blockmix_case("DiT-S/2-shaped op B=32 N=256 H=6 D=64 M=16 bf16", 32, 256, 6, 64, 16, bf, (4, 4), graph=True))
        out.append(blockmix_case("DiT-XL/2 512^2 op B=16 N=1024 H=16 D=72 M=16 bf16", 16, 1024, 16, 72, 16, bf, (4, 4)))
        out.append(blockmix_case("DiT-S/2 2048^2 op B=4 N=16384 H=6 D=64 M=64 S=256 bf16", 4, 16384, 6, 64, 64, bf, (8, 8)))
        out.append(blockmix_case("DiT-S/2 4096^2 op B=1 N=65536 H=6 D=64 M=256 S=256 bf16", 1, 65536, 6, 64, 256, bf, (16, 16)))
    return out


if __name__ == "__main__":
    for r in run_extra_configs(full=True):
        print(json.dumps(r))
