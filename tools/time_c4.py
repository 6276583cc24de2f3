#!/usr/bin/env python
"""Per-kernel times of the C4 (Wan) forward, un-normalised and normalised with split q/k pairs, for the shipped library and the
variants under mhla_amd/lib/variants/ (tools/build_variant.sh):  python tools/time_c4.py [variant ...]"""
import ctypes
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, ROOT)
    import mhla_amd
    from mhla_amd import block_distance_weights, block_index_3d
    DEV = "cuda"
    g = torch.Generator().manual_seed(1)
    B, N, H, D = 1, 31500, 12, 128
    mk = lambda: torch.randn(B, N, H, D, generator=g).to(DEV)
    q, k, v, qd, kd = mk().abs(), mk().abs(), mk(), mk().abs(), mk().abs()
    W = block_distance_weights((3, 5, 10), "linear").to(DEV)
    idx = block_index_3d((21, 30, 50), (3, 5, 10)).to(DEV)
    lib = mhla_amd._lib.load()
    out = {}
    # the Wan layer's inference epilogue (wan/mhla_utils.py:356-362): fused into the output kernel vs the unfused composition
    gate = torch.randn(B, N, H, D, generator=g).to(torch.bfloat16).to(DEV)
    nw = (torch.rand(D, generator=g) + 0.5).to(DEV)
    fused = lambda: mhla_amd.mhla_blockmix_wan(q, k, v, W, None, None, nw, 1e-6, gate, torch.bfloat16, normalize=False, block_index=idx)
    unfused = lambda: mhla_amd.rmsnorm_gate(mhla_amd.mhla_blockmix(q, k, v, W, normalize=False, block_index=idx).to(torch.bfloat16), gate, nw, 1e-6)
    for name, fn in (("plain", lambda: mhla_amd.mhla_blockmix(q, k, v, W, normalize=False, block_index=idx)),
                     ("split", lambda: mhla_amd.mhla_blockmix(q, k, v, W, q_den=qd, k_den=kd, block_index=idx)),
                     ("wanfused", fused), ("unfused", unfused)):
        with torch.no_grad():
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            lib.mhla_prof_enable(1)
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            lib.mhla_prof_enable(0)
        buf = ctypes.create_string_buffer(1 << 16)
        lib.mhla_prof_report(buf, len(buf))
        ks = {}
        for line in buf.value.decode().splitlines():
            nm, cnt, tot = line.rsplit(" ", 2)
            ks[nm] = float(tot) / int(cnt) * 1e3
        if name in ("wanfused", "unfused"):   # wall time per call (the unfused path has a torch cast kernel the hook does not see)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.no_grad():
                ev0.record()
                for _ in range(10):
                    fn()
                ev1.record()
            torch.cuda.synchronize()
            ks["WALL_us_per_call"] = ev0.elapsed_time(ev1) * 100.0
        out[name] = ks
    print(json.dumps(out))
    sys.exit(0)
names = sys.argv[1:] or sorted(os.path.basename(p)[len("libmhla_"):-3] for p in glob.glob(os.path.join(ROOT, "mhla_amd/lib/variants/libmhla_*.so")))
for nm in ["shipped"] + names:
    env = dict(os.environ)
    if nm != "shipped":
        env["MHLA_LIB_PATH"] = os.path.join(ROOT, "mhla_amd/lib/variants", f"libmhla_{nm}.so")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        print(nm, "FAILED", r.stderr[-300:])
        continue
    for case, ks in j.items():
        print(f"{nm:14s} {case:8s} kernels {sum(v for k, v in ks.items() if k != 'WALL_us_per_call'):7.1f}  " + "  ".join(f"{k}={v:.1f}" for k, v in ks.items()))
