#!/usr/bin/env python
"""Per-kernel times of the C2 step for every library variant under mhla_amd/lib/variants/ (tools/build_variant.sh), each in its
own process (MHLA_LIB_PATH), next to the shipped library:  python tools/time_variants.py [name ...]"""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# "env:NAME=VALUE" arguments time the shipped library under that environment setting (e.g. env:MHLA_TILE=811)
names = sys.argv[1:] or sorted(os.path.basename(p)[len("libmhla_"):-3] for p in glob.glob(os.path.join(ROOT, "mhla_amd/lib/variants/libmhla_*.so")))
rows = []
for nm in ["shipped"] + names:
    env = dict(os.environ)
    if nm.startswith("env:"):
        k_, v_ = nm[4:].split("=", 1)
        env[k_] = v_
    elif nm != "shipped":
        env["MHLA_LIB_PATH"] = os.path.join(ROOT, "mhla_amd/lib/variants", f"libmhla_{nm}.so")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extra-configs", "--no-step-benches", "--steps", "40"],
                         env=env, capture_output=True, text=True)
    try:
        j = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception:
        print(nm, "FAILED", out.stderr[-400:])
        continue
    k = j["roofline"]["kernels"]
    rows.append((nm, j["roofline"]["gpu_us_per_step"], {n: v["us_per_step"] for n, v in k.items()}))
keys = sorted({n for _, _, k in rows for n in k})
print("variant".ljust(18), "step".rjust(7), " ".join(n.replace("k_", "")[:12].rjust(12) for n in keys))
for nm, st, k in rows:
    print(nm.ljust(18), f"{st:7.1f}", " ".join(f"{k.get(n, 0):12.1f}" for n in keys))
