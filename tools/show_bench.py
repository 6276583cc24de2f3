#!/usr/bin/env python
"""Print the headline fields of a bench.py JSON line (file argument)."""
import json
import sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: r[k] for k in ("value", "ms_per_step", "n_gpus")})
rf = r["roofline"]
print({k: rf.get(k) for k in ("achieved", "frac", "frac_from_gpu_events", "traffic", "traffic_over_algorithmic", "mfma_busy_frac_pmc")})
print("dominant:", rf["dominant_kernel"])
print("kernels:", {k: round(v["us_per_step"], 1) for k, v in rf["kernels"].items()})
for key in ("dit_xl2_train_step", "north_star_c3", "cpu_baseline", "gpu_over_cpu"):
    print(key + ":", r.get(key))
for e in r.get("extra_configs", []) if isinstance(r.get("extra_configs"), list) else []:
    print(f"{e['shape'][:78]:78s} {e['ms']:.4f} ms  frac {e['hbm_frac']:.4f}  graph {e.get('ms_graph_replay')}  dom {e['dominant_kernel']} {e['dominant_kernel_us']:.1f}")
