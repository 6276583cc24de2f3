#!/usr/bin/env python
"""gpurun_out/<round>prof/* (tools/prof_all.sh) -> profiles/<round>_<shape>_trace.md, profiles/<round>_<shape>_pmc.md and
profiles/<round>_pmc_traffic.json (round = argv[1], default r5): {csrc_sha16, shapes: {<shape>: {hbm_bytes_per_step, read, write, kernels: [...]}}} -- the file
bench.py / tools/bench_configs.py take `roofline.traffic` from when it was made from the kernel sources the library was built from."""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r5"
src = os.path.join(ROOT, "gpurun_out", f"{rnd}prof")
dst = os.path.join(ROOT, "profiles")
out = {"shapes": {}}
for js in sorted(glob.glob(os.path.join(src, "*_pmc.json"))):
    shape = os.path.basename(js)[:-len("_pmc.json")]
    rec = json.load(open(js))
    out["csrc_sha16"] = rec["csrc_sha16"]
    out["shapes"][shape] = {"hbm_bytes_per_step": rec["hbm_bytes_per_step"], "hbm_read_bytes_per_step": rec["hbm_read_bytes_per_step"],
                            "hbm_write_bytes_per_step": rec["hbm_write_bytes_per_step"], "kernels": rec["kernels"]}
    for kind in ("trace", "pmc"):
        f = os.path.join(src, f"{shape}_{kind}.md")
        if os.path.exists(f):
            shutil.copy(f, os.path.join(dst, f"{rnd}_{shape}_{kind}.md"))
out["note"] = ("rocprofv3 PMC passes of tools/prof_all.sh; per step; read = 2 x FETCH_SIZE (gfx950: 64 B counted per 128-B request), "
               "write = WRITE_SIZE; shapes: c2 = bench.py, others = tools/run_shape.py <shape>")
json.dump(out, open(os.path.join(dst, f"{rnd}_pmc_traffic.json"), "w"), indent=1)
print({k: round(v["hbm_bytes_per_step"] / 1e6) for k, v in out["shapes"].items()}, out.get("csrc_sha16"))
