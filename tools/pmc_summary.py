#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv) per kernel: mean counter value per dispatch.
Usage: tools/pmc_summary.py <pmc_dir> [--md out.md] [--json out.json]
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide
coalesced reads, so read bytes = 2 x FETCH_SIZE (MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1]
data = collections.defaultdict(dict)
ndisp = collections.defaultdict(int)   # dispatches per kernel in one pass: kernels launched k times per step count k times
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            if "mhla" not in k:
                continue
            k = k.split("(")[0].replace("void ", "").replace("mhla::", "").replace("fast::", "")
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in agg:
        for c, v in agg[k].items():
            data[k][c] = sum(v) / len(v)
            ndisp[k] = max(ndisp[k], len(v))

rows = []
tot_r = tot_w = 0.0
base = min(ndisp.values()) if ndisp else 1
for k, d in sorted(data.items()):
    w = d.get("SQ_WAVES", 0)
    wc = d.get("SQ_WAVE_CYCLES", 0) * 4 / w if w else 0
    mult = max(1, round(ndisp[k] / base))   # launches of this kernel per step (k_csf_state: S and dP)
    rd = d.get("FETCH_SIZE", 0) * 2 * 1024
    wr = d.get("WRITE_SIZE", 0) * 1024
    tot_r += rd * mult
    tot_w += wr * mult
    hit = d.get("TCC_HIT_sum", 0)
    miss = d.get("TCC_MISS_sum", 0)
    rows.append({
        "kernel": k, "launches_per_step": mult, "waves": int(w), "us_per_wave": wc / 2.4e3 if wc else None,
        "active": d.get("SQ_ACTIVE_INST_ANY", 0) * 4 / w / wc if wc else None,
        "wait_any": d.get("SQ_WAIT_ANY", 0) * 4 / w / wc if wc else None,
        "wait_inst": d.get("SQ_WAIT_INST_ANY", 0) * 4 / w / wc if wc else None,
        "mfma_busy_cycles_per_wave": d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / w if w else None,
        "lds_conflict_frac": d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_LDS_IDX_ACTIVE"] if d.get("SQ_LDS_IDX_ACTIVE") else None,
        "hbm_read_bytes": rd, "hbm_write_bytes": wr, "l2_hit": hit / (hit + miss) if hit + miss else None,
    })
import hashlib
import os
_h = hashlib.sha256()
for _fn in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mhla_amd", "csrc", "*"))):
    _h.update(open(_fn, "rb").read())
out = {"csrc_sha16": _h.hexdigest()[:16], "kernels": rows, "hbm_read_bytes_per_step": tot_r, "hbm_write_bytes_per_step": tot_w,
       "hbm_bytes_per_step": tot_r + tot_w,
       "note": "per launch; the per-step totals count a kernel launches_per_step times; read = 2 x FETCH_SIZE KB (gfx950), write = WRITE_SIZE KB"}
if "--json" in sys.argv:
    json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
lines = ["| kernel | launches per step | waves | us/wave | active | wait_any (vmcnt/barrier) | wait_inst (issue) | MFMA busy cyc/wave | LDS conflict frac | HBM read MB | HBM write MB | L2 hit |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|"]
f2 = lambda x: "" if x is None else f"{x:.2f}"
for r in rows:
    lines.append(f"| `{r['kernel']}` | {r['launches_per_step']} | {r['waves']} | {f2(r['us_per_wave'])} | {f2(r['active'])} | {f2(r['wait_any'])} | {f2(r['wait_inst'])} | "
                 f"{'' if r['mfma_busy_cycles_per_wave'] is None else int(r['mfma_busy_cycles_per_wave'])} | {f2(r['lds_conflict_frac'])} | "
                 f"{r['hbm_read_bytes'] / 1e6:.1f} | {r['hbm_write_bytes'] / 1e6:.1f} | {f2(r['l2_hit'])} |")
lines.append(f"\nTotal HBM traffic per step: read {tot_r / 1e6:.0f} MB + write {tot_w / 1e6:.0f} MB = {(tot_r + tot_w) / 1e6:.0f} MB")
md = "\n".join(lines)
if "--md" in sys.argv:
    open(sys.argv[sys.argv.index("--md") + 1], "w").write(md + "\n")
print(md)
