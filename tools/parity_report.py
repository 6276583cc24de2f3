#!/usr/bin/env python
"""profiles/r*_parity_errors.md from gpurun_out/parity_report.json (written by tests/conftest.py at the end of a `-m gpu` session):
the observed errors grouped by the tolerance each comparison ran under.
  python tools/parity_report.py [gpurun_out/parity_report.json] > profiles/r6_parity_errors.md
Also writes profiles/r6_parity_summary.json: the report without its per-comparison list (what bench.py's `targets` block reads)."""
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_report.json")
rep = json.load(open(path))
rows = rep["all"]
tests = {r["test"] for r in rows}
by_tol = collections.defaultdict(list)
for r in rows:
    by_tol[r["tol"]].append(r)
print(f"# Observed parity errors, round 6 (`python -m pytest tests -m gpu` on one MI355X, {len(tests)} tests with recorded comparisons, "
      f"{len(rows)} comparisons)\n")
print("Every `check()` of the GPU suite records `max|got - want| / max|want|` (the rel-err the north-star bound is stated in) and fla's\n"
      "rms-relative error; `tests/conftest.py` writes them to `gpurun_out/parity_report.json` at the end of the session and\n"
      "`tools/parity_report.py` makes this table.  Grouped by the tolerance the comparison ran under (i.e. by dtype / quantity class);\n"
      "`want` is the CPU oracle or a reference fixture.\n")
print("| tolerance in the test | comparisons | largest rel-err | largest rms ratio | worst case |")
print("|---|---|---|---|---|")
for tol in sorted(by_tol):
    rs = by_tol[tol]
    w = max(rs, key=lambda r: r["rel_err"])
    name = w["test"].split("::")[-1]
    print(f"| {tol:g} | {len(rs)} | {w['rel_err']:.2e} | {max(r['rms_ratio'] for r in rs):.2e} | `{name}` :: {w['name']} ({w['dtype']}) |")
print()
agg = rep["by_dtype_and_kind"]
print("| dtype / kind | comparisons | largest rel-err | largest error beyond the final rounding (max(|err| - u|want|) / max|want|) | loosest tolerance used |")
print("|---|---|---|---|---|")
for k in sorted(agg):
    a = agg[k]
    print(f"| {k} | {a['n']} | {a['max_rel_err']:.2e} | {a.get('max_beyond_final_rounding', float('nan')):.2e} | {a['loosest_tol']:g} |")

fam = rep.get("by_baseline_config", {})
if fam:
    print("\n| BASELINE.json configuration (full-size tests) | comparisons | 16-bit results: largest error beyond the final rounding | fp32 results: largest rel-err | worst case |")
    print("|---|---|---|---|---|")
    for k in sorted(fam):
        a = fam[k]
        print(f"| {k} | {a['n']} | {a['results_16bit_max_beyond_final_rounding']:.2e} | {a['results_fp32_max_rel_err']:.2e} | `{a['worst']}` |")
    with open(os.path.join(ROOT, "profiles", "r6_parity_summary.json"), "w") as f:
        json.dump({k: v for k, v in rep.items() if k != "all"}, f, indent=1)
