#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv) per kernel: mean counter value per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            if "mhla" not in k: continue
            k = k.split("(")[0].replace("void mhla::", "")
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("==", f.split("/")[-3] if f.count("/") > 2 else f)
    for k in sorted(agg):
        print("  ", k, {c: round(sum(v) / len(v), 1) for c, v in agg[k].items()}, "n=%d" % len(next(iter(agg[k].values()))))
