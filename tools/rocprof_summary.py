#!/usr/bin/env python
"""Turn a rocprofv3 --kernel-trace --stats result (rocpd sqlite .db) into a small markdown summary
for profiles/.  Usage: tools/rocprof_summary.py <results.db> <out.md> [title]"""
import sqlite3
import sys


def main():
    db_path, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else db_path
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    res = {r[0]: r[1:] for r in db.execute(
        "select name, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x*grid_y*grid_z/ (workgroup_x*workgroup_y*workgroup_z)) from kernels group by name")}
    with open(out, "w") as f:
        f.write(f"# {title}\n\nrocprofv3 --kernel-trace --stats (durations in microseconds; `top_kernels` view of the rocpd database)\n\n")
        f.write("| kernel | calls | total us | avg us | % | VGPR | AGPR | SGPR | LDS B | workgroups |\n|---|---|---|---|---|---|---|---|---|---|\n")
        for name, calls, tot, avg, pct in rows:
            v = res.get(name, ("", "", "", "", ""))
            f.write(f"| `{name}` | {calls} | {tot:.1f} | {avg:.1f} | {pct:.1f} | {v[0]} | {v[1]} | {v[2]} | {v[3]} | {v[4]} |\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
