#!/usr/bin/env python
"""Compact trace of a kernel's memory waits from a device ISA listing (tools/isa_dump.sh writes /tmp/<unit>.s):
L = global load, S = global store, W<n> = s_waitcnt vmcnt(n), M = a run of MFMAs, B = s_barrier, { / } = loop header label /
backward branch.
A `W0` right after a `{` or between the `L`s of a loop is the signature of a software pipeline that hipcc collapsed (loads behind
branches, or a prologue whose load order differs from the loop's): usage: tools/isa_waits.py file.s name-filter"""
import re
import subprocess
import sys


def main():
    path, filt = sys.argv[1], sys.argv[2]
    cur, out, labels = None, {}, {}
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            labels = {}
            continue
        if cur is None:
            continue
        t = line.strip()
        if t.startswith("s_endpgm"):
            cur = None
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            labels[m.group(1)] = len(out[cur])
            out[cur].append("{" if "Loop Header" in line else "")
            continue
        if t.startswith("global_load") or t.startswith("buffer_load"):
            out[cur].append("L")
        elif t.startswith("global_store") or t.startswith("buffer_store"):
            out[cur].append("S")
        elif t.startswith("v_mfma"):
            if not out[cur] or out[cur][-1] != "M":
                out[cur].append("M")   # (a run of matrix instructions, collapsed)
        elif t.startswith("s_barrier"):
            out[cur].append("B")
        elif t.startswith("s_waitcnt"):
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m:
                out[cur].append("W" + m.group(1))
        elif t.startswith("s_cbranch") or t.startswith("s_branch"):
            tgt = t.split()[-1]
            if tgt in labels:
                out[cur].append("}")
    names = [n for n in out if out[n]]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    for n, d in zip(names, dem):
        if re.search(filt, d):
            print(d)
            print("   ", " ".join(x for x in out[n] if x))


if __name__ == "__main__":
    main()
