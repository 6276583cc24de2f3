#!/usr/bin/env python
"""Per-kernel instruction census of a device ISA listing (hipcc --cuda-device-only -S): flat vs global loads, scratch, MFMA,
LDS transposes, s_waitcnt vmcnt(0) -- the quick check that a kernel's loads stayed on the global path.
usage: tools/isa_stats.py capi.s [name-filter]"""
import re, sys, subprocess

def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))

def main():
    path, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    cur, stats, meta = None, {}, {}
    keys = ["flat_load", "global_load", "flat_store", "global_store", "scratch_", "v_mfma", "ds_read_b64_tr", "ds_read", "ds_write", "s_barrier", "v_pk_"]
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1); stats[cur] = dict.fromkeys(keys, 0); stats[cur]["vmcnt0"] = 0; continue
        if cur is None: continue
        s = line.strip()
        if s.startswith(".end_amdhsa_kernel") or s.startswith(".Lfunc_end"): 
            pass
        for k in keys:
            if s.startswith(k): stats[cur][k] += 1
        if s.startswith("s_waitcnt") and "vmcnt(0)" in s: stats[cur]["vmcnt0"] += 1
        # metadata keys of a kernel are sorted: .name comes BEFORE .private_segment_fixed_size / .sgpr_count / .vgpr_count / ...
        m = re.match(r"\.name:\s+(_Z\w+)", s)
        if m: meta["name"] = m.group(1); meta[m.group(1)] = {}
        m = re.match(r"\.(vgpr_count|sgpr_count|vgpr_spill_count|private_segment_fixed_size):\s+(\d+)", s)
        if m and "name" in meta: meta[meta["name"]][m.group(1)] = int(m.group(2))
    dm = demangle(list(stats))
    print("| kernel | VGPR | spill | scratch B | flat ld | global ld | flat st | global st | scratch | mfma | tr | ds_r | ds_w | bar | vmcnt(0) | v_pk |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for k, st in stats.items():
        name = dm.get(k, k)
        if filt not in name: continue
        if st["v_mfma"] == 0 and st["global_load"] + st["flat_load"] == 0: continue
        mt = meta.get(k, {})
        short = re.sub(r"\(.*", "", name).replace("void ", "")
        print(f"| `{short}` | {mt.get('vgpr_count','?')} | {mt.get('vgpr_spill_count','?')} | {mt.get('private_segment_fixed_size','?')} | {st['flat_load']} | {st['global_load']} | {st['flat_store']} | {st['global_store']} | {st['scratch_']} | {st['v_mfma']} | {st['ds_read_b64_tr']} | {st['ds_read']} | {st['ds_write']} | {st['s_barrier']} | {st['vmcnt0']} | {st['v_pk_']} |")

if __name__ == "__main__":
    main()
