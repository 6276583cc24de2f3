#!/bin/bash
# Round-end evidence for every BASELINE.json shape: rocprofv3 --kernel-trace --stats and the PMC passes (FETCH_SIZE / WRITE_SIZE,
# SQ busy / wait, MFMA busy, LDS conflicts -- separate runs, counters only) of bench.py (C2) and of tools/run_shape.py for
# c3 c3f c4b c5 c5b c2b c2c  ->  gpurun_out/${ROUND}prof/<shape>_{trace,pmc}.md and <shape>_pmc.json (ROUND defaults to r5;
# tools/collect_profiles.py copies them to profiles/${ROUND}_* and writes profiles/${ROUND}_pmc_traffic.json).
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/prof_all.sh'            SHAPES="c5 c5b" limits the set
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ROUND=${ROUND:-r5}
OUT=$ROOT/gpurun_out/${ROUND}prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for s in ${SHAPES:-c2 c3 c3f c4b c5 c5b c2b c2c}; do
  if [ "$s" = c2 ]; then
    rocprofv3 --kernel-trace --stats -d "$OUT/$s.trace" -o res -- python3 "$ROOT/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-extra-configs --no-dit-step --no-step-benches > "$OUT/$s.trace.log" 2>&1
    title="bench.py --steps 20 --warmup 3 (C2: B=8 N=4096 H=16 D=64 bf16 M=64)"
    bash "$ROOT/tools/pmc_run.sh" "$OUT/$s.pmc" --no-dit-step --no-step-benches > /dev/null 2>&1
  else
    rocprofv3 --kernel-trace --stats -d "$OUT/$s.trace" -o res -- python3 "$ROOT/tools/run_shape.py" $s 10 > "$OUT/$s.trace.log" 2>&1
    title="tools/run_shape.py $s 10"
    PMC_SCRIPT="tools/run_shape.py" bash "$ROOT/tools/pmc_run.sh" "$OUT/$s.pmc" $s 3 > /dev/null 2>&1
  fi
  db=$(find "$OUT/$s.trace" -name "*.db" | head -1)
  python3 "$ROOT/tools/rocprof_summary.py" "$db" "$OUT/${s}_trace.md" "$title, ${ROUND} code" > /dev/null 2>&1
  python3 "$ROOT/tools/pmc_summary.py" "$OUT/$s.pmc" --md "$OUT/${s}_pmc.md" --json "$OUT/${s}_pmc.json" > /dev/null 2>&1
  rm -rf "$OUT/$s.trace" "$OUT/$s.pmc"
  echo "== $s"; tail -2 "$OUT/${s}_pmc.md"
done
