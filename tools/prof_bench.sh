#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench.py command (C2) -> gpurun_out/bench_prof/summary.md, then the PMC passes
# (tools/pmc_run.sh) -> gpurun_out/bench_pmc/{summary.md,summary.json}
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/bench_prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/run" -o res -- python3 "$ROOT/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-extra-configs > "$OUT/bench.log" 2>&1
db=$(find "$OUT/run" -name "*.db" | head -1)
python3 "$ROOT/tools/rocprof_summary.py" "$db" "$OUT/summary.md" "bench.py --steps 20 --warmup 3 (C2: B=8 N=4096 H=16 D=64 bf16 M=64), round-2 code" > /dev/null 2>&1
find "$OUT/run" -type f -delete
bash "$ROOT/tools/pmc_run.sh" "$ROOT/gpurun_out/bench_pmc" > /dev/null 2>&1
python3 "$ROOT/tools/pmc_summary.py" "$ROOT/gpurun_out/bench_pmc" --md "$ROOT/gpurun_out/bench_pmc/summary.md" --json "$ROOT/gpurun_out/bench_pmc/summary.json" > /dev/null 2>&1
find "$ROOT/gpurun_out/bench_pmc" -name "*.csv" -delete
cat "$OUT/summary.md"; tail -3 "$ROOT/gpurun_out/bench_pmc/summary.md"; tail -1 "$OUT/bench.log" | cut -c1-200
