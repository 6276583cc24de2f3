#!/bin/bash
# rocprofv3 --kernel-trace --stats of a drop-in module (tools/run_module.py) -> gpurun_out/modules/<name>.md
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/modules
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for s in "$@"; do
  python3 "$ROOT/tools/run_module.py" $s 10
  rocprofv3 --kernel-trace --stats -d "$OUT/$s" -o res -- python3 "$ROOT/tools/run_module.py" $s 10 > "$OUT/$s.log" 2>&1
  db=$(find "$OUT/$s" -name "*.db" | head -1)
  python3 "$ROOT/tools/rocprof_summary.py" "$db" "$OUT/$s.md" "run_module.py $s (12 calls)" > /dev/null 2>&1
  find "$OUT/$s" -type f ! -name "*.md" -delete
  cut -c1-150 "$OUT/$s.md" | head -40
done
