#!/bin/bash
# rocprofv3 --kernel-trace --stats of the non-C2 BASELINE shapes (tools/run_shape.py) -> gpurun_out/shapes/<shape>.md
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/shapes
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for s in ${SHAPES:-c3 c4b c5 xl}; do
  rocprofv3 --kernel-trace --stats -d "$OUT/$s" -o res -- python3 "$ROOT/tools/run_shape.py" $s 10 > "$OUT/$s.log" 2>&1
  db=$(find "$OUT/$s" -name "*.db" | head -1)
  python3 "$ROOT/tools/rocprof_summary.py" "$db" "$OUT/$s.md" "run_shape.py $s (10 iterations)" > /dev/null 2>&1
  find "$OUT/$s" -type f ! -name "*.md" -delete
done
cat "$OUT"/*.md
