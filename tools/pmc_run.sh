#!/bin/bash
# Collect rocprofv3 PMC counters for bench.py's kernels in separate passes (counters only: no trace domains
# besides --kernel-trace).  Usage: tools/pmc_run.sh <outdir> [bench args...]
# PMC_SCRIPT=<path relative to the repo root> profiles that script instead of bench.py (its own arguments follow <outdir>).
set -u
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
if [ -n "${PMC_SCRIPT:-}" ]; then CMD="$ROOT/$PMC_SCRIPT $*"; else CMD="$ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs $*"; fi
run() {  # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 $CMD > "$OUT/$name.log" 2>&1
}
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS
run grbm GRBM_GUI_ACTIVE
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1
find "$OUT" -name "*.csv" | head -50
