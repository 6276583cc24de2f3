#!/bin/bash
# Module-level MFMA figure (tools/module_mfma.py): kernel trace + two counter-only PMC passes of the DiT-XL/2 attention module fwd+bwd
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/prof_module_mfma.sh'   ->  gpurun_out/r5_module/{module_dit.md,module_dit.json}
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r5_module
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/tools/module_mfma.py" run 20 > "$OUT/timing.json" 2> "$OUT/timing.err"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$ROOT/tools/module_mfma.py" run 5 > "$OUT/trace.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d "$OUT/sq1" -- python3 "$ROOT/tools/module_mfma.py" run 2 > "$OUT/sq1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d "$OUT/sq2" -- python3 "$ROOT/tools/module_mfma.py" run 2 > "$OUT/sq2.log" 2>&1
python3 "$ROOT/tools/module_mfma.py" summarise "$OUT" "$OUT/module_dit.md" "$OUT/module_dit.json"
find "$OUT" -name "*.csv" -delete; find "$OUT" -type d -empty -delete
