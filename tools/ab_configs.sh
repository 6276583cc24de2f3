#!/bin/bash
# A/B of library variants (tools/build_variant.sh) over tools/bench_configs.py: one line per shape and variant with the per-kernel times.
#   tools/ab_configs.sh NAME [NAME ...]      ("shipped" = the in-tree library)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for v in "$@"; do
  if [ "$v" = shipped ]; then unset MHLA_LIB_PATH; else export MHLA_LIB_PATH=$ROOT/mhla_amd/lib/variants/libmhla_$v.so; fi
  python3 "$ROOT/tools/bench_configs.py" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print('$v'.ljust(8), j['shape'][:58].ljust(58), '%.4f' % j['ms'], ' '.join('%s=%.1f' % (k.replace('k_sp_', '').replace('k_', ''), v) for k, v in sorted(j.get('kernel_us', {}).items())))
"
done
