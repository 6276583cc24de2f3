#!/bin/bash
# Build an experimental variant of the library next to the shipped one: tools/build_variant.sh NAME [-DFOO=1 ...]
#   -> mhla_amd/lib/variants/libmhla_NAME.so   (git-ignored; travels with gpurun).  Timed with tools/time_variants.py.
# Same translation units and flags as mhla_amd/build.py (its own object cache under lib/variants/obj_NAME).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p "$ROOT/mhla_amd/lib/variants"
cd "$ROOT"
MHLA_BUILD_DEFINES="$*" python - "$name" <<'PY'
import sys
from mhla_amd import build as b
name = sys.argv[1]
import os
out = os.path.join(b.LIB_DIR, "variants", f"libmhla_{name}.so")
b.build(force=True, lib=out, obj_dir=os.path.join(b.LIB_DIR, "variants", f"obj_{name}"))
print("built", out)
PY
