#!/bin/bash
# Build an experimental variant of the library next to the shipped one: tools/build_variant.sh NAME [-DFOO=1 ...]
#   -> mhla_amd/lib/variants/libmhla_NAME.so   (git-ignored; travels with gpurun).  Timed with tools/time_variants.py.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p "$ROOT/mhla_amd/lib/variants"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -shared -fPIC -Wall -Wno-unused-function \
  -Xclang -target-feature -Xclang -packed-fp32-ops "$@" "$ROOT/mhla_amd/csrc/capi.hip" -o "$ROOT/mhla_amd/lib/variants/libmhla_$name.so"
echo "built $name"
