#!/bin/bash
# usage: tools/isa_dump.sh capi.hip [name-filter] -- device ISA listing of one translation unit with the library's flags, then the census
set -e
src=$1; filt=${2:-}
out=/tmp/$(basename "$src" .hip).s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -Xclang -target-feature -Xclang -packed-fp32-ops -DMHLA_BUILD_FLAGS='"x"' \
  -I"$(dirname "$0")/../include" --cuda-device-only -S -o "$out" "$(dirname "$0")/../mhla_amd/csrc/$src" 2>/dev/null
python "$(dirname "$0")/isa_stats.py" "$out" "$filt"
