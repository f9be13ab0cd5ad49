#!/bin/bash
# Build a differently configured libdfx.so for A/B runs (select it with DFX_LIBRARY=<path>).
#   scripts/build_variant.sh <out.so> [extra hipcc flags...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$1; shift
mkdir -p "$(dirname "$OUT")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -I$R/include "$@" \
    -o "$OUT" $R/denseflow_amd/csrc/*.hip $R/denseflow_amd/csrc/*.cpp
echo "built $OUT ($*)"
