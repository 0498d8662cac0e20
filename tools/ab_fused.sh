#!/bin/bash
# A/B of library builds on the scenes the fused kernel renders (run via gpurun): tools/ab_fused.sh libA.so libB.so ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for round in 1 2; do
for lib in "$@"; do
  export PATHED_HIP_LIB=$ROOT/pathed_amd/lib/$lib
  python3 $ROOT/tools/ab_config.py scenes/cornell.json 1024 1024 512 2>/dev/null | sed "s|$ROOT/pathed_amd/lib/||"
  python3 $ROOT/tools/ab_config.py scenes/mis-pbrt.json 1024 1024 256 2>/dev/null | sed "s|$ROOT/pathed_amd/lib/||"
  python3 $ROOT/tools/ab_config.py scenes/cornell-glass.json 1024 1024 256 2>/dev/null | sed "s|$ROOT/pathed_amd/lib/||"
done
done
