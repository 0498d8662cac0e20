#!/bin/bash
# A/B of tuning builds inside ONE gpurun call (boxes differ by +-5 %): tools/ab_libs.sh libA.so libB.so ...
# Each library is timed on Cornell (C2), the teapot (C4) and the 5.2 M-triangle stand-in (C5).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python3 $ROOT/tools/make_assets.py --dragon ${DRAGON:-9} > /dev/null
for lib in "$@"; do
  export PATHED_HIP_LIB=$ROOT/pathed_amd/lib/$lib
  python3 $ROOT/tools/ab_config.py scenes/cornell.json 1024 1024 256 | sed "s|$ROOT/pathed_amd/lib/||"
  python3 $ROOT/tools/ab_config.py scenes/teapot.json 1024 1024 256 | sed "s|$ROOT/pathed_amd/lib/||"
  python3 $ROOT/tools/ab_config.py scenes/dragon-standin.json 1920 1080 64 | sed "s|$ROOT/pathed_amd/lib/||"
done
