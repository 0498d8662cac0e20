#!/bin/bash
# A/B of library builds on the two mesh scenes at 256 spp per call (run via gpurun): tools/ab_queues.sh libA.so libB.so ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python3 $ROOT/tools/make_assets.py --dragon ${DRAGON:-9} > /dev/null
for lib in "$@"; do
  export PATHED_HIP_LIB=$ROOT/pathed_amd/lib/$lib
  python3 $ROOT/tools/ab_config.py scenes/teapot.json 1024 1024 256 2>/dev/null | sed "s|$ROOT/pathed_amd/lib/||"
  python3 $ROOT/tools/ab_config.py scenes/dragon-standin.json 1920 1080 256 2>/dev/null | sed "s|$ROOT/pathed_amd/lib/||"
done
