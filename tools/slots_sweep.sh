#!/bin/bash
# Slots per render state (PATHED_MAX_SLOTS, split over the pools) on the two mesh scenes, 256 spp per call (run via gpurun).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python3 $ROOT/tools/make_assets.py --dragon ${DRAGON:-9} > /dev/null
for slots in 4194304 1048576 2097152 8388608 16777216 4194304; do
  export PATHED_MAX_SLOTS=$slots
  echo "== max slots $slots"
  python3 $ROOT/tools/ab_config.py scenes/teapot.json 1024 1024 256 2>/dev/null | grep best
  python3 $ROOT/tools/ab_config.py scenes/dragon-standin.json 1920 1080 256 2>/dev/null | grep best
done
