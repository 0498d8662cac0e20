#!/bin/bash
# Parking parameters of the persistent trace kernel (lanes below which a dry wave parks its rays, patience in steps) on the
# two mesh scenes, 256 spp per call (run via gpurun).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python3 $ROOT/tools/make_assets.py --dragon ${DRAGON:-9} > /dev/null
for cfg in "32 24" "0 24" "16 24" "48 24" "32 8" "32 64" "48 8" "32 24"; do
  set -- $cfg
  export PATHED_SUSPEND_LANES=$1 PATHED_SUSPEND_PATIENCE=$2
  echo "== suspend lanes $1 patience $2"
  python3 $ROOT/tools/ab_config.py scenes/teapot.json 1024 1024 256 2>/dev/null | grep -o "best.*Msamples/s"
  python3 $ROOT/tools/ab_config.py scenes/dragon-standin.json 1920 1080 256 2>/dev/null | grep -o "best.*Msamples/s"
done
