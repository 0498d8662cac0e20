#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python3 $ROOT/tools/make_assets.py --dragon 9 > /dev/null
for cfg in "2 3" "2 4" "1 3" "1 4" "1 5" "1 6"; do
  set -- $cfg
  export PATHED_POOLS=$1 PATHED_TRACE_BLOCKS_PER_CU=$2
  echo "== pools $1 trace blocks/CU $2"
  python3 $ROOT/tools/ab_config.py scenes/teapot.json 1024 1024 256 2>/dev/null | grep best
  python3 $ROOT/tools/ab_config.py scenes/dragon-standin.json 1920 1080 256 2>/dev/null | grep best
done
