#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python3 $ROOT/tools/make_assets.py --dragon ${SUBDIV:-9} > /dev/null
for lib in "$@"; do
  export PATHED_HIP_LIB=$ROOT/pathed_amd/lib/$lib
  echo "== $lib"
  PATHED_POOLS=1 PATHED_MAX_SLOTS=4194304 PATHED_TRACE_BLOCKS_PER_CU=4 DRAGON_SPP=64 python3 $ROOT/tools/dragon_render.py 2>&1 | tail -1
done
