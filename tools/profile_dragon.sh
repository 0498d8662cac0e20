#!/bin/bash
# rocprofv3 --kernel-trace --stats over the large-BVH render (run through gpurun).
# Usage: tools/profile_dragon.sh <tag> [subdiv] [spp] [pools]
TAG=${1:-r1}
SUBDIV=${2:-9}
export DRAGON_SPP=${3:-64}
export PATHED_POOLS=${4:-1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_dragon_$TAG
mkdir -p $OUT
python3 $ROOT/tools/make_assets.py --dragon $SUBDIV > /dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/dragon_render.py > $OUT/render.log 2>&1
tail -1 $OUT/render.log
head -6 $(find $OUT/trace -name "*_kernel_stats.csv" | head -1)
