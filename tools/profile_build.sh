#!/bin/bash
# rocprofv3 kernel stats of the on-GPU BVH builders on the 5.2 M-triangle stand-in (run via gpurun).
# Usage: tools/profile_build.sh <tag>   ->  gpurun_out/prof_<tag>_{lbvh,ploc}/
TAG=${1:-build}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python3 $ROOT/tools/make_assets.py --dragon ${DRAGON:-9} > /dev/null
cd /tmp && export TMPDIR=/tmp
for builder in lbvh ploc; do
  OUT=$ROOT/gpurun_out/prof_${TAG}_$builder
  mkdir -p $OUT
  PATHED_BVH_BUILDER=$builder DRAGON_SPP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/tools/dragon_render.py > $OUT/run.log 2>&1
  tail -1 $OUT/run.log
done
