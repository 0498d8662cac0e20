#!/bin/bash
# Kernel timeline of one large-BVH step (run via gpurun): tools/timeline_dragon.sh <tag>
# rocprofv3 --kernel-trace (no counters: kernels overlap as in production) for PATHED_POOLS=2 and =1; the summary says how
# much of the step each kernel covers, how much of that is overlapped, and how long the GPU idles between launches.
TAG=${1:-t1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SCENE=${2:-scenes/dragon-standin.json}
OUT=$ROOT/gpurun_out/timeline_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $ROOT && python3 tools/make_assets.py --dragon ${DRAGON_LEVEL:-9} > $OUT/assets.log 2>&1
cd /tmp && export TMPDIR=/tmp
ARGS="--scene $ROOT/$SCENE --width 1920 --height 1080 --spp-per-step ${SPP:-64} --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-large-bvh"
for pools in ${POOLS:-2 1}; do
  export PATHED_POOLS=$pools
  echo "== pools $pools" >> $OUT/summary.txt
  rocprofv3 --kernel-trace --output-format csv -d $OUT/p$pools -- python3 $ROOT/bench.py $ARGS > $OUT/p$pools.log 2>&1 || { echo "pools $pools failed"; tail -3 $OUT/p$pools.log; }
  grep -o '"value": [0-9.]*' $OUT/p$pools.log | head -1 >> $OUT/summary.txt
  MS=$(grep -o '"ms_per_step": [0-9.]*' $OUT/p$pools.log | head -1 | cut -d' ' -f2)
  python3 $ROOT/tools/summarize_timeline.py $OUT/p$pools $MS ${SERIES:-} >> $OUT/summary.txt
  echo "pools $pools done"
done
cat $OUT/summary.txt
find $OUT -name "*.csv" -delete
