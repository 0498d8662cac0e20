#!/bin/bash
# Counter passes behind bench.py's roofline block (run through gpurun):  tools/pmc_per_sample.sh [tag]
# One rocprofv3 --pmc pass per counter group, as /opt/skills/guides/MI355X_MICROARCH.md prescribes (FETCH_SIZE and
# WRITE_SIZE do not fit one pass; no tracing domains mixed in beside --kernel-trace), each over exactly the kernels of
# the timed path (bench.py --no-kernel-timing: no counting pass, no event pairs, no probes):
#   cornell_1024   scenes/cornell.json 1024x1024 x 256 spp        SQ_INSTS_VALU | FETCH_SIZE | WRITE_SIZE
#   large_bvh      scenes/dragon-standin.json (21 M triangles, as bench.py times it) 1920x1080 x 64 spp  SQ_INSTS_VALU | FETCH_SIZE | WRITE_SIZE
# tools/summarize_pmc_per_sample.py <tag> turns them into profiles/pmc_per_sample.json (per camera sample).
TAG=${1:-r3}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmcps_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-large-bvh"
run() {
  name=$1; counters=$2; shift 2
  rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py $COMMON "$@" > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; exit 1; }
  find $OUT/$name -name "*_kernel_trace.csv" -delete
  echo "pass $name done"
}
run cornell_valu "SQ_INSTS_VALU SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU" --spp-per-step 256 && \
run cornell_fetch FETCH_SIZE --spp-per-step 256 && \
run cornell_write WRITE_SIZE --spp-per-step 256 && \
python3 $ROOT/tools/make_assets.py --dragon ${DRAGON:-10} > /dev/null && \
run dragon_valu "SQ_INSTS_VALU SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU" --scene scenes/dragon-standin.json --width 1920 --height 1080 --spp-per-step 64 && \
run dragon_fetch FETCH_SIZE --scene scenes/dragon-standin.json --width 1920 --height 1080 --spp-per-step 64 && \
run dragon_write WRITE_SIZE --scene scenes/dragon-standin.json --width 1920 --height 1080 --spp-per-step 64 && \
python3 $ROOT/tools/summarize_pmc_per_sample.py $TAG
