#!/bin/bash
# rocprofv3 passes over bench.py on a GPU box (run through gpurun). Usage: tools/profile_bench.sh <tag> [steps]
# BENCH_ARGS adds bench.py arguments to every pass (e.g. another scene); PMC_SPP sets the spp per step of the PMC passes.
# 1. --kernel-trace --stats   per-kernel durations of the default bench command (the raw trace is dropped: gpurun
#                             brings back at most 64 MiB)
# 2. --pmc FETCH_SIZE         HBM read traffic   (separate pass, as the MI355X guide prescribes)
# 3. --pmc WRITE_SIZE         HBM write traffic
TAG=${1:-r1}
STEPS=${2:-4}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $BENCH_ARGS --steps $STEPS --warmup 1 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
find $OUT/trace -name "*_kernel_trace.csv" -delete
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py $BENCH_ARGS --spp-per-step ${PMC_SPP:-256} --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py $BENCH_ARGS --spp-per-step ${PMC_SPP:-256} --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_write.log 2>&1
find $OUT -name "*_kernel_trace.csv" -delete
find $OUT -name "*.csv" | head -20
tail -2 $OUT/bench_trace.log
