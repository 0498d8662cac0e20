#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command (run through gpurun): tools/profile_bench_stats.sh <tag>
# Writes gpurun_out/prof_<tag>/{kernel_stats.csv, bench.log}; copy them to profiles/<tag>_kernel_stats.csv / _bench.log.
TAG=${1:-r3}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
# a line every 30 s under gpurun_out/: the profiler's post-processing of several thousand launches is silent for minutes
( while true; do date >> $OUT/alive.log; sleep 30; done ) &
TICKER=$!
trap 'kill $TICKER 2>/dev/null' EXIT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline $BENCH_ARGS > $OUT/bench.log 2> $OUT/bench.err || { echo "profiled bench failed"; tail -5 $OUT/bench.err; exit 1; }
find $OUT/trace -name "*_kernel_trace.csv" -delete
python3 - <<PY
import csv, glob
files = glob.glob("$OUT/trace/*/*_kernel_stats.csv")
rows = list(csv.reader(open(files[0])))
with open("$OUT/kernel_stats.csv", "w", newline="") as handle:
    csv.writer(handle).writerows(rows[:16])
print(open("$OUT/kernel_stats.csv").read())
PY
rm -rf $OUT/trace
