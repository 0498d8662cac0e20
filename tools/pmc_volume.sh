#!/bin/bash
# Counters of the volume path kernel on the reference's participating-media scene (run through gpurun): tools/pmc_volume.sh
# One --pmc pass per counter group over ONE 64-spp call, then the same call untimed by the profiler for the rate.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_volume
rm -rf $OUT; mkdir -p $OUT
python3 $ROOT/tools/make_assets.py > /dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
ARGS="--scene scenes/cornell-medium.json --integrator VolumePathTracer --spp 64"
for pass in "valu:SQ_INSTS_VALU SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU" "fetch:FETCH_SIZE" "write:WRITE_SIZE" "wait:SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  name=${pass%%:*}; counters=${pass#*:}
  rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $OUT/$name -- python3 $ROOT/tools/render_once.py $ARGS > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; exit 1; }
  find $OUT/$name -name "*_kernel_trace.csv" -delete
done
python3 $ROOT/tools/render_once.py $ARGS --spp 1024 | tee $OUT/rate.log
python3 - <<PY
import collections, csv, glob, re
samples = 1024 * 1024 * 64
totals = collections.defaultdict(float)
for path in glob.glob("$OUT/*/*/*_counter_collection.csv"):
    for row in csv.DictReader(open(path)):
        if "k_path_volume" in row["Kernel_Name"]:
            totals[row["Counter_Name"]] += float(row["Counter_Value"])
rate = float(re.search(r"= ([0-9.]+) Msamples/s", open("$OUT/rate.log").read()).group(1))
valu = totals["SQ_INSTS_VALU"] / samples
print("k_path_volume: %.1f VALU wave-instructions per camera sample, lane utilisation %.3f" % (valu, totals["SQ_THREAD_CYCLES_VALU"] / max(totals["SQ_ACTIVE_INST_VALU"], 1) / 64))
print("  HBM bytes per camera sample: %.1f (2 x FETCH_SIZE + WRITE_SIZE, KiB)" % ((2 * totals["FETCH_SIZE"] + totals["WRITE_SIZE"]) * 1024 / samples))
print("  waves waiting: %.3f of wave cycles (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)" % (totals["SQ_WAIT_INST_ANY"] / max(totals["SQ_WAVE_CYCLES"], 1)))
print("  at %.1f Msamples/s: %.0f G VALU wave-instructions per second" % (rate, valu * rate / 1e3))
PY
