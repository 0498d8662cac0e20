#!/bin/bash
# Lane utilisation and HBM traffic of the shade stage, per-slot k_shade against the split stage (k_vertex + k_regen), on the
# 5.2 M-triangle stand-in (run through gpurun): one rocprofv3 --pmc pass per counter group and kernel organisation.
#   tools/pmc_split_vs_per_slot.sh  ->  gpurun_out/pmc_split/summary.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_split
rm -rf $OUT; mkdir -p $OUT
python3 $ROOT/tools/make_assets.py --dragon 9 > /dev/null
cd /tmp && export TMPDIR=/tmp
COMMON="--steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-large-bvh --scene scenes/dragon-standin.json --width 1920 --height 1080 --spp-per-step 64"
for kernel in per-slot split; do
  export PATHED_SHADE_KERNEL=$kernel
  for group in "valu:SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVES" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
    name=${group%%:*}; counters=${group#*:}
    rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $OUT/${kernel}_$name -- python3 $ROOT/bench.py $COMMON > $OUT/${kernel}_$name.log 2>&1 || { echo "pass $kernel $name failed"; tail -3 $OUT/${kernel}_$name.log; exit 1; }
    find $OUT/${kernel}_$name -name "*_kernel_trace.csv" -delete
    echo "pass $kernel $name done"
  done
done
python3 - <<PY > $OUT/summary.txt
import collections, csv, glob
base = "$OUT"
samples = 1920 * 1080 * 64
for kernel in ("per-slot", "split"):
    totals = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for name in ("valu", "fetch", "write"):
        for path in glob.glob("%s/%s_%s/*/*_counter_collection.csv" % (base, kernel, name)):
            for row in csv.DictReader(open(path)):
                k = row["Kernel_Name"].split("(")[0].replace("void ", "")
                if "pathed::" not in k: continue
                entry = totals[k][row["Counter_Name"]]
                entry[0] += 1; entry[1] += float(row["Counter_Value"])
    print("== shade_kernel = %s (5.2 M triangles, 1920x1080 x 64 spp, one call)" % kernel)
    for k in sorted(totals):
        c = totals[k]
        launches = max(v[0] for v in c.values())
        lanes = c["SQ_THREAD_CYCLES_VALU"][1] / c["SQ_ACTIVE_INST_VALU"][1] / 64.0 if c["SQ_ACTIVE_INST_VALU"][1] else 0.0
        hbm = (2 * c["FETCH_SIZE"][1] + c["WRITE_SIZE"][1]) * 1024.0
        print("  %-48s launches %4d  VALU wave-instr/sample %7.2f  lane utilisation %.3f  HBM bytes/sample %7.1f (per launch %.0f MB)" % (
            k, launches, c["SQ_INSTS_VALU"][1] / samples, lanes, hbm / samples, hbm / launches / 1e6))
PY
cat $OUT/summary.txt
