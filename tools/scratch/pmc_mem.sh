#!/bin/bash
# tools/scratch/pmc_mem.sh <tag> <options>
TAG=$1; OPTIONS=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_mem_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $ROOT/tools/render_once.py --scene scenes/teapot.json --spp 64 --options "$OPTIONS" > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; return 1; }
  find $OUT/$name -name "*_kernel_trace.csv" -delete
}
run mem SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS && \
run size FETCH_SIZE WRITE_SIZE && \
run tcc TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum && \
run busy GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES
python3 - <<PY
import csv, glob, collections
totals = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$OUT/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_shade" in k or "k_trace" in k:
            totals[k[:40]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in totals.items():
    print(k, {n: "%.4g" % v for n, v in sorted(c.items())})
PY
