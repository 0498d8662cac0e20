#!/bin/bash
# PMC passes over the Cornell bench kernels (run via gpurun). Usage: tools/pmc_cornell.sh <tag>
TAG=${1:-c1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --steps 1 --warmup 0 --spp-per-step ${SPP:-256} --no-cpu-baseline --no-kernel-timing > $OUT/$name.log 2>&1
}
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
run p4 SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_WR SQ_WAVES_EQ_64
run p5 GRBM_GUI_ACTIVE
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob("$OUT/p*/")):
    for f in glob.glob(d+"*/*_counter_collection.csv"):
        agg=collections.defaultdict(lambda:[0,0.0])
        for r in csv.DictReader(open(f)):
            k=(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
            agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])
        for k,v in sorted(agg.items()):
            if "k_shade" in k[0] or ("k_trace" in k[0] and "true>" not in k[0]):
                print(d.split("/")[-2], k[0], k[1], "n=%d avg=%.4g"%(v[0], v[1]/v[0]))
PY
