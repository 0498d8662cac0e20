#!/bin/bash
# PMC passes over the large-BVH trace kernel (run via gpurun). Usage: tools/pmc_dragon.sh <tag> <subdiv>
TAG=${1:-d1}
SUBDIV=${2:-9}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
python3 $ROOT/tools/make_assets.py --dragon $SUBDIV > /dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
run() {
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $ROOT/tools/dragon_render.py > $OUT/$name.log 2>&1
}
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
run p2 FETCH_SIZE
run p3 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run p4 SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM
run p5 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob("$OUT/p*/")):
    for f in glob.glob(d+"*/*_counter_collection.csv"):
        agg=collections.defaultdict(lambda:[0,0.0])
        for r in csv.DictReader(open(f)):
            k=(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
            agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])
        for k,v in sorted(agg.items()):
            if "k_trace" in k[0] or "k_shade" in k[0]:
                print(d.split("/")[-2], k[0], k[1], "n=%d avg=%.4g"%(v[0], v[1]/v[0]))
PY
