#!/bin/bash
# Where k_shade's wave-cycles go on a BVH scene (run via gpurun): tools/pmc_shade_detail.sh <tag> [scene.json]
# Issue / wait / instruction-fetch / vector-memory counters of every path kernel, one rocprofv3 --pmc pass per group
# (kernels are serialised under --pmc: the counters are those of each kernel alone on the chip).
TAG=${1:-d1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SCENE=${2:-scenes/dragon-standin.json}
OUT=$ROOT/gpurun_out/pmcdetail_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $ROOT && python3 tools/make_assets.py --dragon ${DRAGON_LEVEL:-9} > $OUT/assets.log 2>&1
cd /tmp && export TMPDIR=/tmp
ARGS="--scene $ROOT/$SCENE --width 1920 --height 1080 --spp-per-step ${SPP:-32} --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-large-bvh"
for pass in "p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
            "p2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F32" \
            "p3 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL" \
            "p4 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
            "p5 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum" \
            "p6 SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"; do
  name=${pass%% *}; counters=${pass#* }
  rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py $ARGS > $OUT/$name.log 2>&1 || { echo "$name failed"; tail -3 $OUT/$name.log; }
done
python3 - <<PY > $OUT/summary.txt
import csv,glob,collections
for d in sorted(glob.glob("$OUT/p*/")):
    agg=collections.defaultdict(lambda:[0,0.0]); dur=collections.defaultdict(list)
    for f in glob.glob(d+"*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k=(r["Kernel_Name"].split("(")[0][-44:], r["Counter_Name"]); agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])
    for f in glob.glob(d+"*/*_kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"].split("(")[0][-44:]].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    for k,v in sorted(dur.items()):
        if "k_shade" in k or "k_trace" in k or "k_path" in k: print(d.split("/")[-2], k, "launches %d avg %.1f us"%(len(v), sum(v)/len(v)/1e3))
    for k,v in sorted(agg.items()):
        if "k_shade" in k[0] or "k_trace" in k[0] or "k_path" in k[0]: print(d.split("/")[-2], k[0], k[1], "avg=%.6g"%(v[1]/v[0]))
PY
cat $OUT/summary.txt
find $OUT -name "*.csv" -delete
