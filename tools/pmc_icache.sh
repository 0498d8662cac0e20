#!/bin/bash
# Instruction-cache counters of the Cornell bench kernels (run via gpurun). Usage: tools/pmc_icache.sh <tag>
TAG=${1:-i1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_INST[A-Z_]*" | sort -u > $OUT/counters.txt
run() {
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --steps 1 --warmup 0 --spp-per-step 32 --no-cpu-baseline --no-kernel-timing > $OUT/$name.log 2>&1
}
run p1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run p2 SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES
cat $OUT/counters.txt | tr '\n' ' '; echo
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
