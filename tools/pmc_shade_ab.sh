#!/bin/bash
# PMC comparison of the two shade kernels on one workload (run via gpurun): tools/pmc_shade_ab.sh <tag> [bench args...]
# Kernels are serialised under --pmc, so durations and counters are those of each kernel alone on the chip.
TAG=${1:-ab}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmcab_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for variant in ${VARIANTS:-per-slot staged}; do
  export PATHED_SHADE_KERNEL=$variant
  for pass in "p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
              "p2 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
    name=${pass%% *}; counters=${pass#* }
    rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $OUT/$variant-$name -- python3 $ROOT/bench.py --steps 1 --warmup 0 --spp-per-step ${SPP:-64} --no-cpu-baseline --no-kernel-timing --no-large-bvh $BENCH_ARGS > $OUT/$variant-$name.log 2>&1 || { echo "$variant $name failed"; tail -3 $OUT/$variant-$name.log; }
  done
done
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob("$OUT/*/")):
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
        if "k_shade" in k[0] or "k_path" in k[0]: print(d.split("/")[-2], k[0], k[1], "avg=%.5g"%(v[1]/v[0]))
PY
find $OUT -name "*.csv" -delete
