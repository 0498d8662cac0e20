#!/bin/bash
# Generic counter passes on one BVH workload (run via gpurun): tools/pmc_passes.sh <tag> <scene.json> "<counters of pass 1>" "<pass 2>" ...
# One rocprofv3 --pmc run per pass (kernels serialised), per-kernel averages of every counter + durations in summary.txt.
TAG=$1; SCENE=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmcpass_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $ROOT && python3 tools/make_assets.py --dragon ${DRAGON_LEVEL:-9} > $OUT/assets.log 2>&1
cd /tmp && export TMPDIR=/tmp
ARGS="--scene $ROOT/$SCENE --width ${WIDTH:-1920} --height ${HEIGHT:-1080} --spp-per-step ${SPP:-32} --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-large-bvh"
n=0
for counters in "$@"; do
  n=$((n+1))
  timeout -k 10 240 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $OUT/p$n -- python3 $ROOT/bench.py $ARGS > $OUT/p$n.log 2>&1 || { echo "pass $n ($counters) failed"; tail -2 $OUT/p$n.log | cut -c1-300; }
  echo "pass $n done"
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
