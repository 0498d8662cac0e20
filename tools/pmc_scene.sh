#!/bin/bash
# Counter passes over ONE render call of one scene (run through gpurun):  tools/pmc_scene.sh <tag> <scene.json> [spp] [integrator] [options]
# One rocprofv3 --pmc pass per counter group (MI355X_MICROARCH.md: no tracing domains beside --kernel-trace), summed per kernel:
#   instructions   SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES  -> VALU wave-instructions per sample, lane utilisation
#   classes        SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32
#   waits          SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU
#   icache         SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES
TAG=${1:-s1}; SCENE=${2:-scenes/cornell.json}; SPP=${3:-64}; INTEGRATOR=${4:-PathTracer}; OPTIONS=${5:-}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_scene_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $ROOT/tools/render_once.py --scene $SCENE --spp $SPP --integrator $INTEGRATOR --options "$OPTIONS" > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; return 1; }
  find $OUT/$name -name "*_kernel_trace.csv" -delete
}
run instructions SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES && \
run classes SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 && \
run waits SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU && \
run icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES
python3 - <<PY
import csv, glob, collections, re
samples = None
for line in open("$OUT/instructions.log"):
    m = re.search(r": (\d+) samples in", line)
    if m: samples = int(m.group(1))
print("== $SCENE, $SPP spp, $INTEGRATOR: %s camera samples" % samples)
totals = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$OUT/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "pathed::" not in k: continue
        totals[k[-110:]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in sorted(totals.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    if c.get("SQ_INSTS_VALU", 0) < 1e6: continue
    print(k)
    v = c["SQ_INSTS_VALU"]
    print("   VALU wave-instructions per sample %.1f | lane utilisation %.3f | transcendental %.3f  fma %.3f  mul %.3f  add %.3f of them | SALU per VALU %.3f" % (
        v / samples, c["SQ_THREAD_CYCLES_VALU"] / max(64.0 * c["SQ_ACTIVE_INST_VALU"], 1.0),
        c["SQ_INSTS_VALU_TRANS_F32"] / v, c["SQ_INSTS_VALU_FMA_F32"] / v, c["SQ_INSTS_VALU_MUL_F32"] / v, c["SQ_INSTS_VALU_ADD_F32"] / v, c["SQ_INSTS_SALU"] / v))
    print("   wait-any share of wave cycles %.3f | icache requests per VALU %.3f, miss rate %.4f (%.0f misses)" % (
        c["SQ_WAIT_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1), c["SQC_ICACHE_REQ"] / v, c["SQC_ICACHE_MISSES"] / max(c["SQC_ICACHE_REQ"], 1), c["SQC_ICACHE_MISSES"]))
PY
