#!/bin/bash
# runs dragon + cornell timing for each experimental library variant
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python3 $ROOT/tools/make_assets.py --dragon ${SUBDIV:-9} > /dev/null
for lib in "$@"; do
  export PATHED_HIP_LIB=$ROOT/pathed_amd/lib/$lib
  echo "== $lib"
  python3 $ROOT/tools/dragon_render.py 2>&1 | tail -1
  python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys,json
j=json.loads(sys.stdin.read()); r=j['roofline']; print('cornell %.1f Msamples/s trace %.3f ms/launch shade_total %.0f' % (j['value'], r['avg_launch_ms'], r['shade_ms_timed']))"
done
