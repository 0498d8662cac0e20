#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python3 $ROOT/tools/make_assets.py --dragon 9 > /dev/null
for c in 256 512 1024 128 256; do
  export PATHED_CHUNKS_PER_PASS=$c
  echo "== chunks per pass $c"
  python3 $ROOT/tools/ab_config.py scenes/teapot.json 1024 1024 1024 2>/dev/null | grep -o "best.*Msamples/s"
  python3 $ROOT/tools/ab_config.py scenes/dragon-standin.json 1920 1080 1024 2>/dev/null | grep -o "best.*Msamples/s"
  python3 $ROOT/tools/ab_config.py scenes/cornell.json 1024 1024 1024 2>/dev/null | grep -o "best.*Msamples/s"
done
