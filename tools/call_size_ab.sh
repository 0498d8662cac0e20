#!/bin/bash
# tools/call_size_ab.sh (through gpurun): this library against pathed_amd/lib/libpathed_hip_prev.so at several call sizes (the wavefront's fill and drain)
cd ${GRAFT_REPO_ROOT:-.}
python tools/make_assets.py --dragon 9 > /dev/null 2>&1
for spp in 64 128 256; do
  echo "## $spp spp per call"
  timeout -k 10 300 python tools/rates.py --scenes C4,C5 --spp $spp --repeats 3 --variants front=shade_kernel:per-slot --builder sah --lib pathed_amd/lib/libpathed_hip_prev.so 2>&1 | grep '^{\|^==' || exit 1
done
