#!/bin/bash
# tools/leaf_size_ab.sh (through gpurun): this library against pathed_amd/lib/libpathed_hip_prev.so (another build), mesh scenes with the SAH builder, long and short calls
cd ${GRAFT_REPO_ROOT:-.}
python tools/make_assets.py --dragon 9 > /dev/null 2>&1
timeout -k 10 400 python tools/rates.py --scenes C4,C5 --spp 256 --repeats 3 --variants default,wave=shade_kernel:wave --builder sah --lib pathed_amd/lib/libpathed_hip_prev.so 2>&1 | grep '^{\|^==' || exit 1
echo "## short calls (16 spp), and small scenes whose own tree is walked (refittable: no hybrid split; intersector bvh)"
timeout -k 10 300 python tools/rates.py --scenes C4 --spp 16 --repeats 3 --variants default --builder sah --lib pathed_amd/lib/libpathed_hip_prev.so 2>&1 | grep '^{\|^==' || exit 1
timeout -k 10 300 python tools/rates.py --scenes GL,GLASS --spp 64 --repeats 3 --variants refit=refittable:1,bvh=intersector:bvh+shade_kernel:per-slot --lib pathed_amd/lib/libpathed_hip_prev.so 2>&1 | grep '^{\|^==' || exit 1
