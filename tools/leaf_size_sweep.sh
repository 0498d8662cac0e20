#!/bin/bash
# tools/leaf_size_sweep.sh (through gpurun): triangles per leaf of the host SAH builder, experiments build (profiles/r5_ab_leaf_size.log)
cd ${GRAFT_REPO_ROOT:-.}
python tools/make_assets.py --dragon 9 > /dev/null 2>&1
for leaf in 4 2 3; do
  echo "== max leaf $leaf"
  PATHED_HIP_LIB=pathed_amd/lib/libpathed_hip_experiments.so PATHED_MAX_LEAF=$leaf timeout -k 10 400 python tools/rates.py --scenes C5,C5close --spp 256 --repeats 2 --variants default --builder sah 2>&1 | grep '^{' || exit 1
done
