#!/bin/bash
# A/B of the split shade stage inside ONE gpurun call: the per-slot kernel against k_vertex + k_regen at several persistent
# grids, on the teapot and the 5.2 M-triangle mesh.  PATHED_HIP_LIB picks another build.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python3 $ROOT/tools/make_assets.py --dragon 9 > /dev/null
# (needs the experiments library: PATHED_HIP_LIB=pathed_amd/lib/libpathed_hip_experiments.so; `make experiments`)
# run <variants> [NAME=value ...]: the variants are an argument of the function, the rest goes to the child's environment
run() { variants=$1; shift; echo "== $variants $*"; env "$@" timeout -k 10 200 python3 $ROOT/tools/ab_shade.py --scenes ${SCENES:-dragon,teapot} --variants $variants 2>&1 | grep -v amdgpu.ids; }
run per-slot,split A=0
for grids in "${@}"; do
  v=${grids%%:*}; r=${grids##*:}
  run split PATHED_VERTEX_GRID=$v PATHED_REGEN_GRID=$r
done
