#!/bin/bash
# tools/ab_compiler_flags.sh (through gpurun): the same rates through several builds of the library, one process per library (profiles/r5_ab_compiler_flags.log; the variant libraries are built by hand with the extra -mllvm flag)
cd ${GRAFT_REPO_ROOT:-.}
python tools/make_assets.py --dragon 9 > /dev/null 2>&1
for lib in libpathed_hip.so libpathed_hip_fA.so libpathed_hip_fB.so libpathed_hip_fC.so libpathed_hip_fD.so; do
  echo "== $lib"
  PATHED_HIP_LIB=pathed_amd/lib/$lib timeout -k 10 200 python tools/rates.py --scenes C2,ON,GGX,C3,GLASS,VOL,C4,C5 --spp 256 --repeats 2 2>&1 | grep '^{' || exit 1
done
