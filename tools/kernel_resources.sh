#!/bin/bash
# Prints VGPR / SGPR / scratch / LDS / occupancy per kernel of libpathed_hip (compile-only).
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -ffp-contract=off --offload-arch=gfx950 -Iinclude -mllvm -instcombine-max-copied-from-constant-users=4000 \
  -Rpass-analysis=kernel-resource-usage -c pathed_amd/csrc/pathed_hip.hip -o /tmp/pathed_hip_res.o 2>&1 \
  | grep -E "Function Name|VGPRs:|TotalSGPRs|Occupancy|ScratchSize|LDS Size" \
  | sed -E 's/^.*remark: +//; s/ *\[-Rpass.*$//' \
  | awk '/Function Name/{if(line)print line; line=$0; next}{line=line" | "$0}END{print line}' \
  | sed -E 's/Function Name: //' | c++filt
