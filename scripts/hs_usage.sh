#!/bin/bash
# register / scratch / spill report of the half-step kernels (cross-compiled, no GPU needed).
# ScratchSize of k_half_step must stay 0: a careless dynamic read of the by-value descriptor
# makes the compiler copy all 3 KB of it to scratch (VGPR spill count alone does not show that).
cd "$(dirname "$0")/../naima_amd/csrc"
for f in nh_halfstep.hip nh_persist.hip; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c -mllvm -amdgpu-kernarg-preload-count=16 \
  -Rpass-analysis=kernel-resource-usage $f -o /tmp/hs_usage.o 2>&1 | \
  grep -A 12 "Function Name: _Z1[15]k_half_step" | grep -E "Name|VGPRs:|Spill|SGPRs:|Occupancy|ScratchSize" | sed 's/.*remark: *//; s/ \[-Rpass.*//'
done
