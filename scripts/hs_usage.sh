#!/bin/bash
# register / LDS / spill report of the half-step kernel (cross-compiled, no GPU needed)
cd "$(dirname "$0")/../naima_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c -mllvm -amdgpu-kernarg-preload-count=16 \
  -Rpass-analysis=kernel-resource-usage nh_halfstep.hip -o /tmp/hs_usage.o 2>&1 | \
  grep -A 12 "Function Name: _Z11k_half_step" | grep -E "Name|VGPRs:|Spill|SGPRs:|Occupancy|LDS"
