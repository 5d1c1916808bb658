#!/bin/bash
# Build libnaima_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles
# without a GPU.  Usage: naima_amd/csrc/build.sh [extra hipcc flags]
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="${NH_OUT:-$here/../libnaima_hip.so}"
tmp="$out.tmp.$$"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
  -Wall -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=16 "$@" \
  "$here/nh_core.hip" "$here/nh_synchrotron.hip" "$here/nh_tables.hip" "$here/nh_device.hip" "$here/nh_moves.hip" "$here/nh_comm.hip" "$here/nh_kelner.hip" "$here/nh_halfstep.hip" "$here/nh_persist.hip" "$here/nh_general.hip" \
  -ldl -lpthread -o "$tmp"
mv -f "$tmp" "$out"  # (atomic: a snapshot of the tree never sees a half-written library)
echo "built $out"
