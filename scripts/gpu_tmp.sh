cd $GRAFT_REPO_ROOT
O=gpurun_out/t11; rm -rf $O; mkdir -p $O
export NH_RUN_SPIN_LIMIT=$((1<<24))
NH_HS_SPLIT=2 timeout 900 python scripts/shared_stress.py cfg3 256 4 3000 > $O/s1.log 2>&1; echo rc=$?; grep -v "^W0\|^\*\*\*\|Setting OMP\|amdgpu.ids\|socket.cpp" $O/s1.log | tail -4 | cut -c1-400
timeout 900 python scripts/shared_stress.py cfg5 512 2 3000 > $O/s2.log 2>&1; echo rc=$?; grep -v "^W0\|^\*\*\*\|Setting OMP\|amdgpu.ids\|socket.cpp" $O/s2.log | tail -4 | cut -c1-400
NH_HS_SPLIT=1 timeout 900 python scripts/shared_stress.py cfg2 246 3 2000 > $O/s3.log 2>&1; echo rc=$?; grep -v "^W0\|^\*\*\*\|Setting OMP\|amdgpu.ids\|socket.cpp" $O/s3.log | tail -4 | cut -c1-400
timeout 900 python scripts/shared_stress.py cfg1 64 8 2000 > $O/s4.log 2>&1; echo rc=$?; grep -v "^W0\|^\*\*\*\|Setting OMP\|amdgpu.ids\|socket.cpp" $O/s4.log | tail -4 | cut -c1-400
