cd $GRAFT_REPO_ROOT
O=gpurun_out/t10; rm -rf $O; mkdir -p $O/w
export NH_RUN_SPIN_LIMIT=$((1<<24))
for cfg in "cfg1 32 3" "cfg3 40 4" "cfg1 32 4" "cfg1 32 3" "cfg3 40 4" "cfg1 32 4" "cfg3 32 2"; do
  set -- $cfg
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$3 --master-addr 127.0.0.1 --master-port 29611 \
    tests/gpu_shared_ranks_worker.py $O/w $1 $2 > $O/$1_$3.log 2>&1
  echo "$1 $3 ranks rc=$?"; grep -v "^W0\|^\*\*\*\|Setting OMP\|amdgpu.ids\|socket.cpp" $O/$1_$3.log | grep -i "probe\|shared loop\|warn\|Assert" | head -12 | cut -c1-400
done
(timeout 1500 python -m pytest tests/test_gpu_loops.py -m gpu -q -x -k "shared_ensemble") > $O/tests.log 2>&1; tail -3 $O/tests.log | cut -c1-300
