# the resident loop over an ensemble shared by two ranks, both on the one GPU of the box
cd $GRAFT_REPO_ROOT
O=gpurun_out/shared; rm -rf $O; mkdir -p $O/w
export NH_RUN_SPIN_LIMIT=$((1<<24))
for cfg in "cfg3 32" "cfg5 64"; do
  set -- $cfg
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29611 \
    tests/gpu_shared_ranks_worker.py $O/w $1 $2 > $O/$1.log 2>&1
  echo "$1 rc=$?"; grep -v "^W0\|^\*\*\*\|Setting OMP" $O/$1.log | tail -12 | cut -c1-300
done
