cd $GRAFT_REPO_ROOT
O=gpurun_out/t8; rm -rf $O; mkdir -p $O
export NAIMA_AMD_DEVICE=0 NAIMA_AMD_COMM=host NH_RUN_SPIN_LIMIT=$((1<<24))
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29661 \
  bench.py --gpus 2 --workload cfg5 --scaling strong --walkers-total 2048 --steps 20 --warmup 5 --no-cpu --no-blobs-run > $O/cfg5_strong2.json 2> $O/cfg5_strong2.err
echo rc=$?; grep -v "^W0\|^\*\*\*\|Setting OMP\|amdgpu.ids\|socket.cpp" $O/cfg5_strong2.err | tail -5 | cut -c1-300; cut -c1-900 $O/cfg5_strong2.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29662 \
  bench.py --gpus 2 --workload cfg1 --steps 20 --warmup 5 --no-cpu --no-blobs-run > $O/cfg1_2.json 2> $O/cfg1_2.err
echo rc=$?; grep -v "^W0\|^\*\*\*\|Setting OMP\|amdgpu.ids\|socket.cpp" $O/cfg1_2.err | tail -5 | cut -c1-300; cut -c1-400 $O/cfg1_2.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29663 \
  bench.py --gpus 2 --workload cfg4 --walkers 64 --steps 4 --warmup 2 --no-cpu --no-blobs-run --min-time 0.05 > $O/cfg4_2.json 2> $O/cfg4_2.err
echo rc=$?; grep -v "^W0\|^\*\*\*\|Setting OMP\|amdgpu.ids\|socket.cpp" $O/cfg4_2.err | tail -5 | cut -c1-300; cut -c1-700 $O/cfg4_2.json
