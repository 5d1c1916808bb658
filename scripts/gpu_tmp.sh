cd $GRAFT_REPO_ROOT
O=gpurun_out/t6; rm -rf $O; mkdir -p $O
(timeout 1800 python -m pytest tests/test_gpu_loops.py -m gpu -q -x -k "shared or resident") > $O/tests.log 2>&1; tail -5 $O/tests.log | cut -c1-300
export NAIMA_AMD_DEVICE=0 NAIMA_AMD_COMM=host NH_HS_SPLIT=1 NH_RUN_SPIN_LIMIT=$((1<<24))
for i in 1 2; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 2965$i \
  bench.py --gpus 2 --walkers 256 --steps 20 --warmup 5 --no-cpu > $O/shared2_$i.json 2> $O/shared2_$i.err
echo rc=$?; tail -3 $O/shared2_$i.err | cut -c1-300; cut -c1-250 $O/shared2_$i.json
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29659 \
  bench.py --gpus 2 --walkers 256 --steps 100 --warmup 5 --no-cpu --no-blobs-run > $O/shared2_100.json 2> $O/shared2_100.err
cut -c1-250 $O/shared2_100.json
unset NAIMA_AMD_COMM NH_HS_SPLIT
timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu --no-blobs-run > $O/one_100.json 2> $O/one.err; cut -c1-250 $O/one_100.json
