cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/t16; rm -rf $O; mkdir -p $O
export NAIMA_AMD_DEVICE=0 NAIMA_AMD_COMM=host NH_HS_SPLIT=1 NH_RUN_SPIN_LIMIT=$((1<<24))
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o shared2 -- python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29671 \
  $GRAFT_REPO_ROOT/bench.py --gpus 2 --walkers 256 --steps 20 --warmup 5 --no-cpu --no-blobs-run > $O/shared2_bench.json 2> $O/err.log )
echo rc=$?; tail -3 $O/err.log | cut -c1-200; cut -c1-300 $O/shared2_bench.json
find $O -name "*stats*" | head; for f in $(find $O -name "*kernel_stats.csv" | head -3); do echo $f; head -6 $f | cut -c1-200; done
