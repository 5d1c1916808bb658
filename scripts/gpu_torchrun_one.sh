# bench.py exactly as the driver launches it for N > 1 (torch.distributed.run), with ONE process
# forced onto the sharded code path: rendezvous through the agent's store, RCCL communicator,
# capture probe, split or in-graph collective
cd $GRAFT_REPO_ROOT
NAIMA_AMD_FORCE_SHARDED=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu 2>&1 | tail -1 | cut -c1-900
