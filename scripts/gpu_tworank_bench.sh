# the bench's multi-rank code path end to end with two processes on the ONE GPU of the box
# (gloo stands in for RCCL, which refuses two ranks per device): flow check, not a number
cd $GRAFT_REPO_ROOT
NAIMA_AMD_COMM=host NAIMA_AMD_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 40 --warmup 4 --no-cpu 2>&1 | tail -3 | cut -c1-700
