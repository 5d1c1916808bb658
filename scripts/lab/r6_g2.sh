cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c; mkdir -p $O
for i in 1 2 3 4 5 6; do
NAIMA_AMD_DEVICE=0 NH_RUN_SPIN_LIMIT=16777216 timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu --min-time 0.3 --no-blobs-run > $O/g2_$i.json 2> $O/g2_$i.err; echo "run $i rc=$?"; cut -c1-150 $O/g2_$i.json; grep -c Traceback $O/g2_$i.err
done
