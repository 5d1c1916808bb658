cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p; rm -rf $O; mkdir -p $O
for W in 0 32; do
NH_SSC_W=$W timeout 600 python bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu --no-blobs-run 2>$O/err$W.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('cfg4 W=$W', round(d['value']), 'walker-steps/s', round(d['ms_per_step']*1e3,1), 'us/step', d['kernels_us_per_launch'])
" || tail -3 $O/err$W.log
done
NH_SSC_W=32 timeout 900 python -m pytest tests -m gpu -q -k "ssc or cfg4" 2>&1 | tail -3
