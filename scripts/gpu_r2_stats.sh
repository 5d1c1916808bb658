cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2prof
mkdir -p $O
CMD="python bench.py --steps 20 --warmup 5 --no-cpu --no-blobs-run"
for i in 1 2 3; do
  rm -f $O/stats_*
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats -- $CMD > $O/bench_under_rocprof.json 2> $O/err_stats.log
  echo "try $i rc=$?"
  if [ -s $O/stats_kernel_stats.csv ]; then break; fi
done
head -8 $O/stats_kernel_stats.csv | cut -c1-200
cut -c1-300 $O/bench_under_rocprof.json
