# kernel-trace stats of the default bench command (first pass of gpu_r2_profile.sh).
# rocprofv3 (ROCm 7.2) sometimes dies with a SIGSEGV of its own inside hipGraphLaunch after a
# few hundred replays of the step graphs (same library, same command: 18:26 passed, 18:48 did
# not; cfg5 passes, cfg3 with --no-blobs passes); the un-profiled command never does.  Tried as
# is first, then with fewer repeated timed regions (--min-time 0.05: the same 20-step region,
# 33 times instead of 326) -- the kernels and their durations are the same.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2prof
mkdir -p $O
CMD="python bench.py --steps 20 --warmup 5 --no-cpu --no-blobs-run"
for extra in "" "" "--min-time 0.05" "--min-time 0.05"; do
  rm -f $O/stats_*
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats -- $CMD $extra > $O/bench_under_rocprof.json 2> $O/err_stats.log
  if [ -s $O/bench_under_rocprof.json ]; then echo "stats pass: $CMD $extra" | tee $O/stats_command.txt; break; fi
  echo "rocprofv3 died ($CMD $extra)"
done
head -6 $O/stats_kernel_stats.csv | cut -c1-200
cut -c1-300 $O/bench_under_rocprof.json
