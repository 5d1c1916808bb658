set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
nproc; cat /sys/fs/cgroup/cpu.max; python -c "import os; print(len(os.sched_getaffinity(0)), os.cpu_count())"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1 -- python bench.py --steps 50 --warmup 5 --no-cpu > gpurun_out/prof/bench_under_rocprof.json 2> gpurun_out/prof/stderr.log
ls -R gpurun_out/prof | head -30
find gpurun_out/prof -name "*stats*" | head -3 | xargs -I{} sh -c 'echo {}; head -20 {}'
