# round 3, first GPU call: the new bench-condition parity tests; what kills rocprofv3 inside
# hipGraphLaunch (python + native backtraces, runtime knobs); the "second sampler" effect
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3diag
rm -rf $O; mkdir -p $O
(time timeout 1200 python -m pytest tests/test_gpu_loops.py -m gpu -q -s -k "benchmarks") > $O/tests.log 2>&1
tail -25 $O/tests.log | cut -c1-400
CMD="bench.py --steps 20 --warmup 5 --no-cpu --no-blobs-run"
run() { tag=$1; shift
  ( cd /tmp && timeout 420 env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O -o $tag -- python -X faulthandler $GRAFT_REPO_ROOT/$CMD > $GRAFT_REPO_ROOT/$O/$tag.json 2> $GRAFT_REPO_ROOT/$O/$tag.err )
  echo "== $tag ($*): exit $? json $(wc -c < $O/$tag.json) bytes"; grep -m1 -A12 "Fatal Python error\|most recent call first" $O/$tag.err | cut -c1-200 | head -16
}
run base A=1
run nocapture DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run hostkernarg HIP_FORCE_DEV_KERNARG=0
run gsteps1 NAIMA_AMD_GSTEPS=1
# native backtrace of the base case
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O -o gdb -- rocgdb -batch -ex "set pagination off" -ex "handle SIGSEGV stop print" -ex run -ex bt -ex "info sharedlibrary" -ex "x/6i \$pc" -ex "info registers rip rsi rdi rdx rcx" --args python $GRAFT_REPO_ROOT/$CMD > $GRAFT_REPO_ROOT/$O/gdb.log 2>&1 )
grep -n -A30 "SIGSEGV" $O/gdb.log | cut -c1-250 | head -60
timeout 600 python scripts/second_sampler.py cfg2 256 400 1 0 1 > $O/second_cfg2.log 2>&1; cat $O/second_cfg2.log | cut -c1-300
timeout 600 python scripts/second_sampler.py cfg3 512 400 1 0 > $O/second_cfg3.log 2>&1; cat $O/second_cfg3.log | cut -c1-300
