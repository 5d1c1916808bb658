cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for tag in "$@"; do
  rm -rf /tmp/prof_$tag; mkdir -p /tmp/prof_$tag
  env $(echo $tag | tr ',' ' ') timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$tag -o r1 -- python bench.py --steps 64 --warmup 8 --no-cpu $BENCH_ARGS > /tmp/prof_$tag/bench.json 2> /tmp/prof_$tag/stderr.log
  echo "== $tag"; python scripts/timeline.py /tmp/prof_$tag
done
