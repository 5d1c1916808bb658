cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1 -- python bench.py --steps 50 --warmup 5 --no-cpu > gpurun_out/prof/bench_under_rocprof.json 2> gpurun_out/prof/stderr.log
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last 600 kernels: steady-state graph replay
tail = rows[-600:]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for a, b in zip(tail[:-1], tail[1:]):
    n = a['Kernel_Name'].split('(')[0][:40]
    dur[n].append(int(a['End_Timestamp']) - int(a['Start_Timestamp']))
    gap[n].append(int(b['Start_Timestamp']) - int(a['End_Timestamp']))
tot = 0
for n in dur:
    d = sum(dur[n]) / len(dur[n]); g = sum(gap[n]) / len(gap[n])
    print('%-42s n=%4d dur=%7.2f us  gap_after=%6.2f us' % (n, len(dur[n]), d / 1e3, g / 1e3))
span = int(tail[-1]['Start_Timestamp']) - int(tail[0]['Start_Timestamp'])
print('span per kernel %.2f us; kernels %d' % (span / 1e3 / (len(tail) - 1), len(tail)))
PY
