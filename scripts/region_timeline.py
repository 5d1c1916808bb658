"""What happens on the GPU between two timed regions: kernels and copies of a rocprofv3 trace
(--kernel-trace --memory-copy-trace --output-format csv), merged by time; prints a window of
consecutive operations from the steady state with the gap before each.

    python scripts/region_timeline.py <dir> [first op (fraction of the run, default 0.7)] [count]
"""
import csv
import glob
import sys

d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.7
count = int(sys.argv[3]) if len(sys.argv) > 3 else 36
ops = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:40]))
for f in glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r['Start_Timestamp']), int(r['End_Timestamp']),
                    'COPY %s %s B' % (r.get('Direction', '?'), r.get('Bytes', r.get('Size', '?')))))
ops.sort()
i0 = int(len(ops) * frac)
t0 = ops[i0][0]
prev = ops[i0 - 1][1] if i0 > 0 else t0
print("%d operations; window from #%d" % (len(ops), i0))
for st, en, nm in ops[i0:i0 + count]:
    print("  +%9.1f us  gap %7.1f  dur %8.1f  %s" % ((st - t0) / 1e3, (st - prev) / 1e3, (en - st) / 1e3, nm))
    prev = en
# totals of the run kernel vs wall
run = [(st, en) for st, en, nm in ops if 'half_step_run' in nm]
if len(run) > 10:
    run = run[len(run) // 2:]
    busy = sum(en - st for st, en in run)
    wall = run[-1][1] - run[0][0]
    print("second half of the run: %d launches of k_half_step_run, busy %.1f %% of the span, mean %.1f us, "
          "mean gap %.1f us" % (len(run), 100.0 * busy / wall, busy / len(run) / 1e3,
                                (wall - busy) / (len(run) - 1) / 1e3))
