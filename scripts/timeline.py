"""per-kernel durations inside the replayed step graphs, from a rocprofv3 kernel trace:
python scripts/timeline.py <dir with *kernel_trace.csv>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].split('(')[0][:34] for r in rows]
st = [int(r['Start_Timestamp']) for r in rows]
en = [int(r['End_Timestamp']) for r in rows]
runs, cur = [], [0]
for k in range(1, len(rows)):
    if st[k] - en[k - 1] < 3000:
        cur.append(k)
    else:
        runs.append(cur)
        cur = [k]
runs.append(cur)
big = [r for r in runs if len(r) >= 16]
dur = collections.defaultdict(list)
span = nk = 0
for r in big:
    for k in r:
        dur[names[k]].append(en[k] - st[k])
    span += en[r[-1]] - st[r[0]]
    nk += len(r)
tot = 0.0
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print('%-36s n=%5d  %7.2f us' % (n, len(v), sum(v) / len(v) / 1e3))
per = {n: len(v) for n, v in dur.items()}
nhalf = max(per.values()) if per else 1
print('graph runs %d, kernels %d, busy span per half-step %.2f us' % (len(big), nk, span / 1e3 / nhalf))
