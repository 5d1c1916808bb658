cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for mode in i is wi wis; do for w in 1 2; do
rm -rf /tmp/pk; mkdir -p /tmp/pk
NH_INT_W=$w timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o r1 -- python scripts/kbench2.py $mode > /tmp/pk/out.txt 2>&1
python - $mode $w <<'PY'
import csv,glob,sys
f=glob.glob('/tmp/pk/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'integrate_tables' in r['Name']:
        print(sys.argv[1], 'W=', sys.argv[2], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
done; done
