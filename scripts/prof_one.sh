# rocprofv3 kernel stats of one bench.py command on the GPU box:
#   bash scripts/prof_one.sh <name> <bench.py arguments ...>   ->   gpurun_out/<name>/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
name=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $O
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o $name -- python $GRAFT_REPO_ROOT/bench.py "$@" > $O/bench.json 2> $O/err.log < /dev/null )
echo "exit $?"
f=$(find $O -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print("%-60s calls %6s avg %10.1f us  %5s %%" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
fi
python -c "import json; d=json.load(open('$O/bench.json')); print('value', d['value'], 'forbidden', d.get('proposals_forbidden_by_prior'), 'of', d.get('proposals_total'))" < /dev/null
