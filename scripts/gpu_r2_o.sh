cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 1200 python scripts/shard_table.py > gpurun_out/r2/shard_table.json 2> gpurun_out/r2/shard_table.err; echo rc=$?
tail -22 gpurun_out/r2/shard_table.err
python -c "
import json
d=json.load(open('gpurun_out/r2/shard_table.json'))
for p in d['projection']: print(p)"
