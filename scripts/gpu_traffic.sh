# HBM traffic of the hot kernels from the TCC (L2 memory-side) counters, one counter
# per pass as MI355X_MICROARCH.md prescribes; plus the final kernel-trace stats.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/traffic
CMD="python bench.py --steps 20 --warmup 4 --no-cpu"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/traffic -o fetch -- $CMD > /dev/null 2> gpurun_out/traffic/err_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/traffic -o write -- $CMD > /dev/null 2> gpurun_out/traffic/err_write.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/traffic -o stats -- python bench.py --steps 100 --warmup 10 --no-cpu > gpurun_out/traffic/bench_under_rocprof.json 2> gpurun_out/traffic/err_stats.log
python - <<'PY'
import csv, collections, json
out={}
for name in ("fetch","write"):
    rows=list(csv.DictReader(open('gpurun_out/traffic/%s_counter_collection.csv'%name)))
    agg=collections.defaultdict(float); n=collections.Counter()
    for r in rows:
        k=r['Kernel_Name'].split('(')[0]
        agg[k]+=float(r['Counter_Value']); n[k]+=1
    out[name]={k: agg[k]/n[k] for k in agg}
    out[name+"_launches"]=dict(n)
json.dump(out, open('gpurun_out/traffic/hbm_counters.json','w'), indent=1)
for k in sorted(out["fetch"], key=lambda k:-out["fetch"][k])[:12]:
    print("%-60s FETCH_SIZE %10.1f  WRITE_SIZE %10.1f  (per launch, counter units = KB)"%(k[:60], out["fetch"][k], out["write"].get(k,0)))
PY
head -14 gpurun_out/traffic/stats_kernel_stats.csv | cut -c1-160
