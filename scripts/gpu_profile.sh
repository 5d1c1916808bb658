# Evidence for profiles/ (one MI355X): rocprofv3 kernel stats of the DRIVER'S command for the
# headline and of the other workloads, fabric-side traffic and SQ counters (separate --pmc passes,
# as MI355X_MICROARCH.md prescribes), un-profiled bench lines, the kernels' own phase stamps.
#   bash scripts/gpu_profile.sh [round tag, default r03]   ->   gpurun_out/<tag>prof/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r06}
O=$GRAFT_REPO_ROOT/gpurun_out/${TAG}prof
rm -rf $O; mkdir -p $O
DRV="bench.py --steps 20 --warmup 5"
prof() { name=$1; shift
  ( cd /tmp && timeout 900 rocprofv3 "$@" --output-format csv -d $O -o $name -- python $GRAFT_REPO_ROOT/$CMD > $O/${name}_bench.json 2> $O/err_$name.log )
  echo "$name: exit $? ($CMD)"
}
# 1. kernel stats: the driver's exact command (headline), then the other workloads
CMD="$DRV"; prof cfg3_stats --kernel-trace --stats
echo "python $CMD" > $O/cfg3_stats_command.txt
for w in cfg1 cfg2 cfg5; do CMD="$DRV --workload $w --no-cpu --no-blobs-run"; prof ${w}_stats --kernel-trace --stats; done
CMD="bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu --no-blobs-run"; prof cfg4_stats --kernel-trace --stats
# 2. counters, one group per pass (headline + cfg5)
for w in cfg3 cfg5; do
  CMD="$DRV --workload $w --no-cpu --no-blobs-run --min-time 0.1"
  prof ${w}_fetch --kernel-trace --pmc FETCH_SIZE
  prof ${w}_write --kernel-trace --pmc WRITE_SIZE
  prof ${w}_tcc --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum
  prof ${w}_sqa --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
  prof ${w}_sqb --kernel-trace --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR
  prof ${w}_sqc --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU
  prof ${w}_sqd --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
done
# (vector-pipe utilisation of the other two workloads' dominant kernels: one SQ pass each)
CMD="$DRV --workload cfg2 --no-cpu --no-blobs-run --min-time 0.1"; prof cfg2_sqa --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
CMD="bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu --no-blobs-run --min-time 0.1"; prof cfg4_sqa --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
# (round 6: the instance with two walkers of a workgroup in flight -- cfg3 at 2048 walkers)
CMD="$DRV --workload cfg3 --walkers 2048 --no-cpu --no-blobs-run --min-time 0.1"; prof cfg3w2048_sqa --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
python - <<PY
import csv, collections, glob, json, os
O = "$O"
for w in ("cfg3", "cfg5", "cfg2", "cfg4", "cfg3w2048"):
    res = {}
    for f in sorted(glob.glob(O + '/%s_*_counter_collection.csv' % w)):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
        for r in rows:
            k = r['Kernel_Name'].split('(')[0]
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
        for k in agg:
            for c in agg[k]:
                res.setdefault(k, {})[c] = agg[k][c] / n[k][c]
                res[k]["launches_" + c] = n[k][c]
    json.dump(res, open(O + '/%s_counters_per_launch.json' % w, 'w'), indent=1)
    for k, v in res.items():
        if 'half_step' in k or 'ic_seed' in k:
            print(w, k, json.dumps({c: round(x, 1) for c, x in v.items() if not c.startswith("launches")}))
PY
# 3. un-profiled bench lines
cd $GRAFT_REPO_ROOT
timeout 600 python $DRV > $O/bench_n1_default.json 2> $O/err_bench.log
for w in cfg1 cfg2 cfg5; do timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu > $O/bench_$w.json 2>> $O/err_bench.log; done
timeout 600 python bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu --no-blobs-run > $O/bench_cfg4.json 2>> $O/err_bench.log
timeout 600 python bench.py --workload cfg3 --walkers 256 --steps 20 --warmup 5 --no-cpu > $O/bench_cfg3_w256.json 2>> $O/err_bench.log
timeout 600 python bench.py --workload cfg5 --scaling strong --walkers-total 2048 --steps 100 --warmup 10 --no-cpu --no-blobs-run > $O/bench_cfg5_strong2048_n1.json 2>> $O/err_bench.log
NAIMA_AMD_RESIDENT=0 timeout 600 python $DRV --no-cpu > $O/bench_n1_per_launch_kernel.json 2>> $O/err_bench.log
NAIMA_AMD_RESIDENT=0 timeout 600 python bench.py --workload cfg5 --steps 20 --warmup 5 --no-cpu > $O/bench_cfg5_per_launch_kernel.json 2>> $O/err_bench.log
NAIMA_AMD_RESIDENT=0 timeout 600 python bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu > $O/bench_cfg2_per_launch_kernel.json 2>> $O/err_bench.log
NAIMA_AMD_RESIDENT=0 timeout 600 python bench.py --workload cfg5 --scaling strong --walkers-total 2048 --steps 100 --warmup 10 --no-cpu --no-blobs-run > $O/bench_cfg5_strong2048_n1_per_launch_kernel.json 2>> $O/err_bench.log
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.load(open(f))
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'] * 1e3, 2), 'us/step', d.get('value_without_blobs', d.get('value_store_blobs')), d['kernels_us_per_launch'], d['loop'][:30])
    except Exception as e:
        print(f, "ERR", e)
PY
# 4. the kernels' own phase stamps
NH_HS_DEBUG=1 timeout 300 python scripts/run_stamps.py cfg3 512 > $O/stamps_cfg3.txt 2>&1
NH_HS_DEBUG=1 timeout 300 python scripts/run_stamps.py cfg5 256 > $O/stamps_cfg5.txt 2>&1
tail -3 $O/err_bench.log
NH_HS_DEBUG=1 timeout 300 python scripts/run_stamps.py cfg3 2048 > $O/stamps_cfg3_w2048.txt 2>&1
NH_HS_DEBUG=1 NAIMA_AMD_RESIDENT=0 timeout 300 python scripts/hs_stamps.py cfg3 512 > $O/per_launch_kernel_phase_stamps_cfg3.txt 2>&1
# 5. ensembles of more walkers than CUs (resident workgroups take several walkers of a half-step: two of
# them in flight -- round 6; NH_RUN_PIPELINE=0: strictly in turn, as in round 5)
for n in 1024 2048 4096; do
  timeout 600 python bench.py --workload cfg3 --walkers $n --steps 20 --warmup 5 --no-cpu --no-blobs-run > $O/bench_cfg3_w$n.json 2>> $O/err_bench.log
  NH_RUN_PIPELINE=0 timeout 600 python bench.py --workload cfg3 --walkers $n --steps 20 --warmup 5 --no-cpu --no-blobs-run > $O/bench_cfg3_w${n}_serial_turns.json 2>> $O/err_bench.log
done
for n in 1024 2048; do
  NH_RUN_MAX_PER_WG=1 timeout 600 python bench.py --workload cfg3 --walkers $n --steps 20 --warmup 5 --no-cpu --no-blobs-run > $O/bench_cfg3_w${n}_per_launch_kernel.json 2>> $O/err_bench.log
done
# 6. bench.py starting its own ranks (round 5): two ranks on this ONE GPU (NAIMA_AMD_DEVICE pins them; each
# plans for half the CUs) -- cfg3's 512 walkers SHARED by the two (strong), 512 each (weak), and the same
# with 100-step regions
(
export NAIMA_AMD_DEVICE=0 NH_RUN_SPIN_LIMIT=$((1<<24))
timeout 600 python bench.py --gpus 2 --scaling strong --walkers-total 512 --steps 20 --warmup 5 --no-cpu > $O/bench_cfg3_shared_two_ranks_one_gpu.json 2>> $O/err_bench.log
timeout 600 python bench.py --gpus 2 --scaling strong --walkers-total 512 --steps 100 --warmup 5 --no-cpu --no-blobs-run > $O/bench_cfg3_shared_two_ranks_one_gpu_steps100.json 2>> $O/err_bench.log
timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu --no-blobs-run > $O/bench_cfg3_weak_two_ranks_one_gpu.json 2>> $O/err_bench.log
timeout 900 python bench.py --gpus 4 --steps 20 --warmup 5 --no-cpu --no-blobs-run --walkers 128 > $O/bench_cfg3_weak_four_ranks_one_gpu_baseline_split.json 2>> $O/err_bench.log
# (round 6: every rung of the exchange ladder refused in turn -- rings by fault injection, RCCL by the probe
# processes: two ranks of one device -- the host-staged all-gather taken; config.exchange.ladder says so)
NAIMA_AMD_COMM=rccl NAIMA_AMD_LADDER_REFUSE=ring NAIMA_AMD_RCCL_PROBE_TIMEOUT=60 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu --no-blobs-run > $O/bench_ladder_rings_refused_rccl_unavailable_host_staged.json 2>> $O/err_bench.log
)
# 6b. BASELINE's cfg4 at its 1024 walkers on one GPU
timeout 900 python bench.py --scaling strong --workload cfg4 --walkers-total 1024 --steps 10 --warmup 2 --no-cpu --no-blobs-run > $O/bench_cfg4_strong1024_n1.json 2>> $O/err_bench.log
# 6c. where a 20-step region's time goes on the host
timeout 300 python scripts/region_host_latency.py cfg3 20 > $O/region_host_latency.txt 2>&1
timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu --no-blobs-run > $O/bench_cfg3_one_process_steps100.json 2>> $O/err_bench.log
# 7. one-GPU projection of the multi-GPU configurations
timeout 2400 python scripts/shard_table.py > $O/shard_table.json 2> $O/shard_table.err
for f in bench_cfg3_w1024 bench_cfg3_w2048 bench_cfg3_w4096 bench_cfg3_w2048_serial_turns bench_ladder_rings_refused_rccl_unavailable_host_staged bench_cfg3_shared_two_ranks_one_gpu bench_cfg3_shared_two_ranks_one_gpu_steps100 bench_cfg3_weak_two_ranks_one_gpu bench_cfg3_weak_four_ranks_one_gpu_baseline_split bench_cfg4_strong1024_n1 bench_cfg3_one_process_steps100; do
  python -c "import json; d=json.load(open('$O/$f.json')); print('$f', round(d['value']), d['ms_per_step'])"
done
