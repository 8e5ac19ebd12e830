# rocprofv3 evidence for profiles/: kernel trace + stats of bench.py, then PMC counters in SEPARATE passes
# (never combined with a trace domain other than --kernel-trace). usage: scripts/gpu_pmc.sh r03
TAG=${1:-r05}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
# (1) per-kernel durations: the headline workload alone; the default line with its extras (real size, clustered data, --mode exact);
#     the exact fp32-MFMA kernel alone at the bench size
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $R/gpurun_out/rocprof_bench_$TAG.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_extras -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/rocprof_bench_extras_$TAG.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_exact -- python3 $R/bench.py --mode exact --steps 10 --warmup 3 --no-cpu-baseline --no-extras --repeats 0 > $R/gpurun_out/rocprof_exact_$TAG.log 2>&1
# (1b) the reference's call shape: ONE query per call, 2 000 calls of the C harness (the single-launch streaming kernel's own duration)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_one -- $R/rag_project_icd10_amd/csrc/icd_selftest --oracle $R/oracle/libicd_oracle.so --skip-cases --bench --auto-only --nq 1 --n 40474 --iters 2000 > $R/gpurun_out/rocprof_one_$TAG.log 2>&1
# (2) PMC passes of the headline workload alone
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  d=$R/gpurun_out/pmc_${TAG}_$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --repeats 0 > $d.log 2>&1
done
# (2b) the exact fp32-MFMA kernel: one pass (MFMA count and busy cycles, clock)
d=$R/gpurun_out/pmcx_${TAG}
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $d -- python3 $R/bench.py --mode exact --steps 4 --warmup 2 --no-cpu-baseline --no-extras --repeats 0 > $d.log 2>&1
# (3) the row-sharded workload (one 1.25 M-row shard): kernel stats + HBM traffic of a coarse launch
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_rowshard -- python3 $R/bench.py --workload rowshard --steps 1 --rowshard-steps 1 > $R/gpurun_out/rocprof_rowshard_$TAG.log 2>&1
# (FETCH_SIZE three times over: VERDICT r4 item 3 asks that consecutive runs of the paced sweep agree within 15 %)
for set in "FETCH_SIZE" "WRITE_SIZE" "FETCH_SIZE:2" "FETCH_SIZE:3"; do
  d=$R/gpurun_out/pmcrs_${TAG}_$(echo $set | tr ':' '_')
  timeout 600 rocprofv3 --pmc ${set%%:*} --kernel-trace --output-format csv -d $d -- python3 $R/bench.py --workload rowshard --steps 1 --rowshard-steps 1 > $d.log 2>&1
done
cd $R
python3 - $TAG <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]
def collect(pattern):
    out = {}
    for f in sorted(glob.glob(pattern, recursive=True)):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row.get('Kernel_Name', '')][row['Counter_Name']].append(float(row['Counter_Value']))
        for k, v in agg.items():
            for c, x in v.items():
                out.setdefault(k, {})[c] = sum(x) / len(x)
    return out
def traffic_of(out):
    t = {}
    for k, v in out.items():
        if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
            # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request: doubled (MI355X_MICROARCH.md, HBM)
            t[k] = {"fetch_size_kib": v['FETCH_SIZE'], "write_size_kib": v['WRITE_SIZE'],
                    "traffic_bytes": 2 * v['FETCH_SIZE'] * 1024 + v['WRITE_SIZE'] * 1024}
    return t
out = collect(f'gpurun_out/pmc_{tag}_*/**/*counter_collection.csv')
json.dump(out, open(f'gpurun_out/{tag}_pmc_counters.json', 'w'), indent=1)
json.dump(traffic_of(out), open(f'gpurun_out/{tag}_pmc_traffic.json', 'w'), indent=1)
ex = collect(f'gpurun_out/pmcx_{tag}/**/*counter_collection.csv')
json.dump({k: v for k, v in ex.items() if 'exact_topk' in k}, open(f'gpurun_out/{tag}_pmc_counters_exact_mode.json', 'w'), indent=1)
rs = collect(f'gpurun_out/pmcrs_{tag}_[FW]*_SIZE/**/*counter_collection.csv')   # (the first FETCH_SIZE pass and the WRITE_SIZE pass)
rs_t = traffic_of(rs)
# the three FETCH_SIZE runs of the row-shard workload, each on its own: per-launch averages of the coarse kernel
runs = []
for d in (f'gpurun_out/pmcrs_{tag}_FETCH_SIZE', f'gpurun_out/pmcrs_{tag}_FETCH_SIZE_2', f'gpurun_out/pmcrs_{tag}_FETCH_SIZE_3'):
    c = collect(d + '/**/*counter_collection.csv')
    for k, v in c.items():
        if 'coarse_flat_kernel' in k and 'FETCH_SIZE' in v and v['FETCH_SIZE'] > 1e5:   # (the shard-sized launches; the gated second-pass launches fetch nothing)
            runs.append(2 * v['FETCH_SIZE'] * 1024)
for k in rs_t:
    if 'coarse_flat_kernel' in k:
        rs_t[k]['fetch_bytes_of_three_consecutive_runs'] = runs
json.dump(rs_t, open(f'gpurun_out/{tag}_pmc_traffic_rowshard.json', 'w'), indent=1)
print('rowshard 2 x FETCH_SIZE per coarse launch, three consecutive runs (GB):', [round(r / 1e9, 2) for r in runs])
for k, v in out.items():
    if 'coarse' in k or 'finalize_kernel<true' in k:
        print(k[:70], {c: round(x, 1) for c, x in v.items()})
for k, v in rs_t.items():
    if 'coarse' in k:
        print('rowshard', k[:70], v)
PY
for f in $(find gpurun_out/prof_$TAG -name "*kernel_stats.csv"); do head -10 $f; cp $f gpurun_out/${TAG}_bench_kernel_stats.csv; done
for f in $(find gpurun_out/prof_${TAG}_one -name "*kernel_stats.csv"); do head -2 $f | cut -c1-160; cp $f gpurun_out/${TAG}_single_query_kernel_stats.csv; done
for f in $(find gpurun_out/prof_${TAG}_extras -name "*kernel_stats.csv"); do cp $f gpurun_out/${TAG}_bench_with_extras_kernel_stats.csv; done
for f in $(find gpurun_out/prof_${TAG}_exact -name "*kernel_stats.csv"); do head -3 $f; cp $f gpurun_out/${TAG}_bench_exact_mode_kernel_stats.csv; done
for f in $(find gpurun_out/prof_${TAG}_rowshard -name "*kernel_stats.csv"); do head -6 $f; cp $f gpurun_out/${TAG}_rowshard_kernel_stats.csv; done
tail -1 gpurun_out/rocprof_bench_$TAG.log | cut -c1-400
