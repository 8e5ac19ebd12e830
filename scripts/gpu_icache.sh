# instruction-cache counters of the coarse kernel (one rocprofv3 --pmc pass, only --kernel-trace beside it)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp
rocprofv3 -L 2>/dev/null | grep -i -E "icache|ifetch|inst_cache|SQC_" | head -40 > $R/gpurun_out/icache_counters.txt
cat $R/gpurun_out/icache_counters.txt | cut -c1-200
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_WAVE_CYCLES"; do
  d=$R/gpurun_out/pmc_icache_$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $d.log 2>&1
  tail -2 $d.log | cut -c1-200
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_icache_*/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        agg[row.get('Kernel_Name', '')][row['Counter_Name']].append(float(row['Counter_Value']))
    for k, v in agg.items():
        if 'coarse' in k or 'finalize_kernel<true' in k:
            print(k[:60], {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
