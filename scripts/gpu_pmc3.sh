export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU"; do
  d=$R/gpurun_out/pmcc_$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $d.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmcc_*/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        agg[row.get('Kernel_Name', '')][row['Counter_Name']].append(float(row['Counter_Value']))
    for k, v in agg.items():
        if 'finalize_kernel<true' in k:
            print(k[:50], {c: round(sum(x) / len(x) / 1e6, 2) for c, x in v.items()})
PY
