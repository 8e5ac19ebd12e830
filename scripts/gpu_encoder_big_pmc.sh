# PMC counters of the batch form of the canonical encoder (scripts/probe/encoder_big_profile.py N strings), per kernel, separate passes
# (never combined with a trace domain other than --kernel-trace). usage: bash scripts/gpu_encoder_big_pmc.sh r06 2000
TAG=${1:-r06}; N=${2:-2000}
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i + 1)); d=/tmp/encpmc_$i; rm -rf $d
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python3 $R/scripts/probe/encoder_big_profile.py $N > $d.log 2>&1 < /dev/null
done
cd $R
python3 - $TAG <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('/tmp/encpmc_*/**/*counter_collection.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        agg[row.get('Kernel_Name', '')][row['Counter_Name']].append(float(row['Counter_Value']))
out = {k: {c: sum(x) / len(x) for c, x in v.items()} for k, v in agg.items() if 'enc_' in k}
json.dump(out, open(f'gpurun_out/{tag}_encoder_big_pmc.json', 'w'), indent=1)
for k, v in out.items():
    if 'linear_big' in k or 'attention' in k:
        print(k[:90]); print('   ', {c: round(x) for c, x in v.items()})
PY
