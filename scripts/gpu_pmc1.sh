export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z0-9_]+|GRBM_[A-Z_]+|TCC_[A-Z0-9_]+|FETCH_SIZE|WRITE_SIZE|MfmaUtil|VALUBusy|SALUBusy|LDSBankConflict)\b" | sort -u | tr '\n' ' ' > $R/gpurun_out/pmc_list.txt
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_LDS SQ_IFETCH SQ_CYCLES"; do
  d=$R/gpurun_out/pmc_$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $d.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections, os
for f in sorted(glob.glob('gpurun_out/pmc_*/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        k=row.get('Kernel_Name','')[:40]
        agg[k][row['Counter_Name']].append(float(row['Counter_Value']))
    print('==', f)
    for k,v in agg.items():
        if 'coarse' in k or 'finalize_kernel<true' in k:
            print(' ', k, {c: round(sum(x)/len(x),1) for c,x in v.items()}, 'n=', len(next(iter(v.values()))))
PY
