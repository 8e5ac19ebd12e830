# A/B of the split-bf16 batch form's GEMM shapes and of the LDS-staged W fragments (ABLATE build in csrc/abe: env ICD_ENCBIG_BF_VAR)
# usage: bash scripts/gpu_encoder_big_bf_sweep.sh r06 "0 1 4 10 11 14"
TAG=${1:-r06}; VARS=${2:-"0 1 4 10 11 14"}
export ICD_SEARCH_LIB=$PWD/rag_project_icd10_amd/csrc/abe/libicdsearch.so
OUT=gpurun_out/${TAG}_encoder_big_bf_sweep.log
: > $OUT
for v in $VARS; do
  echo "== ICD_ENCBIG_BF_VAR=$v (0: shipped; 1: TM 2, TN 3 / 4 / 3; 4: TM 2, TN 6 / 6 / 6; 10: W through LDS, TM 4, TN 3 / 4 / 3; 11: LDS, TM 2, TN 3 / 4 / 3; 14: LDS, TM 2, TN 6 / 6 / 6)" >> $OUT
  ICD_ENCBIG_BF_VAR=$v bash scripts/gpu_encoder_big_profile.sh ${TAG}_bf$v 4000 < /dev/null > /dev/null 2>&1
  python3 - gpurun_out/${TAG}_bf${v}_encoder_big_kernel_stats.csv >> $OUT <<'PY'
import csv, sys
for r in list(csv.reader(open(sys.argv[1])))[1:6]:
    print(f"   {r[0].split('(')[0][-76:]:78s} avg {float(r[3]) / 1e3:8.1f} us")
PY
  grep "ms per pass" gpurun_out/${TAG}_bf${v}_encoder_big_profile.log >> $OUT
done
cat $OUT
