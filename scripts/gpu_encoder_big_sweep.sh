# A/B of the batch form's GEMM shapes (ABLATE build in csrc/abe: env ICD_ENCBIG_VAR, ICD_ENCBIG_LDS): kernel averages per variant
# usage: bash scripts/gpu_encoder_big_sweep.sh r06 "0 1 2 3 4 5 6"
TAG=${1:-r06}; VARS=${2:-"0 1 2 3 4 5 6"}
export ICD_SEARCH_LIB=$PWD/rag_project_icd10_amd/csrc/abe/libicdsearch.so
OUT=gpurun_out/${TAG}_encoder_big_sweep.log
: > $OUT
for v in $VARS; do
  echo "== ICD_ENCBIG_VAR=$v (0: TM 2, TN 3 / 4 / 2, PF 1 = shipped; 1: TNO 3; 2: TM 4, TNO 3; 3: TM 4; 4: PF 2; 5: TM 4, TN 3 / 3 / 3, PF 2; 6: TNO 3, PF 2)" >> $OUT
  ICD_ENCBIG_VAR=$v bash scripts/gpu_encoder_big_profile.sh ${TAG}_v$v 4000 < /dev/null > /dev/null 2>&1
  python3 - gpurun_out/${TAG}_v${v}_encoder_big_kernel_stats.csv >> $OUT <<'PY'
import csv, sys
for r in list(csv.reader(open(sys.argv[1])))[1:6]:
    print(f"   {r[0].split('(')[0][-66:]:68s} avg {float(r[3]) / 1e3:8.1f} us")
PY
  grep "ms per pass" gpurun_out/${TAG}_v${v}_encoder_big_profile.log >> $OUT
done
cat $OUT
