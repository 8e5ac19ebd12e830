# A/B of the flat coarse kernel's stage / select variants (make ABLATE=1 OUT=ab; env ICD_FLAT_VAR), interleaved rounds in one box
# usage: scripts/gpu_ab_flat.sh "0 1 3 ..." [rounds]
VARS=${1:-"0 1 3 5 7 9 17 19 27"}
ROUNDS=${2:-2}
cd rag_project_icd10_amd/csrc/ab
O=$GRAFT_REPO_ROOT/gpurun_out/ab_flat.log
: > $O
for rep in $(seq $ROUNDS); do
  for v in $VARS; do
    echo "### VAR=$v" >> $O
    ICD_FLAT_VAR=$v timeout 120 ./icd_selftest --oracle $GRAFT_REPO_ROOT/oracle/libicd_oracle.so --skip-cases --bench --iters 20 2>&1 | grep -E "mode=auto|parity|FAIL|stamps" >> $O
  done
done
cat $O
