# A/B across build directories (csrc/<dir>/icd_selftest) and flat-kernel variants, interleaved rounds on one box
# usage: scripts/gpu_ab_dirs.sh "ab ab12" "0 11 139" [rounds] [extra selftest args]
DIRS=${1:-"ab"}
VARS=${2:-"0 11"}
ROUNDS=${3:-2}
EXTRA=${4:-""}
O=$GRAFT_REPO_ROOT/gpurun_out/ab_dirs.log
: > $O
for rep in $(seq $ROUNDS); do
  for d in $DIRS; do
    for v in $VARS; do
      echo "### DIR=$d VAR=$v" >> $O
      (cd rag_project_icd10_amd/csrc/$d && ICD_FLAT_VAR=$v timeout 120 ./icd_selftest --oracle $GRAFT_REPO_ROOT/oracle/libicd_oracle.so --skip-cases --bench --iters 20 $EXTRA 2>&1 | grep -E "mode=auto|parity|FAIL|stamps") >> $O
    done
  done
done
cat $O
