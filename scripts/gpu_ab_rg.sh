# A/B of the row-group coarse kernel (coarse_rg_kernel.hpp, env ICD_RG_VAR) against the flat one (ICD_FLAT_VAR) across
# build directories; first the harness's parity cases with the row-group kernel, then interleaved timing rounds.
# usage: scripts/gpu_ab_rg.sh "ab abv" "FLAT=139 RG=0 RG=2 RG=4 RG=1" [rounds] [extra selftest args]
DIRS=${1:-"ab"}
SETS=${2:-"FLAT=139 RG=0"}
ROUNDS=${3:-2}
EXTRA=${4:-""}
O=$GRAFT_REPO_ROOT/gpurun_out/ab_rg.log
: > $O
for d in $DIRS; do
  echo "### DIR=$d parity cases with ICD_RG_VAR=0" >> $O
  (cd rag_project_icd10_amd/csrc/$d && ICD_RG_VAR=0 timeout 300 ./icd_selftest --oracle $GRAFT_REPO_ROOT/oracle/libicd_oracle.so 2>&1 | grep -E "FAIL|passed" | tail -12) >> $O || exit 1
done
for rep in $(seq $ROUNDS); do
  for d in $DIRS; do
    for s in $SETS; do
      echo "### DIR=$d $s" >> $O
      case $s in
        FLAT=*) E="ICD_FLAT_VAR=${s#FLAT=}";;
        RG=*) E="ICD_RG_VAR=${s#RG=}";;
      esac
      (cd rag_project_icd10_amd/csrc/$d && env $E timeout 120 ./icd_selftest --oracle $GRAFT_REPO_ROOT/oracle/libicd_oracle.so --skip-cases --bench --iters 20 $EXTRA 2>&1 | grep -E "mode=auto|parity|FAIL") >> $O
    done
  done
done
cat $O
