# A/B over environment settings of the ablation build's harness (csrc/<dir>/icd_selftest), interleaved rounds on one box
# usage: scripts/gpu_ab_env.sh "ICD_FLAT_VAR=139 ICD_FLAT_LIST=97,ICD_FLAT_BOOT=4 ..." [rounds] [extra selftest args] [build dir, default ab]
# (a set is one token: several variables of one set are joined with commas)
SETS=${1:-"ICD_FLAT_VAR=139"}
ROUNDS=${2:-2}
EXTRA=${3:-""}
DIR=${4:-ab}
O=$GRAFT_REPO_ROOT/gpurun_out/ab_env.log
: > $O
for rep in $(seq $ROUNDS); do
  for s in $SETS; do
    echo "### $s" >> $O
    (cd rag_project_icd10_amd/csrc/$DIR && env ${s//,/ } timeout 120 ./icd_selftest --oracle $GRAFT_REPO_ROOT/oracle/libicd_oracle.so --skip-cases --bench --iters 20 $EXTRA 2>&1 | grep -E "mode=auto|parity|FAIL") >> $O
  done
done
cat $O
