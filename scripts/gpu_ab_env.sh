# A/B over environment settings of the ablation build (make ABLATE=1 OUT=ab): every argument is one "VAR=val,VAR=val" set
# ("-" = none); interleaved rounds in one box. usage: scripts/gpu_ab_env.sh ROUNDS "extra selftest args" set1 set2 ...
ROUNDS=$1; EXTRA=$2; shift 2
cd rag_project_icd10_amd/csrc/ab
O=$GRAFT_REPO_ROOT/gpurun_out/ab_env.log
: > $O
for rep in $(seq $ROUNDS); do
  for set in "$@"; do
    echo "### $set" >> $O
    ( if [ "$set" != "-" ]; then for kv in ${set//,/ }; do export "$kv"; done; fi
      timeout 120 ./icd_selftest --oracle $GRAFT_REPO_ROOT/oracle/libicd_oracle.so --skip-cases --bench --iters 20 $EXTRA 2>&1 | grep -E "mode=auto|FAIL" >> $O )
  done
done
cat $O
