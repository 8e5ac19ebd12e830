# like gpu_ab_env.sh, through the Python binding (index created with max_k = 128): usage: ROUNDS "n nq k" set1 set2 ...
ROUNDS=$1; ARGS=$2; shift 2
export ICD_SEARCH_LIB=$GRAFT_REPO_ROOT/rag_project_icd10_amd/csrc/ab/libicdsearch.so
O=gpurun_out/ab_env_py.log
: > $O
for rep in $(seq $ROUNDS); do
  for set in "$@"; do
    echo "### $set" >> $O
    ( if [ "$set" != "-" ]; then for kv in ${set//,/ }; do export "$kv"; done; fi
      timeout 300 python scripts/ab_env_py.py $ARGS 2>&1 | grep -E "^n |Error|error" >> $O )
  done
done
cat $O
