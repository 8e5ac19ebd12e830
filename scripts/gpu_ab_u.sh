# A/B of the flat partition's tiles-per-work-group U (make ABLATE=1 OUT=ab; env ICD_FLAT_U) at corpus sizes given as "n:U,U,..."
# usage: scripts/gpu_ab_u.sh "40474:0 40960:0,100,104 37000:0" [rounds]
SPECS=${1:-"40474:0 40960:0,100,104"}
ROUNDS=${2:-2}
cd rag_project_icd10_amd/csrc/ab
O=$GRAFT_REPO_ROOT/gpurun_out/ab_u.log
: > $O
for rep in $(seq $ROUNDS); do
  for spec in $SPECS; do
    n=${spec%%:*}; us=${spec##*:}
    for u in ${us//,/ }; do
      echo "### n=$n U=$u" >> $O
      if [ "$u" = "0" ]; then
        timeout 120 ./icd_selftest --oracle $GRAFT_REPO_ROOT/oracle/libicd_oracle.so --skip-cases --bench --iters 20 --n $n 2>&1 | grep -E "mode=auto|FAIL" >> $O
      else
        ICD_FLAT_U=$u timeout 120 ./icd_selftest --oracle $GRAFT_REPO_ROOT/oracle/libicd_oracle.so --skip-cases --bench --iters 20 --n $n 2>&1 | grep -E "mode=auto|FAIL" >> $O
      fi
    done
  done
done
cat $O
