cd rag_project_icd10_amd/csrc/ab
O=$GRAFT_REPO_ROOT/gpurun_out/pair_ablate.log
: > $O
for v in 139 1048576 1048577 1048578 1048580 1048584 1048592 1048579 1048583 1048591 1048607; do
  echo "### VAR=$v" >> $O
  ICD_FLAT_VAR=$v timeout 120 ./icd_selftest --oracle $GRAFT_REPO_ROOT/oracle/libicd_oracle.so --skip-cases --bench --iters 20 2>&1 | grep -E "mode=auto|FAIL" >> $O
done
cat $O
