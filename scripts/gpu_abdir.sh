O=gpurun_out/abdir.log
: > $O
for rep in 1 2; do
echo "### product" >> $O
(cd rag_project_icd10_amd/csrc && timeout 120 ./icd_selftest --oracle $GRAFT_REPO_ROOT/oracle/libicd_oracle.so --skip-cases --bench --iters 20 2>&1 | grep -E "mode=auto") >> $O
echo "### product, ICD_NO_PERMUTE=1" >> $O
(cd rag_project_icd10_amd/csrc && ICD_NO_PERMUTE=1 timeout 120 ./icd_selftest --oracle $GRAFT_REPO_ROOT/oracle/libicd_oracle.so --skip-cases --bench --iters 20 2>&1 | grep -E "mode=auto") >> $O
echo "### no gather variant, ICD_NO_PERMUTE=1" >> $O
(cd rag_project_icd10_amd/csrc/ab && ICD_NO_PERMUTE=1 timeout 120 ./icd_selftest --oracle $GRAFT_REPO_ROOT/oracle/libicd_oracle.so --skip-cases --bench --iters 20 2>&1 | grep -E "mode=auto") >> $O
done
cat $O
