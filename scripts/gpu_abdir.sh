O=gpurun_out/abdir.log
: > $O
for rep in 1 2 3; do
for d in rag_project_icd10_amd/csrc rag_project_icd10_amd/csrc/ab; do
echo "### $d" >> $O
(cd $d && timeout 120 ./icd_selftest --oracle $GRAFT_REPO_ROOT/oracle/libicd_oracle.so --skip-cases --bench --iters 20 2>&1 | grep -E "mode=auto") >> $O
done
done
cat $O
