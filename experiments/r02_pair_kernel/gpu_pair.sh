# pair kernel (two waves per SIMD): parity cases of the C++ harness, then timing A/B against the flat kernel
cd rag_project_icd10_amd/csrc/ab
O=$GRAFT_REPO_ROOT/gpurun_out/pair.log
: > $O
echo "### parity cases, pair kernel" >> $O
ICD_FLAT_VAR=1048576 timeout 600 ./icd_selftest --oracle $GRAFT_REPO_ROOT/oracle/libicd_oracle.so 2>&1 | grep -E "FAIL|passed" | head -20 >> $O
for rep in 1 2; do
  for v in 139 1048576; do
    echo "### VAR=$v" >> $O
    ICD_FLAT_VAR=$v timeout 120 ./icd_selftest --oracle $GRAFT_REPO_ROOT/oracle/libicd_oracle.so --skip-cases --bench --iters 20 2>&1 | grep -E "mode=auto|parity|FAIL" >> $O
  done
done
cat $O
