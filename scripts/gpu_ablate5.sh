cd rag_project_icd10_amd/csrc
O=../../gpurun_out/ablate5.log
: > $O
for v in 0 1 512 513 520 1024 1536 528 1552 1544; do echo "### VAR=$v" >> $O; ICD_COARSE_VAR=$v timeout 120 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 10 2>&1 | grep -E "mode=auto|stamps|parity" >> $O; done
for v in 512 1536 1552; do echo "### VAR=$v full parity" >> $O; ICD_COARSE_VAR=$v timeout 300 ./icd_selftest --oracle ../../oracle/libicd_oracle.so 2>&1 | grep -E "FAIL|passed" >> $O; done
cat $O
