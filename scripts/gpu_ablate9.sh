cd rag_project_icd10_amd/csrc
O=../../gpurun_out/ablate9.log
: > $O
for v in 576 6144 6208 2048 2112 576; do echo "### VAR=$v" >> $O; ICD_COARSE_VAR=$v timeout 120 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 10 2>&1 | grep -E "mode=auto|stamps|parity" >> $O; done
for v in 6208 2112; do echo "### VAR=$v full parity" >> $O; ICD_COARSE_VAR=$v timeout 300 ./icd_selftest --oracle ../../oracle/libicd_oracle.so 2>&1 | grep -E "FAIL|passed" >> $O; done
cat $O
