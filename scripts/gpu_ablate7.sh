cd rag_project_icd10_amd/csrc
O=../../gpurun_out/ablate7.log
: > $O
for v in 512 576 544 520 584 552 513; do echo "### VAR=$v" >> $O; ICD_COARSE_VAR=$v timeout 120 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 10 2>&1 | grep -E "mode=auto|stamps|parity" >> $O; done
for v in 512 576; do echo "### VAR=$v full parity" >> $O; ICD_COARSE_VAR=$v timeout 300 ./icd_selftest --oracle ../../oracle/libicd_oracle.so 2>&1 | grep -E "FAIL|passed" >> $O; done
cat $O
