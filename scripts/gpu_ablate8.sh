cd rag_project_icd10_amd/csrc
O=../../gpurun_out/ablate8.log
: > $O
for v in 641 8833 17025 25217 513 576; do echo "### VAR=$v" >> $O; ICD_COARSE_VAR=$v timeout 120 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 10 2>&1 | grep -E "mode=auto|stamps" >> $O; done
cat $O
