cd rag_project_icd10_amd/csrc
O=../../gpurun_out/ablate2.log
: > $O
echo "### default (VAR=0, select v2): full parity + bench" >> $O
timeout 300 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --bench --iters 10 2>&1 | grep -vE "^\[PASS\]" >> $O
echo "### VAR=16 (overlapped select): full parity + bench" >> $O
ICD_COARSE_VAR=16 timeout 300 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --bench --iters 10 2>&1 | grep -vE "^\[PASS\]" >> $O
for v in 1 17 8 24; do echo "### VAR=$v" >> $O; ICD_COARSE_VAR=$v timeout 120 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 10 2>&1 | grep -E "mode=auto|stamps|parity" >> $O; done
cat $O
