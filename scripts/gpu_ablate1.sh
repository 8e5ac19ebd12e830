cd rag_project_icd10_amd/csrc
O=../../gpurun_out/ablate1.log
: > $O
for v in 0 1 2 3 4 8; do echo "### VAR=$v" >> $O; ICD_COARSE_VAR=$v timeout 120 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 10 2>&1 | grep -E "mode=auto|stamps" >> $O; done
for c in 1 2 3 4 6; do echo "### chunks=$c" >> $O; timeout 120 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 10 --chunks $c 2>&1 | grep -E "mode=auto|parity" >> $O; done
cat $O
