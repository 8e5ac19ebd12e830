cd rag_project_icd10_amd/csrc
O=../../gpurun_out/ab.log
: > $O
timeout 600 ./icd_selftest --oracle ../../oracle/libicd_oracle.so 2>&1 | grep -E "FAIL|passed" >> $O
for rep in 1 2; do
echo "### flat" >> $O; timeout 120 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 20 2>&1 | grep -E "mode=auto|parity" >> $O
echo "### old grid VAR=512" >> $O; ICD_COARSE_VAR=512 timeout 120 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 20 2>&1 | grep -E "mode=auto" >> $O
done
echo "### flat nq=125000" >> $O; timeout 300 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 5 --nq 125000 --n 37000 2>&1 | grep -E "mode=auto|parity" >> $O
echo "### old nq=125000" >> $O; ICD_COARSE_VAR=512 timeout 300 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 5 --nq 125000 --n 37000 2>&1 | grep -E "mode=auto|parity" >> $O
echo "### flat nq=1000 n=40474" >> $O; timeout 300 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 10 --nq 1000 --n 40474 2>&1 | grep -E "mode=auto|parity" >> $O
echo "### old nq=1000 n=40474" >> $O; ICD_COARSE_VAR=512 timeout 300 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 10 --nq 1000 --n 40474 2>&1 | grep -E "mode=auto|parity" >> $O
cat $O
