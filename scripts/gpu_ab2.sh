cd rag_project_icd10_amd/csrc
O=../../gpurun_out/ab2.log
: > $O
timeout 600 ./icd_selftest --oracle ../../oracle/libicd_oracle.so 2>&1 | grep -E "FAIL|passed" >> $O
for rep in 1 2; do
echo "### flat + xcd class remap" >> $O; timeout 120 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 20 2>&1 | grep -E "mode=auto|parity" >> $O
echo "### flat, xcd swizzle only" >> $O; ICD_NO_XCD_REMAP=1 timeout 120 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 20 2>&1 | grep -E "mode=auto" >> $O
done
echo "### remap nq=125000" >> $O; timeout 300 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 5 --nq 125000 --n 37000 2>&1 | grep -E "mode=auto|parity" >> $O
echo "### swizzle-only nq=125000" >> $O; ICD_NO_XCD_REMAP=1 timeout 300 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 5 --nq 125000 --n 37000 2>&1 | grep -E "mode=auto|parity" >> $O
echo "### remap nq=1000 n=40474" >> $O; timeout 300 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 10 --nq 1000 --n 40474 2>&1 | grep -E "mode=auto|parity" >> $O
echo "### remap nq=16384 n=1250000" >> $O; timeout 900 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 2 --nq 16384 --n 1250000 2>&1 | grep -E "mode=auto|parity" >> $O
echo "### swizzle-only nq=16384 n=1250000" >> $O; ICD_NO_XCD_REMAP=1 timeout 900 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 2 --nq 16384 --n 1250000 2>&1 | grep -E "mode=auto|parity" >> $O
cat $O
