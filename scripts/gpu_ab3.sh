cd rag_project_icd10_amd/csrc
O=../../gpurun_out/ab3.log
: > $O
for m in 0 1 2 3; do
echo "### XCD_MODE=$m bench shape" >> $O; ICD_XCD_MODE=$m timeout 120 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 20 2>&1 | grep -E "mode=auto" >> $O
done
for m in 0 1 2 3; do
echo "### XCD_MODE=$m nq=16384 n=1250000" >> $O; ICD_XCD_MODE=$m timeout 900 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 2 --nq 16384 --n 1250000 2>&1 | grep -E "mode=auto" >> $O
done
for m in 0 3; do
echo "### XCD_MODE=$m nq=1000 n=40474" >> $O; ICD_XCD_MODE=$m timeout 300 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 10 --nq 1000 --n 40474 2>&1 | grep -E "mode=auto" >> $O
echo "### XCD_MODE=$m nq=40000 n=37000" >> $O; ICD_XCD_MODE=$m timeout 300 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 5 --nq 40000 --n 37000 2>&1 | grep -E "mode=auto" >> $O
done
cat $O
