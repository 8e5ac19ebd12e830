cd rag_project_icd10_amd/csrc
O=../../gpurun_out/flat.log
: > $O
timeout 600 ./icd_selftest --oracle ../../oracle/libicd_oracle.so 2>&1 | grep -E "FAIL|passed|mode=" >> $O
for c in 0 1 2 5; do echo "### chunks=$c" >> $O; timeout 120 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 10 --chunks $c 2>&1 | grep -E "mode=auto|parity" >> $O; done
echo "### nq=125000 n=37000" >> $O
timeout 300 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 5 --nq 125000 --n 37000 2>&1 | grep -E "mode=auto|parity" >> $O
echo "### nq=1000 n=40474" >> $O
timeout 300 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 10 --nq 1000 --n 40474 2>&1 | grep -E "mode=auto|parity" >> $O
echo "### nq=16384 n=1250000" >> $O
timeout 900 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 2 --nq 16384 --n 1250000 2>&1 | grep -E "mode=auto|parity" >> $O
cat $O
cd ../..
bash scripts/gpu_check.sh
