cd rag_project_icd10_amd/csrc
O=../../gpurun_out/kp.log
: > $O
for rep in 1 2 3; do timeout 120 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 20 2>&1 | grep -E "mode=auto|parity" >> $O; done
timeout 300 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 5 --nq 125000 --n 37000 2>&1 | grep -E "mode=auto|parity" >> $O
timeout 300 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 10 --nq 1000 --n 40474 2>&1 | grep -E "mode=auto|parity" >> $O
timeout 600 ./icd_selftest --oracle ../../oracle/libicd_oracle.so 2>&1 | grep -E "FAIL|passed" >> $O
cat $O
