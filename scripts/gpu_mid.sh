cd rag_project_icd10_amd/csrc
for nq in 129 300 1000 3000; do echo "### nq=$nq n=40474"; timeout 300 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 10 --nq $nq --n 40474 2>&1 | grep -E "mode=auto|parity"; done
timeout 120 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 20 2>&1 | grep -E "mode=auto|parity"
timeout 600 ./icd_selftest --oracle ../../oracle/libicd_oracle.so 2>&1 | grep -E "FAIL|passed"
