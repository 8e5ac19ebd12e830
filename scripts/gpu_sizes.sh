cd rag_project_icd10_amd/csrc
O=../../gpurun_out/sizes.log
: > $O
echo "### nq=125000 n=37000 (config 4 per-GPU share)" >> $O
timeout 300 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 5 --nq 125000 --n 37000 2>&1 | grep -E "mode=|parity" >> $O
echo "### nq=40000 n=37000" >> $O
timeout 300 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 5 --nq 40000 --n 37000 2>&1 | grep -E "mode=auto|parity" >> $O
echo "### nq=1000 n=40474 (config 3 search stage)" >> $O
timeout 300 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 10 --nq 1000 --n 40474 2>&1 | grep -E "mode=|parity" >> $O
echo "### nq=1 n=40474 (reference call shape)" >> $O
timeout 300 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 20 --nq 1 --n 40474 2>&1 | grep -E "mode=|parity" >> $O
echo "### nq=16384 n=1250000 (config 5 per-GPU shard, query subset)" >> $O
timeout 900 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 2 --nq 16384 --n 1250000 2>&1 | grep -E "mode=auto|parity" >> $O
cat $O
