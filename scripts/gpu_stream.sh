# streaming-kernel check: parity self-test, tiny batches, the config-5 shard fallback
cd rag_project_icd10_amd/csrc
O=../../gpurun_out/stream.log
: > $O
timeout 600 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --quick 2>&1 | tail -5 >> $O
for nq in 1 2 4 8 16; do
  echo "### nq=$nq n=40474" >> $O
  timeout 300 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 20 --nq $nq --n 40474 2>&1 | grep -E "mode=auto|parity" >> $O
done
echo "### nq=64 n=40474" >> $O
timeout 300 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 20 --nq 64 --n 40474 2>&1 | grep -E "mode=|parity" >> $O
echo "### nq=16384 n=1250000 (config 5 per-GPU shard, query subset)" >> $O
timeout 900 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 2 --nq 16384 --n 1250000 2>&1 | grep -E "mode=auto|parity" >> $O
cat $O
