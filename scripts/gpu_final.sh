# round-end evidence: tests, smoke, bench (default + rowshard), e2e, the timed full build, rocprof kernel stats + PMC passes
# usage: scripts/gpu_final.sh r03
TAG=${1:-r04}
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6) > gpurun_out/pytest_gpu.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > gpurun_out/smoke.log
(timeout 900 python bench.py --steps 50 --warmup 5 2>/dev/null | tail -1) > gpurun_out/bench.log
(timeout 600 python bench.py --workload rowshard --steps 3 2>/dev/null | tail -1) > gpurun_out/bench_rowshard.log
(timeout 600 python scripts/bench_e2e.py 2>/dev/null | tail -60) > gpurun_out/e2e.json
(timeout 600 python scripts/bench_build.py 2>/dev/null) > gpurun_out/build_full.json
(timeout 900 python scripts/bench_encoder_corpus.py 2>/dev/null) > gpurun_out/e2e_encoder_corpus.json
(timeout 600 python scripts/probe/family_fin_ab.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/family_fin_ab.log
(timeout 600 python scripts/probe/anisotropic_probe.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/anisotropic_probe.log
# two ranks on this box's ONE device: RCCL refuses duplicate devices; the gloo control flow with the HIP index on one device is what can run
(ICD_BENCH_BACKEND=gloo ICD_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --rows-per-gpu 400000 --rowshard-queries 20000 2>&1 | tail -2) > gpurun_out/bench_2rank_one_device.log
bash scripts/gpu_pmc.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1
# one-off logs of the round, reproducible in the same call (copied to profiles/ by hand: see profiles/README.md)
(timeout 300 python scripts/probe/encoder_batch_probe.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/encoder_batch.log
(timeout 300 python scripts/probe/query_latency.py 2>&1 | grep "NER o") > gpurun_out/query_latency.log
(timeout 300 python scripts/probe/sparse_incident.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/sparse_incident.log
(timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k family_corpus_is_certified -s 2>&1 | grep "family corpus") > gpurun_out/family_probe.log
for f in bench_kernel_stats.csv bench_with_extras_kernel_stats.csv bench_exact_mode_kernel_stats.csv rowshard_kernel_stats.csv pmc_counters.json pmc_counters_exact_mode.json pmc_traffic.json pmc_traffic_rowshard.json; do cp gpurun_out/${TAG}_$f gpurun_out/final_${TAG}_$f 2>/dev/null; done
cat gpurun_out/pytest_gpu.log gpurun_out/smoke.log; cut -c1-700 gpurun_out/bench.log; cut -c1-400 gpurun_out/bench_rowshard.log; grep -A14 stages_ms gpurun_out/e2e.json; cut -c1-600 gpurun_out/bench_2rank_one_device.log; tail -30 gpurun_out/pmc_$TAG.log | cut -c1-400
