# round-end evidence: tests, smoke, bench (default + rowshard), e2e, rocprof kernel stats + PMC passes
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6) > gpurun_out/pytest_gpu.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > gpurun_out/smoke.log
(timeout 600 python bench.py --steps 50 --warmup 5 2>&1 | tail -1) > gpurun_out/bench.log
(timeout 600 python bench.py --workload rowshard --steps 3 2>&1 | tail -1) > gpurun_out/bench_rowshard.log
(timeout 600 python scripts/bench_e2e.py 2>/dev/null | tail -50) > gpurun_out/e2e.json
bash scripts/gpu_pmc.sh r02 > gpurun_out/pmc_r02.log 2>&1
cat gpurun_out/pytest_gpu.log gpurun_out/smoke.log; cut -c1-600 gpurun_out/bench.log; cut -c1-300 gpurun_out/bench_rowshard.log; grep -A12 stages_ms gpurun_out/e2e.json; tail -25 gpurun_out/pmc_r02.log | cut -c1-400
