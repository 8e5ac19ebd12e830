(timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -5) > gpurun_out/pytest_gpu.log
(timeout 600 python scripts/bench_e2e.py 2>/dev/null | tail -60) > gpurun_out/e2e.json
cat gpurun_out/pytest_gpu.log; grep -A16 stages_ms gpurun_out/e2e.json
