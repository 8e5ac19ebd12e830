mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "config2 or config3 or full_size" 2>&1 | tail -30) > gpurun_out/n2_tests.log
(timeout 600 python scripts/bench_e2e.py 2>&1 | tail -45) > gpurun_out/e2e.log
cat gpurun_out/n2_tests.log gpurun_out/e2e.log
