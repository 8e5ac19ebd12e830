mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -5) > gpurun_out/pytest_gpu.log
(timeout 600 python bench.py --steps 50 --warmup 5 2>&1 | tail -1) > gpurun_out/bench.log
cat gpurun_out/pytest_gpu.log gpurun_out/bench.log
