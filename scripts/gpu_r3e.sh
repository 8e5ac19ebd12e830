# round 3: suite + bench (default) + rowshard
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -25) > gpurun_out/pytest_gpu.log
(timeout 600 python bench.py --steps 20 --warmup 5 2>gpurun_out/bench.err | tail -1) > gpurun_out/bench.log
(timeout 600 python bench.py --workload rowshard --steps 3 2>&1 | tail -1) > gpurun_out/bench_rowshard.log
cat gpurun_out/pytest_gpu.log; cut -c1-1200 gpurun_out/bench.log; cut -c1-600 gpurun_out/bench_rowshard.log
