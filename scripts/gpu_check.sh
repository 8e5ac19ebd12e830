# GPU validation: parity tests, smoke, the bench line (default + row-sharded at N = 1)
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -40) > gpurun_out/pytest_gpu.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5) > gpurun_out/smoke.log
(timeout 600 python bench.py --steps 50 --warmup 5 2>&1 | tail -1) > gpurun_out/bench.log
(timeout 600 python bench.py --workload rowshard --steps 2 2>&1 | tail -1) > gpurun_out/bench_rowshard.log
cat gpurun_out/pytest_gpu.log gpurun_out/smoke.log gpurun_out/bench.log gpurun_out/bench_rowshard.log
