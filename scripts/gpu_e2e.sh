mkdir -p gpurun_out
(timeout 900 python scripts/bench_e2e.py 2> gpurun_out/e2e.err) > gpurun_out/e2e.json
tail -5 gpurun_out/e2e.err; cat gpurun_out/e2e.json
