# round 3, first kernel-side check: the GPU suite, the family-corpus probe, the bench line
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -25) > gpurun_out/pytest_gpu.log
(timeout 600 python scripts/probe/family_corpus_probe.py large 2>&1 | tail -40) > gpurun_out/family_probe_large.log
(timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench.log
cat gpurun_out/pytest_gpu.log; cat gpurun_out/family_probe_large.log; cut -c1-1500 gpurun_out/bench.log
