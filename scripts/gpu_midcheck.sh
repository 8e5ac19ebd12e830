# mid-round check: the GPU suite and the fuzzer (different seed per call: gpu_midcheck.sh [seed])
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -8) > gpurun_out/midcheck_tests.log
(timeout 900 python scripts/gpu_fuzz.py --cases 48 --seed ${1:-11} 2>&1 | tail -60) > gpurun_out/fuzz.log
cat gpurun_out/midcheck_tests.log; tail -4 gpurun_out/fuzz.log; grep -c FAIL gpurun_out/fuzz.log
