mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -8) > gpurun_out/midcheck_tests.log
(timeout 900 python scripts/gpu_fuzz.py --cases 48 --seed 11 2>&1 | tail -60) > gpurun_out/fuzz.log
bash scripts/gpu_ab_dirs.sh "ab" "0 139 155 1163 1179" 2 > /dev/null 2>&1
cat gpurun_out/midcheck_tests.log; tail -4 gpurun_out/fuzz.log; grep -c FAIL gpurun_out/fuzz.log; grep -E "DIR=|coarse=|stamps" gpurun_out/ab_dirs.log | cut -c1-300
