# fuzz soak: several seeds of scripts/gpu_fuzz.py (usage: scripts/gpu_soak.sh "61 62 63" [cases])
SEEDS=${1:-"61 62 63"}
CASES=${2:-48}
mkdir -p gpurun_out
: > gpurun_out/fuzz_soak.log
for seed in $SEEDS; do (timeout 900 python scripts/gpu_fuzz.py --cases $CASES --seed $seed 2>&1 | grep -E "FAIL|gpu_fuzz") >> gpurun_out/fuzz_soak.log; done
cat gpurun_out/fuzz_soak.log
