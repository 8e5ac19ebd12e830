(timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4) > gpurun_out/pytest_gpu.log
(timeout 600 python scripts/gpu_fuzz.py --cases 48 --seed 51 2>&1 | tail -2) > gpurun_out/fuzz.log
(timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1) > gpurun_out/bench_q.log
cat gpurun_out/pytest_gpu.log gpurun_out/fuzz.log; python - <<'PY'
import json
l=json.loads(open('gpurun_out/bench_q.log').read())
print(l['value'], l['ms_per_step'], l['kernel_ms'], l['extra']['windows'], l['ids_exact'])
PY
