for n in 37000 40474 37000 40474; do (timeout 300 python bench.py --n $n --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1) > gpurun_out/b_$n.log; python - $n <<'PY'
import json,sys
l=json.loads(open(f'gpurun_out/b_{sys.argv[1]}.log').read())
print(sys.argv[1], round(l['ms_per_step'],4), l['kernel_ms'], round(l['roofline']['frac'],4), l['coarse_chunks'])
PY
done
