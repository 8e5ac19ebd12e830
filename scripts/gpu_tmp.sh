set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "services_end_to_end or build" 2>&1 | tail -2
python scripts/bench_build.py > gpurun_out/build_packed.json 2> gpurun_out/build_packed.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/build_packed.json")); print(d["build_s"], d["rows_per_s"], d["synchronised_run"])
PY
