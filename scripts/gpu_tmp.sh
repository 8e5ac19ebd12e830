set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_encoder_gpu.py tests/test_ner_gpu.py -x -q -m gpu 2>&1 | tail -2
python scripts/bench_e2e.py > gpurun_out/e2e_packed.json 2> gpurun_out/e2e_packed.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/e2e_packed.json")); print(d["stages_ms"], d["pipeline_strings_per_s"])
PY
