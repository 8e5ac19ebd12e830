set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python scripts/probe/sparse_incident.py > gpurun_out/sparse_incident.log 2>&1
cat gpurun_out/sparse_incident.log
python scripts/gpu_fuzz.py --seed 77 --cases 40 > gpurun_out/fuzz77.log 2>&1
tail -2 gpurun_out/fuzz77.log
