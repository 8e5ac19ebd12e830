set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "family_corpus_is_certified" -s 2>&1 | grep "family corpus" > gpurun_out/family_probe.log
cat gpurun_out/family_probe.log
python scripts/gpu_fuzz.py --seed 78 --cases 60 > gpurun_out/fuzz78.log 2>&1
tail -1 gpurun_out/fuzz78.log
