cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/pytest_full.log 2>&1
grep -E "^(FAILED|ERROR)|^E  " gpurun_out/pytest_full.log | head -30
tail -3 gpurun_out/pytest_full.log
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "family_corpus_is_certified" -s 2>&1 | grep "family corpus" > gpurun_out/family_probe.log
cat gpurun_out/family_probe.log
