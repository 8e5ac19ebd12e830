set -e
cd $GRAFT_REPO_ROOT
python scripts/probe/gemm_m_probe.py > gpurun_out/gemm_m.log 2>&1
cat gpurun_out/gemm_m.log
