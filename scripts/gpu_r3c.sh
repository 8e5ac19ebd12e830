# round 3: the second pass after the placement fix, s_setprio A/B, bare MFMA stream per shape
mkdir -p gpurun_out
(timeout 600 python scripts/probe/family_corpus_probe.py large 2>&1 | grep -v amdgpu.ids | tail -40) > gpurun_out/family_probe_large.log
bash scripts/gpu_ab_flat.sh "34971 166043" 3 > /dev/null 2>&1
cp gpurun_out/ab_flat.log gpurun_out/ab_setprio.log
(timeout 300 scripts/probe/bare_mfma 90 2>&1) > gpurun_out/bare_mfma.log
cat gpurun_out/family_probe_large.log; cat gpurun_out/ab_setprio.log | cut -c1-200; cat gpurun_out/bare_mfma.log
