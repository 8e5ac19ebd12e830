bash scripts/gpu_ab_flat.sh "34971 1083547" 6 > /dev/null 2>&1
grep -E "###|coarse=|FAIL" gpurun_out/ab_flat.log | sed -e 's/.*coarse=\([0-9.]*\).*fallback=\([0-9]*\).*/\1 fb=\2/' | paste - - 
grep -c PASS gpurun_out/ab_flat.log; grep -c FAIL gpurun_out/ab_flat.log
