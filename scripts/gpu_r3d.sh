# s_setprio A/B, more rounds (the first three were 2:1)
bash scripts/gpu_ab_flat.sh "34971 166043" 8 > /dev/null 2>&1
grep -E "###|coarse=" gpurun_out/ab_flat.log | sed -e 's/.*coarse=\([0-9.]*\).*/\1/' | paste - - 
