bash scripts/gpu_ab_dirs.sh "abp abw" "x" 6 > /dev/null 2>&1
grep -E "###|finalize=|FAIL" gpurun_out/ab_dirs.log | sed -e 's/.*finalize=\([0-9.]*\).*/\1/' | paste - - 
grep -c PASS gpurun_out/ab_dirs.log
