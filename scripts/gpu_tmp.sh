timeout -k 10 120 scripts/probe/gemm_x 2>&1 | tail -8
