timeout -k 10 120 scripts/probe/gemm_x 2>&1 | tail -8
timeout -k 10 60 scripts/probe/bare_mfma 90 2>&1 | tail -2
(timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "config4_per_gpu or end_to_end" 2>&1 | tail -4)
