python - <<'PY'
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
from conftest import icd_levels, unit_rows
from rag_project_icd10_amd._native import IcdIndex, MODE_AUTO
corpus, levels = unit_rows(37000, 768, 1234), icd_levels(37000, 1235)
q = unit_rows(10000, 768, 4321)
idx = IcdIndex(corpus, levels, max_nq=10000, max_k=10)
for nbad in (0, 8, 100, 200, 1000, 3000):
    qq = q.copy()
    if nbad: qq[:nbad, 3] = 1e6
    dq = torch.from_numpy(qq).cuda()
    for _ in range(3): idx.search_reweighted(dq, 10, MODE_AUTO)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): idx.search_reweighted(dq, 10, MODE_AUTO)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    idx.set_profiling(True); idx.profile_summary()
    for _ in range(5): idx.search_reweighted(dq, 10, MODE_AUTO)
    torch.cuda.synchronize(); pr = idx.profile_summary(); idx.set_profiling(False)
    print("flagged %4d: %.3f ms per step (fallback %d) | coarse %.3f finalize %.3f exact %.3f exact_fin %.3f" % (nbad, dt * 1e3, idx.stats()["last_fallback"], pr["ms_coarse"], pr["ms_finalize"], pr["ms_exact"], pr["ms_exact_finalize"]))
idx.close()
PY
cd rag_project_icd10_amd/csrc
timeout 300 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 10 --nq 1000 --n 40474 2>&1 | grep -E "mode=exact|parity"
timeout 300 ./icd_selftest --oracle ../../oracle/libicd_oracle.so --skip-cases --bench --iters 3 --nq 10000 --n 37000 2>&1 | grep -E "mode=exact|parity"
timeout 600 ./icd_selftest --oracle ../../oracle/libicd_oracle.so 2>&1 | grep -E "FAIL|passed"
cd ../..
bash scripts/gpu_check.sh
