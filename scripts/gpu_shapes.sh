# per-kernel split (hipEvents inside the library) over sizes and k: where the time goes away from the headline shape
python - <<'PY' > gpurun_out/shapes.log 2>&1
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
from conftest import icd_levels, unit_rows
from rag_project_icd10_amd._native import IcdIndex, MODE_AUTO
q = unit_rows(10000, 768, 4321)
for n in (37000, 40474):
    corpus, levels = unit_rows(n, 768, 1234), icd_levels(n, 1235)
    idx = IcdIndex(corpus, levels, max_nq=10000, max_k=128)
    for nq, k in ((10000, 10), (10000, 20), (10000, 32), (10000, 64), (10000, 100), (1000, 10), (1000, 100)):
        dq = torch.from_numpy(q[:nq]).cuda()
        for _ in range(3): idx.search_reweighted(dq, k, MODE_AUTO)
        torch.cuda.synchronize()
        idx.set_profiling(True); idx.profile_summary()
        t0 = time.perf_counter()
        it = 20
        for _ in range(it): idx.search_reweighted(dq, k, MODE_AUTO)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / it
        p = idx.profile_summary(); idx.set_profiling(False)
        st = idx.stats()
        print("n %5d nq %5d k %3d: %.3f ms %.2f Mq/s | prep %.3f coarse %.3f finalize %.3f exact %.3f exact_fin %.3f | lists %d fallback %d" % (
            n, nq, k, dt * 1e3, nq / dt / 1e6, p["ms_prep"], p["ms_coarse"], p["ms_finalize"], p["ms_exact"], p["ms_exact_finalize"], st["last_chunks"], st["last_fallback"]), flush=True)
    idx.close()
PY
cat gpurun_out/shapes.log
