python - <<'PY'
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
from conftest import icd_levels, unit_rows
from rag_project_icd10_amd._native import IcdIndex, MODE_AUTO
corpus, levels = unit_rows(40474, 768, 1234), icd_levels(40474, 1235)
q = unit_rows(10000, 768, 4321)
idx = IcdIndex(corpus, levels, max_nq=10000, max_k=128)
for nq, k in ((10000, 10), (10000, 20), (10000, 32), (10000, 48), (10000, 64), (10000, 100), (1000, 20), (1000, 100), (16, 20), (1, 100), (1000, 128)):
    dq = torch.from_numpy(q[:nq]).cuda()
    for _ in range(3): idx.search_reweighted(dq, k, MODE_AUTO)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    it = 20 if nq < 5000 else 5
    for _ in range(it): idx.search_reweighted(dq, k, MODE_AUTO)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / it
    print("nq %5d k %3d: %.3f ms  %.3f Mq/s (mode %d lists %d fallback %d)" % (nq, k, dt * 1e3, nq / dt / 1e6, idx.stats()["last_mode"], idx.stats()["last_chunks"], idx.stats()["last_fallback"]))
idx.close()
PY
