import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from conftest import icd_levels, unit_rows
from rag_project_icd10_amd._native import IcdIndex, MODE_AUTO
corpus, q = unit_rows(37000, 1024, 1), unit_rows(10000, 1024, 2)
idx = IcdIndex(corpus, icd_levels(37000, 3), max_nq=10000, max_k=10)
dq = torch.from_numpy(q).cuda()
s0 = idx.search_reweighted(dq, 10, MODE_AUTO)
for rep in range(2):
    for _ in range(3): idx.search_reweighted(dq, 10, MODE_AUTO)
    torch.cuda.synchronize(); idx.set_profiling(True); idx.profile_summary()
    for _ in range(10): idx.search_reweighted(dq, 10, MODE_AUTO)
    torch.cuda.synchronize(); p = idx.profile_summary(); idx.set_profiling(False)
    print(os.environ.get("ICD_1024_FULL", "-"), "dim 1024 coarse ms", round(p["ms_coarse"], 4), "fallback", idx.stats()["last_fallback"])
np.save("/tmp/ids_%s.npy" % os.environ.get("ICD_1024_FULL", "x"), s0[2].cpu().numpy())
