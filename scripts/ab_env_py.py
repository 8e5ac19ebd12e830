#!/usr/bin/env python3
"""One timing of the search through the Python binding with the ablation library (ICD_SEARCH_LIB=.../ab/libicdsearch.so):
per-kernel split for (n, nq, k) given on the command line; the environment carries the A/B switches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import icd_levels, unit_rows
from rag_project_icd10_amd._native import IcdIndex, MODE_AUTO
n, nq, k = (int(x) for x in sys.argv[1:4])
q = unit_rows(nq, 768, 4321)
corpus, levels = unit_rows(n, 768, 1234), icd_levels(n, 1235)
idx = IcdIndex(corpus, levels, max_nq=nq, max_k=128)
dq = torch.from_numpy(q).cuda()
for _ in range(3): idx.search_reweighted(dq, k, MODE_AUTO)
torch.cuda.synchronize()
idx.set_profiling(True); idx.profile_summary()
t0 = time.perf_counter(); it = 20
for _ in range(it): idx.search_reweighted(dq, k, MODE_AUTO)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / it
p = idx.profile_summary(); st = idx.stats()
print("n %5d nq %5d k %3d: %.3f ms | prep %.3f coarse %.3f finalize %.3f exact %.3f exact_fin %.3f | lists %d fallback %d" % (
    n, nq, k, dt * 1e3, p["ms_prep"], p["ms_coarse"], p["ms_finalize"], p["ms_exact"], p["ms_exact_finalize"], st["last_chunks"], st["last_fallback"]))
