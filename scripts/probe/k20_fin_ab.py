"""A/B of the rescoring walk at the serving path's k = 20 (and k = 10, 32) on the headline data: icd_debug_set_family_order
bit 1 = windows of more than 16 rows walked two lanes per row (32 rows per pass) instead of four (16 per pass)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import icd_levels, unit_rows
from rag_project_icd10_amd import _native
from rag_project_icd10_amd._native import IcdIndex, MODE_AUTO
lib = _native.load_library()
corpus, queries = unit_rows(37000, 768, 1234), unit_rows(10000, 768, 4321)
idx = IcdIndex(corpus, icd_levels(37000, 1235), max_nq=10000, max_k=32)
dq = torch.from_numpy(queries).cuda()
for k in (10, 20, 32):
    ref = None
    for rnd in range(3):
        for bits in (1, 3):
            lib.icd_debug_set_family_order(bits)
            for _ in range(5): out = idx.search_reweighted(dq, k, MODE_AUTO)
            torch.cuda.synchronize()
            idx.set_profiling(True); idx.profile_summary()
            t0 = time.perf_counter()
            for _ in range(20): out = idx.search_reweighted(dq, k, MODE_AUTO)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20 * 1e3
            p = idx.profile_summary(); idx.set_profiling(False)
            ref = ref or out
            print(f"k={k} pair_walk={bits >> 1}: {dt:.4f} ms finalize {p['ms_finalize']:.4f} same {all(torch.equal(a, b) for a, b in zip(out, ref))}", flush=True)
lib.icd_debug_set_family_order(3)
idx.close()
