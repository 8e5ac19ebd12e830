"""A/B of the wide-window finalize on the family corpus (300 x 124 rows, cosine 0.99, 10 000 queries, k = 10 / 20):
the per-index option family_order: 0 = batch order (round 3), 1 = family order (shipped)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import family_rows, icd_levels
from rag_project_icd10_amd import _native
from rag_project_icd10_amd._native import IcdIndex, MODE_AUTO

lib = _native.load_library()
corpus, queries = family_rows(300, 124, 768, 0.10, 10000, 7)
levels = icd_levels(len(corpus), 8)
dq = torch.from_numpy(queries).cuda()
ref = None
for k in (10, 20):
    idx = IcdIndex(corpus, levels, max_nq=10000, max_k=20)
    for rnd in range(2):
        for bits in (0, 1):
            idx.set_option("family_order", bits)
            for _ in range(3):
                out = idx.search_reweighted(dq, k, MODE_AUTO)
            torch.cuda.synchronize()
            idx.set_profiling(True); idx.profile_summary()
            t0 = time.perf_counter()
            for _ in range(10):
                out = idx.search_reweighted(dq, k, MODE_AUTO)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10 * 1e3
            p = idx.profile_summary(); idx.set_profiling(False); st = idx.stats()
            if ref is None or ref[0] != k:
                ref = (k, out)
            same = all(torch.equal(a, b) for a, b in zip(out, ref[1]))
            print(f"k={k} bits={bits}: {dt:.3f} ms | coarse {p['ms_coarse']:.3f} finalize {p['ms_finalize']:.3f} | wide {st['wide_mode']} fallback {st['last_fallback']} same_as_first {same}", flush=True)
    idx.close()
