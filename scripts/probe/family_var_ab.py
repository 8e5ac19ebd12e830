"""family corpus (300 x 124, cosine 0.99, 10 000 queries, k = 10) in wide mode under ICD_FLAT_VAR of the A/B library
(ICD_SEARCH_LIB=rag_project_icd10_amd/csrc/ab/libicdsearch.so): coarse-kernel variants where a third of a list's tiles are
bootstrap tiles."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import family_rows, icd_levels, unit_rows
from rag_project_icd10_amd._native import IcdIndex, MODE_AUTO
corpus, queries = family_rows(300, 124, 768, 0.10, 10000, 7)
idx = IcdIndex(corpus, icd_levels(len(corpus), 8), max_nq=10000, max_k=20)
dq = torch.from_numpy(queries).cuda()
for _ in range(5): idx.search_reweighted(dq, 10, MODE_AUTO)
torch.cuda.synchronize()
idx.set_profiling(True); idx.profile_summary()
t0 = time.perf_counter()
for _ in range(20): idx.search_reweighted(dq, 10, MODE_AUTO)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20 * 1e3
p = idx.profile_summary(); st = idx.stats()
print(f"VAR={os.environ.get('ICD_FLAT_VAR')}: family {dt:.3f} ms coarse {p['ms_coarse']:.4f} finalize {p['ms_finalize']:.4f} wide {st['wide_mode']} fallback {st['last_fallback']}", flush=True)
idx.close()
