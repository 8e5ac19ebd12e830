"""Per-kernel durations of the second coarse pass when only a handful of queries need it (run under
rocprofv3 --kernel-trace --stats): 300 families x 124 rows, spread 0.35 (about 15 of 10 000 queries are flagged), and the
tight families (all flagged)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts", "probe"))
from conftest import icd_levels
from family_corpus_probe import family
from rag_project_icd10_amd._native import IcdIndex, MODE_AUTO

spread = float(sys.argv[1]) if len(sys.argv) > 1 else 0.35
corpus, queries = family(300, 124, 768, spread, 10000, 7)
idx = IcdIndex(corpus, icd_levels(len(corpus), 8), max_nq=10000, max_k=20)
idx.set_second_pass(True, adaptive=False)
dq = torch.from_numpy(queries).cuda()
for _ in range(12):
    idx.search_reweighted(dq, 10, MODE_AUTO)
torch.cuda.synchronize()
print(idx.stats())
