#!/usr/bin/env python3
"""20 wide-mode batches (10 000 queries, k = 10) on the 300 x 124 family corpus: the workload of a rocprofv3 kernel summary."""
import os
import sys


ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from rag_project_icd10_amd._native import IcdIndex  # noqa: E402
from test_gpu_parity import _tight_family_corpus, icd_levels  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 10
corpus, queries = _tight_family_corpus(300, 124, 768, 0.10, 10000, 7)
idx = IcdIndex(corpus, icd_levels(len(corpus), 8), max_nq=10000, max_k=20)
dq = torch.from_numpy(queries).cuda()
for _ in range(20):
    idx.search_reweighted(dq, k)
torch.cuda.synchronize()
print(idx.stats(), file=sys.stderr)
