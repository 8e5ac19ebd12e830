#!/usr/bin/env python3
"""Uncertified queries per 200 000 Gaussian queries (20 batches of 10 000 x 40 474 rows) and the time per batch, by k, with
the automatic list plan and with denser plans (icd_index_set_chunks): what an uncertified query costs at larger k (the
streaming fallback with 64-entry lists: ~0.15 ms for two queries) against what more lists cost the coarse pass."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import icd_levels, unit_rows  # noqa: E402
from rag_project_icd10_amd._native import MODE_AUTO, IcdIndex  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40474
corpus = unit_rows(n, 768, 1234)
idx = IcdIndex(corpus, icd_levels(n, 1235), max_nq=10000, max_k=128)
batches = [torch.from_numpy(unit_rows(10000, 768, 5000 + s)).cuda() for s in range(20)]
for _ in range(60):
    idx.search_reweighted(batches[0], 10, MODE_AUTO)
torch.cuda.synchronize()
for k in (10, 20, 32, 48, 64, 100):
    for chunks in (0, 12, 16, 20, 24):
        if chunks and chunks * 16 < 2 * k:
            continue
        idx.set_chunks(chunks)
        for b in batches[:3]:
            idx.search_reweighted(b, k, MODE_AUTO)
        torch.cuda.synchronize()
        tot, lists = 0, 0
        t0 = time.perf_counter()
        for b in batches:
            idx.search_reweighted(b, k, MODE_AUTO)
            st = idx.stats()
            tot += st["last_fallback"]
            lists = st["last_chunks"]
        dt = (time.perf_counter() - t0) / len(batches)
        print(f"n {n} k {k:3d} plan {'auto' if not chunks else 'chunks=%d' % chunks:10s} lists {lists:2d}: {tot:4d} uncertified in 200 000 queries, {dt * 1e3:.3f} ms per batch (incl. one stats() wait)", flush=True)
idx.close()
