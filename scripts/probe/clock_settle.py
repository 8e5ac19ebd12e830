#!/usr/bin/env python3
"""How long does the step time take to settle after a process starts? Chunks of 10 steps, each chunk bracketed by a
synchronize, right after index creation (bench.py's default is 5 warm-up steps)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import icd_levels, unit_rows
from rag_project_icd10_amd._native import IcdIndex, MODE_AUTO
q = torch.from_numpy(unit_rows(10000, 768, 4321)).cuda()
idx = IcdIndex(unit_rows(37000, 768, 1234), icd_levels(37000, 1235), max_nq=10000, max_k=10)
torch.cuda.synchronize()
out = []
for chunk in range(40):
    t0 = time.perf_counter()
    for _ in range(10):
        idx.search_reweighted(q, 10, MODE_AUTO)
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / 10 * 1e3)
print("ms per step, chunks of 10 steps:", " ".join(f"{x:.3f}" for x in out))
