import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from conftest import icd_levels, unit_rows
from rag_project_icd10_amd._native import IcdIndex, MODE_AUTO
corpus = unit_rows(37000, 768, 1234)
idx = IcdIndex(corpus, icd_levels(37000, 1235), max_nq=10000, max_k=10)
tot = 0
for seed in range(20):
    q = torch.from_numpy(unit_rows(10000, 768, 5000 + seed)).cuda()
    idx.search_reweighted(q, 10, MODE_AUTO)
    tot += idx.stats()["last_fallback"]
print(os.environ.get("ICD_FLAT_VAR", "product"), "fallbacks in 200000 queries:", tot)
