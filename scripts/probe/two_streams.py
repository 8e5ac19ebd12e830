import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
from conftest import icd_levels, unit_rows
from rag_project_icd10_amd._native import IcdIndex, MODE_AUTO
corpus, levels = unit_rows(37000, 768, 1234), icd_levels(37000, 1235)
q = torch.from_numpy(unit_rows(10000, 768, 4321)).cuda()
A = IcdIndex(corpus, levels, max_nq=10000, max_k=10)
B = IcdIndex(corpus, levels, max_nq=10000, max_k=10)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def seq_one():
    A.search_reweighted(q, 10, MODE_AUTO)
def seq_halves():
    A.search_reweighted(q[:5000], 10, MODE_AUTO); A.search_reweighted(q[5000:], 10, MODE_AUTO)
def par_halves():
    with torch.cuda.stream(s1): A.search_reweighted(q[:5000], 10, MODE_AUTO)
    with torch.cuda.stream(s2): B.search_reweighted(q[5000:], 10, MODE_AUTO)
def par_full():   # two full batches concurrently (20000 queries)
    with torch.cuda.stream(s1): A.search_reweighted(q, 10, MODE_AUTO)
    with torch.cuda.stream(s2): B.search_reweighted(q, 10, MODE_AUTO)
for name, fn, nq in (("one call of 10000", seq_one, 10000), ("two calls of 5000, one stream", seq_halves, 10000), ("two calls of 5000, two streams", par_halves, 10000), ("two calls of 10000, two streams", par_full, 20000)):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print(f"{name}: {dt*1e3:.3f} ms -> {nq/dt/1e6:.2f} Mq/s")
