"""Probe: certification on anisotropic unit rows x = normalise(mu0 + s e / sqrt(dim)) (mean pairwise cosine 1 / (1 + s^2)),
centred against uncentred fp16 image (the per-index flag center), several s / batch sizes."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import icd_levels
from rag_project_icd10_amd import _native
from rag_project_icd10_amd._native import IcdIndex, MODE_AUTO, MODE_EXACT

lib = _native.load_library()
rng = np.random.default_rng(31)
dim = 768
mu0 = rng.standard_normal(dim).astype(np.float32); mu0 /= np.linalg.norm(mu0)
def rows(m, s):
    x = mu0[None, :] + (s / np.sqrt(dim)) * rng.standard_normal((m, dim)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return np.ascontiguousarray(x, dtype=np.float32)
for s in (0.07, 0.2, 0.5, 1.0):
    for n, nq in ((20000, 2000), (37000, 10000)):
        corpus, queries = rows(n, s), rows(nq, s)
        levels = icd_levels(n, 32)
        dq = torch.from_numpy(queries).cuda()
        for center in (1, 0):
            idx = IcdIndex(corpus, levels, max_nq=nq, max_k=20, center=bool(center))
            for k in (10,):
                for _ in range(3):
                    out = idx.search_reweighted(dq, k, MODE_AUTO)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    out = idx.search_reweighted(dq, k, MODE_AUTO)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 5 * 1e3
                st = idx.stats()
                ex = idx.search_reweighted(dq, k, MODE_EXACT)
                same = all(torch.equal(a, b) for a, b in zip(out, ex))
                print(f"s={s} n={n} nq={nq} k={k} centered={st['centered']} share={st['mean_share']:.4f}: {dt:.3f} ms lists {st['last_chunks']} second_pass {st['last_second_pass']} x{st['last_second_pass_lists']} "
                      f"exact_research {st['last_fallback']} wide {st['wide_mode']} exact_equal {same}", flush=True)
            idx.close()
