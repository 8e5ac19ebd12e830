#!/usr/bin/env python3
"""k = 100 (the largest k /query can ask for: top_k * 2 with top_k <= 50, reference models/icd_models.py:138) on the headline data:
coarse lists of 24 candidates (about k / 6 lists per query: shipped) against lists of 16 (about k / 4 lists: the k <= 64 form),
the per-index option wide_from (k above it takes the lists of 24). 10 000 queries x 37 000 rows; kernel times by the library's events; every query checked against
the oracle."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    import torch
    import oracle as orc
    from bench import icd_levels, unit_rows
    from rag_project_icd10_amd._native import IcdIndex
    n, nq, dim = 37000, 10000, 768
    corpus, levels = unit_rows(n, dim, 1234), icd_levels(n, 1235)
    queries = unit_rows(nq, dim, 4321)
    dq = torch.from_numpy(queries).cuda()
    for k in (100, 64, 50, 40):
        os_, oi = orc.flat_ip_topk(corpus, queries, k)
        want = orc.reweight(os_, oi, levels)
        for wide_from in (32, 128):   # (32: shipped since round 6; 128: lists of 16 at every k; round 5 shipped 64)
            wide = k > wide_from
            idx = IcdIndex(corpus, levels, max_nq=nq, max_k=100)
            idx.set_option("wide_from", wide_from)
            for _ in range(30):
                idx.search_reweighted(dq, k)
            torch.cuda.synchronize()
            idx.set_profiling(True, every=2)
            idx.profile_summary()
            t0 = time.perf_counter()
            for _ in range(20):
                out = idx.search_reweighted(dq, k)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 20 * 1e3
            prof = idx.profile_summary()
            st = idx.stats()
            ok = bool(np.array_equal(out[2].cpu().numpy(), want[2]) and out[0].cpu().numpy().tobytes() == want[0].tobytes())
            print(f"k {k} lists of {'24' if wide else '16'}: {ms:.3f} ms per step | coarse {prof['ms_coarse']:.3f} finalize {prof['ms_finalize']:.3f} exact {prof['ms_exact']:.3f} + {prof['ms_exact_finalize']:.3f} | "
                  f"lists per query {st['last_chunks']} second pass {st['last_second_pass']} re-searched {st['last_fallback']} | exact {ok}", flush=True)
            idx.close()


if __name__ == "__main__":
    main()
