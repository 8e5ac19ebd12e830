#!/usr/bin/env python3
"""What the streaming kernel's two fallback launches cost per clean search, and what a flagged query costs without them.

10 000 Gaussian queries x 37 000 x 768, k = 10. (a) clean batches with all four fallback launches (the non-adaptive test
setting) against the steady state of the default setting (the streaming pair left out after 96 clean searches),
interleaved; (b) a batch with 1 / 8 / 64 unanswerable (all-zero) queries in either state. Wall time per search over
synchronised groups. Output: profiles/rNN_sparse_fallback_policy.log
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    from rag_project_icd10_amd._native import IcdIndex
    rng = np.random.default_rng(3)
    n, nq, d, k = 37000, 10000, 768, int(os.environ.get("ICD_PROBE_K", "10"))
    corpus = rng.standard_normal((n, d), dtype=np.float32)
    corpus /= np.linalg.norm(corpus, axis=1, keepdims=True)
    levels = rng.integers(1, 4, n).astype(np.int32)
    q = rng.standard_normal((nq, d), dtype=np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    idx = IcdIndex(corpus, levels, max_nq=nq, max_k=k)
    dq = torch.from_numpy(q).cuda()

    def run(queries, reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            idx.search_reweighted(queries, k)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    def state(adaptive):
        idx.set_second_pass(True, adaptive=adaptive)
        for _ in range(6):                                 # (evaluated clean searches: the second pass's launches are dropped in both states)
            idx.search_reweighted(dq, k)
            idx.stats()
        for _ in range(120):
            idx.search_reweighted(dq, k)
        idx.stats()
        idx.search_reweighted(dq, k)
        return idx.stats()

    for _ in range(150):
        idx.search_reweighted(dq, k)
    torch.cuda.synchronize()
    print("clean batches, ms per search (50 per window), interleaved:")
    for rep in range(4):
        st = state(False)
        t_all = run(dq, 50)
        st2 = state(True)
        t_two = run(dq, 50)
        print(f"  all four launches (armed={st['sparse_fallback_armed']}, second pass armed={st['second_pass_armed']}): {t_all:.4f}   "
              f"streaming pair left out (armed={st2['sparse_fallback_armed']}, second pass armed={st2['second_pass_armed']}): {t_two:.4f}   saved {1e3 * (t_all - t_two):.1f} us")
    print("a batch with unanswerable (all-zero) queries, ms for that one search:")
    for bad in (1, 8, 32, 40, 41, 64, 128, 256, 512):
        dirty = dq.clone()
        dirty[torch.arange(bad, device="cuda") * 17] = 0
        row = []
        for adaptive in (False, True):
            ts = []
            for _ in range(3):
                st = state(adaptive)
                ts.append(run(dirty, 1))
                fb = idx.stats()["last_fallback"]
            row.append((st["sparse_fallback_armed"], min(ts), fb))
        print(f"  {bad:4d} flagged: with the streaming pair (armed={row[0][0]}) {row[0][1]:.3f} ms (fallback {row[0][2]}), "
              f"without (armed={row[1][0]}) {row[1][1]:.3f} ms (fallback {row[1][2]})")
    idx.close()


if __name__ == "__main__":
    main()
