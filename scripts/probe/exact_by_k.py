#!/usr/bin/env python3
"""exact_by_k.py - `--mode exact` (the fp32-MFMA kernel, the parity anchor and the re-search of uncertified queries) by k, over the
reference's k range (/query searches top_k * 2 with top_k <= 50: models/icd_models.py:138, services/multi_diagnosis_service.py:153),
at the bench size (10 000 x 37 000 x 768) and for a short batch (512): certified NARROW lists (lists of 32 over row-strided chunks,
the per-index option exact_narrow, default on) against lists of KP >= k. Every line is checked against the oracle on a query sample.
Format of profiles/r04_exact_mode_by_k.log; 'frac' = of the 157.3 TFLOP/s fp32 MFMA peak."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc  # noqa: E402  (the checker)
from rag_project_icd10_amd import _native  # noqa: E402
from rag_project_icd10_amd._native import MODE_EXACT, IcdIndex  # noqa: E402


def main():
    lib = _native.load_library()
    n, dim = 37000, 768
    rng = np.random.default_rng(1234)
    corpus = rng.standard_normal((n, dim), dtype=np.float32)
    corpus /= np.linalg.norm(corpus, axis=1, keepdims=True)
    queries = np.random.default_rng(4321).standard_normal((10000, dim), dtype=np.float32)
    queries /= np.linalg.norm(queries, axis=1, keepdims=True)
    levels = np.ones(n, np.int32)
    index = IcdIndex(corpus, levels, max_nq=10000, max_k=100)
    dq = torch.from_numpy(queries).cuda()
    sample = np.arange(0, 10000, 40)
    for narrow in (1, 0):
        index.set_option("exact_narrow", narrow)
        for k in (10, 20, 33, 50, 64, 100):
            if narrow == 0 and k <= 32:
                continue
            for nq in (10000, 512):
                q = dq[:nq]
                for _ in range(2):
                    index.search(q, k, MODE_EXACT)
                torch.cuda.synchronize()
                index.set_profiling(True)
                index.profile_summary()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    s, i = index.search(q, k, MODE_EXACT)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 5
                prof = index.profile_summary()
                index.set_profiling(False)
                st = index.stats()
                sel = sample[sample < nq]
                os_, oi = orc.flat_ip_topk(corpus, queries[sel], k)
                ok_i = bool(np.array_equal(i.cpu().numpy()[sel], oi))
                ok_s = bool(s.cpu().numpy()[sel].tobytes() == os_.tobytes())
                frac = 2.0 * nq * n * dim / (ms * 1e-3) / 1e12 / 157.3
                print(f"{'narrow' if narrow else 'kp>=k '} k {k} nq {nq} ms {ms:.3f} chunks {st['last_chunks']} exact {prof['ms_exact']:.5f} fin {prof['ms_exact_finalize']:.5f} "
                      f"re-searched {st['last_fallback']} frac {frac:.3f} ok {ok_i} {ok_s}", flush=True)


if __name__ == "__main__":
    main()
