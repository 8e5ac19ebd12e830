"""Probe: search time and fallback counts on a corpus of FAMILIES of near-identical rows in code order (the shape the ICD
corpus has: semantic_text repeats the ancestors' names), for the serving path's k = 2 top_k, at several family
tightnesses. Gaussian unit rows never fail the certificate; this is where the fallback paths decide the latency."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import icd_levels
from rag_project_icd10_amd._native import IcdIndex, MODE_AUTO

def family(nfam, per, dim, spread, nq, seed):
    rng = np.random.default_rng(seed)
    cent = rng.standard_normal((nfam, dim)).astype(np.float32)
    x = np.repeat(cent, per, axis=0) + spread * rng.standard_normal((nfam * per, dim)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    q = cent[rng.integers(0, nfam, nq)] + spread * rng.standard_normal((nq, dim)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return np.ascontiguousarray(x, dtype=np.float32), np.ascontiguousarray(q, dtype=np.float32)

def main(spreads=(0.35, 0.1, 0.03, 0.01), shapes=((1000, 10), (1000, 20), (1000, 40), (100, 20), (1, 20)), max_nq=1000, fresh=False, second_pass=True, adaptive=True):
    for spread in spreads:
        corpus, queries = family(300, 124, 768, spread, max_nq, 7)
        n = corpus.shape[0]
        idx = IcdIndex(corpus, icd_levels(n, 8), max_nq=max_nq, max_k=128)
        idx.set_second_pass(second_pass, adaptive)
        for nq, k in shapes:
            if fresh:   # a fresh index per shape: the FIRST large batch (narrow plan + second pass), never wide mode
                idx.close()
                idx = IcdIndex(corpus, icd_levels(n, 8), max_nq=max_nq, max_k=128)
                idx.set_second_pass(second_pass, adaptive)
                idx.set_chunks(0)
            dq = torch.from_numpy(queries[:nq]).cuda()
            for _ in range(3): idx.search_reweighted(dq, k, MODE_AUTO)
            torch.cuda.synchronize()
            idx.set_profiling(True); idx.profile_summary()
            t0 = time.perf_counter(); it = 10
            for _ in range(it): idx.search_reweighted(dq, k, MODE_AUTO)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / it
            p = idx.profile_summary(); idx.set_profiling(False); st = idx.stats()
            print("spread %.2f n %d nq %4d k %3d: %.3f ms | prep %.3f coarse %.3f finalize(+second pass) %.3f exact %.3f exact_fin %.3f | lists %d second pass %d (x %d lists) fallback %d wide_mode %d" % (
                spread, n, nq, k, dt * 1e3, p["ms_prep"], p["ms_coarse"], p["ms_finalize"], p["ms_exact"], p["ms_exact_finalize"],
                st["last_chunks"], st["last_second_pass"], st["last_second_pass_lists"], st["last_fallback"], st["wide_mode"]), flush=True)
        idx.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "large":
        print("## second pass OFF (round 2's behaviour: uncertified queries take the exact re-search)")
        main(spreads=(0.1,), shapes=((10000, 10), (10000, 20)), max_nq=10000, second_pass=False)
        print("## second pass ON, adaptive list count OFF: every batch = narrow plan + second pass (what the FIRST large batch of an index costs)")
        main(spreads=(0.35, 0.1), shapes=((4000, 10), (10000, 10), (10000, 20)), max_nq=10000, adaptive=False)
        print("## second pass ON, adaptive (default): repeated batches on one index switch to the wide partition")
        main(spreads=(0.35, 0.1), shapes=((4000, 10), (4000, 20), (10000, 10), (10000, 20)), max_nq=10000)
    else:
        main()
