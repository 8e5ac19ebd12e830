"""Probe: 10 000-query batches on a corpus of tight families (scripts/probe/family_corpus_probe.py), by the number of
candidate lists per query asked for with icd_index_set_chunks (0 = the automatic partition: 5-8 lists at this size)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests"); sys.path.insert(0, "scripts/probe")
from conftest import icd_levels, unit_rows
from family_corpus_probe import family
from rag_project_icd10_amd._native import IcdIndex, MODE_AUTO

def run(name, corpus, queries):
    n = corpus.shape[0]
    idx = IcdIndex(corpus, icd_levels(n, 8), max_nq=10000, max_k=64)
    for chunks in (0, 12, 20, 30):
        idx.set_chunks(chunks)
        for nq, k in ((10000, 10), (10000, 20)):
            dq = torch.from_numpy(queries[:nq]).cuda()
            for _ in range(2): idx.search_reweighted(dq, k, MODE_AUTO)
            torch.cuda.synchronize()
            idx.set_profiling(True); idx.profile_summary()
            t0 = time.perf_counter(); it = 5
            for _ in range(it): idx.search_reweighted(dq, k, MODE_AUTO)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / it
            p = idx.profile_summary(); idx.set_profiling(False); st = idx.stats()
            print("%s chunks %2d nq %5d k %3d: %.3f ms | prep %.3f coarse %.3f finalize %.3f exact %.3f exact_fin %.3f | lists %d fallback %d" % (
                name, chunks, nq, k, dt * 1e3, p["ms_prep"], p["ms_coarse"], p["ms_finalize"], p["ms_exact"], p["ms_exact_finalize"],
                st["last_chunks"], st["last_fallback"]), flush=True)
    idx.close()

if __name__ == "__main__":
    c, q = family(300, 124, 768, 0.1, 10000, 7)
    run("families(cos 0.99)", c, q)
    run("gaussian", unit_rows(37200, 768, 1), unit_rows(10000, 768, 2))
