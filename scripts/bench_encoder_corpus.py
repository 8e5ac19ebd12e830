#!/usr/bin/env python3
"""Search measured on ENCODER-MADE embeddings: the corpus is built from ICD-shaped strings by the same
encoder that encodes the queries, as the reference does (tools/build_database.py:217-222: encode_query(semantic_text) per
record) - not Gaussian rows.

    synthetic CSV of the real one's shape (tests/golden/csv_shape.json; scripts/bench_build.py synth_csv)
      -> DatabaseBuilder.build_full_database (batched BERT-base forward on ROCm, 40 474 rows) -> HBM index
    queries: (a) the 1 000 golden diagnosis strings (tests/golden/diagnosis_strings.txt), encode_query_batch;
             (b) 10 000 evenly spaced rows' own semantic_text (every query has an exact twin and a family in the corpus)
    for (a) and (b), k = 10 and 20: first batch of a fresh state, steady state, the library's counters (certified by the
    first coarse pass / taken by the second pass / left to the exact re-search, wide mode), kernel split, and equality of
    ids and scores with the library's own fp32-MFMA exact mode (the oracle comparison is tests/test_encoder_gpu.py's).
    BASELINE configs[2] (strings -> matches in one call) on THAT corpus.

No model weights are available offline: the encoder is the seeded random-init BERT-base of text2vec-base-chinese's shape
(its embeddings are far more anisotropic than a trained model's: mean pairwise cosine reported). Prints one JSON object:
`python scripts/bench_encoder_corpus.py > profiles/rNN_e2e_encoder_corpus.json`.
"""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def main():
    import torch
    from bench_build import synth_csv
    os.environ.setdefault("EMBEDDING_MODEL_NAME", "shibing624/text2vec-base-chinese")
    os.environ.setdefault("ICD_EMBEDDING_ALLOW_SYNTHETIC", "1")
    tmp = tempfile.mkdtemp(prefix="icd_enc_")
    os.environ["MILVUS_DB_PATH"] = os.path.join(tmp, "db")
    os.environ["MILVUS_COLLECTION_NAME"] = "icd10_enc"
    shape = json.load(open(os.path.join(ROOT, "tests", "golden", "csv_shape.json"), encoding="utf-8"))
    csv_path = os.path.join(tmp, "icd_synthetic_shape.csv")
    synth_csv(csv_path, shape)
    from rag_project_icd10_amd._native import MODE_AUTO, MODE_EXACT
    from rag_project_icd10_amd.tools.build_database import DatabaseBuilder
    sync = torch.cuda.synchronize
    b = DatabaseBuilder()
    b.initialize_services()
    b.initialize_services = lambda: None
    t0 = time.perf_counter()
    assert b.build_full_database(csv_path, rebuild=True)
    sync()
    t_build = time.perf_counter() - t0
    ms, es = b.milvus_service, b.embedding_service
    recs = ms.client.records
    corpus = ms.client.matrix()
    n = corpus.shape[0]
    rng = np.random.default_rng(5)
    sa, sb = rng.integers(0, n, 4000), rng.integers(0, n, 4000)
    cos = np.einsum("ij,ij->i", corpus[sa], corpus[sb])
    nb = np.einsum("ij,ij->i", corpus[:-1:37], corpus[1::37])
    strings = [l.strip() for l in open(os.path.join(ROOT, "tests", "golden", "diagnosis_strings.txt"), encoding="utf-8") if l.strip()]
    rows_b = np.linspace(0, n - 1, 10000).astype(np.int64)
    qa = es.encode_query_batch(strings, batch_size=256, to_device=True)
    qb = es.encode_query_batch([recs[i]["semantic_text"] for i in rows_b], batch_size=2048, to_device=True)
    sync()
    out = {"what": "search on encoder-made embeddings: corpus and queries from the same (synthetic-weight) BERT-base encoder",
           "corpus_rows": int(n), "build_s": round(t_build, 3), "encoder": es.get_model_info(),
           "anisotropy": {"mean_cosine_random_pairs": float(cos.mean()), "p99_cosine_random_pairs": float(np.quantile(cos, 0.99)),
                          "mean_cosine_code_order_neighbours": float(nb.mean())},
           "runs": []}
    for name, dq in (("a: 1000 golden diagnosis strings", qa), ("b: 10000 corpus rows' own semantic_text", qb)):
        for k in (10, 20):
            assert ms.release_collection()["success"] and ms.load_collection()   # a fresh index: nothing learnt from earlier batches
            index = ms._ready_index()
            sync()
            t0 = time.perf_counter()
            adj, raw, ids, lv = index.search_reweighted(dq, k, MODE_AUTO)
            sync()
            first_ms = (time.perf_counter() - t0) * 1e3
            st0 = index.stats()
            for _ in range(5):
                index.search_reweighted(dq, k, MODE_AUTO)
            sync()
            index.set_profiling(True)
            index.profile_summary()
            t0 = time.perf_counter()
            it = 10
            for _ in range(it):
                adj, raw, ids, lv = index.search_reweighted(dq, k, MODE_AUTO)
            sync()
            steady_ms = (time.perf_counter() - t0) / it * 1e3
            prof = index.profile_summary()
            index.set_profiling(False)
            st = index.stats()
            xa, xr, xi, xl = index.search_reweighted(dq, k, MODE_EXACT)
            sync()
            nq = int(dq.shape[0])
            out["runs"].append({
                "queries": name, "nq": nq, "top_k": k,
                "first_batch": {"ms": round(first_ms, 3), "flagged_by_first_pass": int(st0["last_second_pass"]) if st0["last_second_pass_lists"] else None,
                                "second_pass_queries": int(st0["last_second_pass"]), "second_pass_lists": int(st0["last_second_pass_lists"]),
                                "exact_research_queries": int(st0["last_fallback"]), "wide_mode": int(st0["wide_mode"]), "lists_per_query": int(st0["last_chunks"])},
                "steady_state": {"ms_per_step": round(steady_ms, 4), "second_pass_queries": int(st["last_second_pass"]),
                                 "exact_research_queries": int(st["last_fallback"]), "certified_without_exact_research": nq - int(st["last_fallback"]),
                                 "wide_mode": int(st["wide_mode"]), "lists_per_query": int(st["last_chunks"]), "mode": int(st["last_mode"]),
                                 "kernel_ms": {kk: round(v, 5) for kk, v in prof.items() if kk != "count"}},
                "equals_exact_mode": {"ids": bool(torch.equal(ids, xi)), "adjusted_scores": bool(torch.equal(adj, xa)), "raw_scores": bool(torch.equal(raw, xr))},
            })
    # BASELINE configs[2] on this corpus: strings -> matches in one call
    from rag_project_icd10_amd.services.multi_diagnosis_service import MultiDiagnosisService
    md = MultiDiagnosisService(es, ms)
    md.match_diagnoses_batch(strings, top_k=10, confidence_statistics=True)
    ts = []
    for _ in range(3):
        sync()
        t0 = time.perf_counter()
        md.match_diagnoses_batch(strings, top_k=10, confidence_statistics=True)
        sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    out["config3_strings_to_matches_one_call_ms"] = round(min(ts), 3)
    out["config3_strings_per_s"] = round(len(strings) / (min(ts) / 1e3), 1)
    out["device"] = torch.cuda.get_device_name(0)
    print(json.dumps(out, ensure_ascii=False, indent=1))


if __name__ == "__main__":
    main()
