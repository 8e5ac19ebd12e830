#!/usr/bin/env python3
"""BASELINE configs[2] (SURVEY.md section 8d "Config 3"): 1 MI355X end to end.

    1 000 diagnosis strings -> tokenise -> BERT-base forward on ROCm (batch 256, length-bucketed)
    -> exact top-10 over a 40 474 x 768 corpus -> ICD level reweight (fused in the finalize kernel)
    -> hierarchical rescoring on the host (HierarchicalSimilarityService.batch_calculate_similarities).

The strings are tests/golden/diagnosis_strings.txt (sampled from the real CSV with default_rng(2025), half of
them perturbed - generator: tests/golden/make_golden.py). No model weights are available offline, so the
encoder is the seeded random-init BERT-base of text2vec-base-chinese's shape with the character tokenizer
("synthetic encoder"): the FLOPs and the memory traffic are those of the real model, the vectors are not.
The corpus is 40 474 random unit rows with the real level histogram, inserted through
MilvusService.insert_records like the reference's build does.

Also timed: the reference's call shape for the same work (one encode_query + one search per string).
Prints one JSON object; `python scripts/bench_e2e.py > profiles/rNN_e2e_config3.json`.
"""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    os.environ.setdefault("EMBEDDING_MODEL_NAME", "shibing624/text2vec-base-chinese")
    os.environ.setdefault("ICD_EMBEDDING_ALLOW_SYNTHETIC", "1")
    tmp = tempfile.mkdtemp(prefix="icd_e2e_")
    os.environ["MILVUS_DB_PATH"] = os.path.join(tmp, "db")
    os.environ["MILVUS_COLLECTION_NAME"] = "icd10_e2e"
    from rag_project_icd10_amd.services.embedding_service import EmbeddingService
    from rag_project_icd10_amd.services.hierarchical_similarity_service import HierarchicalSimilarityService
    from rag_project_icd10_amd.services.milvus_service import MilvusService

    strings = [l.strip() for l in open(os.path.join(ROOT, "tests", "golden", "diagnosis_strings.txt"), encoding="utf-8") if l.strip()]
    n, dim, k = 40474, 768, 10
    es = EmbeddingService()
    ms = MilvusService(embedding_service=es)
    rng = np.random.default_rng(1234)
    corpus = rng.standard_normal((n, dim), dtype=np.float32)
    corpus /= np.linalg.norm(corpus, axis=1, keepdims=True)
    r = np.random.default_rng(1235).random(n)
    levels = np.where(r < 0.1243, 1, np.where(r < 0.4234, 2, 3))
    t0 = time.perf_counter()
    for s in range(0, n, 128):   # the reference's insert granularity (tools/build_database.py:183-192)
        recs = [{"code": f"S{i:05d}.{int(levels[i])}", "preferred_zh": f"合成疾病{i}", "level": int(levels[i]),
                 "parent_code": "", "category_path": f"S{i:05d}", "semantic_text": f"合成疾病{i} | ICD-10: S{i:05d}"}
                for i in range(s, min(n, s + 128))]
        assert ms.insert_records(recs, list(corpus[s:s + 128]))
    assert ms.load_collection()
    t_build = time.perf_counter() - t0
    hs = HierarchicalSimilarityService(embedding_service=es)
    from rag_project_icd10_amd.services.multi_diagnosis_service import MultiDiagnosisService
    md = MultiDiagnosisService(es, ms)
    os.environ.setdefault("ICD_NER_ALLOW_SYNTHETIC", "1")
    from rag_project_icd10_amd.services.medical_ner_service import MedicalNERService
    ner = MedicalNERService()
    if not ner.use_model:
        ner = None

    def sync():
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def batched():
        st = {}
        t = time.perf_counter()
        prepared = [f"query: {q}" for q in strings]
        ids = es._tokenize(prepared)
        st["tokenise_ms"] = (time.perf_counter() - t) * 1e3
        t = time.perf_counter()
        qv = es.encode_query_batch(strings, batch_size=256, to_device=True)
        sync()
        st["tokenise_plus_encode_ms"] = (time.perf_counter() - t) * 1e3
        t = time.perf_counter()
        adj, raw, idx, lv = ms.search_batch(qv, top_k=k)
        sync()
        st["search_reweight_ms"] = (time.perf_counter() - t) * 1e3
        t = time.perf_counter()
        hits = ms.search_batch(qv, top_k=k, as_dicts=True)
        st["search_plus_dict_marshalling_ms"] = (time.perf_counter() - t) * 1e3
        t = time.perf_counter()
        for q, h in zip(strings, hits):
            hs.batch_calculate_similarities(q, {}, h)
        st["hierarchical_rescoring_host_ms"] = (time.perf_counter() - t) * 1e3
        # row N2: the same rescoring with its arithmetic on the device (per-query string rules on the host, one launch)
        t = time.perf_counter()
        a2, r2, i2, l2 = ms.search_batch(qv, top_k=2 * k)
        outs = hs.rescore_live_hits_batch(strings, a2, i2, ms.row_tags())
        sync()
        st["search_2k_plus_hierarchical_rescoring_device_ms"] = (time.perf_counter() - t) * 1e3
        t = time.perf_counter()
        qp = [hs.query_params(q) for q in strings]
        st["of_which_query_string_rules_host_ms"] = (time.perf_counter() - t) * 1e3
        def best_of(fn, reps=3):   # (10 000 Python objects per call: the collector's pauses are not the pipeline's)
            ts = []
            for _ in range(reps):
                t = time.perf_counter()
                fn()
                sync()
                ts.append((time.perf_counter() - t) * 1e3)
            return min(ts)
        st["match_diagnoses_batch_total_ms"] = best_of(lambda: md.match_diagnoses_batch(strings, top_k=k, vectors=qv))
        # row N3: the confidence service's cosine + score statistics for the batch (two launches), alone and inside the match
        t = time.perf_counter()
        stats = md.confidence_service.score_statistics_batch(outs[1], outs[0], top_k=k)
        coh = md.confidence_service.semantic_coherence_batch(qv)
        sync()
        st["confidence_statistics_device_ms"] = (time.perf_counter() - t) * 1e3
        st["match_diagnoses_batch_with_confidence_statistics_ms"] = best_of(
            lambda: md.match_diagnoses_batch(strings, top_k=k, vectors=qv, confidence_statistics=True))
        # row N4: the NER token classifier (BERT-base shape) over the same strings: one padded batch vs one string per forward
        if ner is not None:
            st["ner_batch_1000_strings_ms"] = best_of(lambda: ner.extract_medical_entities_batch(strings), reps=2)
            t = time.perf_counter()
            for q in strings[:50]:
                ner.extract_medical_entities(q)
            sync()
            st["ner_one_string_per_call_ms_per_string"] = (time.perf_counter() - t) * 1e3 / 50
        # tokenise + encode + everything above, in one call
        st["strings_to_matches_one_call_ms"] = best_of(lambda: md.match_diagnoses_batch(strings, top_k=k, confidence_statistics=True))
        del ids, qp, outs, stats, coh
        return st, hits

    import gc
    batched()   # warm-up (kernel load, allocator)
    gc.collect()
    gc.freeze()   # (the 40 474 record dicts and everything made so far leave the collector's working set)
    t0 = time.perf_counter()
    stages, hits = batched()
    total = time.perf_counter() - t0
    # reference call shape: one encode + one search per string
    m = 100
    t0 = time.perf_counter()
    ref_hits = [ms.search(es.encode_query(q), top_k=k) for q in strings[:m]]
    t_ref = time.perf_counter() - t0
    same = sum([h["code"] for h in a] == [h["code"] for h in b] for a, b in zip(hits[:m], ref_hits))
    out = {
        "config": "BASELINE configs[2]: 1000 diagnosis strings -> encode -> search(top_k=10) -> level reweight -> hierarchical rescoring",
        "encoder": es.get_model_info(), "encoder_dtype": os.getenv("ICD_EMBEDDING_DTYPE", "fp32"),
        "corpus_rows": n, "dim": dim, "top_k": k, "strings": len(strings),
        "stages_ms": {kk: round(v, 3) for kk, v in stages.items()},
        "all_stages_above_ms": round(total * 1e3, 3),
        "pipeline_strings_per_s": round(len(strings) / (stages["strings_to_matches_one_call_ms"] / 1e3), 1),
        "reference_call_shape": {"strings": m, "total_ms": round(t_ref * 1e3, 3), "strings_per_s": round(m / t_ref, 1),
                                 "same_codes_as_batched": f"{same}/{m}"},
        "index_build_from_vectors_s": round(t_build, 2),
        "device": torch.cuda.get_device_name(0) if torch.cuda.is_available() else "cpu",
    }
    print(json.dumps(out, ensure_ascii=False, indent=1))


if __name__ == "__main__":
    main()
