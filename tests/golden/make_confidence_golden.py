#!/usr/bin/env python3
"""Golden fixture for SURVEY.md row N3 (confidence-service cosines and score statistics), produced by RUNNING THE
REFERENCE'S OWN services/multidimensional_confidence_service.py (unchanged, imported from /root/reference):

    python tests/golden/make_confidence_golden.py

Functions run (reference file services/multidimensional_confidence_service.py):
    _assess_model_uncertainty          :936-963    np.mean / np.std / max over the candidates' 'score'
    _calculate_prediction_variance     :1087-1099  np.var over the candidates' 'score' (0.1 for fewer than two)
    _calculate_confidence_interval     :1101-1114  confidence -/+ 1.96 sqrt(variance), clamped to [0, 1]
    _calculate_semantic_factors        :257-296    only its 'semantic_coherence' entry:
                                                   sklearn cosine_similarity([encode_query(query)], [encode_query(title)])

Only loguru is replaced (absent here; a no-op logger). The embedding service handed to the reference is a table of
seeded unit vectors (text -> fp32 vector as a list of Python floats, the type EmbeddingService.encode_query returns):
the cosine is what is pinned, not the encoder. Candidate records come in the two shapes the reference produces:
the live one (services/multi_diagnosis_service.py:178-186: code / title / score / level - no 'preferred_zh', so the
reference embeds the EMPTY string as "the candidate") and the offline one (with 'preferred_zh').

Only DATA is written: confidence_cases.json (score lists, their outputs; texts and cosines) and confidence_vectors.npz.
"""
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
DIM = 768


def stub_modules():
    class _Logger:
        def __getattr__(self, name):
            return lambda *a, **k: None
    loguru = types.ModuleType("loguru")
    loguru.logger = _Logger()
    sys.modules["loguru"] = loguru


class TableEmbedding:
    """encode_query(text) -> list of Python floats (a unit fp32 vector), from a seeded table"""

    def __init__(self, texts, seed):
        rng = np.random.default_rng(seed)
        base = rng.standard_normal(DIM).astype(np.float32)
        self.table = {}
        for t in texts:
            v = (0.6 * base + rng.standard_normal(DIM).astype(np.float32)).astype(np.float32)
            v /= np.linalg.norm(v)
            self.table[t] = v.astype(np.float32)

    def encode_query(self, text):
        return self.table[text].tolist()


def main():
    stub_modules()
    sys.path.insert(0, REF)
    from services.multidimensional_confidence_service import MultiDimensionalConfidenceService

    rng = np.random.default_rng(20251004)
    # ---- score statistics --------------------------------------------------------------------------------------
    lengths = [0, 1, 2, 3, 5, 7, 8, 9, 10, 10, 10, 15, 16, 17, 20, 20, 24, 31, 32, 33, 50, 64, 100, 127, 128]
    score_cases = []
    svc = MultiDimensionalConfidenceService()
    for rep in range(8):
        for n in lengths:
            kind = (rep + n) % 4
            if kind == 0:
                s = rng.uniform(0.3, 1.0, n)
            elif kind == 1:
                s = np.sort(rng.uniform(0.0, 1.4, n))[::-1]          # best first, like a result list
            elif kind == 2:
                s = np.full(n, rng.uniform(0.2, 0.9))                 # all equal: std 0
            else:
                s = rng.uniform(0.55, 0.56, n) * rng.choice([1.0, 1.2, 0.8], n)
            scores = [float(x) for x in s]
            recs = [{"code": f"X{i:02d}", "title": "t", "score": sc, "level": 1} for i, sc in enumerate(scores)]
            mu = svc._assess_model_uncertainty(recs)
            pv = svc._calculate_prediction_variance(None, recs)
            conf = float(rng.uniform(0.0, 1.0))
            ci = svc._calculate_confidence_interval(conf, pv)
            score_cases.append({"scores": scores, "model_uncertainty": float(mu), "prediction_variance": float(pv),
                                "confidence": conf, "confidence_interval": [float(ci[0]), float(ci[1])]})
    # a record without 'score' counts as 0 (r.get('score', 0))
    recs = [{"code": "A", "score": 0.9}, {"code": "B"}, {"code": "C", "score": 0.5}]
    missing = {"records": recs, "model_uncertainty": float(svc._assess_model_uncertainty(recs)),
               "prediction_variance": float(svc._calculate_prediction_variance(None, recs))}

    # ---- semantic coherence ------------------------------------------------------------------------------------
    queries = [f"诊断{i}" for i in range(24)]
    titles = [f"疾病名称{i}" for i in range(24)]
    emb = TableEmbedding(queries + titles + [""], seed=77)
    svc = MultiDimensionalConfidenceService(embedding_service=emb)
    coh = []
    for i, q in enumerate(queries):
        live = [{"code": "C1", "title": titles[i], "score": 0.8, "level": 1}]
        offline = [{"code": "C1", "preferred_zh": titles[i], "score": 0.8, "level": 1}]
        c_live = svc._calculate_semantic_factors(q, live)["semantic_coherence"]
        c_off = svc._calculate_semantic_factors(q, offline)["semantic_coherence"]
        coh.append({"query": q, "title": titles[i], "live_shape": float(c_live), "offline_shape": float(c_off)})
    none = svc._calculate_semantic_factors(queries[0], [])["semantic_coherence"]

    out = {"score_cases": score_cases, "missing_score": missing, "coherence": coh, "coherence_no_candidates": float(none),
           "numpy": np.__version__}
    with open(os.path.join(HERE, "confidence_cases.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False)
    texts = queries + titles + [""]
    np.savez_compressed(os.path.join(HERE, "confidence_vectors.npz"),
                        vectors=np.stack([emb.table[t] for t in texts]), texts=np.array(texts))
    print(f"{len(score_cases)} score cases, {len(coh)} cosine cases written")


if __name__ == "__main__":
    main()
