"""SURVEY.md row N3 on the CPU: the oracle's restatement of numpy's summation order against numpy itself, and the
oracle + the host-side mirror (rag_project_icd10_amd/services/multidimensional_confidence_service.py) against the
fixture produced by running the reference's services/multidimensional_confidence_service.py
(tests/golden/make_confidence_golden.py -> confidence_cases.json, confidence_vectors.npz)."""
import json
import os

import numpy as np
import pytest

from rag_project_icd10_amd.services.multidimensional_confidence_service import MultiDimensionalConfidenceService

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def cases():
    return json.load(open(os.path.join(GOLD, "confidence_cases.json"), encoding="utf-8"))


@pytest.fixture(scope="module")
def table():
    z = np.load(os.path.join(GOLD, "confidence_vectors.npz"))
    return {str(t): v for t, v in zip(z["texts"], z["vectors"])}


class TableEmbedding:
    def __init__(self, table):
        self.table = table

    def encode_query(self, text):
        return self.table[text].tolist()


def test_oracle_sum_order_is_numpys(confidence_oracle):
    co = confidence_oracle
    rng = np.random.default_rng(5)
    for n in list(range(1, 140)) + [255, 256, 257, 1000]:
        for _ in range(3):
            a = [float(x) for x in rng.uniform(-0.2, 1.4, n)]
            assert co.np_mean(a) == float(np.mean(a))
            assert co.np_var(a) == float(np.var(a))
            assert co.np_std(a) == float(np.std(a))


def test_oracle_equals_reference_fixture(confidence_oracle, cases, table):
    co = confidence_oracle
    for c in cases["score_cases"]:
        s = c["scores"]
        assert co.assess_model_uncertainty(s) == c["model_uncertainty"]
        assert co.prediction_variance(s) == c["prediction_variance"]
        assert list(co.confidence_interval(c["confidence"], c["prediction_variance"])) == c["confidence_interval"]
    m = cases["missing_score"]
    s = [r.get("score", 0) for r in m["records"]]
    assert co.assess_model_uncertainty(s) == m["model_uncertainty"] and co.prediction_variance(s) == m["prediction_variance"]
    empty = [float(x) for x in table[""]]
    for c in cases["coherence"]:
        q = [float(x) for x in table[c["query"]]]
        assert abs(co.semantic_coherence(q, empty) - c["live_shape"]) <= 1e-14
        assert abs(co.semantic_coherence(q, [float(x) for x in table[c["title"]]]) - c["offline_shape"]) <= 1e-14


def test_host_mirror_equals_reference_fixture(cases, table):
    svc = MultiDimensionalConfidenceService()
    for c in cases["score_cases"]:
        recs = [{"code": "X", "title": "t", "score": s, "level": 1} for s in c["scores"]]
        assert svc._assess_model_uncertainty(recs) == c["model_uncertainty"]
        assert svc._calculate_prediction_variance(None, recs) == c["prediction_variance"]
        assert list(svc._calculate_confidence_interval(c["confidence"], c["prediction_variance"])) == c["confidence_interval"]
    m = cases["missing_score"]
    assert svc._assess_model_uncertainty(m["records"]) == m["model_uncertainty"]
    assert svc._calculate_prediction_variance(None, m["records"]) == m["prediction_variance"]
    assert svc.semantic_coherence("x", [{"score": 1.0}]) == 0.0            # no embedding service
    svc = MultiDimensionalConfidenceService(embedding_service=TableEmbedding(table))
    assert svc.semantic_coherence(cases["coherence"][0]["query"], []) == cases["coherence_no_candidates"]
    for c in cases["coherence"]:
        live = [{"code": "C1", "title": c["title"], "score": 0.8, "level": 1}]
        off = [{"code": "C1", "preferred_zh": c["title"], "score": 0.8, "level": 1}]
        assert abs(svc.semantic_coherence(c["query"], live) - c["live_shape"]) <= 1e-14
        assert abs(svc.semantic_coherence(c["query"], off) - c["offline_shape"]) <= 1e-14
    assert svc.semantic_coherence("not in the table", [{"preferred_zh": "x"}]) == 0.0   # the reference swallows the failure
