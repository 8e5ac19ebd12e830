"""SURVEY.md row N3 on the GPU: icd_score_stats against numpy (bit for bit) and against the reference-run fixture,
icd_cosine_rows against the oracle and the fixture (1e-14: the summation order of a dot product is not pinned)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def test_score_stats_equal_numpy_bitwise(torch_cuda):
    torch = torch_cuda
    from rag_project_icd10_amd._native import score_stats
    rng = np.random.default_rng(11)
    for k in (1, 2, 5, 7, 8, 9, 10, 16, 20, 33, 64, 100, 128):
        nq = 257
        s = rng.uniform(0.0, 1.4, (nq, k))
        s[3] = 0.625                                            # all equal
        s[4] = np.sort(s[4])[::-1]
        valid = rng.integers(0, k + 1, nq)                      # hits that exist per query (a prefix)
        valid[:8] = k
        order = np.where(np.arange(k)[None, :] < valid[:, None], 0, -1).astype(np.int32)
        for use in sorted({1, min(5, k), k}):
            out = score_stats(torch.from_numpy(s).cuda(), torch.from_numpy(order).cuda(), use).cpu().numpy()
            for q in range(nq):
                n = min(use, int(valid[q]))
                a = [float(x) for x in s[q, :n]]
                if n == 0:
                    want = [0.0, 0.0, 0.0, 0.0, 0.0, 0.1]
                else:
                    sd = float(np.std(a))
                    mu = min((1.0 - min(sd, 0.5) / 0.5) * 0.6 + max(a) * 0.4, 1.0)
                    want = [float(np.mean(a)), sd, float(np.var(a)), max(a), mu, float(np.var(a)) if n > 1 else 0.1]
                assert out[q].tolist() == want, (k, use, q, n)
    # no order tensor: every entry exists
    s = rng.uniform(0.0, 1.0, (10, 10))
    out = score_stats(torch.from_numpy(s).cuda()).cpu().numpy()
    assert out[:, 0].tolist() == [float(np.mean(r.tolist())) for r in s]


def test_score_stats_equal_reference_fixture(torch_cuda):
    torch = torch_cuda
    from rag_project_icd10_amd._native import score_stats
    cases = json.load(open(os.path.join(GOLD, "confidence_cases.json"), encoding="utf-8"))["score_cases"]
    k = 128
    s = np.zeros((len(cases), k))
    order = np.full((len(cases), k), -1, np.int32)
    for i, c in enumerate(cases):
        n = len(c["scores"])
        s[i, :n] = c["scores"]
        order[i, :n] = np.arange(n)
    out = score_stats(torch.from_numpy(s).cuda(), torch.from_numpy(order).cuda()).cpu().numpy()
    for i, c in enumerate(cases):
        assert out[i, 4] == c["model_uncertainty"], i
        assert out[i, 5] == c["prediction_variance"], i


def test_cosine_rows(torch_cuda, confidence_oracle):
    torch = torch_cuda
    co = confidence_oracle
    from rag_project_icd10_amd._native import cosine_rows
    z = np.load(os.path.join(GOLD, "confidence_vectors.npz"))
    table = {str(t): v for t, v in zip(z["texts"], z["vectors"])}
    cases = json.load(open(os.path.join(GOLD, "confidence_cases.json"), encoding="utf-8"))["coherence"]
    x = torch.from_numpy(np.stack([table[c["query"]] for c in cases])).cuda()
    y = torch.from_numpy(np.stack([table[c["title"]] for c in cases])).cuda()
    live = cosine_rows(x, torch.from_numpy(table[""]).cuda()).cpu().numpy()
    off = cosine_rows(x, y).cpu().numpy()
    for i, c in enumerate(cases):
        assert abs(live[i] - c["live_shape"]) <= 1e-14 and abs(off[i] - c["offline_shape"]) <= 1e-14
    # unnormalised rows, a zero row, a dimension that is not a multiple of 64
    rng = np.random.default_rng(3)
    a = (rng.standard_normal((37, 200)) * 3.0).astype(np.float32)
    b = (rng.standard_normal((37, 200)) * 0.01).astype(np.float32)
    a[5] = 0.0
    got = cosine_rows(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
    for i in range(37):
        assert abs(got[i] - co.semantic_coherence([float(v) for v in a[i]], [float(v) for v in b[i]])) <= 1e-14
    assert got[5] == 0.0


def test_batch_service_matches_per_call(torch_cuda):
    """MultiDimensionalConfidenceService: the batch entry points against its own per-call methods."""
    torch = torch_cuda
    from rag_project_icd10_amd.services.multidimensional_confidence_service import MultiDimensionalConfidenceService
    z = np.load(os.path.join(GOLD, "confidence_vectors.npz"))
    table = {str(t): v for t, v in zip(z["texts"], z["vectors"])}

    class Emb:
        def encode_query(self, text):
            return table[text].tolist()

        def encode_query_batch(self, texts, batch_size=256, to_device=False):
            t = torch.from_numpy(np.stack([table[x] for x in texts]))
            return t.cuda() if to_device else t.numpy()

    svc = MultiDimensionalConfidenceService(embedding_service=Emb())
    queries = [t for t in table if t.startswith("诊断")]
    titles = [t.replace("诊断", "疾病名称") for t in queries]
    qv = Emb().encode_query_batch(queries, to_device=True)
    live = svc.semantic_coherence_batch(qv).cpu().numpy()
    off = svc.semantic_coherence_batch(qv, titles).cpu().numpy()
    for i, q in enumerate(queries):
        assert abs(live[i] - svc.semantic_coherence(q, [{"title": titles[i]}])) <= 1e-14
        assert abs(off[i] - svc.semantic_coherence(q, [{"preferred_zh": titles[i]}])) <= 1e-14
    rng = np.random.default_rng(8)
    s = rng.uniform(0.2, 1.3, (50, 10))
    st = svc.score_statistics_batch(torch.from_numpy(s).cuda(), top_k=5).cpu().numpy()
    for i in range(50):
        recs = [{"score": float(v)} for v in s[i, :5]]
        assert st[i, 4] == svc._assess_model_uncertainty(recs) and st[i, 5] == svc._calculate_prediction_variance(None, recs)
