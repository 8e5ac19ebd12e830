"""The CPU oracle itself: known answers, the canonical order, cross-check against float64."""
import numpy as np
import pytest

from conftest import icd_levels, unit_rows


def test_chain_score_is_sequential_fma(oracle):
    rng = np.random.default_rng(0)
    q = rng.standard_normal(768).astype(np.float32)
    c = rng.standard_normal((5, 768)).astype(np.float32)
    got = oracle.scores(q, c)
    for i in range(5):
        acc = np.float32(0.0)
        for d in range(768):  # fma in float64 then one rounding == fmaf for float32 operands
            acc = np.float32(np.float64(q[d]) * np.float64(c[i, d]) + np.float64(acc))
        assert got[i].tobytes() == acc.tobytes()


def test_known_answer_small(oracle):
    corpus = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0.5, 0.5, 0, 0], [1, 0, 0, 0], [-1, 0, 0, 0]], np.float32)
    q = np.array([[1, 0, 0, 0]], np.float32)
    s, i = oracle.flat_ip_topk(corpus, q, 4)
    assert i.tolist() == [[0, 3, 2, 1]]          # tie (rows 0 and 3) broken by lower id
    assert s.tolist() == [[1.0, 1.0, 0.5, 0.0]]
    s, i = oracle.flat_ip_topk(corpus, q, 8)      # n < k: padded with -inf / -1
    assert i[0, 5:].tolist() == [-1, -1, -1] and np.isneginf(s[0, 5:]).all()
    s, i = oracle.flat_ip_topk(corpus, q, 2, id_base=100)
    assert i.tolist() == [[100, 103]]


def test_nan_rows_skipped(oracle):
    corpus = unit_rows(50, 32, 1)
    corpus[7, 3] = np.nan
    q = unit_rows(3, 32, 2)
    s, i = oracle.flat_ip_topk(corpus, q, 50)
    assert (i[:, -1] == -1).all() and not (i == 7).any()


def test_matches_float64_ranking(oracle):
    corpus, q = unit_rows(3000, 768, 3), unit_rows(16, 768, 4)
    s, i = oracle.flat_ip_topk(corpus, q, 10)
    exact = q.astype(np.float64) @ corpus.astype(np.float64).T
    for r in range(16):
        ref = np.argsort(-exact[r], kind="stable")[:10]
        assert set(ref) == set(i[r])  # gaps in this data are >> 1e-7
        assert np.max(np.abs(exact[r, i[r]] - s[r])) < 5e-7
        assert (np.diff(s[r]) <= 0).all()


def test_reference_shaped_search_agrees(oracle):
    corpus, lv, q = unit_rows(2000, 768, 5), icd_levels(2000, 6), unit_rows(8, 768, 7)
    s, i = oracle.flat_ip_topk(corpus, q, 10)
    adj, raw, ids, _ = oracle.reweight(s, i, lv)
    for r in range(8):
        hits = oracle.reference_shaped_search(corpus, lv, q[r], 10)
        assert [h[2] for h in hits] == ids[r].tolist()
        assert np.allclose([h[0] for h in hits], adj[r], atol=1e-6)


def test_reweight_is_stable_and_uses_float64(oracle):
    raw = np.array([[0.9, 0.8, 0.75, 0.6, 0.6]], np.float32)
    ids = np.array([[0, 1, 2, 3, 4]], np.int64)
    levels = np.array([3, 1, 2, 2, 2], np.int32)   # weights .8 1.2 1 1 1
    adj, oraw, oid, olv = oracle.reweight(raw, ids, levels)
    assert oid.tolist() == [[1, 2, 0, 3, 4]]        # 0.96, 0.75, 0.72.., 0.6, 0.6 (tie keeps raw order)
    assert adj[0, 0] == float(np.float32(0.8)) * 1.2 and adj[0, 2] == float(np.float32(0.9)) * 0.8
    assert olv.tolist() == [[1, 2, 3, 2, 2]]
    # padded hits stay at the end
    adj, _, oid, _ = oracle.reweight(np.array([[0.5, -np.inf]], np.float32), np.array([[2, -1]], np.int64), levels)
    assert oid.tolist() == [[2, -1]] and np.isneginf(adj[0, 1])
    assert [oracle.level_weight(l) for l in (1, 2, 3, 0, 7)] == [1.2, 1.0, 0.8, 1.0, 1.0]
    assert oracle.lib().icd_oracle_level_weight(1) == 1.2 and oracle.lib().icd_oracle_level_weight(9) == 1.0


def test_merge_of_shards_equals_full_search(oracle):
    corpus, q = unit_rows(1000, 64, 8), unit_rows(9, 64, 9)
    full_s, full_i = oracle.flat_ip_topk(corpus, q, 7)
    parts = [oracle.flat_ip_topk(corpus[a:b], q, 7, id_base=a) for a, b in ((0, 300), (300, 301), (301, 1000))]
    s, i = oracle.merge(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]), 7)
    assert (i == full_i).all() and s.tobytes() == full_s.tobytes()


@pytest.mark.parametrize("nthreads", [1, 3])
def test_threads_do_not_change_results(oracle, nthreads):
    corpus, q = unit_rows(500, 128, 10), unit_rows(20, 128, 11)
    a = oracle.flat_ip_topk(corpus, q, 5, nthreads=nthreads)
    b = oracle.flat_ip_topk(corpus, q, 5, nthreads=2)
    assert a[0].tobytes() == b[0].tobytes() and (a[1] == b[1]).all()
