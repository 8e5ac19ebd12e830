"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(ctypes -> libicdsearch.so), against the CPU oracle on the same seeded inputs.

Bar: ids bit-exact; scores bit-identical fp32 (tolerance 0 - stricter than the 1e-5 the north star
allows, possible because kernels and oracle share one canonical summation order); reweighted scores
bit-identical float64. At BASELINE.json's full sizes the oracle only checks a query sample and the
rest is covered by size-independent properties (sortedness, idempotence, mode agreement, monotone
prefix in k, shard-merge equivalence).
"""
import os

import numpy as np
import pytest

import perf_records
from conftest import icd_levels, unit_rows

pytestmark = pytest.mark.gpu

from rag_project_icd10_amd import _native  # noqa: E402
from rag_project_icd10_amd._native import MODE_AUTO, MODE_EXACT, IcdIndex  # noqa: E402


def _bits(a):
    return np.ascontiguousarray(a).tobytes()


def _check(oracle, index, corpus, levels, queries, k, mode, id_base=0):
    s, i = index.search(queries, k, mode)
    os_, oi = oracle.flat_ip_topk(corpus, queries, k, id_base=id_base)
    assert np.array_equal(i, oi), f"id mismatch rows {np.nonzero((i != oi).any(1))[0][:5]}"
    assert _bits(s) == _bits(os_)
    adj, raw, ids, lv = index.search_reweighted(queries, k, mode)
    want = oracle.reweight(os_, oi, levels, id_base=id_base)
    assert np.array_equal(ids, want[2]) and _bits(adj) == _bits(want[0]) and _bits(raw) == _bits(want[1])
    assert np.array_equal(lv, want[3])
    return index.stats()


CASES = [  # n, nq, dim, k, kind
    (100, 3, 768, 5, "gauss"), (7, 2, 768, 10, "gauss"), (1000, 37, 768, 10, "gauss"), (2049, 5, 768, 1, "gauss"),
    (4000, 129, 768, 12, "gauss"), (1500, 20, 1024, 10, "gauss"), (6000, 200, 768, 10, "clustered"),
    (128, 128, 768, 10, "gauss"), (129, 1, 768, 10, "gauss"),
]


@pytest.mark.parametrize("mode", [MODE_AUTO, MODE_EXACT])
@pytest.mark.parametrize("n,nq,dim,k,kind", CASES)
def test_parity_small(oracle, n, nq, dim, k, kind, mode):
    corpus, levels, queries = unit_rows(n, dim, 1 + n, kind), icd_levels(n, 2 + n), unit_rows(nq, dim, 3 + n, kind)
    idx = IcdIndex(corpus, levels, max_nq=max(nq, 1), max_k=max(k, 1))
    st = _check(oracle, idx, corpus, levels, queries, k, mode)
    assert st["n"] == n and st["dim"] == dim
    if mode == MODE_AUTO:
        # batches of <= 16 queries take the exact streaming kernel (no coarse pass); larger ones the fp16 path
        assert st["fast_path"] == 1 and st["last_mode"] == (MODE_EXACT if nq <= 16 else MODE_AUTO)
    idx.close()


@pytest.mark.parametrize("k", [17, 64, 100])
def test_large_k(oracle, k):
    corpus, levels, queries = unit_rows(5000, 768, 40), icd_levels(5000, 41), unit_rows(9, 768, 42)
    idx = IcdIndex(corpus, levels, max_nq=16, max_k=100)
    st = _check(oracle, idx, corpus, levels, queries, k, MODE_AUTO)
    assert st["last_mode"] == MODE_AUTO      # k <= 100 stays on the certified fp16 path
    idx.close()


def test_exact_ties_and_id_base(oracle):
    corpus = unit_rows(3001, 768, 50)
    corpus[5::5] = corpus[np.arange(5, 3001, 5) // 2]       # duplicate rows: exact score ties, broken by id
    levels, queries = icd_levels(3001, 51), unit_rows(70, 768, 52)
    for mode in (MODE_AUTO, MODE_EXACT):
        idx = IcdIndex(corpus, levels, max_nq=70, max_k=10, id_base=1_000_000)
        _check(oracle, idx, corpus, levels, queries, 10, mode, id_base=1_000_000)
        idx.close()


def test_heavy_duplicates_through_the_coarse_path(oracle):
    """40 identical corpus rows that are every query's best hits: all coarse scores tie, the ballot bisection
    of the select cannot split them and must fall back to the unique-key ranking; ids come out row-ascending."""
    corpus = unit_rows(5000, 768, 55)
    corpus[100:140] = corpus[100]
    rng = np.random.default_rng(56)
    queries = corpus[100][None, :] + 0.03 * rng.standard_normal((150, 768)).astype(np.float32)
    queries = (queries / np.linalg.norm(queries, axis=1, keepdims=True)).astype(np.float32)
    levels = icd_levels(5000, 57)
    idx = IcdIndex(corpus, levels, max_nq=150, max_k=10)
    st = _check(oracle, idx, corpus, levels, queries, 10, MODE_AUTO)
    assert st["last_mode"] == MODE_AUTO
    s, i = idx.search(queries, 10, MODE_AUTO)
    assert np.array_equal(i, np.tile(np.arange(100, 110), (150, 1)))
    idx.close()


@pytest.mark.parametrize("dups", [False, True])
def test_lists_long_enough_for_the_synchronised_compaction(oracle, dups):
    """Lists of 32 tiles (two lists per query over 8 192 rows): the coarse kernel's all-queries compaction at tile 24
    runs, with unique scores and with a fifth of the rows duplicated (score ties at the KP-th place of a buffer take its
    tie path: as many of the tied entries as fit, the tied score as the bound). 300 queries: more than one query tile."""
    n = 8192
    corpus = unit_rows(n, 768, 71)
    if dups:
        corpus[5::5] = corpus[np.arange(5, n, 5) // 2]
    levels, queries = icd_levels(n, 72), unit_rows(300, 768, 73)
    idx = IcdIndex(corpus, levels, max_nq=300, max_k=10)
    idx.set_chunks(2)
    st = _check(oracle, idx, corpus, levels, queries, 10, MODE_AUTO)
    assert st["last_mode"] == MODE_AUTO and st["last_chunks"] >= 2
    if not dups:
        assert st["last_fallback"] == 0
    idx.close()


@pytest.mark.parametrize("n,nq,dim,k,mode", [
    (3000, 1, 1024, 10, MODE_AUTO), (3000, 3, 1024, 64, MODE_EXACT), (2500, 2, 96, 7, MODE_AUTO),
    (9000, 40, 768, 10, MODE_EXACT), (9000, 64, 768, 100, MODE_EXACT), (700, 16, 768, 128, MODE_EXACT),
    (257, 7, 2048, 10, MODE_EXACT),
])
def test_streaming_kernel_variants(oracle, n, nq, dim, k, mode):
    """small batches take stream_topk: every (queries-per-pass, list length) instantiation, several passes,
    dims other than 768, k up to the ABI maximum"""
    corpus, levels, queries = unit_rows(n, dim, 90 + n), icd_levels(n, 91 + n), unit_rows(nq, dim, 92 + n)
    idx = IcdIndex(corpus, levels, max_nq=64, max_k=128)
    st = _check(oracle, idx, corpus, levels, queries, k, mode)
    assert st["last_mode"] == MODE_EXACT
    idx.close()


def test_dense_fallback_list_takes_the_mfma_exact_kernel(oracle):
    """more than 256 uncertified queries: the device-side gate routes the fallback to exact_topk, not stream_topk;
    100 of them: stream_topk in 13 passes of 8"""
    corpus, levels = unit_rows(3000, 768, 95), icd_levels(3000, 96)
    queries = unit_rows(900, 768, 97)
    queries[::3] = 0.0            # 300 all-zero queries: every score ties at 0, the certificate cannot separate a top-k
    idx = IcdIndex(corpus, levels, max_nq=900, max_k=10)
    st = _check(oracle, idx, corpus, levels, queries, 10, MODE_AUTO)
    assert st["last_fallback"] == 300
    st = _check(oracle, idx, corpus, levels, queries[:300], 10, MODE_AUTO)
    assert st["last_fallback"] == 100
    idx.close()


def test_every_query_flagged_in_a_large_batch(oracle):
    """all 6000 queries fail certification (the corpus is 10 distinct rows, 200 copies each: every candidate list
    overflows with exact ties): the dense fallback's partial lists must be sized for the whole batch (its list count is
    fitted to the workspace, independently of the sparse path's 32 lists); ties come out row id ascending"""
    corpus, levels = np.repeat(unit_rows(10, 768, 98), 200, axis=0), icd_levels(2000, 99)
    queries = unit_rows(6000, 768, 100)
    idx = IcdIndex(corpus, levels, max_nq=6000, max_k=10)
    idx.set_second_pass(False)                       # (the second coarse pass would certify most of them: next block)
    s, i = idx.search(queries, 10, MODE_AUTO)
    st = idx.stats()
    assert st["last_fallback"] == 6000
    sample = np.arange(0, 6000, 37)
    os_, oi = oracle.flat_ip_topk(corpus, queries[sample], 10)
    assert np.array_equal(i[sample], oi) and _bits(s[sample]) == _bits(os_)
    se, ie = idx.search(queries, 10, MODE_EXACT)
    assert np.array_equal(i, ie) and _bits(s) == _bits(se)
    # with the second pass (the default): ~21 lists of 16 hold all 200 copies of a query's best row, the 256-candidate
    # window certifies most queries from them; what is left takes the dense fallback; same bits either way
    idx.set_second_pass(True)
    s2, i2 = idx.search(queries, 10, MODE_AUTO)
    st2 = idx.stats()
    assert st2["last_second_pass"] == 6000 and st2["last_fallback"] < 6000
    assert np.array_equal(i2, i) and _bits(s2) == _bits(s)
    idx.close()


def _family_corpus(nfam, per, dim, seed):
    """rows in CODE ORDER: families of `per` near-identical rows sit next to each other, like the ICD corpus whose
    semantic_text repeats the ancestors' names (tools/build_database.py:156-171)"""
    rng = np.random.default_rng(seed)
    cent = rng.standard_normal((nfam, dim)).astype(np.float32)
    x = np.repeat(cent, per, axis=0) + 0.35 * rng.standard_normal((nfam * per, dim)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    q = cent[rng.integers(0, nfam, 256)] + 0.35 * rng.standard_normal((256, dim)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return np.ascontiguousarray(x, dtype=np.float32), np.ascontiguousarray(q, dtype=np.float32)


@pytest.mark.parametrize("k", [10, 20])
def test_tight_families_are_certified_by_the_wide_window(oracle, k):
    """300 families of 124 rows with mutual cosine ~0.99: a query has its whole family within 2 eps of its k-th best,
    more candidates than the rescoring window sized for k <= 32 holds (32 or 64), so the first finalize pass flags every
    query. The second pass (same kernel, 256-candidate window, over the flagged list) certifies them from the same coarse
    lists; before it every one of these queries took the exact re-search (2.7 / 5.2 ms per 1 000 queries)."""
    rng = np.random.default_rng(7)
    cent = rng.standard_normal((300, 768)).astype(np.float32)
    corpus = np.repeat(cent, 124, axis=0) + 0.1 * rng.standard_normal((300 * 124, 768)).astype(np.float32)
    corpus = np.ascontiguousarray(corpus / np.linalg.norm(corpus, axis=1, keepdims=True), dtype=np.float32)
    queries = cent[rng.integers(0, 300, 600)] + 0.1 * rng.standard_normal((600, 768)).astype(np.float32)
    queries = np.ascontiguousarray(queries / np.linalg.norm(queries, axis=1, keepdims=True), dtype=np.float32)
    levels = icd_levels(corpus.shape[0], 9)
    idx = IcdIndex(corpus, levels, max_nq=600, max_k=k)
    st = _check(oracle, idx, corpus, levels, queries, k, MODE_AUTO)
    assert st["last_mode"] == MODE_AUTO and st["last_fallback"] <= 6, st["last_fallback"]
    # a handful of queries only: the small-batch shapes take the same route
    for nq in (40, 130):
        _check(oracle, idx, corpus, levels, queries[:nq], k, MODE_AUTO)
    idx.close()


def test_more_lists_certify_tight_families_in_a_large_batch(oracle):
    """4 096 queries have 8 candidate lists of 16: fewer candidates than a 124-row family of near-identical rows has
    members, so most queries fail the first pass's certificate (and, with the second coarse pass switched off, take the
    exact re-search); with ~20 lists per query (icd_index_set_chunks: the partition the second pass and wide mode use)
    none does. Bit-identical results either way."""
    rng = np.random.default_rng(7)
    cent = rng.standard_normal((300, 768)).astype(np.float32)
    corpus = np.repeat(cent, 124, axis=0) + 0.1 * rng.standard_normal((300 * 124, 768)).astype(np.float32)
    corpus = np.ascontiguousarray(corpus / np.linalg.norm(corpus, axis=1, keepdims=True), dtype=np.float32)
    queries = cent[rng.integers(0, 300, 4096)] + 0.1 * rng.standard_normal((4096, 768)).astype(np.float32)
    queries = np.ascontiguousarray(queries / np.linalg.norm(queries, axis=1, keepdims=True), dtype=np.float32)
    levels = icd_levels(corpus.shape[0], 9)
    idx = IcdIndex(corpus, levels, max_nq=4096, max_k=20)
    idx.set_second_pass(False)
    s0, i0 = idx.search(queries, 20, MODE_AUTO)
    assert idx.stats()["last_fallback"] > 0.5 * 4096
    idx.set_chunks(20)
    s1, i1 = idx.search(queries, 20, MODE_AUTO)
    st = idx.stats()
    assert st["last_fallback"] == 0 and st["last_chunks"] >= 12
    assert np.array_equal(i0, i1) and _bits(s0) == _bits(s1)
    sample = np.arange(0, 4096, 16)
    os_, oi = oracle.flat_ip_topk(corpus, queries[sample], 20)
    assert np.array_equal(i1[sample], oi) and _bits(s1[sample]) == _bits(os_)
    idx.close()


def test_corpus_in_code_order_stays_on_the_fast_path(oracle):
    """A query's whole family is contiguous in the corpus. The fp16 copy is stored in a permuted row order so the family
    spreads over the candidate lists; without it one list holds the family, ends on a bound inside it and the
    certificate fails for most queries (results stay exact either way)."""
    corpus, queries = _family_corpus(300, 120, 768, 77)
    levels = icd_levels(len(corpus), 78)
    idx = IcdIndex(corpus, levels, max_nq=256, max_k=10)
    st = _check(oracle, idx, corpus, levels, queries, 10, MODE_AUTO)
    idx.close()
    assert st["last_mode"] == MODE_AUTO and st["last_fallback"] <= 256 // 8
    from rag_project_icd10_amd import _native
    idx = IcdIndex(corpus, levels, max_nq=256, max_k=10, permute=False)   # per-index option: fp16 copy in row order
    st2 = _check(oracle, idx, corpus, levels, queries, 10, MODE_AUTO)
    idx.close()
    assert st2["last_fallback"] >= st["last_fallback"]


def test_one_index_many_call_shapes(oracle):
    """the same handle serves batches of very different sizes, k and modes back to back (workspaces, shared thresholds
    and device-side counters carry nothing over from call to call)"""
    corpus, levels = unit_rows(20000, 768, 110), icd_levels(20000, 111)
    queries = unit_rows(2500, 768, 112)
    bad = queries.copy()
    bad[::7, 5] = 2e5
    idx = IcdIndex(corpus, levels, max_nq=2500, max_k=64)
    plan = [(2500, 10, MODE_AUTO, queries), (1, 10, MODE_AUTO, queries), (300, 5, MODE_AUTO, bad), (2500, 10, MODE_AUTO, bad),
            (17, 12, MODE_AUTO, queries), (700, 10, MODE_EXACT, queries), (129, 64, MODE_AUTO, queries), (2500, 10, MODE_AUTO, queries),
            (64, 1, MODE_EXACT, bad), (1000, 10, MODE_AUTO, queries)]
    for nq, k, mode, src in plan:
        q = src[:nq]
        s, i = idx.search(q, k, mode)
        sample = np.unique(np.concatenate([np.arange(0, nq, max(1, nq // 40)), [nq - 1]]))
        os_, oi = oracle.flat_ip_topk(corpus, q[sample], k)
        assert np.array_equal(i[sample], oi), (nq, k, mode)
        assert _bits(s[sample]) == _bits(os_), (nq, k, mode)
    idx.close()


@pytest.mark.parametrize("k", [13, 20, 32, 48, 64, 100])
def test_larger_k_on_the_fast_path(oracle, k):
    """/query searches top_k * 2 (F7): k up to 100 (everything /query can ask for) stays on the certified fp16 path (about k / 4 lists per query, one to
    four rescoring candidates per lane); larger k takes the exact kernel"""
    corpus, levels, queries = unit_rows(20000, 768, 120), icd_levels(20000, 121), unit_rows(700, 768, 122)
    idx = IcdIndex(corpus, levels, max_nq=700, max_k=128)
    st = _check(oracle, idx, corpus, levels, queries, k, MODE_AUTO)
    assert st["last_mode"] == MODE_AUTO and st["last_fallback"] <= 70
    st = _check(oracle, idx, corpus, levels, queries, 101, MODE_AUTO)   # beyond: 32 lists of 16 cannot certify, exact kernels
    assert st["last_mode"] == MODE_EXACT
    idx.close()


def test_dim_other_than_fast_path(oracle):
    corpus, levels, queries = unit_rows(900, 64, 60), icd_levels(900, 61), unit_rows(11, 64, 62)
    idx = IcdIndex(corpus, levels, max_nq=16, max_k=10)
    assert idx.stats()["fast_path"] == 0
    _check(oracle, idx, corpus, levels, queries, 5, MODE_AUTO)
    idx.close()
    with pytest.raises(_native.IcdError):
        IcdIndex(unit_rows(10, 50, 1))                          # dim % 32 != 0 -> ICD_ERR_UNSUPPORTED


def test_unnormalised_and_nonfinite_inputs(oracle):
    rng = np.random.default_rng(70)
    corpus = (rng.standard_normal((2000, 768)) * rng.uniform(0.1, 30, (2000, 1))).astype(np.float32)
    levels = icd_levels(2000, 71)
    queries = (rng.standard_normal((33, 768)) * 5).astype(np.float32)
    idx = IcdIndex(corpus, levels, max_nq=64, max_k=10)
    _check(oracle, idx, corpus, levels, queries, 10, MODE_AUTO)   # error bound scales with ||q|| * rmax
    queries[3, 10] = 1e6                                          # beyond fp16's range: the query's own power-of-two scale absorbs it
    _check(oracle, idx, corpus, levels, queries, 10, MODE_AUTO)
    idx.close()
    corpus[17, 5] = 7e4                                           # beyond fp16's range: the corpus scale absorbs it
    idx = IcdIndex(corpus, levels, max_nq=64, max_k=10)
    assert idx.stats()["fast_path"] == 1
    _check(oracle, idx, corpus, levels, queries[:5], 10, MODE_AUTO)
    idx.close()
    corpus[18, 6] = np.nan                                        # non-finite corpus: no fp16 copy, exact kernels only
    idx = IcdIndex(corpus, levels, max_nq=64, max_k=10)
    assert idx.stats()["fast_path"] == 0
    _check(oracle, idx, corpus, levels, queries[:5], 10, MODE_AUTO)
    idx.close()


@pytest.mark.parametrize("cscale,qscale", [(1e-5, 1.0), (1e-6, 1.0), (1e-30, 1e-6), (1.0, 1e-6), (1e-3, 1e-5), (1e12, 1e10), (3e-38, 1.0)])
def test_tiny_and_huge_norms_stay_exact(oracle, cscale, qscale):
    """The fp16 images are power-of-two scaled (queries per row, the corpus as a whole), so data far below fp16's normal
    range (2^-14) or above its largest value keeps 11 significant bits and the certificate's relative error bound holds;
    round 1 converted unscaled and certified wrong top-k for a corpus scaled by 1e-5. Near fp32's own underflow
    (3e-38) the canonical chain itself rounds on the subnormal grid: the bound's absolute term takes over, nothing is
    certified, results still come from the exact kernels."""
    corpus, levels = unit_rows(20000, 768, 500) * np.float32(cscale), icd_levels(20000, 501)
    queries = unit_rows(300, 768, 502) * np.float32(qscale)
    idx = IcdIndex(corpus, levels, max_nq=300, max_k=10)
    assert idx.stats()["fast_path"] == 1
    st = _check(oracle, idx, corpus, levels, queries, 10, MODE_AUTO)
    assert st["last_mode"] == MODE_AUTO
    if cscale * qscale > 1e-30:
        assert st["last_fallback"] <= 30                     # certified on the fast path, not rescued by the fallback
    _check(oracle, idx, corpus, levels, queries[:5], 10, MODE_AUTO)
    idx.close()


def test_mixed_magnitudes_within_one_corpus_and_batch(oracle):
    """only SOME rows / queries are tiny: the corpus scale comes from its largest component, so the tiny rows' fp16
    images do sink into the subnormal range - covered by the bound's absolute term relative to the scaled norms"""
    rng = np.random.default_rng(510)
    corpus, levels = unit_rows(20000, 768, 511), icd_levels(20000, 512)
    corpus[::3] *= np.float32(1e-6)
    corpus[1::7] *= np.float32(1e-3)
    corpus[:, :96] *= np.float32(1e-5)                        # subnormal-sized components inside normal rows
    queries = unit_rows(400, 768, 513) * (10.0 ** rng.uniform(-6, 2, (400, 1))).astype(np.float32)
    queries[::5] *= -1.0
    idx = IcdIndex(corpus, levels, max_nq=400, max_k=10)
    st = _check(oracle, idx, corpus, levels, queries, 10, MODE_AUTO)
    assert st["last_mode"] == MODE_AUTO and st["fast_path"] == 1
    _check(oracle, idx, corpus, levels, queries, 5, MODE_EXACT)
    idx.close()


def test_every_query_flagged_on_a_small_corpus_fits_the_workspace():
    """n < 4096 rows, a full batch of all-zero queries (every score ties at 0 -> every query fails the
    certificate): the fallback's fp32-MFMA kernel runs over the whole batch and its list count must be sized against
    the workspace (it used to write nq * 24 * 16 entries into a 2 M-entry buffer)."""
    n, nq, k = 3000, 16384, 10
    corpus, levels = unit_rows(n, 768, 520), icd_levels(n, 521)
    idx = IcdIndex(corpus, levels, max_nq=nq, max_k=k)
    queries = np.zeros((nq, 768), np.float32)
    queries[::1000] = unit_rows(17, 768, 522)
    s, i = idx.search(queries, k, MODE_AUTO)
    st = idx.stats()
    zero = np.ones(nq, bool)
    zero[::1000] = False
    assert (s[zero] == 0).all() and (i[zero] == np.arange(k)).all()          # ties: row id ascending
    assert st["last_fallback"] >= zero.sum()
    s2, i2 = idx.search(queries[::1000], k, MODE_EXACT)
    assert np.array_equal(i[::1000], i2) and _bits(s[::1000]) == _bits(s2)
    idx.close()


def test_real_corpus_size_and_reference_call_shape(oracle):
    """N = 40 474 (the real CSV's row count), nq = 1 per call - how MilvusService.search drives the engine."""
    corpus, levels = unit_rows(40474, 768, 80), icd_levels(40474, 81)
    queries = unit_rows(6, 768, 82)
    idx = IcdIndex(corpus, levels, max_nq=256, max_k=100)
    for r in range(3):
        _check(oracle, idx, corpus, levels, queries[r:r + 1], 5, MODE_AUTO)
    _check(oracle, idx, corpus, levels, queries, 10, MODE_AUTO)
    _check(oracle, idx, corpus, levels, queries[:2], 100, MODE_AUTO)   # /query's top_k*2 upper bound (F7)
    idx.close()


def test_full_size_config2_properties(oracle):
    """BASELINE configs[1]: 10 000 queries x 37 000 x 768, k = 10."""
    n, nq, k = 37000, 10000, 10
    corpus = unit_rows(n, 768, 1234)
    levels = icd_levels(n, 1235)
    queries = unit_rows(nq, 768, 4321)
    idx = IcdIndex(corpus, levels, max_nq=nq, max_k=k)
    s, i = idx.search(queries, k, MODE_AUTO)
    st = idx.stats()
    assert (i >= 0).all() and (i < n).all()
    assert (np.diff(s, axis=1) <= 0).all()                                    # sorted best-first
    assert all(len(set(r)) == k for r in i[::97])                              # no duplicate ids
    s2, i2 = idx.search(queries, k, MODE_AUTO)
    assert np.array_equal(i, i2) and _bits(s) == _bits(s2)                     # idempotent
    os_, oi = oracle.flat_ip_topk(corpus, queries, k)                          # the oracle on EVERY query of the batch
    assert np.array_equal(i, oi) and _bits(s) == _bits(os_)
    recall = np.mean([len(set(a) & set(b)) / k for a, b in zip(i, oi)])
    assert recall == 1.0                                                       # recall@10 vs the exact reference
    se, ie = idx.search(queries[:1024], k, MODE_EXACT)                         # fast path == exact kernel
    assert np.array_equal(ie, i[:1024]) and _bits(se) == _bits(s[:1024])
    s5, i5 = idx.search(queries[:512], 5, MODE_AUTO)                           # top-5 is a prefix of top-10
    assert np.array_equal(i5, i[:512, :5]) and _bits(s5) == _bits(s[:512, :5])
    assert st["last_fallback"] <= nq // 100                                    # certification rarely fails on this data
    adj, raw, ids, lv = idx.search_reweighted(queries, k, MODE_AUTO)
    want = oracle.reweight(s, i, levels)
    assert np.array_equal(ids, want[2]) and _bits(adj) == _bits(want[0])
    assert (np.diff(adj, axis=1) <= 0).all()
    idx.close()


def test_config5_shard_size_properties(oracle):
    """BASELINE configs[4] per-GPU shard: 1 250 000 x 768 rows (3.84 GB fp32 + 1.92 GB fp16 in HBM), a few hundred
    queries: oracle on a query sample, the rest through properties; ids carry the shard's id_base."""
    n, nq, k, base = 1_250_000, 300, 10, 3 * 1_250_000
    rng = np.random.default_rng(1234 + 3)
    corpus = rng.standard_normal((n, 768), dtype=np.float32)
    corpus /= np.linalg.norm(corpus, axis=1, keepdims=True)
    levels = icd_levels(n, 1238)
    queries = unit_rows(nq, 768, 4321)
    idx = IcdIndex(corpus, levels, max_nq=nq, max_k=k, id_base=base)
    s, i = idx.search(queries, k, MODE_AUTO)
    st = idx.stats()
    assert st["last_mode"] == MODE_AUTO
    assert (i >= base).all() and (i < base + n).all() and (np.diff(s, axis=1) <= 0).all()
    sample = np.unique(np.concatenate([np.arange(0, nq, 4), [nq - 1]]))          # 76 of the 300 queries, spread over all three query tiles
    assert len(sample) >= 64
    os_, oi = oracle.flat_ip_topk(corpus, queries[sample], k, id_base=base)
    assert np.array_equal(i[sample], oi) and _bits(s[sample]) == _bits(os_)
    s1, i1 = idx.search(queries[:3], k, MODE_AUTO)                              # tiny batch: streaming kernel, same answer
    assert np.array_equal(i1, i[:3]) and _bits(s1) == _bits(s[:3])
    adj, raw, ids, lv = idx.search_reweighted(queries[sample], k, MODE_AUTO)
    want = oracle.reweight(os_, oi, levels, id_base=base)
    assert np.array_equal(ids, want[2]) and _bits(adj) == _bits(want[0])
    idx.close()


def test_device_tensor_path_and_merge_kernel(oracle):
    import torch
    n, nq, k = 5000, 300, 10
    corpus, levels, queries = unit_rows(n, 768, 90), icd_levels(n, 91), unit_rows(nq, 768, 92)
    dq = torch.from_numpy(queries).cuda()
    full = IcdIndex(torch.from_numpy(corpus).cuda(), levels, max_nq=nq, max_k=k)   # corpus handed over on device
    s, i = full.search(dq, k)
    os_, oi = oracle.flat_ip_topk(corpus, queries, k)
    assert s.is_cuda and np.array_equal(i.cpu().numpy(), oi) and _bits(s.cpu().numpy()) == _bits(os_)
    # row-sharded: 3 uneven shards with global ids -> gather -> merge kernel == single index
    bounds = [(0, 1700), (1700, 1701), (1701, n)]
    parts = []
    for lo, hi in bounds:
        sh = IcdIndex(corpus[lo:hi], levels[lo:hi], max_nq=nq, max_k=k, id_base=lo)
        ps, pi = sh.search(dq, k)
        parts.append((ps, pi, sh.lookup_levels(pi)))
        sh.close()
    adj, raw, ids, lv = _native.merge_topk(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]),
                                           torch.stack([p[2] for p in parts]), k)
    want = oracle.reweight(os_, oi, levels)
    assert np.array_equal(ids.cpu().numpy(), want[2]) and _bits(adj.cpu().numpy()) == _bits(want[0])
    assert _bits(raw.cpu().numpy()) == _bits(want[1]) and np.array_equal(lv.cpu().numpy(), want[3])
    a2 = full.search_reweighted(dq, k)
    assert torch.equal(a2[2], ids) and torch.equal(a2[0], adj)
    full.close()


def test_sharded_search_single_rank_nccl(oracle):
    """ShardedSearch over the HIP index with a 1-rank RCCL group (the multi-rank logic is covered by the gloo tests)."""
    import torch
    import torch.distributed as dist
    from rag_project_icd10_amd.sharded import ROW_SHARD, ShardedSearch
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        n, nq, k = 3000, 64, 10
        corpus, levels, queries = unit_rows(n, 768, 95), icd_levels(n, 96), unit_rows(nq, 768, 97)
        idx = IcdIndex(corpus, levels, max_nq=nq, max_k=k)
        eng = ShardedSearch.from_index(idx, ROW_SHARD)
        adj, raw, ids, lv = eng.search_reweighted(torch.from_numpy(queries).cuda(), k)
        os_, oi = oracle.flat_ip_topk(corpus, queries, k)
        want = oracle.reweight(os_, oi, levels)
        assert np.array_equal(ids.cpu().numpy(), want[2]) and _bits(adj.cpu().numpy()) == _bits(want[0])
        idx.close()
    finally:
        dist.destroy_process_group()


def test_group_c_abi_with_a_one_rank_rccl_communicator(oracle):
    """icd_group_* (include/icd_search.h): the row-sharded and the query-sharded search behind the C ABI with RCCL opened by
    the library itself - a ONE-rank communicator here (ncclCommInitRank + the grouped ncclAllGather really run; two ranks
    cannot share this box's one GPU). Must equal the oracle and the torch.distributed engine of ShardedSearch."""
    import torch
    from rag_project_icd10_amd._native import GROUP_QUERY_SHARD, GROUP_ROW_SHARD, IcdGroup, group_unique_id
    n, nq, k = 5000, 300, 10
    corpus, levels, queries = unit_rows(n, 768, 195), icd_levels(n, 196), unit_rows(nq, 768, 197)
    os_, oi = oracle.flat_ip_topk(corpus, queries, k, id_base=7000)
    want = oracle.reweight(os_, oi, levels, id_base=7000)
    idx = IcdIndex(corpus, levels, max_nq=128, max_k=k, id_base=7000)       # (max_nq < nq: the wrapper / the library slice)
    dq = torch.from_numpy(queries).cuda()
    uid = group_unique_id()
    assert len(uid) == 128 and any(uid)
    for mode in (GROUP_ROW_SHARD, GROUP_QUERY_SHARD):
        for with_comm in (True, False):
            grp = IcdGroup(idx, mode, rank=0, world=1, unique_id=group_unique_id() if with_comm else None)
            adj, raw, ids, lv = grp.search(dq, k)
            torch.cuda.synchronize()
            assert np.array_equal(ids.cpu().numpy(), want[2]) and _bits(adj.cpu().numpy()) == _bits(want[0])
            assert _bits(raw.cpu().numpy()) == _bits(want[1]) and np.array_equal(lv.cpu().numpy(), want[3])
            grp.close()
    with pytest.raises(ValueError):
        IcdGroup(idx, GROUP_ROW_SHARD, rank=0, world=2)                      # more than one rank needs rank 0's id
    idx.close()


def test_query_sharded_search_single_rank_nccl_and_config3_share(oracle):
    """BASELINE configs[3]: the corpus replicated, the query batch sharded over the ranks (QUERY_SHARD over the HIP index,
    RCCL group of one rank here; two ranks in tests/test_sharded_cpu.py), at the per-GPU share of the config: 125 000
    queries x 37 000 rows. The oracle checks a query sample; the rest is covered by properties (sorted, valid ids,
    reweighted order consistent with the raw hits, idempotent, agreement with the plain index call)."""
    import torch
    import torch.distributed as dist
    from rag_project_icd10_amd.sharded import QUERY_SHARD, ShardedSearch
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = "29613"
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        n, nq, k = 37000, 125000, 10
        corpus, levels = unit_rows(n, 768, 1234), icd_levels(n, 1235)
        g = torch.Generator(device="cuda")
        g.manual_seed(4321)
        dq = torch.randn((nq, 768), generator=g, device="cuda", dtype=torch.float32)
        dq /= dq.norm(dim=1, keepdim=True)
        idx = IcdIndex(corpus, levels, max_nq=nq, max_k=k)
        eng = ShardedSearch.from_index(idx, QUERY_SHARD)
        adj, raw, ids, lv = eng.search_reweighted(dq, k, gather=True)
        st = idx.stats()
        assert st["last_mode"] == MODE_AUTO and st["last_nq"] == nq and st["last_fallback"] <= nq // 100
        assert adj.shape == (nq, k) and bool((ids >= 0).all()) and bool((ids < n).all())
        assert bool((adj[:, 1:] <= adj[:, :-1]).all())                               # reweighted order
        tl = torch.from_numpy(levels).cuda()
        w = torch.tensor([1.0, 1.2, 1.0, 0.8], dtype=torch.float64, device="cuda")
        assert torch.equal(lv.long(), tl[ids].long()) and torch.equal(adj, raw.double() * w[lv.long()])   # adj = raw x weight(level)
        a2, r2, i2, l2 = idx.search_reweighted(dq, k)                               # same as the plain call, idempotent
        assert torch.equal(i2, ids) and torch.equal(a2, adj)
        sample = torch.arange(0, nq, 977, device="cuda")
        qs = dq[sample].cpu().numpy()
        os_, oi = oracle.flat_ip_topk(corpus, qs, k)
        want = oracle.reweight(os_, oi, levels)
        assert np.array_equal(ids[sample].cpu().numpy(), want[2]) and _bits(adj[sample].cpu().numpy()) == _bits(want[0])
        assert _bits(raw[sample].cpu().numpy()) == _bits(want[1])
        idx.close()
    finally:
        dist.destroy_process_group()


def test_config4_per_gpu_share_at_full_size():
    """BASELINE configs[4] at its stated per-GPU size, END TO END through the driver's command: `python bench.py --gpus 1
    --workload rowshard` in a process of its own with ICD_SHARDED_ENGINE=native - a 1 250 000 x 768 shard generated on the
    device, the 100 000-query batch in slices of 16 384 through ShardedSearch(ROW) -> icd_group_search (the C-ABI group: local
    top-k, the all-gather step, merge + reweight), one pass; the 56-query sample is checked against the oracle over the whole
    shard (ids, raw and adjusted scores bit for bit), every slice's output is sorted, nearly everything stays certified, and the
    line names the engine that produced it."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["ICD_SHARDED_ENGINE"] = "native"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "rowshard", "--steps", "1", "--rowshard-steps", "1"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["config"]["engine"] == "icd_group (C ABI)" and line["config"]["collective_ranks"] == 1 and line["n_gpus"] == 1
    assert line["config"]["queries"] == 100_000 and line["config"]["rows_per_gpu"] == 1_250_000 and "configs[4]" in line["config"]["workload"]
    assert line["ids_exact_on_sample"] and line["raw_scores_exact_on_sample"] and line["adjusted_scores_exact_on_sample"] and line["adjusted_sorted"]
    assert line["sample_queries"] == 56 and line["sample_slices"] == 7 and "oracle" in line["sample_checked_against"]   # 8 of every slice, the 1 696-query last one included
    assert line["fallback_queries_last_slice"] <= 20 and line["value"] > 1e5
    # the paced coarse sweep of a 1.25 M-row shard: 0.46-0.47 of the fp16 MFMA peak on a quiet box (profiles/r05_bench_rowshard_n1.json);
    # a fraction below 0.35 is a regression of the kernel, not box noise
    assert 0.35 < line["roofline"]["frac"] < 1.0, line["roofline"]


def test_config0_composed_one_string_per_call():
    """BASELINE configs[0] as ONE workload, through the bench's own leg (bench.config0_extra): 100 golden diagnosis strings, per
    string encode_query (one string per call) -> MilvusService.search(top_k=5) (one query per call) over the 40 474-row
    database DatabaseBuilder builds in the same run (reference: services/multi_diagnosis_service.py:152-153,
    services/milvus_service.py:280-285, tools/build_database.py:217-222). Every hit of every string equals the oracle over the
    stored corpus bit for bit, and a stored row IS encode_query of its text."""
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    ctx = bench.Ctx()
    obj, es = bench.config0_extra(ctx)
    assert "error" not in obj, obj
    assert obj["strings"] == 100 and obj["top_k"] == 5 and obj["corpus_rows"] == 40474 and obj["parity_checked_strings"] == 100
    assert obj["hits_exact"] and obj["stored_row_equals_encode_query_of_its_text"] and obj["batch_arithmetic"] == "canonical"
    assert es is not None and obj["strings_per_sec"] > 200          # (measured ~1 900 strings/s; the CPU composition runs at ~5)


def test_config2_shape_batched_equals_one_at_a_time(tmp_path, monkeypatch):
    """BASELINE configs[2] at its stated size: the 1 000 golden diagnosis strings -> encoder on ROCm -> search over
    40 474 rows -> level reweight -> hierarchical rescoring. The batched path (one encoder batch, one search_batch, the
    device-side rescoring) must return what the reference's call shape returns (encode_query + search + the golden-pinned
    batch_calculate_similarities, one string at a time)."""
    from conftest import GOLDEN
    monkeypatch.setenv("MILVUS_DB_PATH", str(tmp_path / "db"))
    monkeypatch.setenv("MILVUS_COLLECTION_NAME", "icd10_cfg2")
    monkeypatch.setenv("EMBEDDING_MODEL_NAME", "shibing624/text2vec-base-chinese")
    monkeypatch.setenv("ICD_EMBEDDING_ALLOW_SYNTHETIC", "1")
    from rag_project_icd10_amd.services.embedding_service import EmbeddingService
    from rag_project_icd10_amd.services.milvus_service import MilvusService
    from rag_project_icd10_amd.services.multi_diagnosis_service import MultiDiagnosisService
    strings = [l.strip() for l in open(os.path.join(GOLDEN, "diagnosis_strings.txt"), encoding="utf-8") if l.strip()]
    assert len(strings) == 1000
    n = 40474
    es = EmbeddingService()
    ms = MilvusService(embedding_service=es)
    corpus, levels = unit_rows(n, 768, 1234), icd_levels(n, 1235)
    letters = "ABCEIJKNSZ"
    for s0 in range(0, n, 128):
        recs = [{"code": f"{letters[i % 10]}{i % 100:02d}.{(i // 7) % 10}" + ("9" if i % 11 == 0 else ""), "preferred_zh": f"合成疾病{i}",
                 "level": int(levels[i]), "parent_code": "", "category_path": "", "semantic_text": f"合成疾病{i}"}
                for i in range(s0, min(n, s0 + 128))]
        assert ms.insert_records(recs, list(corpus[s0:s0 + 128]))
    assert ms.load_collection()
    md = MultiDiagnosisService(es, ms)
    k = 10
    strings = strings + ["待查", "？", " 疑似 ", "肺炎待查", "高血压 糖尿病 肿瘤 感染"]    # empty clean query (exact-match rule), markers, chapter keywords
    # BOTH legs encode for themselves: the batched leg through encode_query_batch inside match_diagnoses_batch, the reference's
    # leg through encode_query, one string per call (services/multi_diagnosis_service.py:152-153). Round 5 handed both the same
    # vectors and could not see the packed split-bf16 forward flip a near-tie; the canonical batch path gives
    # every string the bits encode_query gives it, so the two legs agree on EVERY string, bit for bit.
    assert es.batch_arithmetic() == "canonical"
    batched = md.match_diagnoses_batch(strings, top_k=k)                            # additive entry point (row N2)
    assert len(batched) == len(strings)
    assert sum(1 for m in batched if m.candidates) >= len(strings) - 5
    for i in range(len(strings)):                                                   # the reference call shape on every string
        hits = ms.search(es.encode_query(strings[i]), top_k=k * 2)
        one = md._match_from_hits(strings[i], hits, k)
        got = batched[i]
        assert [c.code for c in got.candidates] == [c.code for c in one.candidates], strings[i]
        for a, b in zip(got.candidates, one.candidates):
            assert a.score == b.score and a.enhanced_score == b.enhanced_score and a.original_score == b.original_score
            assert a.similarity_factors == b.similarity_factors and a.title == b.title
        assert got.match_confidence == one.match_confidence


def _tight_family_corpus(nfam, per, dim, spread, nq, seed):
    """families of near-identical rows IN CODE ORDER, the shape the ICD corpus has (semantic_text repeats the ancestors'
    names, /root/reference/tools/build_database.py:156-171): spread 0.10 -> mutual cosine 0.99"""
    rng = np.random.default_rng(seed)
    cent = rng.standard_normal((nfam, dim)).astype(np.float32)
    x = np.repeat(cent, per, axis=0) + spread * rng.standard_normal((nfam * per, dim)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    q = cent[rng.integers(0, nfam, nq)] + spread * rng.standard_normal((nq, dim)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return np.ascontiguousarray(x, dtype=np.float32), np.ascontiguousarray(q, dtype=np.float32)


@pytest.fixture
def no_gc():
    """timed sections of a few ms: a garbage collection of this long pytest process inside one is a 10+ ms spike"""
    import gc
    gc.collect()
    gc.disable()
    yield
    gc.enable()


@pytest.mark.parametrize("k", [10, 20])
def test_family_corpus_is_certified_within_the_call(oracle, k, no_gc):
    """300 families x 124 rows of mutual cosine 0.99, 10 000 queries, a FRESH index, no icd_index_set_chunks. The narrow plan
    gives a query 5-8 lists of 16 candidates, fewer than its family has rows inside 2 eps of each other, so its first
    coarse pass certifies nothing. (a) Default: icd_index_create has asked the corpus itself (2 048 of its rows searched as
    one batch) and the FIRST user batch already runs on the wide partition: <= 2 ms, <= 1 % on the exact re-search. (b)
    Without that probe: the second coarse pass (flagged queries only, ~20 lists each, sized on the device) certifies them
    inside the same call (8.9 / 18 ms per batch when every query took the exact re-search), and the next large batch
    starts wide. Bit-exact results either way."""
    import time
    import torch
    from rag_project_icd10_amd import _native
    corpus, queries = _tight_family_corpus(300, 124, 768, 0.10, 10000, 7)
    n = corpus.shape[0]
    levels = icd_levels(n, 8)
    dq = torch.from_numpy(queries).cuda()
    sample = np.arange(0, len(queries), 10)
    os_, oi = oracle.flat_ip_topk(corpus, queries[sample], k)
    want = oracle.reweight(os_, oi, levels)
    # (a) the default: the corpus-shape probe at create
    idx_a = IcdIndex(corpus, levels, max_nq=10000, max_k=20)
    assert idx_a.stats()["wide_mode"] == 1 and idx_a.stats()["last_nq"] == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out_a = idx_a.search_reweighted(dq, k)                            # the FIRST user batch of the index
    torch.cuda.synchronize()
    fresh_ms = (time.perf_counter() - t0) * 1e3
    st = idx_a.stats()
    assert st["last_mode"] == MODE_AUTO and st["wide_mode"] == 1 and st["last_second_pass_lists"] == 0
    assert st["last_fallback"] <= 0.01 * len(queries), st
    assert np.array_equal(out_a[2].cpu().numpy()[sample], want[2]) and _bits(out_a[0].cpu().numpy()[sample]) == _bits(want[0])
    each = []
    for _ in range(9):
        t0 = time.perf_counter()
        idx_a.search_reweighted(dq, k)
        torch.cuda.synchronize()
        each.append((time.perf_counter() - t0) * 1e3)
    print(f"family corpus k={k}: fresh index with the create probe: first batch {fresh_ms:.2f} ms, then {' '.join('%.2f' % x for x in each)}")
    # wall-clock limits live in tests/test_zz_perf_gpu.py (run last): a slow box must not stop `pytest -x` in front of parity tests
    perf_records.record(f"family_k{k}_fresh_index_batches_ms", each)
    idx_a.close()
    # (b) the same without the probe: the second coarse pass inside the call
    idx = IcdIndex(corpus, levels, max_nq=10000, max_k=20, probe=False)
    assert idx.stats()["wide_mode"] == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    adj, raw, ids, lv = idx.search_reweighted(dq, k)                  # the FIRST large batch of the index
    torch.cuda.synchronize()
    first_ms = (time.perf_counter() - t0) * 1e3
    st = idx.stats()
    assert st["last_mode"] == MODE_AUTO and st["wide_mode"] == 0
    assert st["last_second_pass"] >= 0.9 * len(queries) and st["last_second_pass_lists"] >= 16   # the first pass certified (almost) nothing
    assert st["last_fallback"] <= 0.01 * len(queries), st                                        # ... the second nearly everything
    assert np.array_equal(ids.cpu().numpy()[sample], want[2]) and _bits(adj.cpu().numpy()[sample]) == _bits(want[0])
    assert _bits(raw.cpu().numpy()[sample]) == _bits(want[1])
    assert all(torch.equal(x, y) for x, y in zip(out_a, (adj, raw, ids, lv)))
    # the second large batch: planned wide from the start (the counters of the first one have arrived), same results
    a2, r2, i2, l2 = idx.search_reweighted(dq, k)
    torch.cuda.synchronize()
    st2 = idx.stats()
    assert st2["wide_mode"] == 1 and st2["last_second_pass_lists"] == 0 and st2["last_fallback"] <= 0.01 * len(queries)
    assert torch.equal(i2, ids) and torch.equal(a2, adj) and torch.equal(r2, raw)
    for _ in range(3):
        idx.search_reweighted(dq, k)
    torch.cuda.synchronize()
    each = []
    for _ in range(7):
        t0 = time.perf_counter()
        idx.search_reweighted(dq, k)
        torch.cuda.synchronize()
        each.append((time.perf_counter() - t0) * 1e3)
    wide_ms = sorted(each)[len(each) // 2]
    st3 = idx.stats()
    print(f"family corpus k={k}: first batch {first_ms:.2f} ms (second pass for {st['last_second_pass']} queries, {st['last_fallback']} exact), "
          f"wide-mode batches {wide_ms:.2f} ms (each: {' '.join('%.2f' % x for x in each)}; last: wide {st3['wide_mode']} lists {st3['last_chunks']} exact {st3['last_fallback']})")
    perf_records.record(f"family_k{k}_first_large_batch_ms", [first_ms])
    perf_records.record(f"family_k{k}_wide_mode_batches_ms", each)
    # a Gaussian batch on the same index is still exact (wide mode costs speed, never results), and a small batch is untouched
    g = unit_rows(4000, 768, 99)
    _check(oracle, idx, corpus, levels, g[:300], k, MODE_AUTO)
    idx.close()


def test_second_pass_switch_and_gaussian_batches_skip_it(oracle):
    """Gaussian data: nothing is flagged, the second pass's two launches find an empty list, wide mode never turns on;
    with the switch off a family corpus takes the exact re-search like before and returns the same bits"""
    import torch
    corpus, levels, queries = unit_rows(20000, 768, 5), icd_levels(20000, 6), unit_rows(4096, 768, 7)
    idx = IcdIndex(corpus, levels, max_nq=4096, max_k=10)
    for _ in range(2):
        st = _check(oracle, idx, corpus, levels, queries[:2500], 10, MODE_AUTO)
        assert st["last_second_pass"] == 0 and st["wide_mode"] == 0 and st["last_fallback"] == 0 and st["last_second_pass_lists"] > 0
    idx.close()
    fc, fq = _tight_family_corpus(100, 124, 768, 0.10, 6000, 11)           # 6 000 queries x 97 tiles: 6-7 lists of 16 per query
    fl = icd_levels(len(fc), 12)
    from rag_project_icd10_amd import _native
    idx = IcdIndex(fc, fl, max_nq=6000, max_k=10, probe=False)             # (the probe would start this corpus on the wide plan)
    a1 = [t.cpu().numpy() for t in idx.search_reweighted(torch.from_numpy(fq).cuda(), 10)]
    st1 = idx.stats()
    idx.set_second_pass(False)
    a0 = [t.cpu().numpy() for t in idx.search_reweighted(torch.from_numpy(fq).cuda(), 10)]
    st0 = idx.stats()
    assert st0["last_second_pass_lists"] == 0 and st0["last_fallback"] >= st1["last_second_pass"] > 0
    assert all(_bits(x) == _bits(y) for x, y in zip(a0, a1))
    os_, oi = oracle.flat_ip_topk(fc, fq[:200], 10)
    assert np.array_equal(a1[2][:200], oracle.reweight(os_, oi, fl)[2])
    idx.close()


def test_second_pass_launches_are_dropped_after_clean_searches_and_rearmed(oracle):
    """the second pass's two launches ride along while nothing is known about the corpus and after any search that flagged
    more queries than the streaming kernel takes cheaply; a few consecutive clean searches drop them (a Gaussian corpus pays
    nothing in steady state), one dirty search re-arms them. Results are exact in every state."""
    import torch
    corpus, levels, queries = unit_rows(20000, 768, 5), icd_levels(20000, 6), unit_rows(2500, 768, 7)
    idx = IcdIndex(corpus, levels, max_nq=2500, max_k=10)
    dq = torch.from_numpy(queries).cuda()
    lists = []
    for _ in range(7):
        idx.search_reweighted(dq, 10)
        st = idx.stats()                                   # (waits for the search: the next one sees its counters)
        lists.append(st["last_second_pass_lists"])
    assert all(x > 0 for x in lists[:4]) and all(x == 0 for x in lists[5:]), lists
    assert idx.stats()["second_pass_armed"] == 0
    zeros = torch.zeros((2500, 768), device="cuda")       # every score ties at 0: the first finalize certifies nothing
    a0 = idx.search_reweighted(zeros, 10)
    st = idx.stats()
    assert st["last_second_pass_lists"] == 0 and st["last_fallback"] == 2500        # disarmed: straight to the exact re-search
    a1 = idx.search_reweighted(zeros, 10)                                           # ... which re-armed the second pass
    st = idx.stats()
    assert st["last_second_pass_lists"] > 0 and st["second_pass_armed"] == 1
    assert all(torch.equal(x, y) for x, y in zip(a0, a1))
    assert bool((a0[2][:, 0] == 0).all()) and bool((a0[2][0] == torch.arange(10, device="cuda")).all())   # ties: row id ascending
    _check(oracle, idx, corpus, levels, queries[:300], 10, MODE_AUTO)
    idx.close()


def test_streaming_fallback_launches_are_dropped_after_a_long_clean_run_and_an_incident_is_exact(oracle):
    """the exact re-search behind every AUTO search is four launches; after 96 consecutive searches with nothing flagged
    (counted on the device: searches the host never waited for count too) the streaming kernel's two are left out and the
    fp32-MFMA kernel takes any count. A flagged query in that state is still answered exactly, restarts the run and
    doubles the required length."""
    import torch
    need = 96
    corpus, levels, queries = unit_rows(6000, 768, 15), icd_levels(6000, 16), unit_rows(96, 768, 17)
    idx = IcdIndex(corpus, levels, max_nq=96, max_k=10)
    dq = torch.from_numpy(queries).cuda()

    def clean(count):
        for _ in range(count):                             # enqueued back to back, whether or not the host sees them complete
            idx.search_reweighted(dq, 10)
        return idx.stats()                                 # (waits for the last search: the NEXT one sees the run length)

    want = idx.search_reweighted(dq, 10)
    st = idx.stats()
    assert st["sparse_fallback_armed"] == 1 and st["last_fallback"] == 0
    assert clean(need - 6)["sparse_fallback_armed"] == 1   # a run of need - 5
    clean(5)
    got = idx.search_reweighted(dq, 10)
    st = idx.stats()
    assert st["sparse_fallback_armed"] == 0 and st["last_fallback"] == 0
    assert all(torch.equal(x, y) for x, y in zip(got, want))
    dirty = dq.clone()
    dirty[5] = 0                                           # every score ties at 0: not certifiable
    dirty[40] = 0
    a0 = idx.search_reweighted(dirty, 10)
    st = idx.stats()
    assert st["sparse_fallback_armed"] == 0 and st["last_fallback"] == 2          # ... the MFMA kernel alone took them
    a1 = idx.search_reweighted(dirty, 10)                                         # re-armed: the streaming kernel again
    st = idx.stats()
    assert st["sparse_fallback_armed"] == 1 and st["last_fallback"] == 2
    assert all(torch.equal(x, y) for x, y in zip(a0, a1))
    assert bool((a0[2][5] == torch.arange(10, device="cuda")).all())              # ties: row id ascending
    keep = [i for i in range(96) if i not in (5, 40)]
    assert all(torch.equal(x[keep], y[keep]) for x, y in zip(a0, want))
    clean(need + 10)                                       # `need` clean searches are not enough a second time (2 * need now)
    idx.search_reweighted(dq, 10)
    assert idx.stats()["sparse_fallback_armed"] == 1
    clean(need)
    idx.search_reweighted(dq, 10)
    assert idx.stats()["sparse_fallback_armed"] == 0
    idx.set_second_pass(True, adaptive=False)              # the test hook's non-adaptive setting keeps all four launches
    idx.search_reweighted(dq, 10)
    assert idx.stats()["sparse_fallback_armed"] == 1
    _check(oracle, idx, corpus, levels, queries, 10, MODE_AUTO)
    idx.close()


def test_threads_sharing_a_handle_are_serialised(oracle):
    """ctypes releases the GIL: four threads enter icd_index_search_reweighted on ONE handle at once (same stream). The
    library serialises them (a mutex around the handle's host state; the device work queues up in stream order): every
    thread gets the results a single-threaded caller gets"""
    import threading
    import torch
    corpus, levels = unit_rows(12000, 768, 21), icd_levels(12000, 22)
    idx = IcdIndex(corpus, levels, max_nq=600, max_k=10)
    batches = [torch.from_numpy(unit_rows(nq, 768, 30 + i)).cuda() for i, nq in enumerate((600, 37, 256, 8))]
    want = [[t.clone() for t in idx.search_reweighted(b, 10)] for b in batches]
    torch.cuda.synchronize()
    errors = []

    def worker(i):
        try:
            for _ in range(25):
                got = idx.search_reweighted(batches[i], 10)
                torch.cuda.synchronize()
                if not all(torch.equal(a, b) for a, b in zip(got, want[i])):
                    errors.append(i)
                idx.stats()
        except Exception as exc:   # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    _check(oracle, idx, corpus, levels, batches[0].cpu().numpy()[:100], 10, MODE_AUTO)
    idx.close()


def test_device_hier_rescoring_matches_host():
    """icd_hier_rescore ALONE (through the C ABI) against HierarchicalSimilarityService.batch_calculate_similarities - the
    host method the reference-generated fixture pins (tests/golden/hier_cases.json, test_golden_host_logic.py) - on all
    1 000 golden diagnosis strings + 5 edge strings: final order, enhanced score, the record's score after the uncertainty
    boost, the applied boost, and the six similarity factors, bit for bit (Python doubles). The hit lists are synthetic
    (the search is not under test): 20 live-shaped hits per string with adjusted scores on both sides of the 0.95 / 1.0
    rules, negative scores, exact ties (stable order), codes of every chapter, ".9" codes and hit lists shorter than k.
    Reference: services/hierarchical_similarity_service.py:475-518,520-579, services/uncertainty_diagnosis_service.py:190-238."""
    import torch
    from conftest import GOLDEN
    from rag_project_icd10_amd.services.hierarchical_similarity_service import HierarchicalSimilarityService, SimilarityFactors
    strings = [l.strip() for l in open(os.path.join(GOLDEN, "diagnosis_strings.txt"), encoding="utf-8") if l.strip()]
    strings = strings + ["待查", "？", " 疑似 ", "肺炎待查", "高血压 糖尿病 肿瘤 感染"]
    nq, k, nrows = len(strings), 20, 6000
    assert nq == 1005
    rng = np.random.default_rng(99)
    letters = "ABCEIJKNSZQ"
    # unique codes (a hit is found again by its code): every chapter letter of the table and two outside it; one code in
    # five matches the uncertainty service's \\.9\\d*$ (".9" + digits), the others have another digit behind the dot or a
    # letter at the end; every 17th is a level-1 shaped code without a dot
    codes = []
    for i in range(nrows):
        L = letters[i % 11]
        if i % 17 == 0:
            codes.append(f"{L}{i:05d}")
        elif i % 5 == 0:
            codes.append(f"{L}{i % 100:02d}.9{i:05d}" if i % 10 else f"{L}{i % 100:02d}.9")   # (".9" itself is not unique: see below)
        else:
            codes.append(f"{L}{i % 100:02d}.{(i // 3) % 9}{i:05d}" + ("x" if i % 13 == 0 else ""))
    seen = set()
    for i, c in enumerate(codes):                                                         # the bare ".9" codes repeat: keep the first of each
        if c in seen:
            codes[i] = f"{c}{i:05d}"
        seen.add(codes[i])
    assert len(set(codes)) == nrows
    recs = [{"code": c, "preferred_zh": f"合成{i}", "level": 1 + i % 3, "parent_code": "", "category_path": "", "semantic_text": ""}
            for i, c in enumerate(codes)]
    ids = np.stack([rng.choice(nrows, k, replace=False) for _ in range(nq)]).astype(np.int64)
    adj = np.sort(rng.uniform(0.2, 1.25, (nq, k)), axis=1)[:, ::-1].copy()               # search order: adjusted score descending
    adj[::7, 3] = adj[::7, 2]                                                            # exact ties
    adj[5::11, :4] = np.asarray([0.97, 0.9500000000000001, 0.95, 0.9499999999999999])    # around the > 0.95 rule
    adj[3::13, -2:] = [-0.01, -0.3]                                                      # negative adjusted scores
    raw = (adj / 1.2).astype(np.float32)
    ids[9::19, 15:] = -1                                                                 # short hit lists (n < k): padded with -1
    adj[9::19, 15:] = -np.inf
    hs = HierarchicalSimilarityService()
    tags = torch.from_numpy(np.asarray([hs.row_tag(c) for c in codes], np.uint8)).cuda()
    order, enh, score, vs, hb, boost = hs.rescore_live_hits_batch(strings, torch.from_numpy(adj).cuda(), torch.from_numpy(ids).cuda(), tags)
    order, enh, score, vs, hb, boost = (t.cpu().numpy() for t in (order, enh, score, vs, hb, boost))
    sc = 0.3 if hs.embedding_service else 0.5
    for q, text in enumerate(strings):
        nhit = int((ids[q] >= 0).sum())
        hits = [{"code": recs[i]["code"], "title": recs[i]["preferred_zh"], "score": float(adj[q, j]), "original_score": float(raw[q, j]),
                 "metadata": {"level": recs[i]["level"], "parent_code": "", "category_path": "", "semantic_text": "",
                              "has_complication": False, "main_code": "", "secondary_code": ""}}
                for j, i in enumerate(ids[q][:nhit])]
        pos = {h["code"]: j for j, h in enumerate(hits)}
        want = hs.batch_calculate_similarities(text, {}, [dict(h) for h in hits])
        assert len(want) == nhit and (order[q, nhit:] < 0).all(), text
        ctx = hs.query_params(text)[1]
        for j, (rec, s_host, f_host) in enumerate(want):
            assert order[q, j] == pos[rec["code"]], (text, j)                            # final order (stable sorts included)
            assert enh[q, j] == s_host == rec["enhanced_score"], (text, j)               # bit-exact doubles
            assert score[q, j] == rec["score"], (text, j)                                # after the uncertainty boost
            assert boost[q, j] == rec.get("uncertainty_boost", 0.0), (text, j)
            assert SimilarityFactors(vs[q, j], hb[q, j], 0.0, sc, 0.0, ctx) == f_host, (text, j)


def test_services_end_to_end_on_gpu(oracle, tmp_path, monkeypatch):
    """csv slice -> DatabaseBuilder (batched encode on ROCm) -> MilvusService.search: every row retrieves itself,
    and the hit dicts have the reference's shape and its level reweight."""
    from conftest import GOLDEN
    monkeypatch.setenv("MILVUS_DB_PATH", str(tmp_path / "db"))
    monkeypatch.setenv("MILVUS_COLLECTION_NAME", "icd10_test")
    monkeypatch.setenv("EMBEDDING_MODEL_NAME", "shibing624/text2vec-base-chinese")
    monkeypatch.setenv("ICD_EMBEDDING_ALLOW_SYNTHETIC", "1")
    from rag_project_icd10_amd.tools.build_database import DatabaseBuilder
    b = DatabaseBuilder()
    b.initialize_services()
    recs = b.load_csv_data(os.path.join(GOLDEN, "csv_slice.csv"))
    assert b.vectorize_and_index(recs) is True
    ms, es = b.milvus_service, b.embedding_service
    assert ms.get_collection_stats()["num_entities"] == len(recs) and ms.get_collection_load_state()["loaded"]
    corpus, levels = ms.client.matrix(), ms.client.levels()
    for r in (0, 2, 57, len(recs) - 1):
        vec = es.encode_query(recs[r]["semantic_text"])
        hits = ms.search(vec, top_k=5)
        assert len(hits) == 5 and recs[r]["code"] in [h["code"] for h in hits]
        assert set(hits[0]) == {"code", "title", "score", "original_score", "metadata"}
        assert set(hits[0]["metadata"]) == {"has_complication", "main_code", "secondary_code", "level", "parent_code",
                                            "category_path", "semantic_text"}
        os_, oi = oracle.flat_ip_topk(corpus, vec[None], 5)
        adj, raw, ids, _ = oracle.reweight(os_, oi, levels)
        assert [h["code"] for h in hits] == [recs[i]["code"] for i in ids[0]]
        assert [h["score"] for h in hits] == adj[0].tolist()                     # exact Python doubles
        assert [h["original_score"] for h in hits] == [float(x) for x in raw[0]]
        assert all(h["score"] == h["original_score"] * ms._calculate_level_weight(h["metadata"]["level"]) for h in hits)
        best_raw = max(hits, key=lambda h: h["original_score"])
        assert best_raw["code"] == recs[r]["code"] and best_raw["original_score"] > 0.9999
    ver = b.verify_database()
    assert ver["search_test"]["results_count"] == 5
    # batched serving path == looping the reference-shaped call
    from rag_project_icd10_amd.services.multi_diagnosis_service import MultiDiagnosisService
    res = MultiDiagnosisService(es, ms).match_multiple_diagnoses("霍乱，伤寒；副伤寒", top_k=3)
    assert res["extracted_diagnoses"] == ["霍乱", "伤寒", "副伤寒"] and res["total_matches"] == 9
    # /query's request path stays on the device when there are no entities to match; the host path (dict marshalling +
    # batch_calculate_similarities per diagnosis, the reference's own shape) must give the same DiagnosisMatch objects
    assert ms.supports_device_rescoring()
    ms.supports_device_rescoring = lambda: False
    res_host = MultiDiagnosisService(es, ms).match_multiple_diagnoses("霍乱，伤寒；副伤寒", top_k=3)
    del ms.supports_device_rescoring
    assert [m.model_dump() for m in res["matches"]] == [m.model_dump() for m in res_host["matches"]]
    assert {k: v for k, v in res.items() if k != "matches"} == {k: v for k, v in res_host.items() if k != "matches"}
    for m in res["matches"]:
        single = ms.search(es.encode_query(m.diagnosis_text), 6)
        assert {c.code for c in m.candidates} <= {h["code"] for h in single}
    # ... and with an NER service (its rules here): the reference's default, ENHANCED text mode - the text's segments embedded in one batch
    # for the boundary confidences, a diagnosis that is a boundary's text searched with that vector. Every DiagnosisMatch is what
    # the one-at-a-time calls give: encode_query -> search(2 top_k) -> rescoring with the diagnosis's own entities.
    from rag_project_icd10_amd.services.medical_ner_service import MedicalNERService
    ner = MedicalNERService(use_model=False)
    md = MultiDiagnosisService(es, ms, ner_service=ner)
    text = "霍乱，伤寒；副伤寒；细菌性食物中毒"
    res_e = md.match_multiple_diagnoses(text, top_k=3)
    assert res_e["processing_mode"] == "enhanced" and len(res_e["extracted_diagnoses"]) >= 2
    detector = md.text_processor._enhanced_processor.boundary_detector
    reused = 0
    for d, m in zip(res_e["extracted_diagnoses"], res_e["matches"]):
        vec = es.encode_query(d)
        cached = detector.cached_vector(d)
        if cached is not None:
            reused += 1
            assert np.array_equal(np.asarray(cached), vec), d          # the batch gave the string the one-string call's bits
        want = md._match_from_hits(d, ms.search(vec, 6), 3, ner.extract_medical_entities(d))
        assert m.model_dump() == want.model_dump(), d
    assert reused >= 2
    assert ms.release_collection()["success"] and ms.get_collection_load_state()["loaded"] is False
    assert ms.load_collection() is True and ms.disconnect()["success"]


@pytest.mark.parametrize("k", [10, 20])
def test_anisotropic_embeddings_are_certified_on_a_centred_image(oracle, k):
    """Sentence embeddings share a large common component (the synthetic encoder's rows have mean pairwise cosine 0.98; here 0.96; at 0.995 the scores of a whole corpus lie within 3e-4 of each other, below the fp32 chain's own error bound: nothing but the exact kernels can answer). The
    scores of a query then differ in the third digit while the fp16 certificate's window is relative to the full norm: most
    queries fail it and take the exact re-search. icd_index_create centres the fp16 image in that case (q.c = q.(c - mu) +
    q.mu, the second term constant per query): the same exact results, and the fast path certifies again."""
    import torch
    from rag_project_icd10_amd import _native
    rng = np.random.default_rng(31)
    n, nq, dim = 20000, 2000, 768
    mu0 = rng.standard_normal(dim).astype(np.float32)
    mu0 /= np.linalg.norm(mu0)

    def rows(m):   # unit rows = the common direction + noise of norm ~0.2: cosine of two rows ~0.96
        x = mu0[None, :] + (0.2 / np.sqrt(dim)) * rng.standard_normal((m, dim)).astype(np.float32)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        return np.ascontiguousarray(x, dtype=np.float32)

    corpus, queries = rows(n), rows(nq)
    levels = icd_levels(n, 32)
    os_, oi = oracle.flat_ip_topk(corpus, queries, k)
    want = oracle.reweight(os_, oi, levels)
    dq = torch.from_numpy(queries).cuda()
    lib = _native.load_library()
    fall = {}
    for center in (1, 0):
        idx = IcdIndex(corpus, levels, max_nq=nq, max_k=k, center=bool(center))
        st = idx.stats()
        assert st["centered"] == center and st["mean_share"] > 0.5 and abs(st["rmax"] - 1.0) < 1e-3
        for _ in range(2):
            adj, raw, ids, lv = idx.search_reweighted(dq, k)
        torch.cuda.synchronize()
        st = idx.stats()
        assert st["last_mode"] == MODE_AUTO
        assert np.array_equal(ids.cpu().numpy(), want[2]) and _bits(adj.cpu().numpy()) == _bits(want[0]) and _bits(raw.cpu().numpy()) == _bits(want[1])
        fall[center] = int(st["last_fallback"])
        idx.close()
    print(f"anisotropic corpus (cosine ~0.96), k={k}: exact re-search for {fall[1]} of {nq} queries centred, {fall[0]} uncentred")
    assert fall[1] <= nq // 20 and fall[1] <= fall[0]
    # an isotropic corpus stays as it was
    idx = IcdIndex(unit_rows(3000, 768, 33), icd_levels(3000, 34), max_nq=64, max_k=10)
    assert idx.stats()["centered"] == 0 and idx.stats()["mean_share"] < 0.05
    idx.close()


@pytest.mark.parametrize("flagged,k", [(41, 10), (120, 10), (300, 20), (700, 20)])
def test_short_flagged_lists_take_the_chunked_exact_research(oracle, flagged, k):
    """A family of 400 near-identical rows is larger than any rescoring window (256): the queries that land in it cannot be
    certified from candidate lists and take the exact re-search - more than 40 of them the fp32-MFMA kernel, whose short
    flagged list is cut into up to 2 048 / KP row chunks (finalize<false, ., 4> merges them) and whose lists start at the
    threshold finalize hands over (the exact score of the query's k-th coarse candidate; ties with it must pass: the
    family's scores tie in many bits). Every flagged query and a sample of the others against the oracle, bit for bit."""
    import torch
    rng = np.random.default_rng(90 + flagged)
    n, nq, dim, fam = 37000, 10000, 768, 400
    corpus = unit_rows(n, dim, 91)
    f = rng.standard_normal(dim).astype(np.float32)
    f /= np.linalg.norm(f)

    def near(m):
        x = f[None, :] + (0.02 / np.sqrt(dim)) * rng.standard_normal((m, dim)).astype(np.float32)
        return np.ascontiguousarray(x / np.linalg.norm(x, axis=1, keepdims=True), dtype=np.float32)

    at = rng.choice(n, fam, replace=False)
    corpus[at] = near(fam)
    corpus[at[:8]] = corpus[at[8]]                      # exact duplicates inside the family: score ties, decided by the row id
    queries = unit_rows(nq, dim, 92)
    where = rng.choice(nq, flagged, replace=False)
    queries[where] = near(flagged)
    levels = icd_levels(n, 93)
    check = np.concatenate([where, rng.choice(nq, 64, replace=False)])
    os_, oi = oracle.flat_ip_topk(corpus, queries[check], k)
    want = oracle.reweight(os_, oi, levels)
    idx = IcdIndex(corpus, levels, max_nq=nq, max_k=k)
    dq = torch.from_numpy(queries).cuda()
    for _ in range(2):
        adj, raw, ids, lv = idx.search_reweighted(dq, k)
    torch.cuda.synchronize()
    st = idx.stats()
    assert st["last_mode"] == MODE_AUTO and flagged <= st["last_fallback"] <= flagged + 8, st
    assert np.array_equal(ids.cpu().numpy()[check], want[2])
    assert _bits(adj.cpu().numpy()[check]) == _bits(want[0]) and _bits(raw.cpu().numpy()[check]) == _bits(want[1])
    idx.close()


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_query_sharded_unpack_kernel_matches_the_host_concatenation(world):
    """icd_group_search, query-sharded with gather: ONE kernel scatters the all-gathered padded slices to the [nq][k] outputs
    (it replaced four hipMemcpyAsync per rank). Its index arithmetic restates shard_bounds; checked here for any world size on
    one GPU through icd_unpack_query_slices against sharded.py's host-side concatenation rule (nq % world != 0 and
    nq < world included). No reference counterpart: the reference is a single process (main.py:753-758)."""
    import ctypes

    import torch
    from rag_project_icd10_amd.sharded import shard_bounds
    lib = _native.load_library()
    rng = np.random.default_rng(7 + world)
    for nq, k in ((1, 1), (5, 10), (world - 1 or 1, 3), (17, 10), (1000, 7), (4097, 100)):
        adj = rng.standard_normal((nq, k))
        raw = rng.standard_normal((nq, k)).astype(np.float32)
        ids = rng.integers(-1, 2 ** 40, (nq, k))
        lv = rng.integers(0, 4, (nq, k)).astype(np.int32)
        width = -(-nq // world)
        per = width * k
        g_adj, g_ids = np.full((world, per), np.nan), np.full((world, per), -7, np.int64)
        g_raw, g_lv = np.full((world, per), np.nan, np.float32), np.full((world, per), -7, np.int32)
        for r in range(world):
            lo, hi = shard_bounds(nq, world, r)
            m = (hi - lo) * k
            g_adj[r, :m], g_ids[r, :m] = adj[lo:hi].ravel(), ids[lo:hi].ravel()
            g_raw[r, :m], g_lv[r, :m] = raw[lo:hi].ravel(), lv[lo:hi].ravel()
        blob = g_adj.tobytes() + g_ids.tobytes() + g_raw.tobytes() + g_lv.tobytes()
        gathered = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
        o_adj = torch.empty((nq, k), dtype=torch.float64, device="cuda")
        o_raw = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        o_ids = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        o_lv = torch.empty((nq, k), dtype=torch.int32, device="cuda")
        vp = ctypes.c_void_p
        rc = lib.icd_unpack_query_slices(0, vp(gathered.data_ptr()), world, nq, k, vp(o_adj.data_ptr()), vp(o_raw.data_ptr()),
                                               vp(o_ids.data_ptr()), vp(o_lv.data_ptr()), vp(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, lib.icd_last_error()
        torch.cuda.synchronize()
        assert _bits(o_adj.cpu().numpy()) == _bits(adj) and _bits(o_raw.cpu().numpy()) == _bits(raw), (world, nq, k)
        assert np.array_equal(o_ids.cpu().numpy(), ids) and np.array_equal(o_lv.cpu().numpy(), lv), (world, nq, k)


@pytest.mark.parametrize("n", [7, 100, 129, 1000, 40474, 300000])
def test_one_and_two_queries_take_the_single_launch_kernel_and_match_the_oracle(oracle, n):
    """The reference's call shape - ONE query per MilvusService.search call (services/milvus_service.py:280-285) - and up to eight
    (a /query request batches the searches of its D diagnoses, services/multi_diagnosis_service.py:98-103,153; up to four take this kernel):
    stream_topk_kernel<ONE> folds the list reduction and finalize into the streaming launch (last-arriver ticket). Bit-equal
    to the oracle and to the general four-operation path (set_option("stream_one", 0)), over corpus sizes from one
    work-group to multi-step sweeps, k = 1 ... 16, exact ties (duplicate rows) included; repeated calls (the ticket only
    ever counts up)."""
    dim = 768
    corpus, levels = unit_rows(n, dim, 41 + n), icd_levels(n, 42 + n)
    if n >= 100:
        corpus[5::7] = corpus[2]          # duplicate rows: exact score ties, broken by row id
    queries = unit_rows(8, dim, 43 + n)
    queries[5] = corpus[2]                # a query that IS the duplicated row
    idx = IcdIndex(corpus, levels, max_nq=64, max_k=16)
    lib = _native.load_library()
    try:
        for k in (1, 5, 10, 16):
            for lo, hi in ((0, 1), (1, 2), (2, 4), (4, 6), (5, 6), (1, 4), (0, 4), (0, 5), (0, 6), (0, 8)):   # 1, 2, 3-4 and 5-8 queries per call
                q = queries[lo:hi]
                os_, oi = oracle.flat_ip_topk(corpus, q, k)
                want = oracle.reweight(os_, oi, levels)
                idx.set_option("stream_one", 1)
                for rep in range(2):
                    s, i = idx.search(q, k)
                    got = idx.search_reweighted(q, k)
                    assert np.array_equal(i, oi) and _bits(s) == _bits(os_), (n, k, lo, hi, rep)
                    assert np.array_equal(got[2], want[2]) and _bits(got[0]) == _bits(want[0]) and _bits(got[1]) == _bits(want[1])
                    assert np.array_equal(got[3], want[3])
                assert idx.stats()["last_mode"] == MODE_EXACT and idx.stats()["last_fallback"] == 0
                idx.set_option("stream_one", 0)
                old = idx.search_reweighted(q, k)
                assert all(_bits(a) == _bits(b) for a, b in zip(old, got))
    finally:
        idx.close()


@pytest.mark.parametrize("dim", [256, 768, 1024])
def test_a_host_callers_one_query_in_the_kernel_arguments_and_the_polled_completion(oracle, dim):
    """ONE query per call from HOST memory (MilvusClient.search(data=[query_vector.tolist()]), services/milvus_service.py:280-285):
    the vector travels in the single-launch kernel's arguments and the call returns on a polled completion word
    (the per-index option host_one, bits 1 and 2). All four settings are bit-equal to the oracle, call after call (the word is a
    sequence number), also interleaved with batch calls, calls of two queries and device-resident calls on the same handle;
    1024-d vectors do not fit the arguments and take the copy; k > 16 leaves the single-launch kernel (the copy is enqueued
    after all)."""
    import torch
    n = 5000
    corpus, levels = unit_rows(n, dim, 81 + dim), icd_levels(n, 82 + dim)
    corpus[9::11] = corpus[4]
    queries = unit_rows(40, dim, 83 + dim)
    queries[3] = corpus[4]
    idx = IcdIndex(corpus, levels, max_nq=64, max_k=32)
    lib = _native.load_library()
    want = {}
    for k in (1, 10, 16, 20):
        os_, oi = oracle.flat_ip_topk(corpus, queries, k)
        want[k] = (os_, oi, oracle.reweight(os_, oi, levels))
    dq = torch.from_numpy(queries).cuda()
    try:
        for bits in (3, 0, 1, 2, 3):
            idx.set_option("host_one", bits)
            for k in (10, 1, 16, 20):
                os_, oi, rw = want[k]
                for j in range(12):
                    s, i = idx.search(queries[j:j + 1], k)
                    assert np.array_equal(i, oi[j:j + 1]) and _bits(s) == _bits(os_[j:j + 1]), (dim, bits, k, j)
                    got = idx.search_reweighted(queries[j:j + 1], k)
                    assert all(_bits(a) == _bits(b[j:j + 1]) for a, b in zip(got, rw)), (dim, bits, k, j)
                    if j % 4 == 1:     # a batch, two queries and a device-resident query in between
                        s2, i2 = idx.search(queries[:40], k)
                        assert np.array_equal(i2, oi) and _bits(s2) == _bits(os_)
                        s3, i3 = idx.search(queries[j:j + 2], k)
                        assert np.array_equal(i3, oi[j:j + 2]) and _bits(s3) == _bits(os_[j:j + 2])
                        s4, i4 = idx.search(dq[j:j + 1], k)
                        torch.cuda.synchronize()
                        assert np.array_equal(i4.cpu().numpy(), oi[j:j + 1]) and _bits(s4.cpu().numpy()) == _bits(os_[j:j + 1])
    finally:
        idx.close()


def test_one_query_host_calls_stay_exact_beside_a_busy_gpu(oracle):
    """the polled completion of a host caller's one query gives up after 0.5 ms and waits for the stream instead: with large
    batches of ANOTHER index running on another stream from another thread, one-query host calls (their kernel queued behind
    0.3-ms coarse launches) take both exits - every result stays exact, and the batches' too"""
    import threading
    import torch
    dim = 768
    corpus_a, levels_a = unit_rows(20000, dim, 301), icd_levels(20000, 302)
    corpus_b, levels_b = unit_rows(20000, dim, 303), icd_levels(20000, 304)
    qa = unit_rows(48, dim, 305)
    qb = unit_rows(6000, dim, 306)
    want_s, want_i = oracle.flat_ip_topk(corpus_a, qa, 10)
    wb_s, wb_i = oracle.flat_ip_topk(corpus_b, qb[:64], 10)
    ia = IcdIndex(corpus_a, levels_a, max_nq=64, max_k=16)
    ib = IcdIndex(corpus_b, levels_b, max_nq=6000, max_k=16)
    stop, errors, batches = threading.Event(), [], [0]

    def load():
        try:
            st = torch.cuda.Stream()
            dq = torch.from_numpy(qb).cuda()
            with torch.cuda.stream(st):
                while not stop.is_set():
                    s, i = ib.search(dq, 10)
                    st.synchronize()
                    batches[0] += 1
                    if batches[0] % 50 == 0 and not (np.array_equal(i[:64].cpu().numpy(), wb_i) and _bits(s[:64].cpu().numpy()) == _bits(wb_s)):
                        errors.append("batch result differs")
        except Exception as exc:   # pragma: no cover
            errors.append(repr(exc))
    th = threading.Thread(target=load)
    th.start()
    try:
        import time
        t_end, calls = time.time() + 30.0, 0
        while (batches[0] < 40 or calls < 400) and time.time() < t_end and not errors:
            j = calls % 48
            s, i = ia.search(qa[j:j + 1], 10)
            assert np.array_equal(i, want_i[j:j + 1]) and _bits(s) == _bits(want_s[j:j + 1]), (calls, j)
            calls += 1
    finally:
        stop.set()
        th.join(timeout=60)
        ia.close()
        ib.close()
    assert not errors, errors
    assert batches[0] >= 20, batches[0]


@pytest.mark.parametrize("k", [33, 50, 64, 100])
def test_exact_mode_above_k32_runs_certified_narrow_lists_and_stays_exact(oracle, k):
    """ICD_MODE_EXACT at k > 32 (the k range /query can ask for: top_k * 2 with top_k <= 50, models/icd_models.py:138,
    services/multi_diagnosis_service.py:153): lists of 32 over row-strided chunks + the certificate of finalize.hpp
    (narrow_check), re-search of what it cannot clear. Bit-equal to the oracle and to the KP >= k lists
    (set_option("exact_narrow", 0)) on Gaussian rows, on families of near-identical NEIGHBOURING rows, on hundreds of exact
    duplicates of one row, and on a corpus built so that ONE list holds more than 32 members of a query's top-k (the
    certificate must flag it: last_fallback > 0)."""
    import torch
    lib = _native.load_library()
    n, dim, nq = 9000, 768, 200
    rng = np.random.default_rng(100 + k)
    gauss = unit_rows(n, dim, 300 + k)
    fam = np.repeat(rng.standard_normal((n // 120, dim)).astype(np.float32), 120, axis=0)
    fam = fam + 0.1 * rng.standard_normal(fam.shape).astype(np.float32)
    fam /= np.linalg.norm(fam, axis=1, keepdims=True)
    dup = gauss.copy()
    dup[rng.choice(n, 700, replace=False)] = gauss[17]
    queries = unit_rows(nq, dim, 400 + k)
    queries[:40] = fam[np.arange(40) * 120 + 5] + 0.05 * rng.standard_normal((40, dim)).astype(np.float32)
    queries[40] = gauss[17]
    queries /= np.linalg.norm(queries, axis=1, keepdims=True)
    try:
        pn = None
        for name, corpus in (("gauss", gauss), ("families", np.ascontiguousarray(fam[:n // 120 * 120])), ("duplicates", dup)):
            levels = icd_levels(len(corpus), 9)
            idx = IcdIndex(corpus, levels, max_nq=nq, max_k=100)
            os_, oi = oracle.flat_ip_topk(corpus, queries, k)
            want = oracle.reweight(os_, oi, levels)
            for narrow in (1, 0):
                idx.set_option("exact_narrow", narrow)
                s, i = idx.search(queries, k, MODE_EXACT)
                got = idx.search_reweighted(queries, k, MODE_EXACT)
                assert np.array_equal(i, oi) and _bits(s) == _bits(os_), (name, narrow)
                assert np.array_equal(got[2], want[2]) and _bits(got[0]) == _bits(want[0]) and np.array_equal(got[3], want[3]), (name, narrow)
                if narrow:
                    st = idx.stats()
                    pn = st["last_chunks"]
                    assert st["last_mode"] == MODE_EXACT and st["last_fallback"] == 0, (name, st)   # rows that spread: nothing to re-search
            idx.close()
        # the adversarial corpus: 45 copies of query 7 at rows 3, 3 + P, 3 + 2 P, ... = ONE strided chunk holds 45 members of its top-k
        adv = gauss.copy()
        adv[3 + pn * np.arange(45)] = queries[7]
        levels = icd_levels(n, 9)
        idx = IcdIndex(adv, levels, max_nq=nq, max_k=100)
        os_, oi = oracle.flat_ip_topk(adv, queries, k)
        s, i = idx.search(queries, k, MODE_EXACT)
        st = idx.stats()
        assert st["last_chunks"] == pn and st["last_fallback"] >= 1, st          # the certificate saw the full list
        assert np.array_equal(i, oi) and _bits(s) == _bits(os_)
        idx.close()
    finally:
        pass


def test_device_resident_searches_can_be_captured_into_a_hip_graph(oracle):
    """Device-in / device-out searches allocate nothing and never synchronise (include/icd_search.h): a batch through the fp16
    fast path and a one-query call are captured into ONE HIP graph and replayed; the replayed results are the oracle's, eager
    searches and icd_index_stats still work afterwards, and a host-buffer call inside a capture is refused with a message
    (it would synchronise). The reference has no counterpart (it calls Milvus per query: services/milvus_service.py:280-285)."""
    import torch
    n, dim, k = 20000, 768, 10
    corpus, levels = unit_rows(n, dim, 71), icd_levels(n, 72)
    queries = unit_rows(2000, dim, 73)
    idx = IcdIndex(corpus, levels, max_nq=2000, max_k=10)
    dq = torch.from_numpy(queries).cuda()
    os_, oi = oracle.flat_ip_topk(corpus, queries, k)
    want = oracle.reweight(os_, oi, levels)
    for _ in range(2):                                   # warm-up outside the capture (kernel attributes, workspaces)
        idx.search_reweighted(dq, k)
        idx.search_reweighted(dq[:1], k)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        big = idx.search_reweighted(dq, k)
        one = idx.search_reweighted(dq[7:8], k)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(big[2].cpu().numpy(), want[2]) and _bits(big[0].cpu().numpy()) == _bits(want[0])
    assert np.array_equal(one[2].cpu().numpy(), want[2][7:8]) and _bits(one[0].cpu().numpy()) == _bits(want[0][7:8])
    # new inputs in the captured buffers, replayed: the graph reads the tensors it captured
    dq.copy_(torch.from_numpy(queries[::-1].copy()).cuda())
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(big[2].cpu().numpy(), want[2][::-1])
    # eager use of the same index afterwards, stats included
    st = _check(oracle, idx, corpus, levels, queries[:300], k, MODE_AUTO)
    assert st["last_mode"] == MODE_AUTO
    # a host-buffer call synchronises: inside a capture it is refused, and the capture survives without it
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        with pytest.raises(_native.IcdError) as e:
            idx._lib and _native._check(idx._lib, idx._lib.icd_index_search(idx._h, queries.ctypes.data, 1, k, 0, MODE_AUTO, os_.ctypes.data, oi.ctypes.data, 0,
                                                                   _native._current_stream_ptr(0)))
        assert "captured" in str(e.value)
        idx.search_reweighted(dq[:1], k)
    g2.replay()
    torch.cuda.synchronize()
    idx.close()


def test_pack_winners_carries_int64_ids_as_bit_patterns():
    """icd_pack_winners (row N2: the winners of a rescored batch for the host in one array; reference
    services/multi_diagnosis_service.py:147-176 builds Candidate objects from them): plane 0 holds the int64 ids' BIT PATTERNS,
    so ids beyond 2^53 - a shard's id_base is an arbitrary int64 in the C ABI - come back exact (a double would round them:
    ADVICE r5); the other planes equal plain torch gathers / slices."""
    import torch
    from rag_project_icd10_amd import _native
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    nq, k, kk = 257, 20, 7
    base = (1 << 60) + 12345
    ids = torch.randint(0, 40474, (nq, k), generator=g, device="cuda", dtype=torch.int64) + base
    raw = torch.rand((nq, k), generator=g, device="cuda", dtype=torch.float32)
    adj, enh, vs, hb, boost = (torch.rand((nq, k), generator=g, device="cuda", dtype=torch.float64) for _ in range(5))
    order = torch.argsort(torch.rand((nq, k), generator=g, device="cuda"), dim=1).to(torch.int32)
    order[5, 3:] = -1                                                    # a query with three hits only
    out = _native.pack_winners(order, ids, raw, adj, enh, vs, hb, boost, kk).cpu()
    o = order[:, :kk].long().clamp(min=0)
    assert out.shape == (8, nq, kk)
    got_ids = out[0].contiguous().view(torch.int64)
    assert torch.equal(got_ids, torch.gather(ids, 1, o).cpu()) and int(got_ids.min()) >= base
    assert float((got_ids.double() - got_ids.double().round()).abs().max()) == 0.0 and not torch.equal(got_ids.double().long(), got_ids)   # (a double could not have held them)
    assert torch.equal(out[1], torch.gather(raw, 1, o).double().cpu()) and torch.equal(out[2], torch.gather(adj, 1, o).cpu())
    assert torch.equal(out[3], order[:, :kk].double().cpu())
    for plane, t in zip((4, 5, 6, 7), (enh, vs, hb, boost)):
        assert torch.equal(out[plane], t[:, :kk].cpu())


@pytest.mark.gpu
@pytest.mark.parametrize("nq", [30, 2000])
def test_host_query_arrays_pageable_pinned_and_offset_views_give_the_device_calls_results(oracle, nq):
    """The ABI takes host pointers (include/icd_search.h: `queries_on_device` 0): a large query array goes through the runtime's
    copy - pageable or PINNED, whole or a view that starts in the middle of an allocation - a small one through the handle's pinned
    block; every form gives what the device-resident call gives (bench.py extra.host_buffers prices them)."""
    import torch
    n, dim, k = 20000, 768, 10
    corpus, levels = unit_rows(n, dim, 171), icd_levels(n, 172)
    queries = unit_rows(nq + 5, dim, 173)
    idx = IcdIndex(corpus, levels, max_nq=nq, max_k=10)
    try:
        want = [t.cpu().numpy() for t in idx.search_reweighted(torch.from_numpy(queries[5:]).cuda(), k)]
        os_, oi = oracle.flat_ip_topk(corpus, queries[5:], k)
        w_adj, w_raw, w_ids, w_lv = oracle.reweight(os_, oi, levels)
        assert np.array_equal(want[2], w_ids) and np.array_equal(want[0], w_adj)
        pinned = torch.from_numpy(queries).pin_memory()
        forms = {"pageable": queries[5:].copy(), "pinned": pinned[5:].numpy(), "pageable view": queries[5:]}
        assert forms["pinned"].ctypes.data == pinned.data_ptr() + 5 * dim * 4
        for name, arr in forms.items():
            for _ in range(2):
                got = idx.search_reweighted(arr, k)
                for g, w in zip(got, want):
                    assert np.array_equal(g, w), (name, nq)
    finally:
        idx.close()
