"""MilvusService.search / insert_records against the fixture produced by RUNNING the reference's own
services/milvus_service.py (tests/golden/make_milvus_golden.py: the reference's Python over a labelled
stand-in for the Milvus FLAT/IP engine). Pins, bit for bit (Python doubles included): the raw-top-k ->
x level weight -> stable re-sort order, the result-dict shape, [] on a missing collection and on an
engine exception, None -> "" and the defaults of insert_records, ValueError on a length mismatch.

CPU: oracle/icd_oracle.c (+ our dict builder) equals the fixture -> the oracle is pinned by the reference.
GPU: rag_project_icd10_amd.services.milvus_service.MilvusService.search on the HIP index equals the fixture.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def gold():
    cases = json.load(open(os.path.join(GOLDEN, "milvus_search_cases.json"), encoding="utf-8"))
    vec = np.load(os.path.join(GOLDEN, "milvus_search_vectors.npz"))
    return cases, {k: vec[k] for k in vec.files}


@pytest.fixture(scope="module")
def gold_batch():
    out = json.load(open(os.path.join(GOLDEN, "milvus_batch_cases.json"), encoding="utf-8"))["out"]
    return out, np.load(os.path.join(GOLDEN, "milvus_batch_vectors.npz"))["batch_queries"]


class _DimProbe:
    def __init__(self, dim):
        self.dim = dim

    def encode_query(self, text):
        return np.zeros(self.dim, np.float32)


def _service(tmp, gold, collection="icd10", fill=True):
    cases, vec = gold
    os.environ["MILVUS_MODE"] = "local"
    os.environ["MILVUS_DB_PATH"] = str(tmp)
    os.environ["MILVUS_COLLECTION_NAME"] = collection
    from rag_project_icd10_amd.services.milvus_service import MilvusService
    svc = MilvusService(_DimProbe(cases["dimension"]))
    if fill:
        recs, corpus = cases["records"], vec["corpus"]
        for b in range(0, len(recs), 128):   # the reference's insert batches (tools/build_database.py:183-192)
            assert svc.insert_records(recs[b:b + 128], [corpus[i] for i in range(b, min(b + 128, len(recs)))]) is True
    return svc


def test_insert_records_rows_match_the_reference(tmp_path, gold):
    cases, vec = gold
    svc = _service(tmp_path, gold)
    assert svc.client.count == len(cases["inserted_rows"])
    for ours, ref in zip(svc.client.records, cases["inserted_rows"]):
        assert ours == ref                                   # None -> "", defaults of .get(), field names
    with pytest.raises(ValueError) as e:
        svc.insert_records(cases["records"][:2], [vec["corpus"][0]])
    assert "ValueError: " + str(e.value) == cases["insert_length_mismatch"]
    assert svc.insert_records(cases["records"][:1], [[0.0] * cases["dimension"]]) is cases["insert_list_embedding"]
    # reloading the store from disk gives the same rows back
    from rag_project_icd10_amd.corpus_store import CorpusStore
    st = CorpusStore.open(str(tmp_path), "icd10", cases["dimension"])
    assert st.records == cases["inserted_rows"]
    assert np.array_equal(st.matrix(), vec["corpus"])


def test_level_weights_match_the_reference(tmp_path, gold, oracle):
    cases, _ = gold
    svc = _service(tmp_path, gold, fill=False)
    for lv, w in cases["level_weights"].items():
        assert svc._calculate_level_weight(int(lv)) == w
        assert oracle.level_weight(int(lv)) == w
        assert oracle.lib().icd_oracle_level_weight(int(lv)) == w


def test_oracle_and_dict_builder_equal_the_reference_output(tmp_path, gold, oracle):
    """oracle.flat_ip_topk + oracle.reweight (the checker of every GPU parity test) -> our dict builder == fixture."""
    cases, vec = gold
    svc = _service(tmp_path, gold)
    levels = svc.client.levels()
    assert np.array_equal(levels, np.asarray([r["level"] for r in cases["inserted_rows"]], np.int32))
    for case in cases["cases"]:
        q = vec["queries"][case["query_index"]]
        raw, ids = oracle.flat_ip_topk(vec["corpus"], q[None], case["top_k"])
        adj, oraw, oid, _ = oracle.reweight(raw, ids, levels)
        got = svc._hits_to_dicts(adj[0], oraw[0], oid[0])
        assert got == case["out"], (case["query_index"], case["top_k"])
        for h in got:                                        # types of the reference's dict (floats are Python floats)
            assert type(h["score"]) is float and type(h["original_score"]) is float
    # n < k: every row comes back, nothing padded
    tiny = cases["tiny"]
    svc_t = _service(tmp_path / "t", gold, collection="tiny", fill=False)
    assert svc_t.insert_records(tiny["records"], [vec["tiny"][i] for i in range(3)]) is True
    raw, ids = oracle.flat_ip_topk(vec["tiny"], vec["queries"][tiny["query_index"]][None], tiny["top_k"])
    adj, oraw, oid, _ = oracle.reweight(raw, ids, svc_t.client.levels())
    assert svc_t._hits_to_dicts(adj[0], oraw[0], oid[0]) == tiny["out"]


def test_oracle_equals_the_reference_on_the_batch_fixture(tmp_path, gold, gold_batch, oracle):
    """the 160-query batch (the reference's search looped): oracle + dict builder == the reference's dicts, k = 10 and 20"""
    _, vec = gold
    out, batch = gold_batch
    svc = _service(tmp_path, gold)
    levels = svc.client.levels()
    for k in (10, 20):
        raw, ids = oracle.flat_ip_topk(vec["corpus"], batch, k)
        adj, oraw, oid, _ = oracle.reweight(raw, ids, levels)
        for q in range(len(batch)):
            assert svc._hits_to_dicts(adj[q], oraw[q], oid[q]) == out[str(k)][q], (q, k)


def test_error_paths_return_empty_lists_like_the_reference(tmp_path, gold, monkeypatch):
    cases, vec = gold
    svc = _service(tmp_path, gold)
    svc.client.drop()                                        # collection gone (milvus_service.py:275-277)
    assert svc.search(vec["queries"][1], 5) == cases["missing_collection"] == []
    svc2 = _service(tmp_path / "e", gold)

    def boom():
        raise RuntimeError("engine failure")
    monkeypatch.setattr(svc2, "_ready_index", boom)          # engine exception (milvus_service.py:318-320)
    assert svc2.search(vec["queries"][0], 5) == cases["engine_exception"] == []


@pytest.mark.gpu
def test_hip_search_equals_the_reference_output(tmp_path, gold):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    cases, vec = gold
    svc = _service(tmp_path, gold)
    for case in cases["cases"]:
        q = vec["queries"][case["query_index"]]
        assert svc.search(q, case["top_k"]) == case["out"], (case["query_index"], case["top_k"])
    assert svc.search(vec["queries"][0]) == [c for c in cases["cases"] if c["query_index"] == 0 and c["top_k"] == 10][0]["out"]
    # the additive batch entry point returns the same dicts, one list per query
    for k in (1, 5, 10):
        want = {c["query_index"]: c["out"] for c in cases["cases"] if c["top_k"] == k}
        got = svc.search_batch(vec["queries"], k, as_dicts=True)
        for qi, out in want.items():
            assert got[qi] == out, (qi, k)
    tiny = cases["tiny"]
    svc_t = _service(tmp_path / "t", gold, collection="tiny", fill=False)
    assert svc_t.insert_records(tiny["records"], [vec["tiny"][i] for i in range(3)]) is True
    assert svc_t.search(vec["queries"][tiny["query_index"]], tiny["top_k"]) == tiny["out"]
    svc_t.client.drop()
    assert svc_t.search(vec["queries"][1], 5) == cases["missing_collection"]


@pytest.mark.gpu
def test_hip_mfma_path_equals_the_reference_output_on_a_batch(tmp_path, gold, gold_batch):
    """The reference's own output against the fp16-MFMA coarse kernel + certification + canonical rescoring: 160 queries in
    one search_batch call take ICD_MODE_AUTO's coarse path (batches of <= 16 take the streaming kernel), among them the
    designed duplicate-row / equal-adjusted-score / all-zero queries, which must come back through the exact fallback
    with the same dicts. Host and device inputs."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from rag_project_icd10_amd._native import MODE_AUTO
    out, batch = gold_batch
    svc = _service(tmp_path, gold)
    for k in (10, 20):
        for q in (batch, torch.from_numpy(batch).cuda()):
            got = svc.search_batch(q, k, as_dicts=True)
            st = svc._ready_index().stats()
            assert st["last_mode"] == MODE_AUTO and st["last_nq"] == len(batch)      # the coarse MFMA pass ran
            assert st["last_fallback"] < len(batch) // 4                             # ... and certified most of the batch
            assert len(got) == len(batch)
            for qi in range(len(batch)):
                assert got[qi] == out[str(k)][qi], (qi, k)
