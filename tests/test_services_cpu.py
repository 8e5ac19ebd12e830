"""Drop-in service contracts that need no GPU: prefixes, shapes, persistence, error behaviour, API surface."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

from rag_project_icd10_amd.corpus_store import CorpusStore


@pytest.fixture(scope="module")
def emb():
    os.environ["EMBEDDING_MODEL_NAME"] = "shibing624/text2vec-base-chinese"
    from rag_project_icd10_amd.services.embedding_service import EmbeddingService
    return EmbeddingService(allow_synthetic=True, device="cpu")


def test_model_load_failure_raises_like_reference():
    from rag_project_icd10_amd.services.embedding_service import EmbeddingService
    with pytest.raises(Exception):
        EmbeddingService(allow_synthetic=False, device="cpu")   # no weights offline -> startup aborts


def test_embedding_prefixes_and_shapes(emb):
    assert emb._prepare_text_for_embedding("霍乱") == "passage: 霍乱"
    assert emb._prepare_text_for_embedding("query: 霍乱") == "query: 霍乱"
    assert emb._prepare_text_for_embedding("passage:x") == "passage:x"
    v = emb.encode_query("急性胃肠炎")
    assert v.shape == (768,) and v.dtype == np.float32 and abs(np.linalg.norm(v) - 1) < 1e-5
    # encode_query == encode of "query: "+text ; encode_single adds "passage: " (not comparable)
    assert np.allclose(v, emb._encode_prepared(["query: 急性胃肠炎"], 1)[0], atol=1e-6)
    assert np.allclose(emb.encode_single("query: 急性胃肠炎"), v, atol=1e-6)
    assert not np.allclose(emb.encode_single("急性胃肠炎"), v, atol=1e-3)
    out = emb.encode_batch(["霍乱", "伤寒", "急性胃肠炎的长一点的描述"], show_progress=False)
    assert isinstance(out, list) and isinstance(out[0], list) and isinstance(out[0][0], float) and len(out[0]) == 768
    assert emb.encode_batch([]) == []
    assert np.allclose(out[1], emb.encode_single("伤寒"), atol=2e-6)           # batching does not change a row
    rec = emb.encode_icd_record({"preferred_zh": " ", "code": "A00"})
    assert np.allclose(rec, emb.encode_single("ICD代码 A00"), atol=1e-6)
    info = emb.get_model_info()
    assert info["loaded"] and info["embedding_dimension"] == 768 and info["synthetic"] is True
    t = emb.test_embedding()
    assert t["success"] and t["embedding_shape"] == (768,) and len(t["sample_values"]) == 5
    qb = emb.encode_query_batch(["霍乱", "急性胃肠炎"])
    assert qb.shape == (2, 768) and np.allclose(qb[1], v, atol=2e-6)


def test_packed_encoder_equals_the_padded_hf_forward(emb, monkeypatch):
    """batches above 32 strings run the BERT encoder over packed tokens (embedding_service._PackedBert): the same
    embeddings as transformers' padded BertModel forward with its attention mask - mean and CLS pooling, any chunking,
    one-token and over-long strings; the attention groups cover every sequence once"""
    from rag_project_icd10_amd.services.embedding_service import _PackedBert
    texts = [l.rstrip("\n") for l in open(os.path.join(GOLDEN, "diagnosis_strings.txt"), encoding="utf-8")][:90]
    texts += ["", " ", "肺", "x" * 128, "高血压" * 60]
    assert emb._packed is not None
    packed = emb.encode_query_batch(texts, batch_size=256)
    monkeypatch.setattr(emb, "PACK_TOKENS", 300)
    chunked = emb.encode_query_batch(texts, batch_size=256)
    monkeypatch.setattr(emb, "_packed", None)
    padded = emb.encode_query_batch(texts, batch_size=256)
    monkeypatch.undo()
    assert packed.shape == (95, 768) and np.max(np.abs(packed - padded)) <= 2e-6 and np.max(np.abs(chunked - padded)) <= 2e-6
    assert np.max(np.abs(np.linalg.norm(packed, axis=1) - 1)) <= 1e-5
    monkeypatch.setattr(emb, "pooling", "cls")
    cls_packed = emb.encode_query_batch(texts, batch_size=256)
    monkeypatch.setattr(emb, "_packed", None)
    monkeypatch.setattr(emb.model, "pooling", "cls")
    assert np.max(np.abs(cls_packed - emb.encode_query_batch(texts, batch_size=256))) <= 2e-6
    monkeypatch.undo()
    assert np.allclose(emb.encode_batch(["query: " + t for t in texts[:40]], show_progress=False), packed[:40], atol=2e-6)   # (/embed: the same forward for more than 32 texts)
    for lengths in ([51, 40, 39, 20, 20, 19, 3], [7], [9] * 40, list(range(128, 2, -1))):
        groups = _PackedBert.plan_groups(lengths)
        assert sum(c for _, c, _ in groups) == len(lengths) and len(groups) <= _PackedBert.MAX_GROUPS
        assert all(lengths[f] == longest and f == sum(c for _, c, _ in groups[:i]) for i, (f, c, longest) in enumerate(groups))


def test_packed_encoder_covers_the_roberta_family():
    """the reference's default checkpoint (intfloat/multilingual-e5-large-instruct) is XLM-R: the same encoder stack with
    positions numbered from padding_idx + 1. The packed forward's hidden states equal transformers' padded forward token for
    token (seeded random-init XLM-R and BERT of a small shape; a decoder or relative positions are refused)"""
    import torch
    from transformers import BertConfig, BertModel, XLMRobertaConfig, XLMRobertaModel
    from rag_project_icd10_amd.services.embedding_service import _PackedBert
    rng = np.random.default_rng(0)
    seqs = sorted([[0] + list(rng.integers(3, 500, int(L))) + [2] for L in rng.integers(1, 40, 50)], key=len, reverse=True)
    width = len(seqs[0])
    for make, pad in ((lambda: XLMRobertaModel(XLMRobertaConfig(vocab_size=500, hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
                                                                  intermediate_size=128, max_position_embeddings=70, pad_token_id=1,
                                                                  type_vocab_size=1), add_pooling_layer=False), 1),
                      (lambda: BertModel(BertConfig(vocab_size=500, hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
                                                    intermediate_size=128, max_position_embeddings=70), add_pooling_layer=False), 0)):
        torch.manual_seed(0)
        model = make().eval()
        assert _PackedBert.supported(model)
        x, (lengths, starts, _, _) = _PackedBert(model).hidden_states(seqs, "cpu")
        ids = torch.full((len(seqs), width), pad)
        mask = torch.zeros((len(seqs), width), dtype=torch.long)
        for r, sq in enumerate(seqs):
            ids[r, :len(sq)] = torch.tensor(sq)
            mask[r, :len(sq)] = 1
        with torch.no_grad():
            ref = model(input_ids=ids, attention_mask=mask).last_hidden_state
        worst = max(float((x[starts[r]:starts[r + 1]] - ref[r, :lengths[r]]).abs().max()) for r in range(len(seqs)))
        assert worst <= 5e-6, (type(model).__name__, worst)
    assert not _PackedBert.supported(BertModel(BertConfig(vocab_size=50, hidden_size=32, num_hidden_layers=1, num_attention_heads=2,
                                                          intermediate_size=64, is_decoder=True)))


def test_corpus_store_roundtrip(tmp_path):
    st = CorpusStore.open(str(tmp_path), "icd10", 8)
    assert not st.exists()
    rows = [{"code": f"A0{i}", "preferred_zh": "甲", "level": 1 + i % 3} for i in range(5)]
    st.append(rows[:2], np.arange(16, dtype=np.float32).reshape(2, 8))
    st.append(rows[2:], np.arange(16, 40, dtype=np.float32).reshape(3, 8))
    st2 = CorpusStore.open(str(tmp_path), "icd10", 8)
    assert st2.count == 5 and st2.matrix().shape == (5, 8) and st2.matrix()[4, 7] == 39
    assert st2.levels().tolist() == [1, 2, 3, 1, 2] and st2.records[3]["code"] == "A03"
    with pytest.raises(ValueError):
        CorpusStore.open(str(tmp_path), "icd10", 16)
    st2.drop()
    assert not CorpusStore.open(str(tmp_path), "icd10", 8).exists()


def test_milvus_service_contracts_without_gpu(tmp_path, monkeypatch):
    import torch
    monkeypatch.setenv("MILVUS_DB_PATH", str(tmp_path / "db"))
    monkeypatch.setenv("MILVUS_COLLECTION_NAME", "t")
    from rag_project_icd10_amd.services.milvus_service import MilvusService

    class Emb:
        def encode_query(self, t):
            return np.ones(64, np.float32) / 8

    svc = MilvusService(Emb())
    assert svc.dimension == 64 and svc.collection_name == "t" and svc.client is not None
    assert MilvusService.__init__.__code__.co_varnames[:2] == ("self", "embedding_service")
    st = svc.get_collection_stats()
    assert st == {"collection_name": "t", "exists": True, "dimension": 64, "num_entities": 0}
    assert svc.search(np.ones(64, np.float32), 5) == []                       # empty collection -> []
    recs = [{"code": "A00", "preferred_zh": "霍乱", "level": 1, "main_code": None, "secondary_code": None}]
    with pytest.raises(ValueError):
        svc.insert_records(recs, [])
    assert svc.insert_records(recs, [[0.0] * 64]) is False                    # list instead of ndarray (reference quirk)
    assert svc.insert_records(recs, [np.ones(64, np.float32)]) is True
    assert svc.get_collection_stats()["num_entities"] == 1
    assert svc.client.records[0]["main_code"] == "" and svc.client.records[0]["secondary_code"] == ""
    assert svc._calculate_level_weight(1) == 1.2 and svc._calculate_level_weight(5) == 1.0
    mem = svc.get_memory_usage()
    assert mem["estimated_memory_mb"] == 1 * 64 * 4 / (1024 * 1024)
    if not torch.cuda.is_available():
        assert svc.search(np.ones(64, np.float32), 5) == []                   # engine unavailable -> [] (never raises)
        assert svc.load_collection() is False
    assert svc.test_connection()["connected"] is True
    hc = svc.health_check()
    assert set(hc) >= {"healthy", "connection", "load_state", "memory_usage", "timestamp"}
    assert svc.disconnect()["success"] and svc.client is None
    assert svc.release_collection() == {"success": False, "message": "客户端未连接"}
    # reopen: rows persisted
    svc2 = MilvusService.__new__(MilvusService)
    svc2.config = svc.config; svc2.collection_name = "t"; svc2.embedding_service = None; svc2.dimension = 64
    svc2.client = None; svc2._index = None; svc2._index_rows = 0
    svc2._connect()
    assert svc2.client.count == 1
    monkeypatch.setenv("MILVUS_MODE", "remote")
    with pytest.raises(ValueError):
        MilvusService(Emb())


def test_api_surface_with_stub_services():
    from fastapi.testclient import TestClient
    from rag_project_icd10_amd.api import app as appmod

    class Emb:
        def encode_batch(self, texts, show_progress=True):
            return [[0.0, 1.0]] * len(texts)

        def encode_query_batch(self, qs, **kw):
            return np.zeros((len(qs), 2), np.float32)

        def get_model_info(self):
            return {"loaded": True, "model_name": "stub"}

    class Mil:
        def search_batch(self, v, k, as_dicts=False):
            hit = lambda c, s: {"code": c, "title": "t" + c, "score": s, "original_score": s, "metadata": {"level": 2}}
            return [[hit("I21.9", 0.9), hit("I21", 0.5), hit("K29.7", -0.1)][:k] for _ in range(len(v))]

        def test_connection(self):
            return {"connected": True}

        def get_collection_stats(self):
            return {"num_entities": 3}

        def disconnect(self):
            return {}

    appmod.install_services(Emb(), Mil())   # the lifespan only builds real services when none are installed
    with TestClient(appmod.app) as client:
        r = client.post("/embed", json={"texts": ["a", "b"]})
        assert r.status_code == 200 and r.json() == {"embeddings": [[0.0, 1.0], [0.0, 1.0]], "model": "stub"}
        r = client.post("/query", json={"text": "高血压，糖尿病", "top_k": 1})
        body = r.json()
        assert r.status_code == 200 and body["is_multi_diagnosis"] and body["extracted_diagnoses"] == ["高血压", "糖尿病"]
        assert len(body["candidates"]) == 1 and len(body["diagnosis_matches"]) == 2
        cand = body["diagnosis_matches"][0]["candidates"][0]
        assert set(cand) == {"code", "title", "score", "level", "parent_code", "enhanced_score", "original_score", "similarity_factors"}
        assert set(cand["similarity_factors"]) == {"vector_similarity", "hierarchy_boost", "entity_match_score",
                                                   "semantic_coherence", "category_alignment", "context_relevance"}
        assert client.post("/query", json={"text": "", "top_k": 5}).status_code == 422      # min_length=1
        assert client.post("/query", json={"text": "x", "top_k": 51}).status_code == 422     # le=50
        assert client.get("/health").json() == {"status": "healthy", "milvus_connected": True,
                                                "embedding_model_loaded": True, "total_records": 3}
        assert client.get("/stats").json()["milvus"] == {"num_entities": 3}
        appmod.install_services(None, None, None)
        r = client.post("/query", json={"text": "x"})
        assert r.status_code == 500 and "服务未就绪" in r.json()["detail"]                     # 503 swallowed into 500 (main.py:361-363)
        assert client.post("/embed", json={"texts": ["a"]}).status_code == 500


def test_sentence_transformers_checkpoint_config_is_honoured(tmp_path):
    """a local sentence-transformers checkpoint: max_seq_length from sentence_bert_config.json, pooling from
    1_Pooling/config.json (mean / cls); any other pooling mode must raise instead of embedding differently"""
    from transformers import BertConfig, BertModel, BertTokenizerFast
    from rag_project_icd10_amd.services.embedding_service import EmbeddingService
    d = tmp_path / "st_model"
    d.mkdir()
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "query", ":", "passage"] + [chr(c) for c in range(0x4e00, 0x4e00 + 200)]
    (d / "vocab.txt").write_text("\n".join(vocab), encoding="utf-8")
    BertTokenizerFast(str(d / "vocab.txt"), do_lower_case=True).save_pretrained(str(d))
    BertModel(BertConfig(vocab_size=len(vocab), hidden_size=32, num_hidden_layers=1, num_attention_heads=2, intermediate_size=64,
                         max_position_embeddings=64), add_pooling_layer=False).save_pretrained(str(d))
    (d / "sentence_bert_config.json").write_text(json.dumps({"max_seq_length": 12, "do_lower_case": False}))
    (d / "modules.json").write_text(json.dumps([{"idx": 0, "name": "0", "path": "", "type": "sentence_transformers.models.Transformer"},
                                                {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"}]))
    (d / "1_Pooling").mkdir()
    pool = {"word_embedding_dimension": 32, "pooling_mode_cls_token": False, "pooling_mode_mean_tokens": True,
            "pooling_mode_max_tokens": False, "pooling_mode_mean_sqrt_len_tokens": False}
    (d / "1_Pooling" / "config.json").write_text(json.dumps(pool))
    os.environ["EMBEDDING_MODEL_NAME"] = str(d)
    try:
        emb = EmbeddingService(allow_synthetic=False, device="cpu")
        info = emb.get_model_info()
        assert info["max_seq_length"] == 12 and info["pooling"] == "mean" and info["synthetic"] is False
        long_text = "".join(chr(0x4e00 + i) for i in range(60))
        assert len(emb._tokenize(["query: " + long_text])[0]) == 12          # truncated at the checkpoint's max_seq_length
        v_mean = emb.encode_query(long_text)
        pool.update(pooling_mode_cls_token=True, pooling_mode_mean_tokens=False)
        (d / "1_Pooling" / "config.json").write_text(json.dumps(pool))
        emb_cls = EmbeddingService(allow_synthetic=False, device="cpu")
        assert emb_cls.get_model_info()["pooling"] == "cls"
        assert not np.allclose(emb_cls.encode_query(long_text), v_mean, atol=1e-4)
        pool.update(pooling_mode_cls_token=False, pooling_mode_max_tokens=True)
        (d / "1_Pooling" / "config.json").write_text(json.dumps(pool))
        with pytest.raises(ValueError):
            EmbeddingService(allow_synthetic=True, device="cpu")              # not even with synthetic weights allowed
    finally:
        os.environ["EMBEDDING_MODEL_NAME"] = "shibing624/text2vec-base-chinese"


def test_corpus_store_survives_a_torn_append(tmp_path):
    """an append that died after extending the data files but before its manifest: the surplus bytes are cut on the next
    load / append, so later rows do not shift against their metadata; a file shorter than committed raises"""
    st = CorpusStore.open(str(tmp_path), "icd10", 4)
    rows = [{"code": f"B{i}", "preferred_zh": "乙", "level": 1 + i % 3} for i in range(6)]
    st.append(rows[:3], np.arange(12, dtype=np.float32).reshape(3, 4))
    d = os.path.join(str(tmp_path), "icd10")
    with open(os.path.join(d, "corpus.f32"), "ab") as f:           # the torn append: two rows of vectors, one level, half a line
        np.full(8, 99, np.float32).tofile(f)
    with open(os.path.join(d, "levels.i32"), "ab") as f:
        np.asarray([7], np.int32).tofile(f)
    with open(os.path.join(d, "meta.jsonl"), "ab") as f:
        f.write(b'{"code": "TORN"')
    st2 = CorpusStore.open(str(tmp_path), "icd10", 4)
    assert st2.count == 3 and os.path.getsize(os.path.join(d, "corpus.f32")) == 3 * 4 * 4
    st2.append(rows[3:], np.arange(12, 24, dtype=np.float32).reshape(3, 4))
    st3 = CorpusStore.open(str(tmp_path), "icd10", 4)
    assert st3.count == 6 and [r["code"] for r in st3.records] == [f"B{i}" for i in range(6)]
    assert np.array_equal(st3.matrix(), np.arange(24, dtype=np.float32).reshape(6, 4))
    assert st3.levels().tolist() == [1, 2, 3, 1, 2, 3]
    with open(os.path.join(d, "corpus.f32"), "r+b") as f:
        f.truncate(5 * 4 * 4)
    with pytest.raises(ValueError):
        CorpusStore.open(str(tmp_path), "icd10", 4)


def test_corpus_store_stale_handle_never_truncates_committed_rows(tmp_path):
    """two handles of one store (build_database next to the API): an append through the handle that has not seen the
    other's rows takes them in first - it must not cut them off as "surplus bytes" """
    rows = [{"code": f"C{i}", "preferred_zh": "丙", "level": 1 + i % 3} for i in range(6)]
    a = CorpusStore.open(str(tmp_path), "icd10", 4)
    a.append(rows[:2], np.arange(8, dtype=np.float32).reshape(2, 4))
    b = CorpusStore.open(str(tmp_path), "icd10", 4)          # second handle, sees 2 rows
    a.append(rows[2:4], np.arange(8, 16, dtype=np.float32).reshape(2, 4))   # committed behind b's back
    b.append(rows[4:], np.arange(16, 24, dtype=np.float32).reshape(2, 4))   # stale handle: reloads, then appends
    assert b.count == 6 and [r["code"] for r in b.records] == [f"C{i}" for i in range(6)]
    c = CorpusStore.open(str(tmp_path), "icd10", 4)
    assert c.count == 6 and np.array_equal(c.matrix(), np.arange(24, dtype=np.float32).reshape(6, 4))
    assert np.array_equal(b.matrix(), c.matrix()) and c.levels().tolist() == [1, 2, 3, 1, 2, 3]
    assert os.path.exists(os.path.join(str(tmp_path), "icd10", ".lock"))




def test_trusted_candidate_equals_the_validated_constructor():
    """the batched path builds its 10 000 Candidates per request without one validator call each: same objects, same
    model_dump, and the one rule with teeth (score >= 0, models/icd_models.py:71 of the reference) still raises"""
    from pydantic import ValidationError
    from rag_project_icd10_amd.api import icd_models
    from rag_project_icd10_amd.api.icd_models import Candidate, DiagnosisMatch, trusted_candidate
    from rag_project_icd10_amd.services.hierarchical_similarity_service import SimilarityFactors, trusted_factors
    f = trusted_factors(0.68, 0.089, 0.0, 0.3, 0.0, 0.95)
    assert f == SimilarityFactors(0.68, 0.089, 0.0, 0.3, 0.0, 0.95) and type(f.vector_similarity) is float
    a = trusted_candidate("I21.9", "急性心肌梗死", 1.719, 1.719, 0.85, f)
    b = Candidate(code="I21.9", title="急性心肌梗死", score=1.719, level=1, parent_code="", enhanced_score=1.719, original_score=0.85,
                  similarity_factors=SimilarityFactors(0.68, 0.089, 0.0, 0.3, 0.0, 0.95))
    assert icd_models._trusted_ok is True                      # this pydantic lays objects out as assumed: the fast path is on
    assert a == b and a.model_dump() == b.model_dump() and a.model_dump_json() == b.model_dump_json()
    m = DiagnosisMatch(diagnosis_text="x", candidates=[a, b], match_confidence=0.9)
    assert m.model_dump()["candidates"][0] == m.model_dump()["candidates"][1]
    for bad in (-0.01, float("nan")):
        with pytest.raises(ValidationError):
            trusted_candidate("A00", "霍乱", bad, bad, 0.1, f)
    # the per-query form (one call per hit list) gives the same objects and keeps the rule
    from rag_project_icd10_amd.api.icd_models import trusted_candidates
    from rag_project_icd10_amd.services.hierarchical_similarity_service import trusted_factors_row
    recs = [{"code": "A00", "preferred_zh": "霍乱"}, {"code": "I21.9"}, {}]
    fr = trusted_factors_row([0.9, 0.8, 0.7], [0.045, 0.06, 0.03], 0.5, 0.25)
    assert fr == [SimilarityFactors(0.9, 0.045, 0.0, 0.5, 0.0, 0.25), SimilarityFactors(0.8, 0.06, 0.0, 0.5, 0.0, 0.25), SimilarityFactors(0.7, 0.03, 0.0, 0.5, 0.0, 0.25)]
    got = trusted_candidates(recs, [2, 0, 1], [1.2, 1.1, 0.0], [0.9, 0.8, 0.7], fr)
    want = [Candidate(code=recs[i].get("code", ""), title=recs[i].get("preferred_zh", ""), score=s, level=1, parent_code="", enhanced_score=s,
                      original_score=o, similarity_factors=ff) for i, s, o, ff in zip([2, 0, 1], [1.2, 1.1, 0.0], [0.9, 0.8, 0.7], fr)]
    assert got == want and [c.model_dump() for c in got] == [c.model_dump() for c in want]
    got[0].score = 2.0                                         # assigning a field of one object leaves the others alone
    assert got[1].score == 1.1 and got[0].model_fields_set == want[0].model_fields_set
    with pytest.raises(ValidationError):
        trusted_candidates(recs, [0, 1], [0.5, -1e-9], [0.1, 0.1], fr[:2])


def test_bulk_candidates_c_loop_equals_python_loop_equals_validated_constructor():
    """the Candidate / SimilarityFactors objects of the batched request path (row N2; reference models/icd_models.py:56-87,
    services/multi_diagnosis_service.py:161-175): csrc/fastobj.c builds them in one C loop - the same objects as the Python loop
    and as the validated constructors, field for field, in model_dump() and in model_fields_set; a negative score raises the
    validator's error either way"""
    import random
    import pydantic
    from rag_project_icd10_amd.api import icd_models as M
    from rag_project_icd10_amd.services.hierarchical_similarity_service import SimilarityFactors
    assert M._fastobj is not None, "rag_project_icd10_amd/_fastobj.so is not built (make -C rag_project_icd10_amd/csrc)"
    M.trusted_candidate("A00", "霍乱", 0.5, 0.5, 0.4, None)            # (the one-time layout check the bulk path relies on)
    assert M.trusted_matches_ready()
    rnd = random.Random(4)
    n = 500
    codes, titles = [f"A{i % 100:02d}.{i % 10}" for i in range(n)], [f"疾病{i}" for i in range(n)]
    for kk in (0, 1, 10, 50):
        ids = [rnd.randrange(n) for _ in range(kk)]
        sc, og, vs, hb = ([rnd.random() for _ in range(kk + 3)] for _ in range(4))   # (longer than ids: only the first kk count)
        got = M.bulk_candidates(codes, titles, SimilarityFactors, ids, sc, og, vs, hb, 0.3, 0.25)
        saved, M._fastobj = M._fastobj, None
        try:
            py = M.bulk_candidates(codes, titles, SimilarityFactors, ids, sc, og, vs, hb, 0.3, 0.25)
        finally:
            M._fastobj = saved
        want = [M.Candidate(code=codes[i], title=titles[i], score=sc[j], level=1, parent_code="", enhanced_score=sc[j], original_score=og[j],
                            similarity_factors=SimilarityFactors(vs[j], hb[j], 0.0, 0.3, 0.0, 0.25)) for j, i in enumerate(ids)]
        assert got == py == want and len(got) == kk
        assert [c.model_dump() for c in got] == [c.model_dump() for c in want]
        assert all(c.model_fields_set == w.model_fields_set and type(c.similarity_factors) is SimilarityFactors for c, w in zip(got, want))
    with pytest.raises(pydantic.ValidationError):
        M.bulk_candidates(codes, titles, SimilarityFactors, [1, 2], [0.5, -0.1], [0.5, 0.5], [0.1, 0.1], [0.1, 0.1], 0.3, 0.25)
    with pytest.raises(pydantic.ValidationError):
        M.bulk_candidates(codes, titles, SimilarityFactors, [1], [float("nan")], [0.5], [0.1], [0.1], 0.3, 0.25)
    m = M.trusted_match("诊断", got, 0.5, None)
    assert m == M.DiagnosisMatch(diagnosis_text="诊断", candidates=got, match_confidence=0.5, confidence_factors=None)
