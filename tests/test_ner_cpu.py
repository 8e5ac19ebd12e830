"""SURVEY.md row N4 on the CPU: MedicalNERService / DiagnosisEntityFilter against the fixture made by running the
reference's own modules, and _TokenClassifier (the restated 'simple' aggregation + the padded batch forward) against the
outputs of transformers' own NER pipeline on the same seeded model (tests/golden/make_ner_golden.py)."""
import json
import os
import tempfile

import numpy as np
import pytest

from rag_project_icd10_amd.services.diagnosis_entity_filter import DiagnosisEntityFilter
from rag_project_icd10_amd.services.medical_ner_service import MedicalNERService, _CharOffsetTokenizer, _TokenClassifier

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    return json.load(open(os.path.join(GOLD, "ner_cases.json"), encoding="utf-8"))


def tiny_model(gold):
    """the fixture's seeded two-layer token classifier and its character vocabulary, rebuilt from the stored weights"""
    import torch
    from transformers import BertConfig, BertForTokenClassification, BertTokenizerFast
    p = gold["pipeline"]
    d = tempfile.mkdtemp()
    with open(os.path.join(d, "vocab.txt"), "w", encoding="utf-8") as f:
        f.write("\n".join(p["vocab"]) + "\n")
    tok = BertTokenizerFast(vocab_file=os.path.join(d, "vocab.txt"), do_lower_case=True, model_max_length=512)
    labels = p["labels"]
    cfg = BertConfig(vocab_size=len(p["vocab"]), max_position_embeddings=512, num_labels=len(labels),
                     id2label=dict(enumerate(labels)), label2id={l: i for i, l in enumerate(labels)}, **p["config"])
    model = BertForTokenClassification(cfg).eval()
    z = np.load(os.path.join(GOLD, "ner_tiny_model.npz"))
    model.load_state_dict({k: torch.from_numpy(z[k]) for k in z.files})
    return model, tok, cfg.id2label


def same_entities(got, want, tol=0.0):
    assert list(got.keys()) == list(want.keys())
    for kind in want:
        assert len(got[kind]) == len(want[kind]), kind
        for a, b in zip(got[kind], want[kind]):
            assert {k: v for k, v in a.items() if k != "confidence"} == {k: v for k, v in b.items() if k != "confidence"}
            assert abs(float(a["confidence"]) - b["confidence"]) <= tol


def test_rules_filter_keywords_summaries_equal_the_reference(gold):
    svc = MedicalNERService(use_model=False)
    for c in gold["reference"]["rules"]:
        t = c["text"]
        same_entities(svc.extract_medical_entities(t, filter_drugs=True), c["filtered"])
        same_entities(svc.extract_medical_entities(t, filter_drugs=False), c["unfiltered"])
        assert svc.identify_diagnosis_keywords(t) == c["keywords"]
        if c["summary"] is not None:
            s = svc.get_entity_summary(t)
            info = s.pop("model_info")
            assert s == c["summary"] and info["use_model"] is False and info["fallback_available"] is True
            st = svc.get_filter_stats(t)
            st.pop("filter_config")
            assert st == c["filter_stats"]
    assert svc.extract_medical_entities_batch([c["text"] for c in gold["reference"]["rules"]]) == \
        [svc.extract_medical_entities(c["text"]) for c in gold["reference"]["rules"]]


def test_filter_switches_equal_the_reference(gold):
    ref = gold["reference"]
    for c in ref["filter"]:
        f = DiagnosisEntityFilter(dict(c["config"]))
        out = f.filter_entities(ref["filter_entities_input"], ref["filter_text"])
        assert out == c["out"], c["config"]
        st = f.get_filter_stats(ref["filter_entities_input"], out)
        st.pop("filter_config")
        assert st == c["stats"]
    assert DiagnosisEntityFilter().filter_entities({}, "x") == {}


def test_conversion_of_classifier_groups_equals_the_reference(gold):
    conv = gold["reference"]["conversion"]
    groups = [dict(g, score=np.float32(g["score"])) for g in conv["groups"]]
    svc = MedicalNERService(use_model=False)
    svc.use_model = True
    svc.ner_pipeline = lambda texts: [groups for _ in texts]
    same_entities(svc._extract_entities_with_model("x"), conv["converted"], tol=1e-7)
    same_entities(svc.extract_medical_entities(conv["text"], filter_drugs=True), conv["filtered"], tol=1e-7)


def test_token_classifier_equals_transformers_pipeline(gold):
    model, tok, id2label = tiny_model(gold)
    clf = _TokenClassifier(model, tok, id2label, "cpu", max_batch=16)
    outs = gold["pipeline"]["outputs"]
    texts = [o["text"] for o in outs]
    batched = clf(texts)                      # padded, length-sorted batches
    single = [clf([t])[0] for t in texts]     # the pipeline's own shape: one unpadded string per forward
    for o, b, s in zip(outs, batched, single):
        for got, tol in ((s, 2e-6), (b, 2e-5)):
            assert [(g["entity_group"], g["word"], g["start"], g["end"]) for g in got] == \
                   [(g["entity_group"], g["word"], g["start"], g["end"]) for g in o["groups"]], o["text"]
            assert all(abs(float(g["score"]) - w["score"]) <= tol for g, w in zip(got, o["groups"]))
    assert sum(len(o["groups"]) for o in outs) > 100


def test_token_classifier_over_packed_tokens_equals_transformers_pipeline(gold, monkeypatch):
    """batches above 32 strings: the encoder runs over packed tokens and the classifier head over the same rows
    (_TokenClassifier._packed): the groups of transformers' own pipeline (the fixture), the scores of the padded forward"""
    model, tok, id2label = tiny_model(gold)
    clf = _TokenClassifier(model, tok, id2label, "cpu", max_batch=256)
    assert clf._packed is not None
    outs = gold["pipeline"]["outputs"]
    texts = [o["text"] for o in outs]
    assert len(texts) > 32
    packed = clf(texts)
    monkeypatch.setattr(clf, "PACK_TOKENS", 120)          # several chunks
    chunked = clf(texts)
    monkeypatch.setattr(clf, "_packed", None)
    padded = clf(texts)
    for o, a, b, c in zip(outs, packed, chunked, padded):
        for got in (a, b):
            assert [(g["entity_group"], g["word"], g["start"], g["end"]) for g in got] == \
                   [(g["entity_group"], g["word"], g["start"], g["end"]) for g in o["groups"]], o["text"]
            assert all(abs(float(g["score"]) - w["score"]) <= 2e-5 for g, w in zip(got, o["groups"]))
            assert all(abs(float(g["score"]) - float(w["score"])) <= 2e-6 for g, w in zip(got, c))


def test_model_load_failure_falls_back_to_the_rules(monkeypatch):
    monkeypatch.setenv("MEDICAL_NER_MODEL", "/nonexistent/ner-model")
    monkeypatch.delenv("ICD_NER_ALLOW_SYNTHETIC", raising=False)
    svc = MedicalNERService()                 # use_model defaults to true: the load fails, the rules take over (:93-100)
    assert svc.use_model is False and svc.ner_pipeline is None and svc.get_model_info()["model_loaded"] is False
    assert "disease" in svc.extract_medical_entities("慢性阻塞性肺疾病急性加重期")


def test_char_offset_tokenizer_offsets():
    tk = _CharOffsetTokenizer(21128, max_len=8)
    ids, offsets, tokens, special = tk.encode("A 肺炎 x" * 3)
    assert len(ids) == 8 and special == [1, 0, 0, 0, 0, 0, 0, 1] and tokens[1:4] == ["a", "肺", "炎"]
    assert offsets[1:4] == [(0, 1), (2, 3), (3, 4)] and tk.join(["肺", "炎"]) == "肺 炎"


def test_request_path_extracts_all_entities_in_one_batch():
    """MultiDiagnosisService with an NER service (rules here): ONE extract_medical_entities_batch per request; every
    diagnosis is rescored with its own entities, like the reference's per-diagnosis calls
    (services/multi_diagnosis_service.py:147-158) - they move category_alignment and the hierarchy boost."""
    from rag_project_icd10_amd.services.multi_diagnosis_service import MultiDiagnosisService

    class Emb:
        def encode_query_batch(self, texts, batch_size=256, to_device=False):
            return np.zeros((len(texts), 4), np.float32)

        def encode_query(self, text):
            return np.zeros(4, np.float32)

    def hits():
        hit = {"code": "I21.9", "title": "急性心肌梗死", "score": 0.8, "original_score": 0.8,
               "metadata": {"level": 3, "parent_code": "I21", "semantic_text": "急性心肌梗死"}}
        return [dict(hit), dict(hit, code="I10", title="高血压病", score=0.7, original_score=0.7)]

    class Milvus:
        def search_batch(self, vectors, top_k, as_dicts=False):
            return [hits() for _ in range(len(vectors))]

    calls = []
    ner = MedicalNERService(use_model=False)
    inner = ner.extract_medical_entities_batch
    ner.extract_medical_entities_batch = lambda texts, filter_drugs=True: calls.append(list(texts)) or inner(texts, filter_drugs)
    md = MultiDiagnosisService(Emb(), Milvus(), ner_service=ner)
    got = md.match_multiple_diagnoses("急性心肌梗死；高血压病", top_k=2)
    # with an NER service the text mode is the reference's default, "enhanced": the TEXT's entities first (one call: they are fused
    # with its boundaries into the diagnoses), then ONE batch for the diagnoses' own
    assert got["processing_mode"] == "enhanced" and got["extraction_metadata"]["extraction_method"] == "enhanced"
    assert sorted(got["extracted_diagnoses"]) == sorted(["急性心肌梗死", "高血压病"])
    assert calls == [["急性心肌梗死；高血压病"], got["extracted_diagnoses"]]
    plain = MultiDiagnosisService(Emb(), Milvus())
    for d, m in zip(got["extracted_diagnoses"], got["matches"]):
        want = plain._match_from_hits(d, hits(), 2, MedicalNERService(use_model=False).extract_medical_entities(d))
        assert [(c.code, c.score, c.similarity_factors) for c in m.candidates] == \
               [(c.code, c.score, c.similarity_factors) for c in want.candidates]
    without = plain.match_multiple_diagnoses("急性心肌梗死；高血压病", top_k=2)
    assert without["processing_mode"] == "simple" and without["extracted_diagnoses"] == ["急性心肌梗死", "高血压病"]
    first = without["extracted_diagnoses"].index(got["extracted_diagnoses"][0])
    assert got["matches"][0].candidates[0].score != without["matches"][first].candidates[0].score
