"""The ENHANCED text mode - entities fused with semantic boundaries, the reference's default way of cutting a /query text into
diagnoses (services/enhanced_text_processor.py, services/semantic_boundary_service.py, tools/text_processor.py:35-85) - against
fixtures made by RUNNING the reference's classes (tests/golden/make_text_enhanced_golden.py: rule-based NER, a bag-of-characters
stand-in for the embedding service). Here the boundary detector embeds a text's segments in ONE batch where the reference makes
(3 S - 2) one-string calls."""
import json
import os

import numpy as np
import pytest

from rag_project_icd10_amd.services.enhanced_text_processor import EnhancedTextProcessor
from rag_project_icd10_amd.services.medical_ner_service import MedicalNERService
from rag_project_icd10_amd.services.semantic_boundary_service import SemanticBoundaryDetector
from rag_project_icd10_amd.tools.text_processor import DiagnosisTextProcessor

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "text_enhanced_cases.json"), encoding="utf-8"))["cases"]


def bag_of_characters(text, dim=48):
    v = np.zeros(dim, np.float64)
    for ch in text:
        v += np.random.default_rng(ord(ch)).standard_normal(dim)
    n = float(np.sqrt(np.dot(v, v)))
    return (v / n if n > 0 else v + 1.0 / np.sqrt(dim)).astype(np.float32)


class OneAtATime:
    """the reference's embedding interface: encode_query only"""

    def __init__(self):
        self.calls = 0

    def encode_query(self, text):
        self.calls += 1
        return bag_of_characters(text)


class Batched(OneAtATime):
    """... and the batch call EmbeddingService has here (one call per text's segments)"""

    def __init__(self):
        super().__init__()
        self.batches = []

    def encode_query_batch(self, texts):
        self.batches.append(list(texts))
        return np.stack([bag_of_characters(t) for t in texts])


def tuples(x):
    return [tuple(b) for b in x]


@pytest.mark.parametrize("service", [OneAtATime, Batched])
def test_boundaries_and_their_confidences_are_the_references(service):
    emb = service()
    det = SemanticBoundaryDetector(emb)
    multi = 0
    for c in CASES:
        b = det.detect_diagnosis_boundaries(c["text"])
        assert b == tuples(c["boundaries"]), c["text"]
        assert det.get_boundary_confidence(b) == c["confidences"], c["text"]
        s = det.analyze_text_structure(c["text"])
        assert s == c["structure"], c["text"]
        multi += len(b) > 1
    assert multi >= 20
    # both sides of the 0.75 threshold are in the fixture: a clarity bonus somewhere, none somewhere else
    plain_rule = SemanticBoundaryDetector(None)
    with_bonus = without_bonus = 0
    for c in CASES:
        rule = plain_rule.get_boundary_confidence(tuples(c["boundaries"]))
        for r, got in list(zip(rule, c["confidences"]))[:-1]:
            if r < 1.0:                       # (a rule score of 1.0 hides the bonus behind the clamp)
                with_bonus += got > r
                without_bonus += got == r
    assert with_bonus >= 10 and without_bonus >= 1, (with_bonus, without_bonus)


def test_a_texts_segments_are_embedded_in_one_batch_where_the_reference_makes_3s_minus_2_calls():
    emb = Batched()
    det = SemanticBoundaryDetector(emb)
    for c in CASES:
        emb.batches.clear()
        emb.calls = 0
        det._vectors.clear()
        b = det.detect_diagnosis_boundaries(c["text"])
        det.get_boundary_confidence(b)
        s = len(b)
        assert emb.calls == 0
        if c["reference_encode_calls"]:
            assert c["reference_encode_calls"] >= 3 * s - 2 and s > 1
            assert 1 <= len(emb.batches) <= 2 and sum(len(x) for x in emb.batches) <= 2 * s   # (merged segments are embedded as merged: a second batch)
        else:
            assert not emb.batches


def test_without_an_embedding_service_the_detector_keeps_the_delimiter_segmentation():
    """The reference's detector joins segment DICTS at :233 when it has no embedding service and raises TypeError for every
    multi-segment text (its enhanced processor never asks it without one). Here: the segmentation it would have joined - the
    grouping is one group per segment with embeddings too - and the rule part of the confidences."""
    det = SemanticBoundaryDetector(None)
    raised = 0
    for c in CASES:
        b = det.detect_diagnosis_boundaries(c["text"])
        if c["boundaries_without_embeddings_raises"]:
            raised += 1
            assert b == tuples(c["boundaries"])
        else:
            assert b == tuples(c["boundaries_without_embeddings"])
            assert det.get_boundary_confidence(b) == c["confidences_without_embeddings"]
    assert raised >= 20


def _same_diagnoses(got, want, text):
    assert len(got) == len(want), (text, [g["text"] for g in got], [w["text"] for w in want])
    for g, w in zip(got, want):
        assert g == w, (text, g, w)


def test_enhanced_extraction_is_the_references():
    ner = MedicalNERService(use_model=False)
    enh = EnhancedTextProcessor(Batched(), ner)
    enh0 = EnhancedTextProcessor(None, ner)
    for c in CASES:
        t = c["text"]
        _same_diagnoses(enh.extract_diagnoses_enhanced(t), c["enhanced"], t)
        _same_diagnoses(enh.extract_diagnoses_enhanced(t, filter_drugs=False), c["enhanced_keep_drugs"], t)
        _same_diagnoses(enh0.extract_diagnoses_enhanced(t), c["enhanced_without_embeddings"], t)
        if t.strip():
            assert enh._simple_boundary_detection(t) == tuples(c["simple_boundaries"]), t
            _same_diagnoses(enh._fallback_extraction(t), c["fallback"], t)
            s = enh.get_processing_summary(t)
            s.pop("ner_info")
            s["entity_types_found"] = sorted(s["entity_types_found"])
            assert s == c["summary"], t
    with pytest.raises(ValueError):
        EnhancedTextProcessor(Batched(), None)


def test_the_text_processor_is_enhanced_with_an_ner_service_and_simple_without():
    ner = MedicalNERService(use_model=False)
    tp = DiagnosisTextProcessor(embedding_service=Batched(), ner_service=ner)
    assert tp.get_processing_mode() == "enhanced"
    for c in CASES:
        t = c["text"]
        assert tp.extract_diagnoses(t) == c["processor_extract"], t
        _same_diagnoses(tp.extract_diagnoses_enhanced(t), c["processor_enhanced"], t)
        assert tp.is_multi_diagnosis(t) == c["processor_is_multi"]
        assert c["processor_mode"] == "enhanced"
    assert DiagnosisTextProcessor(embedding_service=Batched()).get_processing_mode() == "simple"            # no classifier: the simple mode
    assert DiagnosisTextProcessor(Batched(), use_enhanced_processing=False, ner_service=ner).get_processing_mode() == "simple"
    os.environ["USE_ENHANCED_TEXT_PROCESSING"] = "false"
    try:
        assert DiagnosisTextProcessor(Batched(), ner_service=ner).get_processing_mode() == "simple"
    finally:
        del os.environ["USE_ENHANCED_TEXT_PROCESSING"]

    class Broken:
        def extract_medical_entities(self, text, filter_drugs=True):
            raise RuntimeError("no classifier today")

        def get_model_info(self):
            return {}
    # a failing NER service costs the text its entities, not the request: delimiter boundaries at confidence 0.5 (:392-420)
    out = DiagnosisTextProcessor(Batched(), ner_service=Broken()).extract_diagnoses_enhanced("高血压病；糖尿病")
    assert [d["text"] for d in out] == ["高血压病", "糖尿病"] and all(d["metadata"].get("is_fallback") for d in out)


def test_a_request_embeds_its_text_once_when_the_diagnoses_are_the_boundaries():
    """/query in the enhanced mode (MultiDiagnosisService with an NER service): the text's segments are embedded in ONE batch for the
    boundary confidences; a diagnosis that IS a boundary's text is searched with that vector (the canonical batch arithmetic gives
    a string the same bits in any call) - the reference embeds every segment three times and every diagnosis once more
    (services/semantic_boundary_service.py:184-192,290-292; services/multi_diagnosis_service.py:152-153)."""
    from rag_project_icd10_amd.services.multi_diagnosis_service import MultiDiagnosisService

    class Emb(Batched):
        def encode_query_batch(self, texts, batch_size=256, to_device=False):
            return super().encode_query_batch(texts)

    seen = []

    class Milvus:
        def search_batch(self, vectors, top_k, as_dicts=False):
            seen.append(np.asarray(vectors))
            hit = {"code": "I10", "title": "高血压病", "score": 0.7, "original_score": 0.7,
                   "metadata": {"level": 3, "parent_code": "I10", "semantic_text": "高血压病"}}
            return [[dict(hit)] for _ in range(len(vectors))]

    emb = Emb()
    md = MultiDiagnosisService(emb, Milvus(), ner_service=MedicalNERService(use_model=False))
    out = md.match_multiple_diagnoses("2型糖尿病伴有多个并发症；慢性肾功能不全；高血压病", top_k=1)
    assert out["processing_mode"] == "enhanced" and len(out["extracted_diagnoses"]) == 3
    assert emb.calls == 0 and len(emb.batches) == 1 and sorted(emb.batches[0]) == sorted(out["extracted_diagnoses"])
    assert np.array_equal(seen[0], np.stack([bag_of_characters(d) for d in out["extracted_diagnoses"]]))
    # a text whose boundary is cut at its disease entities: the pieces are new strings, embedded in a second batch, the rest reused
    emb.batches.clear()
    seen.clear()
    out = md.match_multiple_diagnoses("高血压病 糖尿病 冠状动脉粥样硬化性心脏病；慢性胃炎", top_k=1)
    assert np.array_equal(seen[0], np.stack([bag_of_characters(d) for d in out["extracted_diagnoses"]]))
    assert sum(len(b) for b in emb.batches) <= len(set(out["extracted_diagnoses"])) + 2


def test_a_one_diagnosis_request_classifies_its_text_once():
    """The usual /query request is ONE diagnosis: the text and the diagnosis are the same string. The reference runs its token classifier
    on it twice (the text, :57 of the enhanced processor; the diagnosis, services/multi_diagnosis_service.py:147-150) and embeds it once;
    here the diagnosis takes the text's entities (same string, same switches) - one forward each."""
    from rag_project_icd10_amd.services.multi_diagnosis_service import MultiDiagnosisService

    class Milvus:
        def search_batch(self, vectors, top_k, as_dicts=False):
            hit = {"code": "I10", "title": "高血压病", "score": 0.7, "original_score": 0.7,
                   "metadata": {"level": 3, "parent_code": "I10", "semantic_text": "高血压病"}}
            return [[dict(hit)] for _ in range(len(vectors))]

    class Emb(Batched):
        def encode_query_batch(self, texts, batch_size=256, to_device=False):
            return super().encode_query_batch(texts)

    ner = MedicalNERService(use_model=False)
    calls = []
    inner = ner.extract_medical_entities_batch
    ner.extract_medical_entities_batch = lambda texts, filter_drugs=True: calls.append(list(texts)) or inner(texts, filter_drugs)
    md = MultiDiagnosisService(Emb(), Milvus(), ner_service=ner)
    out = md.match_multiple_diagnoses("原发性高血压病", top_k=1)
    assert out["extracted_diagnoses"] == ["原发性高血压病"] and calls == [["原发性高血压病"]]
    want = MultiDiagnosisService(Emb(), Milvus())._match_from_hits("原发性高血压病", Milvus().search_batch([0], 2, True)[0], 1,
                                                                MedicalNERService(use_model=False).extract_medical_entities("原发性高血压病"))
    assert [(c.code, c.score, c.similarity_factors) for c in out["matches"][0].candidates] == [(c.code, c.score, c.similarity_factors) for c in want.candidates]
