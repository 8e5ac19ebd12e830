"""Host-side logic against fixtures produced by running the reference itself (tests/golden/make_golden.py)."""
import dataclasses
import hashlib
import json
import os
import warnings

import numpy as np
import pytest

from conftest import GOLDEN

from rag_project_icd10_amd.services.hierarchical_similarity_service import (HierarchicalSimilarityService,
                                                                             SimilarityFactors)
from rag_project_icd10_amd.services.uncertainty_diagnosis_service import UncertaintyDiagnosisService
from rag_project_icd10_amd.tools.build_database import DatabaseBuilder
from rag_project_icd10_amd.tools.text_processor import DiagnosisTextProcessor

REF_CSV = "/root/reference/data/ICD_10v601.csv"


def _load(name):
    with open(os.path.join(GOLDEN, name), encoding="utf-8") as f:
        return json.load(f)


def test_csv_slice_records_match_reference():
    recs = DatabaseBuilder().load_csv_data(os.path.join(GOLDEN, "csv_slice.csv"))
    gold = _load("csv_records.json")
    assert recs == gold
    by_code = {r["code"]: r for r in recs}
    a = by_code["A00.001"]
    assert (a["level"], a["parent_code"], a["category_path"]) == (3, "A00.0", "A00 > A00.0 > A00.001")
    assert a["semantic_text"] == "古典生物型霍乱 | 霍乱 | 霍乱,由于01群霍乱弧菌,霍乱生物型所致 | ICD-10: A00.001"
    combo = [r for r in recs if r["has_complication"]]
    assert combo and all("+" in r["code"] and r["secondary_code"] and "*" not in r["secondary_code"] for r in combo)
    assert any(r["code"].startswith("M8") and r["level"] == 1 for r in recs)   # morphology codes: no dot -> level 1


@pytest.mark.skipif(not os.path.exists(REF_CSV), reason="full reference CSV only exists in the build container")
def test_full_csv_digest_matches_reference():
    d = _load("csv_full_digest.json")
    b = DatabaseBuilder()
    full = b.load_csv_data(REF_CSV)
    assert len(full) == d["count"] == 40474
    hist = {}
    for r in full:
        hist[str(r["level"])] = hist.get(str(r["level"]), 0) + 1
    assert hist == d["level_histogram"] == {"1": 5031, "2": 12106, "3": 23337}
    assert sum(r["has_complication"] for r in full) == d["has_complication"] == 1000
    for key, field in (("sha256_semantic_text", "semantic_text"), ("sha256_codes", "code"), ("sha256_parent_codes", "parent_code")):
        assert hashlib.sha256("\n".join(r[field] for r in full).encode()).hexdigest() == d[key]
    assert full[0] == d["first"] and full[2] == d["third"]


def test_batch_size_rule():
    b = DatabaseBuilder()
    for n, want in _load("csv_full_digest.json")["batch_size_rule"].items():
        assert b._calculate_optimal_batch_size(int(n)) == want


def test_hierarchy_parser_cases():
    b = DatabaseBuilder()
    assert b._parse_hierarchy("A00", {}) == (1, "", "A00")
    assert b._parse_hierarchy("A00.0", {}) == (2, "A00", "A00 > A00.0")
    assert b._parse_hierarchy("A00.001", {}) == (3, "A00.0", "A00 > A00.0 > A00.001")
    assert b._parse_hierarchy("A01.003+G01*", {}) == (3, "A01.0", "A01 > A01.0 > A01.003+G01*")
    assert b._parse_hierarchy("B95.61", {}) == (3, "B95", "B95 > B95.61")
    assert b._parse_hierarchy("M800000/0", {}) == (1, "", "M800000/0")
    assert b._build_semantic_text("X1.2", "甲", "X1 > X1.2", {"X1": "甲"}) == "甲 | ICD-10: X1.2"  # duplicate ancestor name dropped


class _FakeEmbedder:
    def __init__(self):
        self.calls = 0

    def encode_query(self, text):
        self.calls += 1
        h = hashlib.sha256(("query: " + text).encode()).digest()
        v = np.frombuffer(h[:16], dtype=np.uint8).astype(np.float32) - 127.5
        return v / np.linalg.norm(v)


def test_hierarchical_rescoring_matches_reference():
    cases = _load("hier_cases.json")
    assert len(cases) == 30
    for c in cases:
        emb = _FakeEmbedder() if c["with_embedder"] else None
        svc = HierarchicalSimilarityService(embedding_service=emb)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = svc.batch_calculate_similarities(c["query"], c["entities"], json.loads(json.dumps(c["candidates"])))
        got = [{"code": r.get("code"), "enhanced_score": float(s), "record_enhanced_score": r.get("enhanced_score"),
                "original_score": r.get("original_score"), "score": r.get("score"),
                "uncertainty_boost": r.get("uncertainty_boost"), "uncertainty_weight": r.get("uncertainty_weight"),
                "factors": dataclasses.asdict(f)} for r, s, f in res]
        assert json.loads(json.dumps(got, default=float)) == c["out"], (c["set"], c["query"])
        assert (emb.calls if emb else 0) == c["encode_calls"]


def test_live_shaped_record_known_answer():
    """SURVEY.md section 8c known answer: live-shaped I21.9 hit -> 1.719, factors (0.68, 0.089, 0, 0.3, 0.95, 0)."""
    svc = HierarchicalSimilarityService(embedding_service=_FakeEmbedder())
    rec = {"code": "I21.9", "title": "急性心肌梗死，未特指", "score": 0.68, "original_score": 0.85,
           "metadata": {"level": 3, "parent_code": "I21", "semantic_text": "x"}}
    score, f = svc.calculate_enhanced_similarity("急性心肌梗死", {"disease": [{"text": "急性心肌梗死", "confidence": 0.95}]}, rec)
    assert abs(score - 1.719) < 1e-9
    assert (f.vector_similarity, round(f.hierarchy_boost, 3), f.entity_match_score, f.semantic_coherence,
            f.category_alignment, f.context_relevance) == (0.68, 0.089, 0.0, 0.3, 0.95, 0.0)
    assert isinstance(SimilarityFactors(np.float32(0.5)).vector_similarity, float)


def test_uncertainty_matches_reference():
    cases = _load("uncertainty_cases.json")
    svc = UncertaintyDiagnosisService()
    for c in cases["detect"]:
        assert svc.detect_uncertainty(c["text"]) == c["out"], c["text"]
    for c in cases["process"]:
        clean, out = svc.process_uncertainty_query(c["text"], json.loads(json.dumps(c["candidates"])))
        assert clean == c["clean_query"] and out == c["out"], c["text"]


def test_text_split_matches_reference():
    tp = DiagnosisTextProcessor()
    for c in _load("text_split_cases.json"):
        assert tp._extract_diagnoses_simple(c["text"]) == c["out"], c["text"]
        assert tp.is_multi_diagnosis(c["text"]) == c["multi"]
    assert tp.get_processing_mode() == "simple"
    assert tp.extract_diagnoses("   ") == []


def test_device_rescoring_host_inputs_equal_the_per_candidate_methods():
    """The per-query numbers and per-row tags the device-side rescoring consumes (row N2) are exactly what the golden-pinned
    per-candidate methods compute for live-shaped hits; the kernel's arithmetic itself is checked bit for bit on the GPU
    (tests/test_gpu_parity.py::test_config2_shape_batched_equals_one_at_a_time)."""
    from rag_project_icd10_amd.services.hierarchical_similarity_service import HierarchicalSimilarityService as H
    hs = H(embedding_service=object())
    strings = [l.strip() for l in open(os.path.join(GOLDEN, "diagnosis_strings.txt"), encoding="utf-8")][:400]
    strings += ["待查", "？", " 疑似 ", "肺炎待查", "高血压 糖尿病 肿瘤 感染", "", "Possible 肺炎?", "不能排除恶性肿瘤，待确诊"]
    for q in strings:
        found = hs.uncertainty_service.detect_uncertainty(q)
        clean = found["clean_text"] if found["has_uncertainty"] else q
        want = [float(found["uncertainty_weight"]) if found["has_uncertainty"] else 0.0,
                float(hs._calculate_context_relevance(clean, {})), 1.0 if clean.strip() == "" else 0.0]
        want += [float(hs._calculate_category_semantic_boost(clean, {}, hs.main_categories[c])) for c in H.CHAPTER_ORDER]
        assert hs.query_params(q) == want, q
    assert H.CHAPTER_ORDER == ("A", "B", "C", "E", "I", "J", "K", "N", "S")
    assert H.row_tag("I21.9") == 4 | 0x80 and H.row_tag("I21.900") == 4 | 0x80 and H.row_tag("I21.91") == 4 | 0x80
    assert H.row_tag("I21") == 4 and H.row_tag("Z99.9") == 15 | 0x80 and H.row_tag("") == 15 and H.row_tag("S06.2") == 8
    assert H.row_tag("A01.003+G01*") == 0 and H.row_tag("M800000/0") == 15
    w = hs.device_weights()
    assert w == [0.20, 0.15, 0.08, 0.04, 0.03, 0.3, 0.15 * 0.3] and H().device_weights()[5] == 0.5


def test_query_params_closed_form_fast_path_equals_the_general_path():
    """HierarchicalSimilarityService.query_params decides the common case (no uncertainty marker, no chapter keyword) in closed
    form: the same twelve numbers as the general path (the per-candidate methods of the reference,
    services/hierarchical_similarity_service.py:293-328,448-473, against a live-shaped hit) on the 1 000 golden strings and the
    edge cases"""
    import re
    from rag_project_icd10_amd.services.hierarchical_similarity_service import HierarchicalSimilarityService
    hs = HierarchicalSimilarityService()
    strings = [l.strip() for l in open(os.path.join(GOLDEN, "diagnosis_strings.txt"), encoding="utf-8") if l.strip()]
    strings += ["", " ", "  \t", "x", "待查", "？", " 疑似 ", "肺炎待查", "高血压 糖尿病 肿瘤 感染", "Possible 肺炎?", "ABC def"]
    fast = [hs.query_params(q) for q in strings]
    cls = type(hs)
    saved = cls._kw_any
    cls._kw_any = re.compile("")          # matches everything: no string takes the closed form
    try:
        general = [hs.query_params(q) for q in strings]
    finally:
        cls._kw_any = saved
    assert fast == general
    assert sum(1 for p in fast if p[0] == 0.0 and not any(p[3:])) > 300   # (the closed form is a common case: half of the golden strings carry a marker)


def test_clean_and_weight_equals_detect_uncertainty():
    """UncertaintyDiagnosisService.clean_and_weight (the batched path's form) against detect_uncertainty (the reference's,
    services/uncertainty_diagnosis_service.py:76-125, pinned by tests/golden/uncertainty_cases.json) on every golden string"""
    from rag_project_icd10_amd.services.uncertainty_diagnosis_service import UncertaintyDiagnosisService
    u = UncertaintyDiagnosisService()
    strings = [l.strip() for l in open(os.path.join(GOLDEN, "diagnosis_strings.txt"), encoding="utf-8") if l.strip()]
    strings += ["", "待查", "？?", " 疑似 肺炎 待查 ", "Possible 肺炎?", "不能排除 肿瘤，考虑 炎症。", "性质待定待定", "排除排除"]
    for s in strings:
        d = u.detect_uncertainty(s)
        assert u.clean_and_weight(s) == (d["clean_text"], d["uncertainty_weight"]), s
