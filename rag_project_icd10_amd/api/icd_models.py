"""Request / response schemas of the /embed and /query surface (pydantic v2).

Field names, defaults and bounds follow the reference's models/icd_models.py: Candidate (:56-87, score
>= 0 at :71), DiagnosisMatch (:90-124), QueryRequest (:135-138, top_k in [1, 50], default 5),
QueryResponse (:141-158), EmbeddingRequest / EmbeddingResponse (:184-192), HealthCheckResponse
(:210-215), convert_numpy_types (:14-37).
"""
from __future__ import annotations

import dataclasses
from typing import Any, List, Optional

import numpy as np
from pydantic import BaseModel, ConfigDict, Field, field_serializer


def convert_numpy_types(obj):
    if isinstance(obj, np.integer):
        return int(obj)
    if isinstance(obj, np.floating):
        return float(obj)
    if isinstance(obj, np.ndarray):
        return obj.tolist()
    if isinstance(obj, dict):
        return {k: convert_numpy_types(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [convert_numpy_types(v) for v in obj]
    if dataclasses.is_dataclass(obj) and not isinstance(obj, type):
        return convert_numpy_types(dataclasses.asdict(obj))
    if hasattr(obj, "__dict__"):
        return {k: convert_numpy_types(v) for k, v in obj.__dict__.items() if not k.startswith("_")}
    return obj


class Candidate(BaseModel):
    model_config = ConfigDict(arbitrary_types_allowed=True)
    code: str = Field(..., description="ICD-10编码")
    title: str = Field(..., description="诊断名称")
    score: float = Field(..., description="相似度分数", ge=0.0)
    level: Optional[int] = Field(default=1, description="ICD层级级别")
    parent_code: Optional[str] = Field(default="", description="父级编码")
    enhanced_score: Optional[float] = Field(default=None, description="增强后的分数")
    original_score: Optional[float] = Field(default=None, description="原始相似度分数")
    similarity_factors: Optional[Any] = Field(default=None, description="相似度计算因子")

    @field_serializer("similarity_factors")
    def _ser_factors(self, value):
        return None if value is None else convert_numpy_types(value)


_CAND_FIELDS = frozenset(("code", "title", "score", "level", "parent_code", "enhanced_score", "original_score", "similarity_factors"))
_CAND_FIELDS_SET = frozenset(_CAND_FIELDS)   # copied into every trusted Candidate of trusted_candidates (its own mutable set, like a validated object's)
_trusted_ok: Optional[bool] = None


def _trusted_candidate_unchecked(code, title, score, enhanced_score, original_score, similarity_factors) -> "Candidate":
    c = Candidate.__new__(Candidate)
    object.__setattr__(c, "__dict__", {"code": code, "title": title, "score": score, "level": 1, "parent_code": "",
                                       "enhanced_score": enhanced_score, "original_score": original_score,
                                       "similarity_factors": similarity_factors})
    object.__setattr__(c, "__pydantic_fields_set__", set(_CAND_FIELDS))
    object.__setattr__(c, "__pydantic_extra__", None)
    object.__setattr__(c, "__pydantic_private__", None)
    return c


def trusted_candidate(code: str, title: str, score: float, enhanced_score: float, original_score: float, similarity_factors) -> "Candidate":
    """A Candidate of a live hit (level 1, parent_code "": SURVEY F8) from values that are ALREADY what the validator would
    produce - Python str / float straight from the batched device path - without one validator call per object
    (10 000 of them per 1 000-string request: 33 -> 9 ms). The one rule with teeth is applied here: score >= 0
    (models/icd_models.py:71 of the reference; a negative score must fail like the validated constructor does).
    The first call checks that the object equals the validated constructor's, field for field and in model_dump();
    if this pydantic lays its objects out differently, every call goes through the validated constructor instead."""
    global _trusted_ok
    if score < 0.0 or score != score:
        return Candidate(code=code, title=title, score=score)   # raises the ValidationError the reference's path raises
    if _trusted_ok is None:
        try:
            a = _trusted_candidate_unchecked(code, title, score, enhanced_score, original_score, similarity_factors)
            b = Candidate(code=code, title=title, score=score, level=1, parent_code="", enhanced_score=enhanced_score,
                          original_score=original_score, similarity_factors=similarity_factors)
            _trusted_ok = bool(a == b and a.model_dump() == b.model_dump() and a.model_fields_set == b.model_fields_set)
        except Exception:
            _trusted_ok = False
    if not _trusted_ok:
        return Candidate(code=code, title=title, score=score, level=1, parent_code="", enhanced_score=enhanced_score,
                         original_score=original_score, similarity_factors=similarity_factors)
    return _trusted_candidate_unchecked(code, title, score, enhanced_score, original_score, similarity_factors)


def trusted_candidates(recs, ids, scores, originals, factors) -> List["Candidate"]:
    """trusted_candidate for one hit list in ONE call (the per-object Python call is most of what is left of the cost):
    recs: the corpus records; ids / scores / originals / factors: parallel sequences of a query's winners. A negative or NaN
    score raises the ValidationError of the validated constructor (the caller degrades the match to an empty one)."""
    if _trusted_ok is not True:   # (not yet checked, or this pydantic differs: the checked path, one object at a time)
        return [trusted_candidate(recs[i].get("code", ""), recs[i].get("preferred_zh", ""), s, s, o, f)
                for i, s, o, f in zip(ids, scores, originals, factors)]
    out = []
    new, setattr_, fields_set = Candidate.__new__, object.__setattr__, _CAND_FIELDS_SET
    for i, s, o, f in zip(ids, scores, originals, factors):
        if not s >= 0.0:
            Candidate(code="", title="", score=s)   # raises
        rec = recs[i]
        c = new(Candidate)
        setattr_(c, "__dict__", {"code": rec.get("code", ""), "title": rec.get("preferred_zh", ""), "score": s, "level": 1,
                                 "parent_code": "", "enhanced_score": s, "original_score": o, "similarity_factors": f})
        setattr_(c, "__pydantic_fields_set__", set(fields_set))   # (its own set: model_fields_set is mutable per object)
        setattr_(c, "__pydantic_extra__", None)
        setattr_(c, "__pydantic_private__", None)
        out.append(c)
    return out


def trusted_matches_ready() -> bool:
    """True once trusted_candidate has checked, on a real object, that this pydantic lays its models out the way the unchecked
    constructors assume"""
    return _trusted_ok is True


_SHARED_CAND_FIELDS = set(_CAND_FIELDS)   # ONE set for every bulk-made Candidate: all eight fields are in it, so pydantic's own
                                          # `fields_set.add(name)` on assignment never changes it; model_copy() copies it


try:   # the same loop in C (csrc/fastobj.c, built by `make -C rag_project_icd10_amd/csrc`): optional
    from .. import _fastobj
except ImportError:   # pragma: no cover - a tree without the built module
    _fastobj = None


def bulk_candidates(codes, titles, factors_cls, ids, scores, originals, vs, hb, sc: float, cr: float) -> List["Candidate"]:
    """The Candidate objects of ONE query's winners with their SimilarityFactors, in one loop without a call per object (row N2:
    the batched request path makes 10 000 of each per 1 000 strings - this loop is most of what that path costs the host).
    codes / titles: the corpus' columns by row; ids / scores / originals / vs / hb: the winners' parallel sequences (live hits:
    level 1, parent "", no entity match, no category alignment - SURVEY F8). The caller has checked trusted_matches_ready().
    A negative or NaN score raises the validated constructor's ValidationError (reference models/icd_models.py:71)."""
    if _fastobj is not None:
        try:
            return _fastobj.bulk_candidates(Candidate, factors_cls, _SHARED_CAND_FIELDS, codes, titles, ids, scores, originals, vs, hb, sc, cr)
        except ValueError:   # (a negative / NaN score: the loop below lets the validated constructor raise what the reference raises)
            pass
        except TypeError:    # (sequences that are not lists: the Python loop takes anything indexable)
            pass
    out = []
    new, onew, setattr_, shared = Candidate.__new__, object.__new__, object.__setattr__, _SHARED_CAND_FIELDS
    for j in range(len(ids)):
        s = scores[j]
        if not s >= 0.0:
            Candidate(code="", title="", score=s)   # raises
        f = onew(factors_cls)
        f.__dict__ = {"vector_similarity": vs[j], "hierarchy_boost": hb[j], "entity_match_score": 0.0, "semantic_coherence": sc,
                      "category_alignment": 0.0, "context_relevance": cr}
        i = ids[j]
        c = new(Candidate)
        setattr_(c, "__dict__", {"code": codes[i], "title": titles[i], "score": s, "level": 1, "parent_code": "",
                                 "enhanced_score": s, "original_score": originals[j], "similarity_factors": f})
        setattr_(c, "__pydantic_fields_set__", shared)
        setattr_(c, "__pydantic_extra__", None)
        setattr_(c, "__pydantic_private__", None)
        out.append(c)
    return out


_trusted_match_ok: Optional[bool] = None
_setattr = object.__setattr__


def _new_match(cls):
    return cls.__new__(cls)


def trusted_match(diagnosis_text: str, candidates: List["Candidate"], match_confidence: float, confidence_factors=None) -> "DiagnosisMatch":
    """DiagnosisMatch(diagnosis_text=, candidates=, match_confidence=, confidence_factors=) from values that are already what its
    validator would produce (a str, a list of Candidate objects, a float in [0, 1]); checked once against the validated
    constructor, like trusted_candidate"""
    global _trusted_match_ok
    if _trusted_match_ok and 0.0 <= match_confidence <= 1.0:
        m = _new_match(DiagnosisMatch)
        _setattr(m, "__dict__", {"diagnosis_text": diagnosis_text, "candidates": candidates, "match_confidence": match_confidence,
                                 "confidence_metrics": None, "confidence_factors": confidence_factors, "confidence_level": None})
        _setattr(m, "__pydantic_fields_set__", {"diagnosis_text", "candidates", "match_confidence", "confidence_factors"})
        _setattr(m, "__pydantic_extra__", None)
        _setattr(m, "__pydantic_private__", None)
        return m
    b = DiagnosisMatch(diagnosis_text=diagnosis_text, candidates=candidates, match_confidence=match_confidence, confidence_factors=confidence_factors)   # (raises outside [0, 1])
    if _trusted_match_ok is None:
        try:
            _trusted_match_ok = True
            a = trusted_match(diagnosis_text, candidates, match_confidence, confidence_factors)
            _trusted_match_ok = bool(a == b and a.model_dump() == b.model_dump() and a.model_fields_set == b.model_fields_set)
        except Exception:
            _trusted_match_ok = False
    return b


class DiagnosisMatch(BaseModel):
    model_config = ConfigDict(arbitrary_types_allowed=True)
    diagnosis_text: str = Field(..., description="提取的诊断文本")
    candidates: List[Candidate] = Field(..., description="匹配的候选结果")
    match_confidence: float = Field(..., description="整体匹配置信度", ge=0.0, le=1.0)
    confidence_metrics: Optional[Any] = Field(default=None)
    confidence_factors: Optional[Any] = Field(default=None)
    confidence_level: Optional[str] = Field(default=None)

    @field_serializer("confidence_metrics", "confidence_factors")
    def _ser_conf(self, value):
        return None if value is None else convert_numpy_types(value)


class QueryRequest(BaseModel):
    text: str = Field(..., description="输入的诊断文本", min_length=1)
    top_k: int = Field(default=5, description="返回候选数量", ge=1, le=50)


class QueryResponse(BaseModel):
    model_config = ConfigDict(arbitrary_types_allowed=True)
    candidates: List[Candidate] = Field(..., description="候选结果列表")
    is_multi_diagnosis: bool = Field(default=False)
    extracted_diagnoses: List[str] = Field(default_factory=list)
    diagnosis_matches: List[DiagnosisMatch] = Field(default_factory=list)


class EmbeddingRequest(BaseModel):
    texts: List[str] = Field(..., description="要向量化的文本列表")


class EmbeddingResponse(BaseModel):
    embeddings: List[List[float]] = Field(..., description="向量列表")
    model: str = Field(..., description="使用的模型名称")


class HealthCheckResponse(BaseModel):
    status: str
    milvus_connected: bool
    embedding_model_loaded: bool
    total_records: int
