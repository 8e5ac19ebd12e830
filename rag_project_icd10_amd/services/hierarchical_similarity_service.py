"""Hierarchical rescoring of the search hits (host side; pure string / scalar rules).

Drop-in for the reference's services/hierarchical_similarity_service.py: same class names, constructor
signature, method names and return shapes. Behaviour follows :69-83 (weights), :93-141 (category
table), :143-219 (calculate_enhanced_similarity), :221-473 (six factors), :475-518 (weighted score),
:520-579 (batch_calculate_similarities), :581-639 (explanation / update_weights) and is pinned by
tests/golden/hier_cases.json, produced by running the reference itself. That includes the reference's
observable quirk that live `MilvusService.search` hits keep level / parent_code / semantic_text under
"metadata" and title under "title", so the top-level reads below see defaults.
"""
from __future__ import annotations

import logging
from dataclasses import dataclass
from typing import Any, Dict, List, Tuple

from .uncertainty_diagnosis_service import UncertaintyDiagnosisService

logger = logging.getLogger(__name__)


@dataclass
class SimilarityFactors:
    vector_similarity: float = 0.0
    hierarchy_boost: float = 0.0
    entity_match_score: float = 0.0
    semantic_coherence: float = 0.0
    category_alignment: float = 0.0
    context_relevance: float = 0.0

    def __post_init__(self):
        for name in ("vector_similarity", "hierarchy_boost", "entity_match_score", "semantic_coherence",
                     "category_alignment", "context_relevance"):
            setattr(self, name, float(getattr(self, name)))


def trusted_factors(vs: float, hb: float, em: float, sc: float, ca: float, cr: float) -> SimilarityFactors:
    """SimilarityFactors from six values that are Python floats already (the batched device path's tolist()): the same
    object as SimilarityFactors(...) without the six float() coercions of __post_init__"""
    f = object.__new__(SimilarityFactors)
    f.__dict__ = {"vector_similarity": vs, "hierarchy_boost": hb, "entity_match_score": em, "semantic_coherence": sc,
                  "category_alignment": ca, "context_relevance": cr}
    return f


def trusted_factors_row(vs_row, hb_row, sc: float, cr: float) -> list:
    """trusted_factors for one query's winners in one call (live hits: no entity match, no category alignment)"""
    new, cls = object.__new__, SimilarityFactors
    out = []
    for vs, hb in zip(vs_row, hb_row):
        f = new(cls)
        f.__dict__ = {"vector_similarity": vs, "hierarchy_boost": hb, "entity_match_score": 0.0, "semantic_coherence": sc,
                      "category_alignment": 0.0, "context_relevance": cr}
        out.append(f)
    return out


@dataclass
class HierarchyInfo:
    level: int = 1
    parent_code: str = ""
    category_path: str = ""
    main_category: str = ""
    sub_category: str = ""
    semantic_keywords: List[str] = None

    def __post_init__(self):
        if self.semantic_keywords is None:
            self.semantic_keywords = []


# first letter of the ICD code -> (chapter name, keywords, semantic weight)   (reference :93-141)
_CHAPTERS = {
    "A": ("某些传染病和寄生虫病", ("感染", "传染", "病毒", "细菌", "寄生虫", "真菌"), 1.1),
    "B": ("肿瘤", ("癌", "瘤", "肿瘤", "恶性", "良性", "转移"), 1.2),
    "C": ("血液及造血器官疾病", ("血液", "贫血", "白血病", "出血", "凝血"), 1.0),
    "E": ("内分泌、营养和代谢疾病", ("糖尿病", "甲状腺", "代谢", "内分泌", "营养"), 1.1),
    "I": ("循环系统疾病", ("心脏", "血管", "高血压", "心肌", "循环"), 1.2),
    "J": ("呼吸系统疾病", ("肺", "呼吸", "咳嗽", "气管", "支气管"), 1.1),
    "K": ("消化系统疾病", ("胃", "肠", "肝", "消化", "腹泻"), 1.0),
    "N": ("泌尿生殖系统疾病", ("肾", "膀胱", "泌尿", "生殖", "尿"), 1.0),
    "S": ("损伤、中毒和外因的某些其他后果", ("损伤", "外伤", "骨折", "中毒", "烧伤"), 0.9),
}
_LEVEL_BOOST = {1: 0.15, 2: 0.20, 3: 0.10}
_ENTITY_GAIN = (("disease", 0.4, 0.2), ("symptom", 0.2, None), ("anatomy", 0.1, None))


def _cosine(u, v) -> float:
    from sklearn.metrics.pairwise import cosine_similarity  # same routine as the reference (:12)
    return cosine_similarity([u], [v])[0][0]


class HierarchicalSimilarityService:
    def __init__(self, embedding_service=None, ner_service=None):
        self.embedding_service = embedding_service
        self.ner_service = ner_service
        self.uncertainty_service = UncertaintyDiagnosisService()
        self.level_weights = {1: 1.2, 2: 1.0, 3: 0.8}
        self.factor_weights = {
            "vector_similarity": 0.50, "hierarchy_boost": 0.20, "entity_match_score": 0.15,
            "semantic_coherence": 0.08, "category_alignment": 0.04, "context_relevance": 0.03,
        }
        self.main_categories = self._load_main_categories()
        self.similarity_cache: Dict[str, Any] = {}

    def _load_main_categories(self) -> Dict[str, Dict[str, Any]]:
        return {k: {"name": n, "keywords": list(kw), "semantic_weight": w} for k, (n, kw, w) in _CHAPTERS.items()}

    # ---- one candidate ------------------------------------------------------------------------------
    def calculate_enhanced_similarity(self, query_text: str, query_entities: Dict[str, List[Dict]],
                                      candidate_record: Dict[str, Any]) -> Tuple[float, SimilarityFactors]:
        factors = SimilarityFactors()
        try:
            title = candidate_record.get("preferred_zh", "").strip()
            exact = title == query_text.strip()
            factors.vector_similarity = self._calculate_vector_similarity(query_text, candidate_record)
            if exact and factors.vector_similarity < 0.9:
                factors.vector_similarity = 1.0
            factors.hierarchy_boost = self._calculate_hierarchy_boost(query_text, query_entities, candidate_record)
            factors.entity_match_score = self._calculate_entity_match_score(query_entities, candidate_record)
            factors.semantic_coherence = self._calculate_semantic_coherence(query_text, candidate_record)
            factors.category_alignment = self._calculate_category_alignment(query_entities, candidate_record)
            factors.context_relevance = self._calculate_context_relevance(query_text, candidate_record)
            score = self._calculate_weighted_score(factors)
            if exact:
                score = max(score, 1.5)
            return float(score), factors
        except Exception as exc:  # degrade to the base score, as the reference does (:214-219)
            logger.error("enhanced similarity failed: %s", exc)
            return float(candidate_record.get("score", 0.0)), factors

    def _calculate_vector_similarity(self, query_text: str, candidate_record: Dict[str, Any]) -> float:
        try:
            if not self.embedding_service:
                return candidate_record.get("score", 0.0)
            if "score" in candidate_record:
                return float(candidate_record["score"])
            qv = self.embedding_service.encode_query(query_text)
            ctext = candidate_record.get("semantic_text", candidate_record.get("preferred_zh", ""))
            cv = self.embedding_service.encode_query(ctext)
            return float(max(_cosine(qv, cv), 0.0))
        except Exception as exc:
            logger.warning("vector similarity failed: %s", exc)
            return candidate_record.get("score", 0.0)

    def _calculate_hierarchy_boost(self, query_text: str, query_entities: Dict[str, List[Dict]],
                                   candidate_record: Dict[str, Any]) -> float:
        try:
            level = candidate_record.get("level", 1)
            code = candidate_record.get("code", "")
            parent = candidate_record.get("parent_code", "")
            boost = self._get_level_boost_factor(level) * 0.3
            chapter = code[0] if code else ""
            if chapter in self.main_categories:
                boost += self._calculate_category_semantic_boost(query_text, query_entities,
                                                                 self.main_categories[chapter]) * 0.4
            if parent:
                boost += self._calculate_parent_child_boost(query_entities, code, parent) * 0.3
            return float(min(boost, 0.3))
        except Exception as exc:
            logger.warning("hierarchy boost failed: %s", exc)
            return 0.0

    def _get_level_boost_factor(self, level: int) -> float:
        return float(_LEVEL_BOOST.get(level, 0.10))

    def _calculate_category_semantic_boost(self, query_text: str, query_entities: Dict[str, List[Dict]],
                                           category_info: Dict[str, Any]) -> float:
        try:
            keywords = category_info.get("keywords", [])
            weight = category_info.get("semantic_weight", 1.0)
            boost = 0.0
            lowered = query_text.lower()
            hits = sum(1 for kw in keywords if kw in lowered)
            if hits > 0:
                boost += (hits / len(keywords)) * 0.3 * weight
            for ent in query_entities.get("disease", []):
                etext = ent.get("text", "").lower()
                ehits = sum(1 for kw in keywords if kw in etext)
                if ehits > 0:
                    boost += (ehits / len(keywords)) * 0.2 * ent.get("confidence", 0.5)
            return float(min(boost, 0.4))
        except Exception as exc:
            logger.warning("category boost failed: %s", exc)
            return 0.0

    def _calculate_parent_child_boost(self, query_entities, code: str, parent_code: str) -> float:
        return 0.1 if (len(code) > len(parent_code) and code.startswith(parent_code)) else 0.0

    def _calculate_entity_match_score(self, query_entities: Dict[str, List[Dict]],
                                      candidate_record: Dict[str, Any]) -> float:
        try:
            haystack = (f"{candidate_record.get('preferred_zh', '').lower()} "
                        f"{candidate_record.get('semantic_text', '').lower()}")
            total = 0.0
            for kind, full_gain, partial_gain in _ENTITY_GAIN:
                for ent in query_entities.get(kind, []):
                    etext = ent.get("text", "").lower()
                    conf = ent.get("confidence", 0.5)
                    if etext in haystack:
                        total += conf * full_gain
                    elif partial_gain is not None and any(w in haystack for w in etext.split()):
                        total += conf * partial_gain
            return float(min(total, 1.0))
        except Exception as exc:
            logger.warning("entity match failed: %s", exc)
            return 0.0

    def _calculate_semantic_coherence(self, query_text: str, candidate_record: Dict[str, Any]) -> float:
        try:
            if not self.embedding_service:
                return 0.5
            sem = candidate_record.get("semantic_text", "")
            if not sem:
                return 0.3
            qv = self.embedding_service.encode_query(query_text)
            sv = self.embedding_service.encode_query(sem)
            return max(_cosine(qv, sv), 0.0)
        except Exception as exc:
            logger.warning("semantic coherence failed: %s", exc)
            return 0.5

    def _calculate_category_alignment(self, query_entities: Dict[str, List[Dict]],
                                      candidate_record: Dict[str, Any]) -> float:
        try:
            code = candidate_record.get("code", "")
            if not code or code[0] not in self.main_categories:
                return 0.0
            keywords = self.main_categories[code[0]].get("keywords", [])
            aligned, count = 0.0, 0
            for _kind, ents in query_entities.items():
                for ent in ents:
                    count += 1
                    etext = ent.get("text", "").lower()
                    if any(kw in etext for kw in keywords):
                        aligned += ent.get("confidence", 0.5)
            return float(aligned / count) if count > 0 else 0.0
        except Exception as exc:
            logger.warning("category alignment failed: %s", exc)
            return 0.0

    def _calculate_context_relevance(self, query_text: str, candidate_record: Dict[str, Any]) -> float:
        try:
            title = candidate_record.get("preferred_zh", "")
            lq, lt = len(query_text), len(title)
            length_sim = 1.0 - abs(lq - lt) / max(lq, lt, 1)
            qs, ts = set(query_text), set(title)
            union = qs | ts
            overlap = len(qs & ts) / len(union) if union else 0
            return max(length_sim * 0.3 + overlap * 0.7, 0.0)
        except Exception as exc:
            logger.warning("context relevance failed: %s", exc)
            return 0.5

    def _calculate_weighted_score(self, factors: SimilarityFactors) -> float:
        try:
            w = self.factor_weights
            base = factors.vector_similarity
            high_precision = base > 0.95
            extra = 0.0
            extra += factors.hierarchy_boost * w["hierarchy_boost"] / 0.2 * (0.5 if high_precision else 1.0)
            extra += factors.entity_match_score * w["entity_match_score"] / 0.15
            if factors.semantic_coherence > base:
                extra += (factors.semantic_coherence - base) * w["semantic_coherence"] / 0.08
            extra += factors.category_alignment * w["category_alignment"] / 0.04
            extra += factors.context_relevance * w["context_relevance"] / 0.03
            if high_precision:
                extra += 0.15
            return float(min(base + extra, 1.8))
        except Exception as exc:
            logger.error("weighted score failed: %s", exc)
            return float(factors.vector_similarity)

    # ---- batch ------------------------------------------------------------------------------------------
    def batch_calculate_similarities(self, query_text: str, query_entities: Dict[str, List[Dict]],
                                     candidate_records: List[Dict[str, Any]]
                                     ) -> List[Tuple[Dict[str, Any], float, SimilarityFactors]]:
        clean_query, candidates = self.uncertainty_service.process_uncertainty_query(query_text, candidate_records)
        results = []
        for rec in candidates:
            try:
                score, factors = self.calculate_enhanced_similarity(clean_query, query_entities, rec)
                out = rec.copy()
                out["enhanced_score"] = score
                out["original_score"] = rec.get("original_score", rec.get("score", 0.0))
                out["similarity_factors"] = factors
                if "uncertainty_boost" in rec:
                    out["uncertainty_boost"] = rec["uncertainty_boost"]
                    out["uncertainty_weight"] = rec["uncertainty_weight"]
                results.append((out, score, factors))
            except Exception as exc:
                logger.error("rescoring of %s failed: %s", rec.get("code", "unknown"), exc)
                results.append((rec, rec.get("score", 0.0), SimilarityFactors()))
        results.sort(key=lambda item: item[1], reverse=True)
        return results

    # ---- additive: a whole batch of queries, arithmetic on the device ---------
    CHAPTER_ORDER = tuple(_CHAPTERS)   # index of a code's first letter in this tuple = low bits of a row tag

    @classmethod
    def row_tag(cls, code: str) -> int:
        """One byte per corpus row for the device-side rescoring: bits 0-3 = index of the code's first letter in
        CHAPTER_ORDER (15 = not a chapter of the table), bit 7 = the code matches the uncertainty service's `\\.9\\d*$`."""
        from .uncertainty_diagnosis_service import _CODE_DOT9
        c = code[0] if code else ""
        tag = cls.CHAPTER_ORDER.index(c) if c in cls.CHAPTER_ORDER else 15
        return tag | (0x80 if _CODE_DOT9.search(code or "") else 0)

    def query_params(self, query_text: str) -> List[float]:
        """The twelve per-query numbers of the device-side rescoring: everything batch_calculate_similarities derives from
        the QUERY STRING alone when the hits are live-shaped (no top-level preferred_zh / level / parent_code /
        semantic_text) and no entities are supplied: the uncertainty weight (0 = no marker), the context
        relevance against an empty title, the exact-match flag (an empty clean query equals the empty title), and the
        category-semantic boost of each of the nine chapters. Same values as the per-candidate methods return (they are
        what the slow path below calls); the common case - no marker, no chapter keyword in the text - is decided by
        two precompiled alternations instead of ~65 substring tests."""
        import re
        cls = type(self)
        if getattr(cls, "_kw_any", None) is None:
            from .uncertainty_diagnosis_service import _MARKER_GROUPS
            cls._kw_any = re.compile("|".join(re.escape(k) for _n, kws, _w in _CHAPTERS.values() for k in kws))
            cls._marker_any = re.compile("|".join(re.escape(m.lower()) for _t, _w, _d, ms in _MARKER_GROUPS for m in ms))
            cls._NO_CATS = [0.0] * len(self.CHAPTER_ORDER)
        lowered_q = query_text.lower()
        if cls._marker_any.search(lowered_q):
            clean, weight = self.uncertainty_service.clean_and_weight(query_text)
            weight = float(weight)
        else:
            clean, weight = query_text, 0.0
            if not cls._kw_any.search(lowered_q):
                # the common case, in closed form: no marker, no chapter keyword. Against the empty title of a live hit
                # _calculate_context_relevance is max((1 - len / max(len, 1)) * 0.3 + 0 * 0.7, 0) = 0.0 for any non-empty text
                # (0.3 for the empty one), the exact-match flag is set by an all-blank query only, every chapter boost is 0
                return [0.0, 0.0 if clean else 0.3, 1.0 if not clean.strip() else 0.0] + cls._NO_CATS
        cr = self._calculate_context_relevance(clean, {})
        exact = 1.0 if "" == clean.strip() else 0.0
        lowered = clean.lower()
        cats = [0.0] * len(self.CHAPTER_ORDER)
        if cls._kw_any.search(lowered):
            for ci, c in enumerate(self.CHAPTER_ORDER):   # _calculate_category_semantic_boost with no entities, inlined
                info = self.main_categories[c]
                kws = info.get("keywords", [])
                hits = 0
                for kw in kws:
                    if kw in lowered:
                        hits += 1
                if hits > 0:
                    cats[ci] = float(min(0.0 + (hits / len(kws)) * 0.3 * info.get("semantic_weight", 1.0), 0.4))
        return [weight, float(cr), exact] + cats

    def device_weights(self) -> List[float]:
        w = self.factor_weights
        return [w["hierarchy_boost"], w["entity_match_score"], w["semantic_coherence"], w["category_alignment"],
                w["context_relevance"], 0.3 if self.embedding_service else 0.5, self._get_level_boost_factor(1) * 0.3]

    def rescore_live_hits_batch(self, queries: List[str], adj, ids, row_tags, id_base: int = 0, q_params=None):
        """batch_calculate_similarities(q, {}, hits) for every query of a batch whose hits are still device tensors
        (adj f64 [nq, k] and ids i64 [nq, k] from MilvusService.search_batch; row_tags from MilvusService.row_tags()).
        Returns device tensors [nq, k] in the final order: (order, enhanced, score, vector_similarity, hierarchy_boost,
        uncertainty_boost) - see include/icd_search.h icd_hier_rescore. Bit-identical to the per-query Python method."""
        import torch
        from .._native import hier_rescore
        if q_params is None:
            q_params = [self.query_params(q) for q in queries]
        qp = torch.tensor(q_params, dtype=torch.float64)
        return hier_rescore(adj, ids, row_tags, qp, self.device_weights(), id_base=id_base)

    # ---- explanation / tuning --------------------------------------------------------------------------
    def get_similarity_explanation(self, factors: SimilarityFactors) -> Dict[str, Any]:
        labels = {
            "vector_similarity": "基础向量相似度", "hierarchy_boost": "ICD-10层级增强分数",
            "entity_match_score": "医学实体匹配分数", "semantic_coherence": "语义一致性分数",
            "category_alignment": "ICD类别对齐分数", "context_relevance": "上下文相关性分数",
        }
        detail = {}
        for name, label in labels.items():
            value = getattr(factors, name)
            weight = self.factor_weights[name]
            detail[name] = {"score": value, "weight": weight, "contribution": value * weight, "description": label}
        return {"total_score": self._calculate_weighted_score(factors), "factors": detail}

    def update_weights(self, new_weights: Dict[str, float]):
        for name, value in new_weights.items():
            if name in self.factor_weights:
                self.factor_weights[name] = value
        total = sum(self.factor_weights.values())
        if total != 1.0:
            for name in self.factor_weights:
                self.factor_weights[name] /= total
