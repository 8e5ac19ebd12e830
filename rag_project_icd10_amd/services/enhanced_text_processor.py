"""Diagnosis extraction from entities AND boundaries - the reference's default text mode (services/enhanced_text_processor.py,
restated; MultiDiagnosisService builds its text processor with use_enhanced_processing=True, services/multi_diagnosis_service.py:44-47).

extract_diagnoses_enhanced (:39-88): the text's medical entities (MedicalNERService, row N4), its diagnosis boundaries and their
confidences (SemanticBoundaryDetector: ONE encoder batch per text here, (3 S - 2) one-string forwards in the reference), fused
(:148-203): a boundary with two or more disease entities is cut at the entities (:205-262), every piece gets its entities, entity
density, types and a confidence (:279-318), then length / confidence filters, a character-Jaccard deduplication and a stable
sort by confidence (:320-379). Any failure falls back to delimiter boundaries at confidence 0.5 (:392-420).

The NER service is INJECTED (the reference constructs its own, :26): its classifier weights are not available offline, so whoever
builds the services decides whether there is one (api/app.py, MultiDiagnosisService). Results against the reference's class over
the same texts, entities and embeddings: tests/test_text_enhanced_cpu.py (fixtures: tests/golden/make_text_enhanced_golden.py).
"""
from __future__ import annotations

import logging
import re
from typing import Any, Dict, List, Tuple

from .semantic_boundary_service import SemanticBoundaryDetector

logger = logging.getLogger(__name__)

Boundary = Tuple[int, int, str]
# (:96-100) semicolons first, commas outside full-width parentheses, plus signs
_SIMPLE_SEPARATORS = (re.compile(r"[；;]"), re.compile(r"[，,](?![^（]*）)"), re.compile(r"[+＋]"))
_SPLIT_KEYWORDS = ("既往", "病史", "术后", "治疗", "保守", "规律", "控制")
_ENTITY_WEIGHT = {"disease": 1.2, "symptom": 0.8}


class EnhancedTextProcessor:
    def __init__(self, embedding_service=None, ner_service=None):
        if ner_service is None:
            raise ValueError("EnhancedTextProcessor needs a MedicalNERService (the reference builds one itself; here it is injected)")
        self.ner_service = ner_service
        self.boundary_detector = SemanticBoundaryDetector(embedding_service)
        self.embedding_service = embedding_service
        self.last_text_entities = None
        self.config = {"min_diagnosis_length": 2, "max_diagnosis_length": 50, "min_entity_confidence": 0.6,
                       "use_semantic_boundary": True, "fallback_to_simple_split": True}

    # ---- the entry point ------------------------------------------------------------------------------------------------------------
    def extract_diagnoses_enhanced(self, text: str, filter_drugs: bool = True) -> List[Dict[str, Any]]:
        if not text or not text.strip():
            return []
        try:
            if self.config["use_semantic_boundary"] and self.embedding_service:
                # the text's entities and its boundaries do not depend on each other: the token classifier runs in a worker thread (its own
                # encoder handle and stream) while this thread embeds the segments - both forwards are latency-bound and overlap
                job = self._pool().submit(self.ner_service.extract_medical_entities, text, filter_drugs=filter_drugs)
                try:
                    boundaries = self.boundary_detector.detect_diagnosis_boundaries(text)
                    confidences = self.boundary_detector.get_boundary_confidence(boundaries)
                    if len(boundaries) == 1:
                        # a one-segment text embeds nothing for its boundaries - and is, as a rule, its own diagnosis: embed it NOW, beside
                        # the classifier (MultiDiagnosisService finds the vector ready; a text that is cut after all has cost one hidden forward)
                        try:
                            self.boundary_detector._embed([boundaries[0][2]])
                        except Exception as exc:
                            logger.debug("speculative embedding failed: %s", exc)
                finally:
                    entities = job.result()
            else:
                entities = self.ner_service.extract_medical_entities(text, filter_drugs=filter_drugs)
                boundaries = self._simple_boundary_detection(text)
                confidences = [0.5] * len(boundaries)
            self.last_text_entities = (text, bool(filter_drugs), entities)   # (MultiDiagnosisService: a diagnosis that IS the text is not classified twice)
            fused = self._fuse_entity_boundary_info(text, entities, boundaries, confidences)
            return self._filter_and_rank_diagnoses(fused)
        except Exception as exc:
            logger.error("enhanced diagnosis extraction failed: %s", exc)
            return self._fallback_extraction(text)

    def _pool(self):
        pool = getattr(self, "_executor", None)
        if pool is None:
            from concurrent.futures import ThreadPoolExecutor
            pool = self._executor = ThreadPoolExecutor(max_workers=1, thread_name_prefix="icd-text-ner")
        return pool

    # ---- delimiter boundaries (no encoder; also the fallback) -----------------------------------------------------------------------
    def _pieces(self, text: str, parts: List[str]) -> List[Boundary]:
        out, pos = [], 0
        for part in parts:
            part = part.strip()
            if part and len(part) >= self.config["min_diagnosis_length"]:
                start = text.find(part, pos)
                if start != -1:
                    out.append((start, start + len(part), part))
                    pos = start + len(part)
        return out

    def _simple_boundary_detection(self, text: str) -> List[Boundary]:
        for sep in _SIMPLE_SEPARATORS:
            parts = sep.split(text)
            if len(parts) > 1:
                found = self._pieces(text, parts)
                if len(found) > 1:
                    return found
        for keyword in _SPLIT_KEYWORDS:     # (:124-146) history / treatment words start a new piece and stay with it
            if keyword in text:
                parts = text.split(keyword)
                if len(parts) > 1:
                    found = self._pieces(text, [parts[0]] + [keyword + p for p in parts[1:]])
                    if len(found) > 1:
                        return found
        return [(0, len(text), text.strip())]

    # ---- fusion --------------------------------------------------------------------------------------------------------------------------
    def _fuse_entity_boundary_info(self, text: str, entities: Dict[str, List[Dict]], boundaries: List[Boundary],
                                   boundary_confidences: List[float]) -> List[Dict[str, Any]]:
        method = self.ner_service.get_model_info().get("extraction_method", "unknown")
        out: List[Dict[str, Any]] = []
        for i, (start, end, boundary_text) in enumerate(boundaries):
            b_conf = boundary_confidences[i] if i < len(boundary_confidences) else 0.5
            for sub in self._extract_sub_diagnoses_from_boundary(boundary_text, entities, start, end):
                stripped = sub["text"].strip()
                info = {"text": stripped, "start_pos": sub["start"], "end_pos": sub["end"], "boundary_confidence": b_conf,
                        "entities": sub["entities"], "entity_density": 0.0, "primary_entity_types": [], "diagnosis_confidence": 0.0,
                        "metadata": {"length": len(stripped), "has_disease_entity": False, "has_symptom_entity": False,
                                     "entity_count": 0, "ner_method": method}}
                total = sum(len(v) for v in sub["entities"].values())
                if total > 0:
                    info["entity_density"] = total / len(sub["text"]) if sub["text"] else 0
                    info["metadata"]["entity_count"] = total
                    for etype, elist in sub["entities"].items():
                        if elist:
                            info["primary_entity_types"].append(etype)
                            if etype == "disease":
                                info["metadata"]["has_disease_entity"] = True
                            elif etype == "symptom":
                                info["metadata"]["has_symptom_entity"] = True
                info["diagnosis_confidence"] = self._calculate_diagnosis_confidence(info)
                out.append(info)
        return out

    def _extract_sub_diagnoses_from_boundary(self, boundary_text: str, entities: Dict[str, List[Dict]], boundary_start: int,
                                             boundary_end: int) -> List[Dict[str, Any]]:
        whole = [{"text": boundary_text, "start": boundary_start, "end": boundary_end,
                  "entities": self._extract_entities_in_boundary(entities, boundary_start, boundary_end)}]
        diseases = [e for e in entities.get("disease", []) if boundary_start <= e.get("start", 0) < boundary_end]
        if len(diseases) <= 1:
            return whole
        diseases.sort(key=lambda e: e.get("start", 0))     # (stable, like the reference's sorted())
        subs: List[Dict[str, Any]] = []
        prev_end = boundary_start
        for i, ent in enumerate(diseases):
            e_start = ent.get("start", boundary_start)
            e_end = ent.get("end", e_start + len(ent.get("text", "")))
            seg_end = min(diseases[i + 1].get("start", boundary_end), boundary_end) if i < len(diseases) - 1 else boundary_end
            seg_start = max(prev_end, e_start - 10)        # up to ten characters of lead-in, never into the previous entity
            piece = boundary_text[seg_start - boundary_start:seg_end - boundary_start].strip()
            if piece and len(piece) >= 2:
                subs.append({"text": piece, "start": seg_start, "end": seg_end,
                             "entities": self._extract_entities_in_boundary(entities, seg_start, seg_end)})
            prev_end = e_end
        return subs or whole

    def _extract_entities_in_boundary(self, entities: Dict[str, List[Dict]], start: int, end: int) -> Dict[str, List[Dict]]:
        # inside the span, or overlapping it (:276-277; the second condition contains the first for non-empty entities)
        return {etype: [e for e in elist
                        if (e.get("start", 0) >= start and e.get("end", 0) <= end) or (e.get("start", 0) < end and e.get("end", 0) > start)]
                for etype, elist in entities.items()}

    def _calculate_diagnosis_confidence(self, info: Dict[str, Any]) -> float:
        c = 0.3
        c += info["boundary_confidence"] * 0.3
        scores = [e.get("confidence", 0.5) * _ENTITY_WEIGHT.get(etype, 0.6) for etype, elist in info["entities"].items() for e in elist]
        if scores:
            c += sum(scores) / len(scores) * 0.4
        n = len(info["text"])
        if 4 <= n <= 20:
            c += 0.1
        elif n < 2:
            c -= 0.2
        if info["entity_density"] > 0.1:
            c += 0.1
        return min(c, 1.0)

    # ---- filters ---------------------------------------------------------------------------------------------------------------------
    def _filter_and_rank_diagnoses(self, diagnoses: List[Dict[str, Any]]) -> List[Dict[str, Any]]:
        lo, hi = self.config["min_diagnosis_length"], self.config["max_diagnosis_length"]
        floor = max(0.4, self.config.get("min_diagnosis_confidence", 0.4))
        kept = [d for d in diagnoses if lo <= len(d["text"]) <= hi and d["diagnosis_confidence"] >= floor]
        return sorted(self._deduplicate_diagnoses(kept), key=lambda d: d["diagnosis_confidence"], reverse=True)

    def _deduplicate_diagnoses(self, diagnoses: List[Dict[str, Any]]) -> List[Dict[str, Any]]:
        if len(diagnoses) <= 1:
            return diagnoses
        kept: List[Dict[str, Any]] = []
        for d in diagnoses:
            for existing in kept:
                if self._text_similarity(d["text"], existing["text"]) > 0.8:
                    if d["diagnosis_confidence"] > existing["diagnosis_confidence"]:   # the better of the two, at the END of the list
                        kept.remove(existing)
                        kept.append(d)
                    break
            else:
                kept.append(d)
        return kept

    @staticmethod
    def _text_similarity(a: str, b: str) -> float:
        if not a or not b:
            return 0.0
        sa, sb = set(a), set(b)
        union = len(sa | sb)
        return len(sa & sb) / union if union > 0 else 0.0

    def _fallback_extraction(self, text: str) -> List[Dict[str, Any]]:
        return [{"text": t.strip(), "start_pos": s, "end_pos": e, "boundary_confidence": 0.5, "entities": {}, "entity_density": 0.0,
                 "primary_entity_types": [], "diagnosis_confidence": 0.5,
                 "metadata": {"length": len(t.strip()), "has_disease_entity": False, "has_symptom_entity": False, "entity_count": 0, "is_fallback": True}}
                for s, e, t in self._simple_boundary_detection(text)]

    # ---- conveniences of the reference's class -----------------------------------------------------------------------------------
    def extract_diagnoses_simple(self, text: str) -> List[str]:
        return [d["text"] for d in self.extract_diagnoses_enhanced(text)]

    def get_processing_summary(self, text: str) -> Dict[str, Any]:
        results = self.extract_diagnoses_enhanced(text)
        types = set()
        for r in results:
            types.update(r["primary_entity_types"])
        return {"original_text": text, "total_diagnoses": len(results),
                "avg_confidence": sum(r["diagnosis_confidence"] for r in results) / len(results) if results else 0,
                "entity_types_found": list(types),
                "high_confidence_count": sum(1 for r in results if r["diagnosis_confidence"] > 0.7),
                "processing_method": "enhanced" if self.config["use_semantic_boundary"] and self.embedding_service else "simple",
                "ner_info": self.ner_service.get_model_info()}
