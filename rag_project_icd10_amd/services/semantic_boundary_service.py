"""Diagnosis boundaries of a medical text (the reference's services/semantic_boundary_service.py, restated; part of row N2 of
SURVEY.md section 8: the encoder calls of one request collected into batches).

What the reference computes, and what it costs there:
  * detect_diagnosis_boundaries (:48-83): delimiter segmentation (:86-136: the FIRST delimiter, in priority order, that leaves more
    than one segment of >= 2 characters; a delimiter that leaves one keeps that one unless a later delimiter splits better - the
    quirk is kept), segments that carry a connection pattern merged into their predecessor (:138-172), then a "semantic clustering"
    (:174-224) that embeds every segment with its own encode_query call and clusters the S segments into S clusters: the
    grouping it returns is always one group per segment, in order (every label is distinct; a failing clustering falls back to the
    same list). Only its cost is real: S batch-1 forwards.
  * get_boundary_confidence (:263-301): a rule score per boundary, + 0.1 when the cosine between the boundary's embedding and the
    next boundary's is below 0.75: two more encode_query calls per adjacent pair - (3 S - 2) forwards per text in all.

Here the S segment texts are embedded ONCE, in ONE batch (EmbeddingService.encode_query_batch: the canonical arithmetic, every
row bit for bit what encode_query returns for the string alone), kept by text, and both steps read them.
The results are the reference's: tests/test_text_enhanced_cpu.py holds them against fixtures made by running the reference's
class over the same texts and embeddings (tests/golden/make_text_enhanced_golden.py).
"""
from __future__ import annotations

import logging
import re
import threading
from collections import OrderedDict
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

logger = logging.getLogger(__name__)

# delimiter -> priority, in the reference's table order (:29-36); sorted() there is stable, so equal priorities keep this order
_DELIMITERS: Tuple[Tuple[str, int], ...] = (("；", 1), (";", 1), ("。", 2), (".", 2), ("，", 3), (",", 3), ("\n", 4), ("+", 5), ("＋", 5), (" ", 6), ("\t", 6))
_CONNECTIONS = tuple(re.compile(p) for p in (r"伴?有?(?:并发|合并)", r"(?:继发|导致|引起)", r"(?:急性|慢性)加重", r"(?:病史|既往史)", r"(?:术后|治疗后)"))
_TERM = re.compile(r"[^，。；\s]{2,}(?:病|症|炎|癌|瘤)")
_DEPENDENT = re.compile(r"(?:伴有|合并|继发)")
_CACHE_ENTRIES = 512

Boundary = Tuple[int, int, str]


class SemanticBoundaryDetector:
    def __init__(self, embedding_service=None):
        self.embedding_service = embedding_service
        self.semantic_threshold = 0.75
        self.min_segment_length = 2
        self.delimiter_priority = dict(_DELIMITERS)
        self.connection_patterns = [p.pattern for p in _CONNECTIONS]
        self._vectors: "OrderedDict[str, np.ndarray]" = OrderedDict()   # text -> embedding, most recent last
        self._lock = threading.Lock()                                   # (requests of a server run in a pool of threads)

    # ---- embeddings: one batch per text, kept by string ---------------------------------------------------------------------------
    def _embed(self, texts: Sequence[str]) -> List[np.ndarray]:
        with self._lock:
            have = {t: self._vectors[t] for t in texts if t in self._vectors}
        missing = [t for t in dict.fromkeys(texts) if t not in have]
        if missing:
            svc = self.embedding_service
            batch = getattr(svc, "encode_query_batch", None)
            rows = batch(missing) if batch is not None else [svc.encode_query(t) for t in missing]
            for t, v in zip(missing, rows):
                have[t] = np.asarray(v)
            with self._lock:
                for t in missing:
                    self._vectors[t] = have[t]
                while len(self._vectors) > _CACHE_ENTRIES:
                    self._vectors.popitem(last=False)
        return [have[t] for t in texts]

    def cached_vector(self, text: str) -> Optional[np.ndarray]:
        """the embedding of `text` if a boundary step of this detector has computed it (MultiDiagnosisService: a diagnosis that IS a
        boundary's text is not embedded a second time - the vector is the one encode_query would return, bit for bit)"""
        with self._lock:
            return self._vectors.get(text)

    @staticmethod
    def _cosine(a: np.ndarray, b: np.ndarray) -> float:
        # sklearn.metrics.pairwise.cosine_similarity of two rows: both normalised in their own dtype, then the dot product
        a, b = np.asarray(a), np.asarray(b)
        if a.dtype != b.dtype or a.dtype.kind != "f":
            a, b = a.astype(np.float64), b.astype(np.float64)
        na, nb = np.sqrt(np.dot(a, a)), np.sqrt(np.dot(b, b))
        a = a / na if na != 0 else a
        b = b / nb if nb != 0 else b
        return float(np.dot(a, b))

    # ---- boundaries -------------------------------------------------------------------------------------------------------------------
    def detect_diagnosis_boundaries(self, text: str) -> List[Boundary]:
        if not text or not text.strip():
            return []
        segments = self._initial_segmentation(text)
        if len(segments) <= 1:
            return [(0, len(text), text.strip())]
        if self.embedding_service:
            try:
                groups = self._semantic_clustering(segments)
            except Exception as exc:   # (the reference logs and keeps the segmentation)
                logger.warning("semantic grouping failed, keeping the delimiter segmentation: %s", exc)
                groups = [[s["text"]] for s in segments]
        else:
            groups = [[s["text"]] for s in segments]
        return self._optimize_boundaries(groups, text)

    def _initial_segmentation(self, text: str) -> List[Dict[str, Any]]:
        segments: List[Dict[str, Any]] = []
        for delimiter, priority in sorted(_DELIMITERS, key=lambda kv: kv[1]):
            if delimiter not in text:
                continue
            parts = text.split(delimiter)
            if len(parts) <= 1:
                continue
            segments, pos = [], 0          # (a delimiter that is tried REPLACES what an earlier one left: the reference's quirk)
            for part in parts:
                part = part.strip()
                if part and len(part) >= self.min_segment_length:
                    start = text.find(part, pos)
                    segments.append({"text": part, "start": start, "end": start + len(part), "delimiter": delimiter, "priority": priority})
                    pos = start + len(part)
            if len(segments) > 1:
                break
        if not segments:
            segments = [{"text": text.strip(), "start": 0, "end": len(text), "delimiter": None, "priority": 0}]
        return self._filter_connection_cases(segments, text)

    def _filter_connection_cases(self, segments: List[Dict[str, Any]], text: str) -> List[Dict[str, Any]]:
        kept: List[Dict[str, Any]] = []
        for seg in segments:
            connected = any(p.search(seg["text"]) for p in _CONNECTIONS)
            if connected and kept:
                prev = kept[-1]
                kept[-1] = {"text": prev["text"] + " " + seg["text"], "start": prev["start"], "end": seg["end"],
                            "delimiter": seg["delimiter"], "priority": min(prev["priority"], seg["priority"])}
            else:
                kept.append(seg)
        return kept

    def _semantic_clustering(self, segments: List[Dict[str, Any]]) -> List[List[str]]:
        """One group per segment, in order - what the reference's S-clusters-of-S-points clustering returns whatever the embeddings
        are (:199-221). The segments are embedded here, in one batch, because the reference embeds them here: a failing encoder
        fails at the same place (and is survived the same way), and get_boundary_confidence finds the vectors ready."""
        texts = [s["text"] for s in segments]
        if len(segments) > 1:
            try:
                self._embed(texts)
            except Exception as exc:
                logger.error("segment embedding failed: %s", exc)
        return [[t] for t in texts]

    def _optimize_boundaries(self, groups: List[List[str]], original_text: str) -> List[Boundary]:
        out: List[Boundary] = []
        cursor = 0
        for group in groups:
            joined = " ".join(group).strip()
            if not joined:
                continue
            start = original_text.find(joined, cursor)
            if start == -1:      # a merged group is not a substring: anchor it at its first segment
                start = original_text.find(group[0].strip(), cursor)
                if start == -1:
                    start = cursor
                end = min(start + len(joined), len(original_text))
            else:
                end = start + len(joined)
            out.append((start, end, joined))
            cursor = end + 1
        return out or [(0, len(original_text), original_text.strip())]

    def get_boundary_confidence(self, boundaries: List[Boundary]) -> List[float]:
        vectors: Optional[List[np.ndarray]] = None
        if self.embedding_service and len(boundaries) > 1:
            try:
                vectors = self._embed([b[2] for b in boundaries])
            except Exception:   # (the reference swallows the encoder's failures pair by pair: no clarity bonus then)
                vectors = None
        out: List[float] = []
        for i, (_, _, text) in enumerate(boundaries):
            c = 0.5
            if len(text) >= 4:
                c += 0.1
            if len(text) >= 8:
                c += 0.1
            if _TERM.search(text):
                c += 0.2
            if not _DEPENDENT.search(text):
                c += 0.1
            if i < len(boundaries) - 1 and vectors is not None:
                try:
                    if self._cosine(vectors[i], vectors[i + 1]) < self.semantic_threshold:
                        c += 0.1
                except Exception:
                    pass
            out.append(min(c, 1.0))
        return out

    def analyze_text_structure(self, text: str) -> Dict[str, Any]:
        boundaries = self.detect_diagnosis_boundaries(text)
        confidences = self.get_boundary_confidence(boundaries)
        return {"original_text": text, "total_boundaries": len(boundaries),
                "boundaries": [{"text": b[2], "start": b[0], "end": b[1], "confidence": c, "length": len(b[2])} for b, c in zip(boundaries, confidences)],
                "avg_confidence": float(np.mean(confidences)) if confidences else 0.0,
                "is_multi_diagnosis": len(boundaries) > 1}
