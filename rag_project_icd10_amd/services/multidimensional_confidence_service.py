"""The confidence service's embedding cosine and score statistics, per call on the host and for a
whole batch on the GPU.

Follows the reference's services/multidimensional_confidence_service.py for exactly these pieces:
    _assess_model_uncertainty          :936-963    np.mean / np.std / max over the candidates' 'score'
    _calculate_prediction_variance     :1087-1099  np.var over the candidates' 'score' (0.1 for fewer than two)
    _calculate_confidence_interval     :1101-1114
    _calculate_semantic_factors        :257-296    ONLY its 'semantic_coherence' entry (:273-280): the cosine of
                                                   encode_query(query_text) and encode_query(best candidate's 'preferred_zh')
NOT reproduced (out of scope): the other ten factors (NER, term lists, complexity, quality), their
merge into overall_confidence and the explanation texts.

A property of the reference kept as it is: the live /query path hands this service records WITHOUT 'preferred_zh'
(services/multi_diagnosis_service.py:178-186 builds them from code / title / score / level), so
`best_candidate.get('preferred_zh', '')` is '' and "the candidate vector" is the embedding of the empty string - one
constant vector. The batch entry point embeds it once; with the candidates' titles given it embeds those in one batch.

Batch entry points (additive): `score_statistics_batch` (icd_score_stats: numpy-identical doubles) and
`semantic_coherence_batch` (icd_cosine_rows), both on device tensors as `MilvusService.search_batch` /
`HierarchicalSimilarityService.rescore_live_hits_batch` return them.
"""
from __future__ import annotations

import logging
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

logger = logging.getLogger(__name__)

STAT_COLUMNS = ("mean", "std", "var", "max", "model_uncertainty", "prediction_variance")


class MultiDimensionalConfidenceService:
    def __init__(self, embedding_service=None, ner_service=None, hierarchical_similarity_service=None):
        self.embedding_service = embedding_service
        self.ner_service = ner_service
        self.hierarchical_similarity_service = hierarchical_similarity_service
        self._empty_vector = None   # encode_query('') on the device, made on first use

    # ---- per call, on the host (the reference's own numpy calls) -----------------------------------------------
    def _assess_model_uncertainty(self, candidate_records: List[Dict[str, Any]]) -> float:
        try:
            if not candidate_records:
                return 0.0
            scores = [r.get("score", 0) for r in candidate_records]
            if not scores:
                return 0.0
            std_score = float(np.std(scores))
            uncertainty_score = 1.0 - min(std_score, 0.5) / 0.5
            score_confidence = max(scores)
            final_uncertainty = (uncertainty_score * 0.6 + score_confidence * 0.4)
            return min(final_uncertainty, 1.0)
        except Exception as exc:
            logger.warning("model uncertainty failed: %s", exc)
            return 0.5

    def _calculate_prediction_variance(self, factors, candidate_records: List[Dict[str, Any]]) -> float:
        try:
            scores = [r.get("score", 0) for r in candidate_records]
            if len(scores) > 1:
                return float(np.var(scores))
            return 0.1
        except Exception:
            return 0.1

    def _calculate_confidence_interval(self, confidence: float, variance: float) -> Tuple[float, float]:
        try:
            margin = 1.96 * float(np.sqrt(variance))
            return (max(0.0, confidence - margin), min(1.0, confidence + margin))
        except Exception:
            return (max(0.0, confidence - 0.1), min(1.0, confidence + 0.1))

    def semantic_coherence(self, query_text: str, candidate_records: List[Dict[str, Any]]) -> float:
        """The 'semantic_coherence' entry of _calculate_semantic_factors (:257-296): 0.0 without candidates, without an
        embedding service, or when anything fails."""
        if not candidate_records or not self.embedding_service:
            return 0.0
        try:
            candidate_text = candidate_records[0].get("preferred_zh", "")
            q = np.asarray(self.embedding_service.encode_query(query_text), dtype=np.float64)
            c = np.asarray(self.embedding_service.encode_query(candidate_text), dtype=np.float64)
            return float(_cosine(q, c))
        except Exception as exc:
            logger.warning("semantic factors failed: %s", exc)
            return 0.0

    # ---- whole batch, on the device ------------------------------------------------------------------------------
    def score_statistics_batch(self, scores, order=None, top_k: Optional[int] = None):
        """scores f64 [nq, k] (device): every query's candidate scores in result order; order (i32 [nq, k], optional):
        entries below 0 mark hits that do not exist; top_k: the statistics run over each query's first top_k hits, like
        the reference's candidates[:top_k]. Returns f64 [nq, 6] (STAT_COLUMNS), bit-identical to the per-call methods."""
        from .._native import score_stats
        return score_stats(scores, order, top_k)

    def semantic_coherence_batch(self, query_vectors, candidate_texts: Optional[Sequence[str]] = None):
        """query_vectors f32 [nq, dim] (device, as encode_query_batch(..., to_device=True) returns them).
        candidate_texts None: the live /query shape, every query against encode_query('') (see the module docstring);
        else one text per query (the best candidates' 'preferred_zh'), embedded in ONE encoder batch. Returns f64 [nq]."""
        from .._native import cosine_rows
        if self.embedding_service is None:
            import torch
            return torch.zeros((query_vectors.shape[0],), dtype=torch.float64, device=query_vectors.device)
        if candidate_texts is None:
            if self._empty_vector is None or self._empty_vector.device != query_vectors.device:
                self._empty_vector = self.embedding_service.encode_query_batch([""], to_device=True)[0].to(query_vectors.device)
            return cosine_rows(query_vectors, self._empty_vector)
        assert len(candidate_texts) == query_vectors.shape[0]
        cand = self.embedding_service.encode_query_batch(list(candidate_texts), to_device=True).to(query_vectors.device)
        return cosine_rows(query_vectors, cand)


def _cosine(x: np.ndarray, y: np.ndarray) -> float:
    """sklearn.metrics.pairwise.cosine_similarity of two rows: each divided by its Euclidean norm (a zero row is left
    alone), then the dot product, in float64."""
    nx = float(np.sqrt(np.einsum("i,i->", x, x)))
    ny = float(np.sqrt(np.einsum("i,i->", y, y)))
    xn = x / (nx if nx != 0.0 else 1.0)
    yn = y / (ny if ny != 0.0 else 1.0)
    return float(np.dot(xn, yn))
