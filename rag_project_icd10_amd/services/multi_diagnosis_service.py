"""Per-diagnosis embed -> search(2k) -> hierarchical rescoring -> top_k: the serving-time caller of the
hot path, reduced to the parts that are in scope.

Follows the reference's services/multi_diagnosis_service.py: match_multiple_diagnoses (:51-125) and
_match_single_diagnosis_enhanced steps 2-4 (:152-175); match confidence is the reference's own
original formula _calculate_match_confidence (:276-304). NOT reproduced (out of scope, SURVEY.md
section 2): the semantic-boundary splitter (delimiter split only) and the 12-factor confidence service beyond row
N3's pieces. The NER service (row N4) is optional: without one, query_entities is {}; the all-device batch path
(match_diagnoses_batch) never uses entities. Difference by design (row N2): all diagnoses of a
request are embedded in ONE encoder batch and searched in ONE search_batch call.
"""
from __future__ import annotations

import gc
import logging
import threading
from typing import Any, Dict, List

import numpy as np

from ..api.icd_models import Candidate, DiagnosisMatch, trusted_candidate, trusted_candidates
from ..tools.text_processor import DiagnosisTextProcessor
from .hierarchical_similarity_service import HierarchicalSimilarityService
from .multidimensional_confidence_service import MultiDimensionalConfidenceService

logger = logging.getLogger(__name__)



# The collector is a PROCESS-WIDE switch: under a threaded server two requests must not toggle it for each other (one
# re-enabling it in the middle of the other's build, or leaving it off for good). A lock-protected depth counter: the
# first request in switches it off (if it was on), the last one out switches it back on.
_gc_lock = threading.Lock()
_gc_depth = 0
_gc_was_on = False


class _gc_paused:
    def __enter__(self):
        global _gc_depth, _gc_was_on
        with _gc_lock:
            if _gc_depth == 0:
                _gc_was_on = gc.isenabled()
                if _gc_was_on:
                    gc.disable()
            _gc_depth += 1

    def __exit__(self, *exc):
        global _gc_depth
        with _gc_lock:
            _gc_depth -= 1
            if _gc_depth == 0 and _gc_was_on:
                gc.enable()
        return False

class MultiDiagnosisService:
    def __init__(self, embedding_service, milvus_service, ner_service=None):
        """ner_service: a MedicalNERService (row N4) or None. The reference always builds one (:28); here it is opt-in
        because its weights are not available offline - with one given, the per-request path extracts the entities of
        all diagnoses of a request in ONE classifier batch and hands them to the rescoring like the reference (:147-158)."""
        self.embedding_service = embedding_service
        self.milvus_service = milvus_service
        self.ner_service = ner_service
        self.hierarchical_similarity = HierarchicalSimilarityService(embedding_service=embedding_service,
                                                                     ner_service=ner_service)
        # the reference's default text mode is "enhanced" (entities + semantic boundaries, :44-47): here whenever there is an NER service
        self.text_processor = DiagnosisTextProcessor(embedding_service=embedding_service, ner_service=ner_service)
        # row N3: only the embedding cosine and the score statistics of the reference's confidence service
        self.confidence_service = MultiDimensionalConfidenceService(
            embedding_service=embedding_service, hierarchical_similarity_service=self.hierarchical_similarity)

    def match_multiple_diagnoses(self, text: str, top_k: int = 5) -> Dict[str, Any]:
        enhanced = self.text_processor.extract_diagnoses_enhanced(text)
        diagnoses = [d["text"] for d in enhanced]
        mode = self.text_processor.get_processing_mode()
        if not diagnoses:
            return {"original_text": text, "extracted_diagnoses": [], "matches": [], "total_matches": 0,
                    "processing_mode": mode,
                    "extraction_metadata": {"enhanced_results_count": 0, "avg_extraction_confidence": 0.0}}
        confs = [d.get("diagnosis_confidence", 0.5) for d in enhanced]
        # one encoder batch + one search batch for the whole request
        matches = None
        if self.ner_service is None and getattr(self.milvus_service, "supports_device_rescoring", lambda: False)():
            # no entities to match (the reference's rescoring then depends on the query string and the hits' codes only):
            # the whole request stays on the device - encode -> search(2 top_k) -> rescoring -> top_k winners come back.
            # Same DiagnosisMatch objects as the host path below (tests/test_gpu_parity.py, all 1 000 golden strings).
            try:
                matches = self.match_diagnoses_batch(diagnoses, top_k=top_k)
            except Exception as exc:
                logger.error("device-side request path failed (%s): host path", exc)
                matches = None
        if matches is not None:
            return {"original_text": text, "extracted_diagnoses": diagnoses, "matches": matches,
                    "total_matches": sum(len(m.candidates) for m in matches), "processing_mode": mode,
                    "extraction_metadata": {"enhanced_results_count": len(enhanced),
                                            "avg_extraction_confidence": sum(confs) / len(confs),
                                            "extraction_method": mode, "drug_filtering_enabled": mode == "enhanced"}}
        # The entities of the request's diagnoses do not depend on their embeddings: the token classifier runs in a worker thread on
        # its own stream WHILE this thread embeds and searches (both forwards are latency-bound - a few work-groups each, csrc/
        # encoder_small.hpp - and overlap almost entirely: a one-diagnosis request with NER 1.14 -> 0.8 ms). Same results either way.
        ner_job = self._ner_pool().submit(self._entities_of, diagnoses) if self.ner_service else None
        vectors = self._embed_diagnoses(diagnoses)
        try:
            hit_lists = self.milvus_service.search_batch(vectors, top_k * 2, as_dicts=True)
        except Exception as exc:
            logger.error("batch search failed: %s", exc)
            hit_lists = [[] for _ in diagnoses]
        entities = ner_job.result() if ner_job is not None else [{} for _ in diagnoses]
        matches = [self._match_from_hits(d, hits, top_k, ents) for d, hits, ents in zip(diagnoses, hit_lists, entities)]
        return {"original_text": text, "extracted_diagnoses": diagnoses, "matches": matches,
                "total_matches": sum(len(m.candidates) for m in matches), "processing_mode": mode,
                "extraction_metadata": {"enhanced_results_count": len(enhanced),
                                        "avg_extraction_confidence": sum(confs) / len(confs),
                                        "extraction_method": mode, "drug_filtering_enabled": mode == "enhanced"}}

    def _embed_diagnoses(self, diagnoses: List[str]):
        """the request's diagnoses embedded in one batch - minus those the enhanced text mode has embedded already: a diagnosis that is a
        boundary's text was embedded when the boundaries were scored (services/semantic_boundary_service.py), and the canonical batch
        arithmetic gives a string the same bits whatever shares its call"""
        detector = getattr(getattr(self.text_processor, "_enhanced_processor", None), "boundary_detector", None)
        known = [detector.cached_vector(d) if detector is not None else None for d in diagnoses]
        missing = [d for d, v in zip(diagnoses, known) if v is None]
        if len(missing) == len(diagnoses):
            return self.embedding_service.encode_query_batch(diagnoses)
        fresh = iter(self.embedding_service.encode_query_batch(missing)) if missing else iter(())
        return np.stack([np.asarray(v) if v is not None else np.asarray(next(fresh)) for v in known])

    def _ner_pool(self):
        pool = getattr(self, "_ner_executor", None)
        if pool is None:
            from concurrent.futures import ThreadPoolExecutor
            pool = self._ner_executor = ThreadPoolExecutor(max_workers=1, thread_name_prefix="icd-ner")
        return pool

    def _entities_of(self, diagnoses: List[str]) -> List[Dict[str, Any]]:
        """the entities of every diagnosis of a request, never raising (the reference extracts inside
        _match_single_diagnosis_enhanced's try, services/multi_diagnosis_service.py:147-158: a failing NER costs that diagnosis its
        entities, never the request - one batch here, so a failure of the batch retries string by string and only the strings that
        still fail go without entities). Runs in the worker thread of _ner_pool, on its own stream when the classifier is on a GPU."""
        entities = [{} for _ in diagnoses]
        stream_ctx = None
        try:
            import torch
            dev = getattr(getattr(self.ner_service, "ner_pipeline", None), "device", "cpu")
            if str(dev).startswith("cuda") and torch.cuda.is_available():
                if getattr(self, "_ner_stream", None) is None:
                    self._ner_stream = torch.cuda.Stream(device=dev)
                stream_ctx = torch.cuda.stream(self._ner_stream)
        except Exception:   # (no torch / no GPU: the caller's stream, which is then the CPU)
            stream_ctx = None
        import contextlib
        # a diagnosis that IS the request's text (the usual one-diagnosis request) was classified when the text was cut: the same string,
        # the same switches, the same entities - not a second forward
        last = getattr(getattr(self.text_processor, "_enhanced_processor", None), "last_text_entities", None)
        if last is not None and last[1] and last[2] is not None and any(d == last[0] for d in diagnoses):
            import copy
            rest = [d for d in diagnoses if d != last[0]]
            rest_entities = iter(self._entities_of(rest)) if rest else iter(())
            return [copy.deepcopy(last[2]) if d == last[0] else next(rest_entities) for d in diagnoses]
        with (stream_ctx if stream_ctx is not None else contextlib.nullcontext()):
            try:
                entities = self.ner_service.extract_medical_entities_batch(diagnoses, filter_drugs=True)
            except Exception as exc:
                logger.error("batch NER failed (%s): one string at a time", exc)
                for i, d in enumerate(diagnoses):
                    try:
                        entities[i] = self.ner_service.extract_medical_entities(d, filter_drugs=True)
                    except Exception as exc1:
                        logger.error("NER failed for %r: %s", d, exc1)
        return entities

    def match_diagnoses_batch(self, diagnoses: List[str], top_k: int = 5, vectors=None,
                              confidence_statistics: bool = False) -> List[DiagnosisMatch]:
        """Additive (row N2): embed -> search(2 top_k) -> level reweight -> hierarchical rescoring for MANY diagnosis
        strings with everything between the tokenizer and the final top_k on the GPU: one encoder batch, one search_batch,
        one rescoring launch; only the top_k winners per string come back and become Candidate objects. Same results as
        _match_from_hits(d, milvus.search(encode_query(d), 2 top_k), top_k) per string (tests/test_gpu_parity.py).
        confidence_statistics=True (row N3) also fills DiagnosisMatch.confidence_factors with the three numbers of the
        reference's confidence service that are in scope - semantic_coherence (the live shape: cosine with the embedding
        of the empty string), model_uncertainty, prediction_variance - computed for the whole batch in two launches."""
        from .hierarchical_similarity_service import trusted_factors_row
        if not diagnoses:
            return []
        if vectors is None:
            vectors = self.embedding_service.encode_query_batch(diagnoses, to_device=True)
        hs = self.hierarchical_similarity
        qps = [hs.query_params(d) for d in diagnoses]   # ([1] is the context relevance the factors report)
        adj, raw, ids, _lv = self.milvus_service.search_batch(vectors, top_k * 2)
        order, enh, score, vs, hb, boost = hs.rescore_live_hits_batch(diagnoses, adj, ids, self.milvus_service.row_tags(),
                                                                       q_params=qps)
        # winners only: gather on the device, one copy to the host, plain Python lists for the object loop
        import torch
        kk = min(top_k, order.shape[1])
        # (ONE copy: eight .tolist() calls were eight stream synchronisations, ~20 us each - a fifth of a one-diagnosis request.
        #  Everything travels in 8-byte slots: float64 - exact for the int32 order and the float32 scores - except the int64 ids,
        #  whose bit patterns ride in plane 0 (an id_base of a shard may exceed 2^53))
        if adj.is_cuda:
            from .. import _native
            packed = _native.pack_winners(order, ids, raw, adj, enh, vs, hb, boost, kk).cpu()   # (one launch: gathers, slices, stack)
        else:
            o = order[:, :kk].long().clamp(min=0)
            packed = torch.stack([torch.gather(ids, 1, o).long().contiguous().view(torch.float64), torch.gather(raw, 1, o).double(), torch.gather(adj, 1, o).double()]
                                 + [t[:, :kk].double() for t in (order, enh, vs, hb, boost)], 0)
        h_ids, h_ord = packed[0].contiguous().view(torch.int64).tolist(), packed[3].long().tolist()
        h_raw, h_adj, h_enh, h_vs, h_hb, h_boost = (packed[i].tolist() for i in (1, 2, 4, 5, 6, 7))
        recs = self.milvus_service.client.records
        sc = 0.3 if hs.embedding_service else 0.5
        conf = None
        if confidence_statistics:
            cs = self.confidence_service
            stats = cs.score_statistics_batch(enh, order, top_k=kk).tolist()
            qv = vectors if torch.is_tensor(vectors) else torch.as_tensor(np.asarray(vectors, dtype=np.float32))
            coh = cs.semantic_coherence_batch(qv.to(adj.device)).tolist()
            conf = [{"semantic_coherence": coh[q], "model_uncertainty": stats[q][4], "prediction_variance": stats[q][5]}
                    for q in range(len(diagnoses))]
        out = []
        # (tens of thousands of acyclic objects are born here: the collector's generation-0 passes over them are pure cost)
        with _gc_paused():
            self._build_matches(out, diagnoses, kk, h_ord, h_enh, h_adj, h_raw, h_boost, h_ids, h_vs, h_hb, recs, sc, qps, conf, trusted_factors_row)
        return out

    def _build_matches(self, out, diagnoses, kk, h_ord, h_enh, h_adj, h_raw, h_boost, h_ids, h_vs, h_hb, recs, sc, qps, conf, trusted_factors_row):
        from ..api.icd_models import bulk_candidates, trusted_match, trusted_matches_ready
        from .hierarchical_similarity_service import SimilarityFactors
        # the corpus' code / title columns (plain lists by row) where the store offers them, and the one-loop constructor once
        # trusted_candidate has checked this pydantic's object layout on a real hit (the first call of a process does)
        cols = getattr(getattr(self.milvus_service, "client", None), "code_title_columns", None)
        codes = titles = None
        if cols is not None and recs is getattr(self.milvus_service.client, "records", None):
            codes, titles = cols()
        for q, diagnosis in enumerate(diagnoses):
            try:
                # how many winners exist (order < 0 from there on); live hits carry level / parent_code under "metadata": the
                # top-level defaults apply (F8). The values are Python floats / str already (tolist() of the device results):
                # the trusted constructors skip the per-object validator but keep its one rule with teeth - a negative score
                # raises and degrades the whole match to an empty one, like the reference's
                n = kk
                row_ord = h_ord[q]
                if row_ord[kk - 1] < 0:   # (the winners come first: most queries have all kk)
                    for j in range(kk):
                        if row_ord[j] < 0:
                            n = j
                            break
                enh_q, adj_q, raw_q, boost_q = h_enh[q], h_adj[q], h_raw[q], h_boost[q]
                if max(boost_q) > 0:      # (an uncertainty marker in the query: the boosted hits report their reweighted score)
                    originals = [adj_q[j] if boost_q[j] > 0 else raw_q[j] for j in range(n)]
                else:
                    originals = raw_q
                if codes is not None and trusted_matches_ready():
                    ids_q = h_ids[q] if n == kk else h_ids[q][:n]
                    cands = bulk_candidates(codes, titles, SimilarityFactors, ids_q, enh_q, originals, h_vs[q], h_hb[q], sc, qps[q][1])
                else:
                    cands = trusted_candidates(recs, h_ids[q][:n], enh_q[:n], originals[:n],
                                               trusted_factors_row(h_vs[q][:n], h_hb[q][:n], sc, qps[q][1]))
                out.append(trusted_match(diagnosis, cands, self._match_confidence_of_scores(enh_q if n == kk else enh_q[:n]), conf[q] if conf is not None else None))
            except Exception as exc:
                logger.error("match failed for %s: %s", diagnosis, exc)
                out.append(DiagnosisMatch(diagnosis_text=diagnosis, candidates=[], match_confidence=0.0))

    def _match_from_hits(self, diagnosis: str, hits: List[Dict[str, Any]], top_k: int,
                         query_entities: Dict[str, Any] = None) -> DiagnosisMatch:
        try:
            rescored = self.hierarchical_similarity.batch_calculate_similarities(diagnosis, query_entities or {}, hits)
            candidates = []
            for rec, score, factors in rescored[:top_k]:
                cand = Candidate(code=rec.get("code", ""), title=rec.get("title", ""), score=float(score))
                cand.level = rec.get("level", 1)
                cand.parent_code = rec.get("parent_code", "")
                cand.enhanced_score = float(score)
                cand.original_score = float(rec.get("original_score", 0.0))
                cand.similarity_factors = factors
                candidates.append(cand)
            return DiagnosisMatch(diagnosis_text=diagnosis, candidates=candidates,
                                  match_confidence=self._calculate_match_confidence(candidates))
        except Exception as exc:  # e.g. a negative score fails Candidate's ge=0 validator
            logger.error("match failed for %s: %s", diagnosis, exc)
            return DiagnosisMatch(diagnosis_text=diagnosis, candidates=[], match_confidence=0.0)

    def _calculate_match_confidence(self, candidates: List[Candidate]) -> float:
        if not candidates:
            return 0.0
        return self._match_confidence_of_scores([c.score for c in candidates])

    @staticmethod
    def _match_confidence_of_scores(scores) -> float:
        """the reference's _calculate_match_confidence (services/multi_diagnosis_service.py:276-304) on the candidates' scores"""
        if not scores:
            return 0.0
        best = max(scores)
        if best > 0.9:
            conf = min(best, 0.95)
        elif len([s for s in scores if s > 0.7]) >= 2:
            conf = best * 0.8
        else:
            conf = best * 0.6
        return round(conf, 3)
