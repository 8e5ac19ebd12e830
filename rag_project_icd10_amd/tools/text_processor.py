"""Diagnosis extraction from a medical text (the reference's tools/text_processor.py).

Restates :29-33 (separator table), :87-109 (_extract_diagnoses_simple), :111-122 (_split_by_separators),
:124-148 (_clean_diagnosis_text), :150-154 (is_multi_diagnosis), :62-85 / :156-192 (the enhanced mode with its fall-backs),
:194-199 (get_processing_mode). Simple mode is pinned by tests/golden/text_split_cases.json.

The ENHANCED mode (NER entities fused with semantic boundaries, services/enhanced_text_processor.py) is the reference's default
(USE_ENHANCED_TEXT_PROCESSING, default true; :35-45). It needs a token classifier; the reference constructs one inside the enhanced
processor, here it is handed in (`ner_service`): with one, the mode is "enhanced" as in the reference - with none (no classifier
weights offline) the class reports "simple", which is also what the reference does when its enhanced processor fails to start (:55-60).
"""
from __future__ import annotations

import logging
import os
import re
from typing import Any, Dict, List

logger = logging.getLogger(__name__)

_SEPARATORS = re.compile(r"[，,；;]|[+＋]|\s+")
_PREFIXES = ("？", "?", "诊断为", "患者")
_SUFFIXES = ("？", "?", "诊断")


class DiagnosisTextProcessor:
    def __init__(self, embedding_service=None, use_enhanced_processing=None, ner_service=None):
        self.medical_separators = [r"[，,；;]", r"[+＋]", r"\s+"]
        self.embedding_service = embedding_service
        if use_enhanced_processing is None:
            use_enhanced_processing = os.getenv("USE_ENHANCED_TEXT_PROCESSING", "true").lower() == "true"
        self.use_enhanced_processing = bool(use_enhanced_processing) and ner_service is not None
        self._enhanced_processor = None
        if self.use_enhanced_processing:
            try:
                from ..services.enhanced_text_processor import EnhancedTextProcessor
                self._enhanced_processor = EnhancedTextProcessor(embedding_service, ner_service)
            except Exception as exc:   # (:55-60: an enhanced processor that does not start leaves the simple mode)
                logger.error("enhanced text processing is off: %s", exc)
                self.use_enhanced_processing = False

    def extract_diagnoses(self, text: str) -> List[str]:
        if not text or not text.strip():
            return []
        if self.use_enhanced_processing and self._enhanced_processor:
            try:
                return self._enhanced_processor.extract_diagnoses_simple(text)
            except Exception as exc:
                logger.warning("enhanced extraction failed, simple extraction instead: %s", exc)
        return self._extract_diagnoses_simple(text)

    def _extract_diagnoses_simple(self, text: str) -> List[str]:
        seen, out = set(), []
        for seg in self._split_by_separators(text):
            clean = self._clean_diagnosis_text(seg)
            if clean and len(clean) >= 2 and clean not in seen:
                seen.add(clean)
                out.append(clean)
        return out

    def _split_by_separators(self, text: str) -> List[str]:
        return [s.strip() for s in _SEPARATORS.split(text) if s and s.strip()]

    def _clean_diagnosis_text(self, text: str) -> str:
        if not text:
            return ""
        text = text.strip()
        for p in _PREFIXES:
            if text.startswith(p):
                text = text[len(p):].strip()
        for s in _SUFFIXES:
            if text.endswith(s):
                text = text[:-len(s)].strip()
        return text

    def is_multi_diagnosis(self, text: str) -> bool:
        return len(self.extract_diagnoses(text)) > 1

    def extract_diagnoses_enhanced(self, text: str, filter_drugs: bool = True) -> List[Dict[str, Any]]:
        if not self.use_enhanced_processing or not self._enhanced_processor:
            return [{"text": t, "diagnosis_confidence": 0.5, "metadata": {"is_simple_extraction": True}}
                    for t in self._extract_diagnoses_simple(text)]
        try:
            return self._enhanced_processor.extract_diagnoses_enhanced(text, filter_drugs=filter_drugs)
        except Exception as exc:
            logger.error("enhanced extraction failed: %s", exc)
            return [{"text": t, "diagnosis_confidence": 0.5, "metadata": {"is_fallback": True}}
                    for t in self._extract_diagnoses_simple(text)]

    def get_processing_mode(self) -> str:
        return "enhanced" if self.use_enhanced_processing and self._enhanced_processor else "simple"
