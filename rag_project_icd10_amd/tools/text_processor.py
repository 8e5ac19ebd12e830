"""Delimiter-based diagnosis extraction (the "simple" mode of the reference's tools/text_processor.py).

Restates :29-33 (separator table), :87-109 (_extract_diagnoses_simple), :111-122 (_split_by_separators),
:124-148 (_clean_diagnosis_text), :150-154 (is_multi_diagnosis), :156-192 (extract_diagnoses_enhanced in
its no-enhanced-processor form), :194-199 (get_processing_mode). The NER / semantic-boundary "enhanced"
mode is out of scope, so this class always reports mode "simple". Pinned by
tests/golden/text_split_cases.json.
"""
from __future__ import annotations

import re
from typing import Any, Dict, List

_SEPARATORS = re.compile(r"[，,；;]|[+＋]|\s+")
_PREFIXES = ("？", "?", "诊断为", "患者")
_SUFFIXES = ("？", "?", "诊断")


class DiagnosisTextProcessor:
    def __init__(self, embedding_service=None, use_enhanced_processing=None):
        self.medical_separators = [r"[，,；;]", r"[+＋]", r"\s+"]
        self.embedding_service = embedding_service
        self.use_enhanced_processing = False  # enhanced (NER) mode is not part of this build
        self._enhanced_processor = None

    def extract_diagnoses(self, text: str) -> List[str]:
        if not text or not text.strip():
            return []
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
        return [{"text": t, "diagnosis_confidence": 0.5, "metadata": {"is_simple_extraction": True}}
                for t in self._extract_diagnoses_simple(text)]

    def get_processing_mode(self) -> str:
        return "simple"
