"""Keeps the diagnosis-related entities of an NER result, drops medicines, equipment and departments.

Follows the reference's services/diagnosis_entity_filter.py (the NER service's post-filter):
config from the environment (:62-71), strict mode = a whitelist of entity types above a confidence threshold
(:105-140), smart mode = per-type rules (:142-204): medicines survive only with a diagnosis keyword within
`context_window` characters and are renamed `drug_related_disease` (:206-239), treatments / procedures survive when they
carry a disease suffix or have diagnosis context (:241-275), equipment when it carries a disease suffix (:277-292),
departments never, lab indicators above min(0.5, threshold) when enabled, everything else above the threshold.
The word lists are the reference's data. Pinned by tests/golden/ner_cases.json (made by running the reference).
"""
from __future__ import annotations

import logging
import os
import re
from typing import Any, Dict, List

logger = logging.getLogger(__name__)

DRUG_DIAGNOSIS_KEYWORDS = frozenset((
    "过敏", "中毒", "不良反应", "副作用", "依赖", "滥用", "耐药", "抗药性", "药物性", "中毒性", "戒断", "成瘾",
    "肝毒性", "肾毒性", "心脏毒性", "神经毒性"))
CONTEXT_KEYWORDS = DRUG_DIAGNOSIS_KEYWORDS | frozenset((
    "诊断", "疑似", "考虑", "排除", "病史", "既往史", "症状", "表现", "发作", "急性", "慢性", "复发",
    "并发症", "合并症", "继发", "原发"))
# medicine names: dosage-form and drug-class suffixes, common prefixes, release forms (re.match: anchored at the start)
DRUG_NAME = re.compile(
    r"(?:.*(?:片|胶囊|注射液|口服液|颗粒|软膏|滴眼液|喷雾剂|素|霉素|西林|沙星|洛尔|普利|沙坦|司汀)$)"
    r"|(?:^[阿氨左右].*)|(?:.*(?:缓释|控释|肠溶).*)")
PURE_TREATMENT = re.compile(r".*(?:手术|切除术|造影|穿刺|化疗|放疗|康复|训练|护理|检查|监测)$")
DISEASE_SUFFIXES = ("病", "症", "炎", "癌", "瘤", "综合征", "性疾病", "功能不全", "功能障碍", "衰竭", "梗死", "出血",
                    "破裂", "穿孔", "狭窄", "扩张", "增生", "萎缩")
STRICT_TYPES = frozenset(("disease", "symptom", "anatomy", "pathology", "injury", "sign", "microbiology"))


def _env_flag(name: str, default: str) -> bool:
    return os.getenv(name, default).lower() == "true"


class DiagnosisEntityFilter:
    def __init__(self, config: Dict = None):
        self.config = self._get_default_config()
        if config:
            self.config.update(config)
        self.drug_diagnosis_keywords = set(DRUG_DIAGNOSIS_KEYWORDS)
        self.disease_suffixes = set(DISEASE_SUFFIXES)

    def _get_default_config(self) -> Dict:
        return {
            "strict_mode": _env_flag("DIAGNOSIS_FILTER_STRICT_MODE", "false"),
            "keep_drug_diseases": _env_flag("KEEP_DRUG_DISEASES", "true"),
            "keep_lab_indicators": _env_flag("KEEP_LAB_INDICATORS", "true"),
            "context_window": int(os.getenv("FILTER_CONTEXT_WINDOW", "20")),
            "confidence_threshold": float(os.getenv("FILTER_CONFIDENCE_THRESHOLD", "0.6")),
            "enable_context_analysis": _env_flag("ENABLE_CONTEXT_ANALYSIS", "true"),
        }

    # ---- entry point -------------------------------------------------------------------------------------------
    def filter_entities(self, entities: Dict[str, List[Dict]], original_text: str) -> Dict[str, List[Dict]]:
        if not entities:
            return {}
        if self.config["strict_mode"]:
            return self._strict_filter(entities)
        return self._smart_filter(entities, original_text)

    def _confident(self, entity_list: List[Dict], threshold: float) -> List[Dict]:
        return [e for e in entity_list if e.get("confidence", 0) >= threshold]

    def _filter_by_confidence(self, entity_list: List[Dict]) -> List[Dict]:
        return self._confident(entity_list, self.config["confidence_threshold"])

    def _strict_filter(self, entities: Dict[str, List[Dict]]) -> Dict[str, List[Dict]]:
        allowed = set(STRICT_TYPES)
        if self.config["keep_lab_indicators"]:
            allowed.add("lab_indicator")
        out = {}
        for kind, items in entities.items():
            kept = self._filter_by_confidence(items) if kind in allowed else []
            if kept:
                out[kind] = kept
        return out

    def _smart_filter(self, entities: Dict[str, List[Dict]], text: str) -> Dict[str, List[Dict]]:
        out = {}
        for kind, items in entities.items():
            if kind == "drug":
                key, kept = "drug_related_disease", self._filter_drug_entities(items, text)
            elif kind in ("treatment", "procedure"):
                key, kept = f"{kind}_related_disease", self._filter_treatment_entities(items, text)
            elif kind in ("equipment", "inspect_equipment"):
                key, kept = f"{kind}_related", self._filter_equipment_entities(items, text)
            elif kind == "department":
                continue
            elif kind == "lab_indicator":
                if not self.config["keep_lab_indicators"]:
                    continue
                key, kept = kind, self._confident(items, min(0.5, self.config["confidence_threshold"]))
            else:
                key, kept = kind, self._filter_by_confidence(items)
            if kept:
                out[key] = kept
        return out

    # ---- per-type rules ----------------------------------------------------------------------------------------
    def _filter_drug_entities(self, entity_list: List[Dict], text: str) -> List[Dict]:
        if not self.config["keep_drug_diseases"]:
            return []
        kept = []
        for entity in entity_list:
            name = entity["text"]
            if DRUG_NAME.match(name):
                continue
            if self.config["enable_context_analysis"]:
                if self._has_diagnosis_context(entity, text):
                    kept.append(entity)
            elif self._has_disease_characteristics(name):
                kept.append(entity)
        return kept

    def _filter_treatment_entities(self, entity_list: List[Dict], text: str) -> List[Dict]:
        kept = []
        for entity in entity_list:
            name = entity["text"]
            if self._has_disease_characteristics(name):
                kept.append(entity)
            elif not PURE_TREATMENT.match(name) and self.config["enable_context_analysis"] \
                    and self._has_diagnosis_context(entity, text):
                kept.append(entity)
        return kept

    def _filter_equipment_entities(self, entity_list: List[Dict], text: str) -> List[Dict]:
        return [e for e in entity_list if self._has_disease_characteristics(e["text"])]

    def _has_diagnosis_context(self, entity: Dict, text: str) -> bool:
        start = entity.get("start", 0)
        end = entity.get("end", len(entity["text"]))
        window = self.config["context_window"]
        context = text[max(0, start - window):min(len(text), end + window)]
        return any(word in context for word in CONTEXT_KEYWORDS)

    def _has_disease_characteristics(self, entity_text: str) -> bool:
        return any(suffix in entity_text for suffix in DISEASE_SUFFIXES)

    # ---- statistics (:302-327) ---------------------------------------------------------------------------------
    def get_filter_stats(self, original_entities: Dict, filtered_entities: Dict) -> Dict[str, Any]:
        before = sum(len(v) for v in original_entities.values())
        after = sum(len(v) for v in filtered_entities.values())
        dropped = {}
        for kind, items in original_entities.items():
            left = len(filtered_entities.get(kind, []))
            renamed = [k for k in filtered_entities if k.startswith(kind)]   # (the type's own key counts again: as the reference)
            if renamed:
                left += sum(len(filtered_entities[k]) for k in renamed)
            if len(items) - left > 0:
                dropped[kind] = len(items) - left
        return {"original_total": before, "filtered_total": after, "filtered_out_total": before - after,
                "filtered_out_by_type": dropped, "filter_config": self.config,
                "filter_efficiency": (before - after) / before if before > 0 else 0}
