#!/usr/bin/env python3
"""Golden fixture for the ENHANCED text mode (entities fused with semantic boundaries; the text side of SURVEY.md row N2):

    python tests/golden/make_text_enhanced_golden.py

Produced by RUNNING THE REFERENCE'S OWN services/semantic_boundary_service.py, services/enhanced_text_processor.py and
tools/text_processor.py (unchanged, imported from /root/reference; only loguru is replaced by a no-op logger), with the
reference's rule-based MedicalNERService (use_model=False: no classifier weights offline) and a deterministic stand-in for the
embedding service (a bag-of-characters vector per string: texts that share characters are close, so the 0.75 cosine threshold of
the boundary confidence is crossed both ways). Only DATA is written: text_enhanced_cases.json.
"""
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

TEXTS = [
    "患者诊断为高血压3级，冠心病，心功能不全", "急性胃肠炎，发热38.5℃，腹泻3天", "2型糖尿病伴有多个并发症；慢性肾功能不全；高血压病",
    "蛋白尿待查 肾功能不全 2型糖尿病伴血糖控制不佳", "肺部阴影 蛋白尿 血肌酐偏高", "肾功能不全，建议进一步检查",
    "高血压病 糖尿病 冠状动脉粥样硬化性心脏病", "慢性肾小球肾炎 尿毒症 贫血", "急性心肌梗死；高血压病；阿司匹林片；药物过敏",
    "冠状动脉粥样硬化性心脏病；不稳定型心绞痛", "慢性阻塞性肺疾病急性加重期", "肝硬化失代偿期+食管胃底静脉曲张破裂出血",
    "甲状腺结节，考虑甲状腺癌可能", "脑梗死后遗症，高血压3级，很高危", "急性阑尾炎，阑尾切除术后，恢复良好", "左肺上叶腺癌术后化疗。既往高血压病史。",
    "胃炎；A", "AB；C", "高血压", "", "   ", "糖尿病\n高血压\n冠心病", "慢性胃炎伴糜烂,十二指肠球部溃疡,幽门螺杆菌感染",
    "高血压病合并糖尿病；继发肾功能不全；慢性心力衰竭急性加重", "患者既往高血压病史10年规律服药控制可", "冠心病＋心律失常＋心功能III级",
    "发热 咳嗽 咳痰 胸痛", "肺炎（社区获得性，重症），呼吸衰竭", "高血压病高血压病高血压病；高血压病", "心悸失眠；头晕头痛；乏力纳差",
    "右侧乳腺增生\t左侧乳腺结节", "腰椎间盘突出症 L4-5；颈椎病 神经根型", "持续性腹痛伴大量呕吐，考虑急性胰腺炎，胆囊结石",
]


def stub_modules():
    class _Logger:
        def __getattr__(self, name):
            return lambda *a, **k: None
    loguru = types.ModuleType("loguru")
    loguru.logger = _Logger()
    sys.modules["loguru"] = loguru


def bag_of_characters(text, dim=48):
    """the stand-in embedding (tests/test_text_enhanced_cpu.py has the same function): float32 unit vector"""
    v = np.zeros(dim, np.float64)
    for ch in text:
        v += np.random.default_rng(ord(ch)).standard_normal(dim)
    n = float(np.sqrt(np.dot(v, v)))
    return (v / n if n > 0 else v + 1.0 / np.sqrt(dim)).astype(np.float32)


class Embedding:
    def __init__(self):
        self.calls = 0

    def encode_query(self, text):
        self.calls += 1
        return bag_of_characters(text)


def plain(obj):
    if isinstance(obj, dict):
        return {k: plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple, set)):
        return [plain(v) for v in (sorted(obj) if isinstance(obj, set) else obj)]
    if isinstance(obj, np.floating):
        return float(obj)
    if isinstance(obj, np.integer):
        return int(obj)
    return obj


def main():
    stub_modules()
    sys.path.insert(0, REF)
    from services.semantic_boundary_service import SemanticBoundaryDetector
    from services.enhanced_text_processor import EnhancedTextProcessor
    from tools.text_processor import DiagnosisTextProcessor

    strings = [l.strip() for l in open(os.path.join(HERE, "diagnosis_strings.txt"), encoding="utf-8") if l.strip()]
    texts = TEXTS + strings[:30] + ["；".join(strings[40:44]), "，".join(strings[50:53]), " ".join(strings[60:65])]
    import random
    rng = random.Random(5)     # seeded combinations of golden strings under every delimiter (and one that is none: 、)
    for _ in range(60):
        texts.append(rng.choice(["；", ";", "，", ",", " ", "+", "。", "\n", "、"]).join(rng.sample(strings, rng.randint(2, 5))))
    emb = Embedding()
    det, det0 = SemanticBoundaryDetector(emb), SemanticBoundaryDetector(None)
    enh, enh0 = EnhancedTextProcessor(emb, use_model_ner=False), EnhancedTextProcessor(None, use_model_ner=False)
    tp = DiagnosisTextProcessor(embedding_service=emb, use_enhanced_processing=True)
    tp._enhanced_processor = enh          # (the reference would build its own with the environment's NER choice: rule-based here, like enh)
    tp.use_enhanced_processing = True
    cases = []
    for t in texts:
        before = emb.calls
        b = det.detect_diagnosis_boundaries(t)
        c = det.get_boundary_confidence(b)
        calls = emb.calls - before
        try:    # (without an embedding service the reference joins segment DICTS at :233 and raises for every multi-segment text)
            b0, b0_raises = det0.detect_diagnosis_boundaries(t), None
        except TypeError as exc:
            b0, b0_raises = None, type(exc).__name__
        summary = enh.get_processing_summary(t) if t.strip() else None
        if summary:
            summary.pop("ner_info")
            summary["entity_types_found"] = sorted(summary["entity_types_found"])
        cases.append({"text": t,
                      "boundaries": plain(b), "confidences": plain(c), "reference_encode_calls": calls,
                      "boundaries_without_embeddings": plain(b0), "boundaries_without_embeddings_raises": b0_raises,
                      "confidences_without_embeddings": plain(det0.get_boundary_confidence(b0)) if b0 is not None else None,
                      "structure": plain(det.analyze_text_structure(t)),
                      "simple_boundaries": plain(enh._simple_boundary_detection(t)) if t.strip() else None,
                      "fallback": plain(enh._fallback_extraction(t)) if t.strip() else None,
                      "enhanced": plain(enh.extract_diagnoses_enhanced(t)),
                      "enhanced_keep_drugs": plain(enh.extract_diagnoses_enhanced(t, filter_drugs=False)),
                      "enhanced_without_embeddings": plain(enh0.extract_diagnoses_enhanced(t)),
                      "summary": plain(summary),
                      "processor_extract": tp.extract_diagnoses(t), "processor_enhanced": plain(tp.extract_diagnoses_enhanced(t)),
                      "processor_mode": tp.get_processing_mode(), "processor_is_multi": bool(tp.is_multi_diagnosis(t))})
    out = os.path.join(HERE, "text_enhanced_cases.json")
    json.dump({"made_by": "tests/golden/make_text_enhanced_golden.py (the reference's classes, rule-based NER, bag-of-characters embeddings)",
               "cases": cases}, open(out, "w", encoding="utf-8"), ensure_ascii=False, indent=0)
    print(len(cases), "cases ->", out, os.path.getsize(out), "bytes;",
          sum(1 for c in cases if len(c["boundaries"]) > 1), "multi-boundary texts;",
          sum(1 for c in cases for x in c["confidences"] if x) , "confidences")


if __name__ == "__main__":
    main()
