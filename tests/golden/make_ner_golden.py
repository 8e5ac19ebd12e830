#!/usr/bin/env python3
"""Golden fixture for SURVEY.md row N4 (medical NER service + diagnosis entity filter):

    python tests/golden/make_ner_golden.py

Part 1 - produced by RUNNING THE REFERENCE'S OWN services/medical_ner_service.py and
services/diagnosis_entity_filter.py (unchanged, imported from /root/reference; only loguru is replaced by a no-op
logger): the rule-based extraction, the filter in smart and strict mode and with the other switches, the conversion
of classifier entity groups (`_extract_entities_with_model` over a canned pipeline output), keywords, summaries, stats.

Part 2 - produced by running transformers' own `pipeline("ner", aggregation_strategy="simple")` (the third-party
dependency the reference delegates to, `services/medical_ner_service.py:71-92`; installed in this image) on a seeded
two-layer BertForTokenClassification with a character vocabulary: what `_TokenClassifier` restates. The model's
weights and vocabulary are part of the fixture so that the test rebuilds exactly this model.

Only DATA is written: ner_cases.json, ner_tiny_model.npz.
"""
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

TEXTS = [
    "急性心肌梗死 高血压病 阿司匹林片 药物过敏 青霉素中毒 心脏手术 手术后综合征 心内科 胸痛",
    "2型糖尿病伴有多个并发症，慢性肾功能不全", "反复咳嗽咳痰3年，加重伴气促1周", "左肺上叶腺癌术后化疗",
    "冠状动脉粥样硬化性心脏病；不稳定型心绞痛", "腰椎间盘突出症 L4-5", "待查", "12345", "", "   ",
    "慢性阻塞性肺疾病急性加重期", "肝硬化失代偿期 食管胃底静脉曲张破裂出血", "甲状腺结节，考虑甲状腺癌可能",
    "患者服用阿莫西林胶囊后出现皮疹，考虑药物过敏", "右侧乳腺增生", "脑梗死后遗症 高血压3级 很高危",
    "持续性腹痛伴大量呕吐", "急性阑尾炎，阑尾切除术后", "颈椎病 神经根型", "心悸失眠 头晕头痛",
]


def stub_modules():
    class _Logger:
        def __getattr__(self, name):
            return lambda *a, **k: None
    loguru = types.ModuleType("loguru")
    loguru.logger = _Logger()
    sys.modules["loguru"] = loguru


def plain(obj):
    if isinstance(obj, dict):
        return {k: plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [plain(v) for v in obj]
    if isinstance(obj, (np.floating,)):
        return float(obj)
    if isinstance(obj, (np.integer,)):
        return int(obj)
    return obj


def reference_part():
    stub_modules()
    sys.path.insert(0, REF)
    from services.medical_ner_service import MedicalNERService
    from services.diagnosis_entity_filter import DiagnosisEntityFilter

    strings = [l.strip() for l in open(os.path.join(HERE, "diagnosis_strings.txt"), encoding="utf-8") if l.strip()]
    texts = TEXTS + strings[:40]
    svc = MedicalNERService(use_model=False)
    rules = []
    for t in texts:
        summary = svc.get_entity_summary(t) if t.strip() else None
        if summary:
            summary.pop("model_info")
        rules.append({"text": t,
                      "filtered": plain(svc.extract_medical_entities(t, filter_drugs=True)),
                      "unfiltered": plain(svc.extract_medical_entities(t, filter_drugs=False)),
                      "keywords": svc.identify_diagnosis_keywords(t),
                      "summary": plain(summary),
                      "filter_stats": plain({k: v for k, v in svc.get_filter_stats(t).items() if k != "filter_config"}) if t.strip() else None})
    # the filter on hand-made entity dicts (the reference's own demo data + more types), every switch
    entities = {
        "disease": [{"text": "急性心肌梗死", "start": 0, "end": 6, "confidence": 0.95}, {"text": "高血压病", "start": 7, "end": 11, "confidence": 0.58}],
        "drug": [{"text": "阿司匹林片", "start": 12, "end": 17, "confidence": 0.92}, {"text": "药物过敏", "start": 18, "end": 22, "confidence": 0.85},
                 {"text": "青霉素中毒", "start": 23, "end": 28, "confidence": 0.78}, {"text": "华法林", "start": 60, "end": 63, "confidence": 0.9}],
        "treatment": [{"text": "心脏手术", "start": 29, "end": 33, "confidence": 0.80}, {"text": "手术后综合征", "start": 34, "end": 40, "confidence": 0.75},
                      {"text": "抗凝", "start": 64, "end": 66, "confidence": 0.7}],
        "procedure": [{"text": "支架植入", "start": 67, "end": 71, "confidence": 0.7}],
        "equipment": [{"text": "起搏器综合征", "start": 72, "end": 78, "confidence": 0.7}, {"text": "起搏器", "start": 79, "end": 82, "confidence": 0.9}],
        "inspect_equipment": [{"text": "CT机", "start": 83, "end": 86, "confidence": 0.9}],
        "department": [{"text": "心内科", "start": 41, "end": 44, "confidence": 0.90}],
        "lab_indicator": [{"text": "肌钙蛋白", "start": 87, "end": 91, "confidence": 0.55}, {"text": "血糖", "start": 92, "end": 94, "confidence": 0.45}],
        "symptom": [{"text": "胸痛", "start": 45, "end": 47, "confidence": 0.85}],
        "other": [{"text": "其他", "start": 95, "end": 97, "confidence": 0.61}],
    }
    text = ("急性心肌梗死 高血压病 阿司匹林片 药物过敏 青霉素中毒 心脏手术 手术后综合征 心内科 胸痛" + " " * 12 +
            "华法林 抗凝 支架植入 起搏器综合征 起搏器 CT机 肌钙蛋白 血糖 其他")
    filt = []
    for cfg in ({}, {"strict_mode": True}, {"strict_mode": True, "keep_lab_indicators": False}, {"keep_drug_diseases": False},
                {"enable_context_analysis": False}, {"keep_lab_indicators": False}, {"confidence_threshold": 0.4},
                {"context_window": 2}, {"strict_mode": True, "confidence_threshold": 0.9}):
        f = DiagnosisEntityFilter(dict(cfg))
        out = f.filter_entities(entities, text)
        stats = f.get_filter_stats(entities, out)
        stats.pop("filter_config")
        filt.append({"config": cfg, "out": plain(out), "stats": plain(stats)})
    # conversion of classifier output (canned entity groups through the reference's _extract_entities_with_model)
    canned = [
        {"entity_group": "DiseaseNameOrComprehensiveCertificate", "score": np.float32(0.93), "word": "急 性 心 肌 梗 死", "start": 0, "end": 6},
        {"entity_group": "DiseaseNameOrComprehensiveCertificate", "score": np.float32(0.97), "word": "心 肌 梗 死", "start": 2, "end": 6},
        {"entity_group": "Drug", "score": np.float32(0.88), "word": "阿 司 ##匹 林", "start": 7, "end": 11},
        {"entity_group": "Symptom", "score": np.float32(0.49), "word": "胸 痛", "start": 12, "end": 14},
        {"entity_group": "Symptom", "score": np.float32(0.8), "word": "痛", "start": 15, "end": 16},
        {"entity_group": "UnknownLabel", "score": np.float32(0.7), "word": "未 知", "start": 17, "end": 19},
        {"entity": "B-BodyParts", "score": np.float32(0.66), "word": "左 肺", "start": 20, "end": 22},
        {"entity_group": "Department", "score": np.float32(0.9), "word": "心 内 科"},
    ]
    svc2 = MedicalNERService(use_model=False)
    svc2.use_model = True
    svc2.ner_pipeline = lambda t: canned
    conv = {"groups": plain(canned), "text": "急性心肌梗死 阿司匹林 胸痛 痛 未知 左肺 心内科",
            "converted": plain(svc2._extract_entities_with_model("x")),
            "filtered": plain(svc2.extract_medical_entities("急性心肌梗死 阿司匹林 胸痛 痛 未知 左肺 心内科", filter_drugs=True))}
    return {"rules": rules, "filter": filt, "filter_entities_input": entities, "filter_text": text, "conversion": conv}


def pipeline_part():
    import tempfile
    import torch
    from transformers import BertConfig, BertForTokenClassification, BertTokenizerFast, pipeline
    strings = [l.strip() for l in open(os.path.join(HERE, "diagnosis_strings.txt"), encoding="utf-8") if l.strip()]
    texts = [t for t in TEXTS if t.strip()] + strings[40:70] + ["abc hello 12 mg", "未登录字𪚥测试", "x" * 700]
    chars = sorted({c for t in texts[:-2] for c in t.lower() if not c.isspace()})
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + chars + ["##" + c for c in "abcdefghijklmnopqrstuvwxyz0123456789"] + ["he", "##llo", "mg"]
    vocab = list(dict.fromkeys(vocab))
    d = tempfile.mkdtemp()
    with open(os.path.join(d, "vocab.txt"), "w", encoding="utf-8") as f:
        f.write("\n".join(vocab) + "\n")
    tok = BertTokenizerFast(vocab_file=os.path.join(d, "vocab.txt"), do_lower_case=True, model_max_length=512)
    labels = ["O", "B-Symptom", "I-Symptom", "B-BodyParts", "I-BodyParts", "B-Drug", "I-Drug",
              "B-DiseaseNameOrComprehensiveCertificate", "I-DiseaseNameOrComprehensiveCertificate", "Department"]
    cfg = BertConfig(vocab_size=len(vocab), hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64,
                     max_position_embeddings=512, num_labels=len(labels), id2label=dict(enumerate(labels)),
                     label2id={l: i for i, l in enumerate(labels)})
    torch.manual_seed(20251004)
    model = BertForTokenClassification(cfg).eval()
    with torch.no_grad():
        model.classifier.weight.mul_(40.0)   # wide logit margins: the argmax does not hinge on the last float bit
    ner = pipeline("ner", model=model, tokenizer=tok, aggregation_strategy="simple", device=-1)
    outs = []
    for t in texts:
        outs.append({"text": t, "groups": plain(ner(t))})
    state = {k: v.numpy() for k, v in model.state_dict().items()}
    np.savez_compressed(os.path.join(HERE, "ner_tiny_model.npz"), **state)
    return {"vocab": vocab, "labels": labels, "outputs": outs,
            "config": {"hidden_size": 32, "num_hidden_layers": 2, "num_attention_heads": 2, "intermediate_size": 64}}


def main():
    import transformers
    out = {"reference": reference_part(), "pipeline": pipeline_part(), "transformers": transformers.__version__}
    with open(os.path.join(HERE, "ner_cases.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False)
    print(len(out["reference"]["rules"]), "rule cases,", len(out["reference"]["filter"]), "filter configs,",
          len(out["pipeline"]["outputs"]), "pipeline cases;",
          sum(len(o["groups"]) for o in out["pipeline"]["outputs"]), "entity groups")


if __name__ == "__main__":
    main()
