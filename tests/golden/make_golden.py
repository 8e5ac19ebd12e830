#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference's own modules.

Run in the build container only (needs /root/reference; never on the GPU box):

    python tests/golden/make_golden.py

The reference's pure-Python parts (CSV -> records, hierarchy parsing, semantic text, hierarchical
rescoring, uncertainty rules, delimiter splitting) are executed as-is; its missing third-party
imports (loguru, dotenv, sentence_transformers, pymilvus, tqdm is present) are replaced by in-memory
stubs that are never called on these code paths. Only DATA (inputs and the reference's outputs) is
written to the fixtures - no reference source text.

Fixtures written:
  csv_slice.csv            a slice of the reference's data file data/ICD_10v601.csv (data, with BOM/CRLF kept)
  csv_records.json         DatabaseBuilder.load_csv_data(csv_slice.csv) -> records
  csv_full_digest.json     digest of load_csv_data over the FULL reference CSV (count, level histogram, sha256)
  hier_cases.json          HierarchicalSimilarityService.batch_calculate_similarities I/O
  uncertainty_cases.json   UncertaintyDiagnosisService.detect_uncertainty / process_uncertainty_query I/O
  text_split_cases.json    DiagnosisTextProcessor._extract_diagnoses_simple I/O
  diagnosis_strings.txt    1000 disease names sampled from the CSV (default_rng(2025)), half perturbed
                           (SURVEY.md section 8d config 1/3 workload)
"""
import dataclasses
import hashlib
import json
import os
import sys
import types

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _stub_modules():
    class _Logger:
        def __getattr__(self, name):
            return lambda *a, **k: None

    loguru = types.ModuleType("loguru")
    loguru.logger = _Logger()
    sys.modules["loguru"] = loguru
    dotenv = types.ModuleType("dotenv")
    dotenv.load_dotenv = lambda *a, **k: None
    sys.modules["dotenv"] = dotenv
    st = types.ModuleType("sentence_transformers")
    st.SentenceTransformer = object
    sys.modules["sentence_transformers"] = st
    pm = types.ModuleType("pymilvus")
    pm.MilvusClient = object
    pm.DataType = object
    sys.modules["pymilvus"] = pm


def _np_default(o):
    import numpy as np
    if isinstance(o, np.floating):
        return float(o)
    if isinstance(o, np.integer):
        return int(o)
    if isinstance(o, np.ndarray):
        return o.tolist()
    raise TypeError(type(o))


def main():
    import numpy as np

    _stub_modules()
    sys.path.insert(0, REF)
    os.chdir("/tmp")  # DatabaseBuilder adds a log sink relative to cwd (stubbed, harmless)
    from services.hierarchical_similarity_service import HierarchicalSimilarityService
    from services.uncertainty_diagnosis_service import UncertaintyDiagnosisService
    from tools.build_database import DatabaseBuilder
    from tools.text_processor import DiagnosisTextProcessor

    csv_path = os.path.join(REF, "data", "ICD_10v601.csv")
    raw = open(csv_path, "rb").read()
    lines = raw.split(b"\r\n")
    header, body = lines[0], [l for l in lines[1:] if l]

    # ---- CSV slice: head, plus rows exercising combo codes, morphology codes, quoted commas ----------
    picked = list(range(0, 260))
    text_lines = [l.decode("utf-8") for l in body]
    combo = [i for i, l in enumerate(text_lines) if "+" in l.split(",")[0] and "*" in l.split(",")[0]][:12]
    morph = [i for i, l in enumerate(text_lines) if l.startswith("M8")][:12]
    for i in combo + morph:
        # keep ancestors so the hierarchy text has parents to cite
        code = text_lines[i].split(",")[0]
        stem = code.split(".")[0]
        anc = [j for j, l in enumerate(text_lines[:i]) if l.split(",")[0] in (stem, stem + "." + code.split(".")[1][:1] if "." in code else stem)]
        picked.extend(anc + [i])
    picked = sorted(set(picked))
    slice_bytes = header + b"\r\n" + b"\r\n".join(body[i] for i in picked) + b"\r\n"
    open(os.path.join(HERE, "csv_slice.csv"), "wb").write(slice_bytes)

    builder = DatabaseBuilder()
    recs = builder.load_csv_data(os.path.join(HERE, "csv_slice.csv"))
    json.dump(recs, open(os.path.join(HERE, "csv_records.json"), "w"), ensure_ascii=False, indent=0)

    full = builder.load_csv_data(csv_path)
    hist = {}
    for r in full:
        hist[str(r["level"])] = hist.get(str(r["level"]), 0) + 1
    digest = {
        "count": len(full),
        "level_histogram": hist,
        "has_complication": sum(1 for r in full if r["has_complication"]),
        "sha256_semantic_text": hashlib.sha256("\n".join(r["semantic_text"] for r in full).encode()).hexdigest(),
        "sha256_codes": hashlib.sha256("\n".join(r["code"] for r in full).encode()).hexdigest(),
        "sha256_parent_codes": hashlib.sha256("\n".join(r["parent_code"] for r in full).encode()).hexdigest(),
        "batch_size_rule": {str(n): builder._calculate_optimal_batch_size(n) for n in (1, 999, 1000, 9999, 10000, 40474, 49999, 50000)},
        "first": full[0], "third": full[2],
    }
    json.dump(digest, open(os.path.join(HERE, "csv_full_digest.json"), "w"), ensure_ascii=False, indent=1)

    # ---- diagnosis strings (workload of configs 1 and 3) ------------------------------------------------
    rng = np.random.default_rng(2025)
    names = [r["preferred_zh"] for r in full]
    idx = rng.choice(len(names), size=1000, replace=False)
    out = []
    for j, i in enumerate(idx):
        s = names[int(i)]
        if j % 2 == 1:
            s = s[:-1] if (j % 4 == 1 and len(s) > 2) else s + "待查"
        out.append(s.replace("\n", " "))
    open(os.path.join(HERE, "diagnosis_strings.txt"), "w", encoding="utf-8").write("\n".join(out) + "\n")

    # ---- hierarchical rescoring -------------------------------------------------------------------------
    class FakeEmbedder:
        """deterministic 16-d embedder so the flattened-record branch (2 encodes) is reproducible"""

        def __init__(self):
            self.calls = 0

        def encode_query(self, text):
            self.calls += 1
            h = hashlib.sha256(("query: " + text).encode()).digest()
            v = np.frombuffer(h[:16], dtype=np.uint8).astype(np.float32) - 127.5
            return v / np.linalg.norm(v)

    def live(code, title, score, orig, level, parent, sem):
        return {"code": code, "title": title, "score": score, "original_score": orig,
                "metadata": {"has_complication": False, "main_code": code, "secondary_code": "", "level": level,
                             "parent_code": parent, "category_path": "", "semantic_text": sem}}

    def flat(code, title, score, level, parent, sem):
        return {"code": code, "preferred_zh": title, "score": score, "level": level, "parent_code": parent,
                "category_path": "", "semantic_text": sem}

    cand_sets = {
        "live_mi": [live("I21.9", "急性心肌梗死，未特指", 0.68, 0.85, 3, "I21", "急性心肌梗死，未特指 | 急性心肌梗死 | ICD-10: I21.9"),
                    live("I21", "急性心肌梗死", 1.02, 0.85, 1, "", "急性心肌梗死 | ICD-10: I21"),
                    live("I47.9", "阵发性心动过速，未特指", 0.576, 0.72, 3, "I47", "x"),
                    live("K29.7", "胃炎，未特指", 0.5, 0.5, 2, "K29", "y"),
                    live("S06.9", "颅内损伤，未特指", 0.97, 0.97, 2, "S06", "z"),
                    live("Z99", "依赖于机器", 0.31, 0.31, 1, "", "w")],
        "flat_mi": [flat("I21.9", "急性心肌梗死，未特指", 0.85, 3, "I21", "急性心肌梗死，未特指 | 急性心肌梗死 | ICD-10: I21.9"),
                    flat("I47.9", "阵发性心动过速，未特指", 0.72, 3, "I47", "阵发性心动过速，未特指 | ICD-10: I47.9"),
                    flat("I25.9", "慢性缺血性心脏病，未特指", 0.68, 3, "I25", "慢性缺血性心脏病 | ICD-10: I25.9"),
                    flat("I21", "急性心肌梗死", 0.96, 1, "", "急性心肌梗死 | ICD-10: I21"),
                    flat("A09.9", "胃肠炎", 0.2, 2, "A09", "")],
        "empty": [],
    }
    queries = [
        ("急性心肌梗死", {"disease": [{"text": "急性心肌梗死", "confidence": 0.95}]}),
        ("急性心肌梗死伴心律失常", {"disease": [{"text": "急性心肌梗死", "confidence": 0.9}, {"text": "心律失常", "confidence": 0.8}],
                         "symptom": [{"text": "胸痛", "confidence": 0.7}], "anatomy": [{"text": "心脏", "confidence": 0.6}]}),
        ("颅内损伤待查", {}),
        ("疑似胃炎？", {"disease": [{"text": "胃炎", "confidence": 0.5}]}),
        ("高血压 糖尿病", {"disease": [{"text": "高血压 糖尿病"}]}),
    ]
    hier = []
    for with_embedder in (True, False):
        for setname, cands in cand_sets.items():
            for q, ents in queries:
                emb = FakeEmbedder() if with_embedder else None
                svc = HierarchicalSimilarityService(embedding_service=emb)
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    res = svc.batch_calculate_similarities(q, ents, json.loads(json.dumps(cands)))
                hier.append({
                    "with_embedder": with_embedder, "set": setname, "query": q, "entities": ents, "candidates": cands,
                    "encode_calls": emb.calls if emb else 0,
                    "out": [{"code": r.get("code"), "enhanced_score": float(s), "record_enhanced_score": r.get("enhanced_score"),
                             "original_score": r.get("original_score"), "score": r.get("score"),
                             "uncertainty_boost": r.get("uncertainty_boost"), "uncertainty_weight": r.get("uncertainty_weight"),
                             "factors": dataclasses.asdict(f)} for r, s, f in res],
                })
    json.dump(hier, open(os.path.join(HERE, "hier_cases.json"), "w"), ensure_ascii=False, indent=0, default=_np_default)

    # ---- uncertainty ---------------------------------------------------------------------------------------
    unc = UncertaintyDiagnosisService()
    ustrings = ["颅内损伤待查", "疑似肺炎", "肺炎", "考虑急性阑尾炎可能", "发热原因不明", "胃炎？", "排除结核",
                "高血压", "不能排除恶性肿瘤，待确诊", "  待查  ", "Possible 肺炎?"]
    ucases = {"detect": [], "process": []}
    for s in ustrings:
        ucases["detect"].append({"text": s, "out": unc.detect_uncertainty(s)})
    ucands = [
        {"code": "S06.9", "preferred_zh": "颅内损伤，未特指", "score": 0.7},
        {"code": "S06.900", "preferred_zh": "未特指的颅内损伤", "score": 0.69},
        {"code": "S06.8", "preferred_zh": "其他颅内损伤", "score": 0.71},
        {"code": "S06.2", "preferred_zh": "弥漫性脑损伤", "score": 0.65},
        {"code": "J18.9", "title": "肺炎，未特指", "score": 0.6},          # live-shaped: no preferred_zh (F8)
        {"code": "J18.901", "title": "肺炎", "score": 0.62},
        {"code": "J15", "title": "细菌性肺炎", "score": 0.5},
    ]
    for s in ustrings:
        cq, outc = unc.process_uncertainty_query(s, json.loads(json.dumps(ucands)))
        ucases["process"].append({"text": s, "candidates": ucands, "clean_query": cq, "out": outc})
    json.dump(ucases, open(os.path.join(HERE, "uncertainty_cases.json"), "w"), ensure_ascii=False, indent=0, default=_np_default)

    # ---- delimiter splitting (reduced /query pipeline) ----------------------------------------------------------
    tp = DiagnosisTextProcessor(use_enhanced_processing=False)
    texts = ["急性胃肠炎", "高血压，糖尿病；冠心病", "高血压+糖尿病", "患者高血压 诊断为糖尿病？", "肺炎, 肺炎 ;肺炎",
             "？待查", "a", "", "   ", "胃炎诊断", "２型糖尿病＋高血压", "诊断为肺炎?"]
    json.dump([{"text": t, "out": tp._extract_diagnoses_simple(t), "multi": len(tp._extract_diagnoses_simple(t)) > 1} for t in texts],
              open(os.path.join(HERE, "text_split_cases.json"), "w"), ensure_ascii=False, indent=0)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
