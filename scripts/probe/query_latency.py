#!/usr/bin/env python3
"""Latency of ONE /query-shaped request (MultiDiagnosisService.match_multiple_diagnoses) on one MI355X, synthetic encoder /
NER weights, 40 474-row corpus: texts of 1, 3 and 8 diagnoses, top_k = 5 (search k = 10), without and with the NER service."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("EMBEDDING_MODEL_NAME", "shibing624/text2vec-base-chinese")
os.environ.setdefault("ICD_EMBEDDING_ALLOW_SYNTHETIC", "1")
os.environ.setdefault("ICD_NER_ALLOW_SYNTHETIC", "1")
os.environ.setdefault("MEDICAL_NER_MODEL", "/nonexistent/ner")
tmp = tempfile.mkdtemp(prefix="icd_q_")
os.environ["MILVUS_DB_PATH"] = os.path.join(tmp, "db")
os.environ["MILVUS_COLLECTION_NAME"] = "icd10_q"


def main():
    import torch
    from rag_project_icd10_amd.services.embedding_service import EmbeddingService
    from rag_project_icd10_amd.services.medical_ner_service import MedicalNERService
    from rag_project_icd10_amd.services.milvus_service import MilvusService
    from rag_project_icd10_amd.services.multi_diagnosis_service import MultiDiagnosisService
    n, dim = 40474, 768
    es = EmbeddingService()
    ms = MilvusService(embedding_service=es)
    rng = np.random.default_rng(1234)
    corpus = rng.standard_normal((n, dim), dtype=np.float32)
    corpus /= np.linalg.norm(corpus, axis=1, keepdims=True)
    for s in range(0, n, 4096):
        recs = [{"code": f"S{i:05d}.{i % 9}", "preferred_zh": f"合成疾病{i}", "level": 1 + i % 3, "parent_code": "",
                 "category_path": f"S{i:05d}", "semantic_text": f"合成疾病{i}"} for i in range(s, min(n, s + 4096))]
        assert ms.insert_records(recs, list(corpus[s:s + 4096]))
    assert ms.load_collection()
    strings = [l.strip() for l in open(os.path.join(ROOT, "tests", "golden", "diagnosis_strings.txt"), encoding="utf-8") if l.strip()]
    for ner in (None, MedicalNERService()):
        md = MultiDiagnosisService(es, ms, ner_service=ner)
        for nd in (1, 3, 8):
            texts = ["，".join(strings[i * nd:(i + 1) * nd]) for i in range(40)]
            for t in texts[:5]:
                md.match_multiple_diagnoses(t, top_k=5)
            torch.cuda.synchronize()
            lat = []
            for t in texts:
                t0 = time.perf_counter()
                out = md.match_multiple_diagnoses(t, top_k=5)
                lat.append((time.perf_counter() - t0) * 1e3)
            lat.sort()
            print(f"NER {'on ' if ner else 'off'} {nd} diagnoses per request ({len(out['extracted_diagnoses'])} extracted, {out['total_matches']} candidates): "
                  f"median {lat[len(lat) // 2]:.2f} ms, p90 {lat[int(len(lat) * 0.9)]:.2f} ms")


if __name__ == "__main__":
    main()
