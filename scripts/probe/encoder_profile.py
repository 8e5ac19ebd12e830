#!/usr/bin/env python3
"""encode_query_batch over the 1 000 golden diagnosis strings, ten times: the workload of a rocprofv3 --kernel-trace --stats run
(which kernels the 52 ms of the fp32 BERT-base forward are made of)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("EMBEDDING_MODEL_NAME", "shibing624/text2vec-base-chinese")
os.environ.setdefault("ICD_EMBEDDING_ALLOW_SYNTHETIC", "1")
import torch  # noqa: E402
from rag_project_icd10_amd.services.embedding_service import EmbeddingService  # noqa: E402

strings = [l.strip() for l in open(os.path.join(ROOT, "tests", "golden", "diagnosis_strings.txt"), encoding="utf-8") if l.strip()][:1000]
es = EmbeddingService()
for _ in range(3):
    es.encode_query_batch(strings)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    es.encode_query_batch(strings)
torch.cuda.synchronize()
print(f"encode_query_batch(1000): {(time.perf_counter() - t0) / 10 * 1e3:.1f} ms", file=sys.stderr)
