#!/usr/bin/env python3
"""The batch form of the canonical encoder (csrc/encoder_big.hpp) under rocprofv3 --kernel-trace --stats: the 1 000 golden strings
(18 290 tokens) through encode_query_batch, 10 timed passes. Prints ms per pass; the kernel table comes from rocprofv3."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("EMBEDDING_MODEL_NAME", "shibing624/text2vec-base-chinese")
os.environ["ICD_EMBEDDING_ALLOW_SYNTHETIC"] = "1"
import torch
from rag_project_icd10_amd.services.embedding_service import EmbeddingService
es = EmbeddingService(allow_synthetic=True, device="cuda")
strings = [l.strip() for l in open(os.path.join(ROOT, "tests", "golden", "diagnosis_strings.txt"), encoding="utf-8") if l.strip()][:1000]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
texts = (strings * ((n + 999) // 1000))[:n]
for _ in range(3):
    es.encode_query_batch(texts, to_device=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    es.encode_query_batch(texts, to_device=True)
torch.cuda.synchronize()
print(f"{n} strings: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per pass")
