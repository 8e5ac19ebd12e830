#!/usr/bin/env python3
"""encode_query_batch over the 1 000 golden strings at several sub-batch sizes (padding waste against GEMM size)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("EMBEDDING_MODEL_NAME", "shibing624/text2vec-base-chinese")
os.environ.setdefault("ICD_EMBEDDING_ALLOW_SYNTHETIC", "1")
import torch  # noqa: E402
from rag_project_icd10_amd.services.embedding_service import EmbeddingService  # noqa: E402

strings = [l.strip() for l in open(os.path.join(ROOT, "tests", "golden", "diagnosis_strings.txt"), encoding="utf-8") if l.strip()][:1000]
es = EmbeddingService()
lens = np.sort([len(x) for x in es._tokenize([f"query: {s}" for s in strings])])
print("token lengths: mean %.1f p50 %d p90 %d p99 %d max %d, sum %d" % (lens.mean(), lens[500], lens[900], lens[990], lens[-1], lens.sum()))
pk = es._packed
for packed in (True, False):
    es._packed = pk if packed else None
    for _ in range(3):
        es.encode_query_batch(strings, batch_size=256)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        es.encode_query_batch(strings, batch_size=256, to_device=True)
    torch.cuda.synchronize()
    print(f"{'packed tokens' if packed else 'padded, batch 256'}: {(time.perf_counter() - t0) / 8 * 1e3:6.1f} ms")
from rag_project_icd10_amd.services.embedding_service import _PackedBert
es._packed = pk
for ratio in (0.9, 0.75, 0.6, 0.5, 0.4, 0.3):
    _PackedBert.GROUP_RATIO = ratio
    groups = _PackedBert.plan_groups(sorted((int(x) for x in lens), reverse=True))
    for _ in range(3):
        es.encode_query_batch(strings, batch_size=256)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        es.encode_query_batch(strings, batch_size=256, to_device=True)
    torch.cuda.synchronize()
    print(f"packed, attention groups within {ratio:.2f} of their longest: {len(groups)} groups, padded attention tokens {sum(c * L for _, c, L in groups)}: {(time.perf_counter() - t0) / 8 * 1e3:6.1f} ms")
_PackedBert.GROUP_RATIO = 0.75
es._packed = None
for bs in (64, 128, 256, 512, 1000):
    padded = 0
    order = lens[::-1]
    for s in range(0, 1000, bs):
        padded += order[s] * len(order[s:s + bs])
    for _ in range(3):
        es.encode_query_batch(strings, batch_size=bs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        es.encode_query_batch(strings, batch_size=bs, to_device=True)
    torch.cuda.synchronize()
    print(f"batch_size {bs:5d}: {(time.perf_counter() - t0) / 8 * 1e3:6.1f} ms  padded tokens {padded} ({padded / lens.sum():.2f} x the real ones)")
