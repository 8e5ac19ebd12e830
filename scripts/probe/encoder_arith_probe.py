#!/usr/bin/env python3
"""The canonical encoder's two arithmetics (ICD_ENCODER_ARITH): fp32-input MFMAs against the split-bf16 form (three bf16 MFMAs per
32 k-values) - accuracy against the framework's fp32 forward of the same weights on the GPU, one string per call and in batches,
bit-equality of the two call shapes under each arithmetic, and ms per 1 000 golden strings / per encode_query call."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("EMBEDDING_MODEL_NAME", "shibing624/text2vec-base-chinese")
os.environ["ICD_EMBEDDING_ALLOW_SYNTHETIC"] = "1"


def main():
    import torch
    from rag_project_icd10_amd.services.embedding_service import EmbeddingService
    strings = [l.strip() for l in open(os.path.join(ROOT, "tests", "golden", "diagnosis_strings.txt"), encoding="utf-8") if l.strip()][:1000]
    ref = None
    vecs = {}
    for arith in ("fp32", "bf16x3"):
        os.environ["ICD_ENCODER_ARITH"] = arith
        es = EmbeddingService(allow_synthetic=True, device="cuda")
        assert es._small is not None and es._small.arithmetic == arith
        if ref is None:   # the framework's padded fp32 forward (no hand-written encoder, no packed one)
            small, packed = es._small, es._packed
            es._small, es._packed = None, None
            ref = es.encode_query_batch(strings, batch_size=32)
            es._small, es._packed = small, packed
        one = np.stack([es.encode_query(t) for t in strings[:200]])
        es.encode_query_batch(strings, to_device=True)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            v = es.encode_query_batch(strings, to_device=True)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        many = v.cpu().numpy()
        lat = []
        for i in range(200):
            t0 = time.perf_counter()
            es.encode_query(strings[i % 100])
            lat.append((time.perf_counter() - t0) * 1e6)
        vecs[arith] = many
        cos_ref = ref.astype(np.float64) @ ref.astype(np.float64).T
        cos = many.astype(np.float64) @ many.astype(np.float64).T
        print(f"{arith}: batch of 1000 strings {sorted(ts)[2]:.1f} ms; encode_query median {sorted(lat)[100]:.0f} us; max |d vector| vs the framework's fp32 forward "
              f"{float(np.max(np.abs(many - ref))):.2e}; max |d cosine| over all pairs {float(np.max(np.abs(cos - cos_ref))):.2e}; "
              f"batch rows == one-string rows: {bool(np.array_equal(one, many[:200]))}", flush=True)
        del es
        torch.cuda.empty_cache()
    print(f"max |d vector| between the two arithmetics {float(np.max(np.abs(vecs['fp32'] - vecs['bf16x3']))):.2e}")


if __name__ == "__main__":
    main()
