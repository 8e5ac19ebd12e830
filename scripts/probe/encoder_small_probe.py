#!/usr/bin/env python3
"""encoder_small_probe.py - latency of the embed step at the reference's own call shape: ONE string per
EmbeddingService.encode_query call (services/embedding_service.py:117-120), and a request's handful, through
  small   the hand-written small-input forward (csrc/encoder_small.hpp, icd_encoder_encode: one graph launch)
  graph   the framework's forward replayed from a HIP graph per (batch, width) bucket (ICD_EMBEDDING_SMALL=0, round 1-4)
on one MI355X, synthetic BERT-base weights (the real model's shapes). Host latency per call (median of 200), the
GPU time of one forward (device outputs, 200 calls between two events) and the largest difference of the embeddings."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("EMBEDDING_MODEL_NAME", "shibing624/text2vec-base-chinese")


def main():
    import torch
    from rag_project_icd10_amd.services.embedding_service import EmbeddingService
    es = EmbeddingService(allow_synthetic=True, device="cuda")
    small = es._small
    assert small is not None
    strings = [l.strip() for l in open(os.path.join(ROOT, "tests", "golden", "diagnosis_strings.txt"), encoding="utf-8") if l.strip()]
    cases = [("1 string, short (%d tokens)", [strings[0]]), ("1 string, long (%d tokens)", ["高血压" * 30]),
             ("3 strings (%d tokens)", strings[1:4]), ("8 strings (%d tokens)", strings[4:12]), ("14 strings (%d tokens)", strings[12:26])]
    for label, texts in cases:
        ntok = sum(len(x) for x in es._tokenize([f"query: {t}" for t in texts]))
        res = {}
        for mode in ("small", "graph"):
            es._small = small if mode == "small" else None
            for _ in range(10):
                es.encode_query_batch(texts)
            torch.cuda.synchronize()
            lat = []
            for _ in range(200):
                t0 = time.perf_counter()
                v = es.encode_query_batch(texts)
                lat.append((time.perf_counter() - t0) * 1e6)
            lat.sort()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                es.encode_query_batch(texts, to_device=True)
            e1.record()
            torch.cuda.synchronize()
            res[mode] = (lat[100], lat[20], lat[180], e0.elapsed_time(e1) / 200 * 1e3, v)
        es._small = small
        diff = float(np.max(np.abs(res["small"][4] - res["graph"][4])))
        print((label % ntok) + ": " + " | ".join(f"{m}: host call median {r[0]:7.1f} us (p10 {r[1]:.1f}, p90 {r[2]:.1f}), back-to-back {r[3]:7.1f} us per forward"
                                                  for m, r in res.items()) + f" | max |d| {diff:.2e}")
    # the reference's call itself: encode_query(one string) -> numpy
    for mode in ("small", "graph"):
        es._small = small if mode == "small" else None
        for s in strings[:20]:
            es.encode_query(s)
        lat = []
        for i in range(200):
            t0 = time.perf_counter()
            es.encode_query(strings[i % 100])
            lat.append((time.perf_counter() - t0) * 1e6)
        lat.sort()
        print(f"encode_query over 100 golden strings, {mode}: median {lat[100]:.1f} us, p10 {lat[20]:.1f}, p90 {lat[180]:.1f}")
    es._small = small


if __name__ == "__main__":
    main()
