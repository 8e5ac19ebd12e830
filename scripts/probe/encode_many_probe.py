#!/usr/bin/env python3
"""What the canonical batch path costs next to the packed split-bf16 one: the 1 000 golden diagnosis
strings through encode_query_batch - canonical (icd_encoder_encode_many: the small-input kernels cut into calls, bit-identical
to one string per call) and fast (ICD_EMBEDDING_BATCH=fast) - and 10 000 corpus-shaped strings (the build's shape).
Synthetic BERT-base weights. Prints ms per batch (median of 5) and the max |d vector| between the two."""
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
    es = EmbeddingService(allow_synthetic=True, device="cuda")
    strings = [l.strip() for l in open(os.path.join(ROOT, "tests", "golden", "diagnosis_strings.txt"), encoding="utf-8") if l.strip()][:1000]
    corpus_like = [f"{s} | {strings[(i * 7) % 1000]} | {strings[(i * 13) % 1000]} | ICD-10: A{i % 100:02d}.{i % 10}" for i, s in enumerate(strings * 10)]
    for name, texts in (("1000 golden strings", strings), ("10000 corpus-shaped strings", corpus_like)):
        toks = sum(len(x) for x in es._tokenize([f"query: {t}" for t in texts]))
        res = {}
        for mode, fast in (("canonical", False), ("fast", True)):
            es.encode_query_batch(texts, to_device=True, fast=fast)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                v = es.encode_query_batch(texts, to_device=True, fast=fast)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            res[mode] = (sorted(ts)[2], v.cpu().numpy())
        d = float(np.max(np.abs(res["canonical"][1] - res["fast"][1])))
        print(f"{name}: {toks} tokens; canonical {res['canonical'][0]:.1f} ms ({res['canonical'][0] * 1e3 / toks:.2f} us/token), "
              f"fast {res['fast'][0]:.1f} ms ({res['fast'][0] * 1e3 / toks:.2f} us/token); max |d vector| {d:.2e}", flush=True)
    one = np.stack([es.encode_query(t) for t in strings[:200]])
    print("canonical batch rows == encode_query rows:", bool(np.array_equal(one, es.encode_query_batch(strings)[:200])))


if __name__ == "__main__":
    main()
