#!/usr/bin/env python3
"""encoder_small_profile.py NTOK [NSEQ] - 200 forwards of the small-input encoder on NSEQ sequences of NTOK tokens in all
(for rocprofv3 --kernel-trace --stats: the per-kernel durations of one forward)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("EMBEDDING_MODEL_NAME", "shibing624/text2vec-base-chinese")


def main():
    import torch
    from rag_project_icd10_amd.services.embedding_service import EmbeddingService
    ntok = int(sys.argv[1]) if len(sys.argv) > 1 else 15
    nseq = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    es = EmbeddingService(allow_synthetic=True, device="cuda")
    rng = np.random.default_rng(1)
    per = ntok // nseq
    ids = [[101] + [int(v) for v in rng.integers(1000, 21000, size=per - 2)] + [102] for _ in range(nseq)]
    for _ in range(200):
        es._small.encode(ids, to_device=True)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
