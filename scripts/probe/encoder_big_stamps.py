#!/usr/bin/env python3
"""encoder_big_stamps.py [NSTRINGS] - where a work-group of the batch form's GEMMs spends its life: clock stamps (s_memtime =
shader clock, s_memrealtime = 100 MHz) of wave 0 of a work-group in the middle of the grid, last layer's four GEMMs, from a
DIAGNOSTIC build (make -C rag_project_icd10_amd/csrc ABLATE=1 OUT=abe; the drains the stamps need cost the wave its overlap of
the first loads with the set-up arithmetic, nothing else)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("EMBEDDING_MODEL_NAME", "shibing624/text2vec-base-chinese")
os.environ["ICD_SEARCH_LIB"] = os.path.join(ROOT, "rag_project_icd10_amd", "csrc", "abe", "libicdsearch.so")
os.environ["ICD_ENC_STAMPS"] = "1"


def main():
    import torch
    from rag_project_icd10_amd import _native
    from rag_project_icd10_amd.services.embedding_service import EmbeddingService
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 450
    es = EmbeddingService(allow_synthetic=True, device="cuda")
    lib = _native.load_library()
    strings = [l.strip() for l in open(os.path.join(ROOT, "tests", "golden", "diagnosis_strings.txt"), encoding="utf-8") if l.strip()][:n]
    ids = es._tokenize([f"query: {t}" for t in strings])
    print(f"{len(ids)} strings, {sum(len(x) for x in ids)} tokens in one pass of the batch form")
    for _ in range(5):
        es._small.encode_many(ids, to_device=True)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    lib.icd_debug_encoder_stamps.argtypes = [C.c_void_p, C.c_void_p]
    assert lib.icd_debug_encoder_stamps(es._small._h, buf) == 0
    st = np.array(buf, dtype=np.uint64).reshape(4, 8, 2).astype(np.int64)
    names = ["QKV (24 K steps)", "attention output (24 K steps)", "FFN up (24 K steps, erf-GELU)", "FFN down (96 K steps in four slices)"]
    labels = ["first kernel argument -> bias, first fragments, epilogue inputs requested", "-> all of them here", "-> first piece (6 of the K steps) done",
              "-> K walk done", "-> epilogue arithmetic done", "-> stores acknowledged"]
    for g in range(4):
        print(names[g])
        for i in range(6):
            dc, dr = st[g, i + 1, 0] - st[g, i, 0], st[g, i + 1, 1] - st[g, i, 1]
            print(f"   {labels[i]:76s} {dc:8d} cycles {dr / 100.0:7.2f} us")
        tc, tr = st[g, 6, 0] - st[g, 0, 0], st[g, 6, 1] - st[g, 0, 1]
        print(f"   {'wave 0 in all':76s} {tc:8d} cycles {tr / 100.0:7.2f} us  ({tc / max(tr, 1) * 100:.0f} MHz)")


if __name__ == "__main__":
    main()
