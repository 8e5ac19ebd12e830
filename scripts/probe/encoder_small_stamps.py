#!/usr/bin/env python3
"""encoder_small_stamps.py [NTOK] - where the time of a small-input GEMM kernel goes: clock stamps (s_memtime = shader clock,
s_memrealtime = 100 MHz) of wave 0 of work-group 1 in the four GEMMs of layer 6, from a DIAGNOSTIC build
(make -C rag_project_icd10_amd/csrc ABLATE=1 OUT=abe; ICD_SEARCH_LIB=.../abe/libicdsearch.so ICD_ENC_STAMPS=1)."""
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
    ntok = int(sys.argv[1]) if len(sys.argv) > 1 else 15
    es = EmbeddingService(allow_synthetic=True, device="cuda")
    lib = _native.load_library()
    rng = np.random.default_rng(1)
    ids = [[101] + [int(v) for v in rng.integers(1000, 21000, size=ntok - 2)] + [102]]
    for _ in range(300):
        es._small.encode(ids, to_device=True)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    lib.icd_debug_encoder_stamps.argtypes = [C.c_void_p, C.c_void_p]
    assert lib.icd_debug_encoder_stamps(es._small._h, buf) == 0
    st = np.array(buf, dtype=np.uint64).reshape(4, 8, 2).astype(np.int64)
    names = ["QKV (LayerNorm folded in; 144 WG x 256, 48 operand loads per lane: four slabs)", "attention output (96 WG x 256)",
             "FFN up (LayerNorm folded in, GELU; 192 WG x 256)", "FFN down (K split four ways: 192 WG x 256)"]
    labels = ["first kernel argument -> loads issued", "-> loads landed", "-> MFMAs issued, statistics done", "-> partial sums in LDS", "-> barrier passed", "-> epilogue stored"]
    print(f"{ntok} tokens; per segment: shader-clock cycles, microseconds by the 100 MHz counter, and the clock they imply")
    for g in range(4):
        print(names[g])
        print(f"   first instruction -> first kernel argument read  {st[g, 0, 0] - st[g, 7, 0]:8d} cycles {(st[g, 0, 1] - st[g, 7, 1]) / 100.0:7.2f} us")
        for i in range(6):
            dc, dr = st[g, i + 1, 0] - st[g, i, 0], st[g, i + 1, 1] - st[g, i, 1]
            print(f"   {labels[i]:32s} {dc:8d} cycles {dr / 100.0:7.2f} us" + (f"  ({dc / dr * 100:.0f} MHz)" if dr > 20 else ""))
        tc, tr = st[g, 6, 0] - st[g, 0, 0], st[g, 6, 1] - st[g, 0, 1]
        print(f"   wave 0 in all                     {tc:8d} cycles {tr / 100.0:7.2f} us  ({tc / max(tr, 1) * 100:.0f} MHz)")
    for g in range(3):
        print(f"   {names[g].split(' (')[0]} start -> {names[g + 1].split(' (')[0]} start: {(st[g + 1, 0, 1] - st[g, 0, 1]) / 100.0:.2f} us")


if __name__ == "__main__":
    main()
