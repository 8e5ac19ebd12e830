#!/usr/bin/env python3
"""Where the 52 ms of the BERT-base fp32 forward over 1 000 diagnosis strings go: the encoder's GEMM shapes through
PyTorch-ROCm's two BLAS backends (fp32), against the 157.3 TFLOP/s fp32 MFMA peak, and the whole forward per backend."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    dev = "cuda"
    M = 20000   # ~1 000 strings x ~20 tokens
    shapes = [("qkv/out proj", M, 768, 768), ("ffn up", M, 3072, 768), ("ffn down", M, 768, 3072)]
    for backend in ("default", "hipblaslt", "hipblas"):
        if backend != "default":
            try:
                torch.backends.cuda.preferred_blas_library(backend)
            except Exception as exc:
                print(backend, "not selectable:", exc)
                continue
        print("backend", backend, "->", torch.backends.cuda.preferred_blas_library())
        for name, m, n, k in shapes:
            a = torch.randn(m, k, device=dev)
            w = torch.randn(n, k, device=dev)
            b = torch.randn(n, device=dev)
            for fn_name, fn in (("linear+bias", lambda: torch.nn.functional.linear(a, w, b)), ("matmul", lambda: a @ w.t())):
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    fn()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 20
                print(f"  {name:14s} {fn_name:12s} {m}x{n}x{k}: {dt * 1e3:.3f} ms  {2 * m * n * k / dt / 1e12:.1f} TFLOP/s ({2 * m * n * k / dt / 1e12 / 157.3:.2f} of fp32 peak)")
    # the whole forward
    os.environ.setdefault("EMBEDDING_MODEL_NAME", "shibing624/text2vec-base-chinese")
    os.environ.setdefault("ICD_EMBEDDING_ALLOW_SYNTHETIC", "1")
    from rag_project_icd10_amd.services.embedding_service import EmbeddingService
    strings = [l.strip() for l in open(os.path.join(ROOT, "tests", "golden", "diagnosis_strings.txt"), encoding="utf-8") if l.strip()][:1000]
    es = EmbeddingService()
    for backend in ("hipblaslt", "hipblas"):
        try:
            torch.backends.cuda.preferred_blas_library(backend)
        except Exception:
            continue
        for _ in range(3):
            es.encode_query_batch(strings)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            es.encode_query_batch(strings)
        torch.cuda.synchronize()
        print(f"encode_query_batch(1 000 strings) with {backend}: {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms")


if __name__ == "__main__":
    main()
