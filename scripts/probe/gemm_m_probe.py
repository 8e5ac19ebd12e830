#!/usr/bin/env python3
"""fp32 linear(+bias) of the packed encoder's four shapes at M = the token count of the 1 000 golden strings and at the
next multiples of 128 / 256: does a ragged M cost hipBLASLt anything?"""
import time
import torch
dev = "cuda"
for name, n, k in (("qkv", 2304, 768), ("attn out", 768, 768), ("ffn up", 3072, 768), ("ffn down", 768, 3072)):
    for m in (18290, 18304, 18432, 20000, 20480):
        a = torch.randn(m + 1, k, device=dev)[:m]
        w = torch.randn(n, k, device=dev)
        b = torch.randn(n, device=dev)
        for _ in range(5):
            torch.nn.functional.linear(a, w, b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            torch.nn.functional.linear(a, w, b)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 30
        print(f"{name:9s} M={m:6d} N={n:5d} K={k:5d}: {dt * 1e3:.3f} ms  {2 * m * n * k / dt / 1e12:6.1f} TFLOP/s")
