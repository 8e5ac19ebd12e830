#!/usr/bin/env python3
"""rowshard_pacing_probe.py - the coarse sweep over a 1.25 M-row shard (BASELINE configs[4] on one GPU: 16 384 queries per
slice), with and without PACING of the work-groups that sweep the same corpus tiles (csrc/coarse_flat_kernel.hpp, VAR 67108864;
the per-index options pacing_shift / pacing_lead).

  python3 scripts/probe/rowshard_pacing_probe.py                      # interleaved A/B of several settings: ms per slice, results equal
  python3 scripts/probe/rowshard_pacing_probe.py --one SHIFT LEAD     # ONE setting, three searches: run under
      rocprofv3 --pmc FETCH_SIZE (and WRITE_SIZE in its own pass) to read the fabric traffic of a coarse launch
No reference counterpart (the reference is a single process over 40 474 rows); the workload is SURVEY 8(d) config 5's shard."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from rag_project_icd10_amd import _native  # noqa: E402
from rag_project_icd10_amd._native import IcdIndex  # noqa: E402


def shard(n, dim, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    c = torch.empty((n, dim), dtype=torch.float32, device="cuda")
    for s in range(0, n, 250_000):
        e = min(n, s + 250_000)
        x = torch.randn((e - s, dim), generator=g, device="cuda", dtype=torch.float32)
        c[s:e] = x / x.norm(dim=1, keepdim=True)
    r = torch.rand(n, generator=g, device="cuda")
    return c, torch.where(r < 0.1243, 1, torch.where(r < 0.4234, 2, 3)).to(torch.int32)


def main():
    lib = _native.load_library()
    n, dim, nq, k = 1_250_000, 768, 16384, 10
    corpus, levels = shard(n, dim, 1234)
    index = IcdIndex(corpus, levels, max_nq=nq, max_k=10)
    del corpus
    g = torch.Generator(device="cuda")
    g.manual_seed(4321)
    q = torch.randn((nq, dim), generator=g, device="cuda")
    q /= q.norm(dim=1, keepdim=True)
    if len(sys.argv) >= 4 and sys.argv[1] == "--one":
        index.set_option("pacing_shift", int(sys.argv[2])); index.set_option("pacing_lead", int(sys.argv[3]))
        for _ in range(3):
            index.search_reweighted(q, k)
        torch.cuda.synchronize()
        return
    settings = [(-1, 1), (2, 2), (3, 2), (2, 4), (4, 2), (1, 4)]
    ref = None
    times = {s: [] for s in settings}
    for rnd in range(3):
        for st in settings:
            index.set_option("pacing_shift", st[0]); index.set_option("pacing_lead", st[1])
            index.search_reweighted(q, k)          # warm
            torch.cuda.synchronize()
            index.set_profiling(True)
            index.profile_summary()
            t0 = time.perf_counter()
            out = None
            for _ in range(3):
                out = index.search_reweighted(q, k)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / 3 * 1e3
            prof = index.profile_summary()
            index.set_profiling(False)
            times[st].append((prof["ms_coarse"], wall))
            got = tuple(x.cpu() for x in out)
            if ref is None:
                ref = got
            else:
                assert all(torch.equal(a, b) for a, b in zip(ref, got)), f"results differ under pacing {st}"
    flop = 2.0 * nq * n * dim
    for st in settings:
        cs = sorted(t[0] for t in times[st])
        ws = sorted(t[1] for t in times[st])
        name = "pacing off" if st[0] < 0 else f"epochs of {1 << st[0]} tiles, lead {st[1]} epochs ({st[1] << st[0]} tiles)"
        print(f"{name:46s}: coarse launch median {cs[1]:7.3f} ms (min {cs[0]:7.3f}) = {flop / (cs[1] * 1e-3) / 1e12 / 2500:.3f} of 2.5 PFLOP/s; "
              f"search {ws[1]:7.3f} ms per slice of {nq}; fallback {index.stats()['last_fallback']}")
    print("results bit-identical across all settings")
    index.set_option("pacing_shift", 3); index.set_option("pacing_lead", 2)


if __name__ == "__main__":
    main()
