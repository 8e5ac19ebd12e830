#!/usr/bin/env python3
"""Headline benchmark: batch exact top-k search over the ICD corpus on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]), per GPU:
    corpus  37 000 x 768 fp32, iid N(0,1) rows L2-normalised, default_rng(1234); levels default_rng(1235)
            from the real CSV histogram;
    queries 10 000 x 768, default_rng(4321 + rank), resident in HBM before the timed region;
    one STEP = one pass of the hot path over the batch: icd_index_search_reweighted (query fp16 image,
    fp16-MFMA coarse top-k', certification + exact fp32 rescoring, level reweight + stable re-sort,
    exact fallback for uncertified queries), k = 10. Results are bit-identical to the exact kernel.
With N > 1 every rank holds a corpus replica and its own query batch (data-parallel, no collective on
the data path: weak scaling); `value` is the whole-job rate = N * nq * K / max-over-ranks time.

The JSON line also carries
    roofline      dominant kernel (coarse_flat_kernel): algorithmic FLOPs 2*nq*n*dim per launch / its mean
                  duration over the timed steps (hipEvents recorded by the library on the search
                  stream), against the dense fp16/bf16 MFMA peak (2.5 PFLOP/s);
    cpu_baseline  the reference's call shape on the host CPU (one query per call: fp32 scan + top-k +
                  level weight + stable sort, oracle/oracle.py reference_shaped_search), rank 0, N = 1
                  only, on a bounded query sample;
    recall_at_10 / ids_exact against the CPU oracle on a query sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS_F16 = 2500.0  # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"


def unit_rows(n, dim, seed):
    x = np.random.default_rng(seed).standard_normal((n, dim), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return np.ascontiguousarray(x, dtype=np.float32)


def icd_levels(n, seed):
    r = np.random.default_rng(seed).random(n)
    return np.where(r < 0.1243, 1, np.where(r < 0.4234, 2, 3)).astype(np.int32)


def cpu_baseline(corpus, levels, queries, k, budget_s=12.0):
    """reference-shaped CPU search, one query per call, on a bounded sample of the same workload"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    import torch
    threads = torch.get_num_threads()
    t0 = time.perf_counter()
    done = 0
    for q in queries[:8]:
        orc.reference_shaped_search(corpus, levels, q, k)
        done += 1
    per = (time.perf_counter() - t0) / done
    m = int(max(16, min(len(queries), budget_s / per)))
    t0 = time.perf_counter()
    for q in queries[:m]:
        orc.reference_shaped_search(corpus, levels, q, k)
    dt = time.perf_counter() - t0
    return {"value": m / dt, "unit": "queries/s", "cores": int(os.cpu_count() or threads), "blas_threads": int(threads),
            "kind": "port", "sample": f"{m} of the {len(queries)} queries, one query per call (reference call shape), "
                                      f"{corpus.shape[0]}x{corpus.shape[1]} fp32 corpus, numpy/BLAS"}


def pmc_traffic(kernel, nq, n):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/rNN_pmc_traffic.json, made by scripts/gpu_pmc2.sh: FETCH_SIZE and WRITE_SIZE collected in their own
    passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950). PMC counters cannot be read from inside
    this process; the number is reported only for the workload it was collected on."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files or (nq, n) != (10000, 37000):
        return None
    try:
        d = json.load(open(files[-1]))
        for name, v in d.items():
            if name.startswith(kernel):
                return float(v["traffic_bytes"])
    except Exception:
        return None
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nq", type=int, default=10000)
    ap.add_argument("--n", type=int, default=37000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--mode", choices=["auto", "exact"], default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from rag_project_icd10_amd._native import MODE_AUTO, MODE_EXACT, IcdIndex

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # test hook (single-GPU boxes): ICD_BENCH_BACKEND=gloo ICD_BENCH_ONE_DEVICE=1 runs the N-rank control flow with
    # every rank on cuda:0 and the reductions on the CPU; the driver's runs use nccl (= RCCL), one GPU per rank
    backend = os.environ.get("ICD_BENCH_BACKEND", "nccl")
    if os.environ.get("ICD_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = dev if backend == "nccl" else torch.device("cpu")
    dim, nq, n, k = 768, args.nq, args.n, args.k
    mode = MODE_AUTO if args.mode == "auto" else MODE_EXACT

    corpus, levels = unit_rows(n, dim, 1234), icd_levels(n, 1235)
    queries = unit_rows(nq, dim, 4321 + rank)
    index = IcdIndex(corpus, levels, device=local_rank, max_nq=nq, max_k=max(k, 10))
    dq = torch.from_numpy(queries).to(dev)
    torch.cuda.synchronize(dev)

    def step():
        return index.search_reweighted(dq, k, mode)

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize(dev)
    index.set_profiling(True)
    index.profile_summary()  # reset the event window
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    prof = index.profile_summary()
    index.set_profiling(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stats = index.stats()

    if rank == 0:
        adj, raw, ids, lv = (x.cpu().numpy() for x in out)
        # parity / recall on a query sample against the CPU oracle
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as orc
        sample = np.arange(0, nq, max(1, nq // 64))[:64]
        os_, oi = orc.flat_ip_topk(corpus, queries[sample], k)
        want = orc.reweight(os_, oi, levels)
        recall = float(np.mean([len(set(a) & set(b)) / k for a, b in zip(ids[sample], oi)]))
        ids_exact = bool(np.array_equal(ids[sample], want[2]))
        max_dscore = float(np.max(np.abs(raw[sample].astype(np.float64) - want[1].astype(np.float64))))

        dom_ms = prof["ms_coarse"] if (mode == MODE_AUTO and stats["last_mode"] == MODE_AUTO) else prof["ms_exact"]
        flops = 2.0 * nq * n * dim
        achieved = flops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        peak = PEAK_TFLOPS_F16 if stats["last_mode"] == MODE_AUTO else 157.3
        line = {
            "metric": "queries_per_sec", "value": world * nq * args.steps / elapsed, "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16" if stats["last_mode"] == MODE_AUTO else "f32",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: {nq} random fp32 768-d queries x {n}x768 corpus, top_k={k}, "
                                   f"search only (pre-embedded, HBM-resident inputs), level reweight fused",
                       "nq_per_gpu": nq, "corpus_rows": n, "dim": dim, "top_k": k, "mode": args.mode,
                       "parallelism": f"query-sharded x{world}, corpus replicated" if world > 1 else "single GPU",
                       "result_arithmetic": "fp32 canonical chain (bit-identical to the CPU oracle)"},
            "recall_at_10": recall, "ids_exact": ids_exact, "max_abs_dscore": max_dscore,
            "fallback_queries": int(stats["last_fallback"]), "coarse_chunks": int(stats["last_chunks"]),
            "kernel_ms": {kname: round(v, 5) for kname, v in prof.items() if kname != "count"},
            "roofline": {"bound": "mfma", "kernel": "coarse_flat_kernel" if stats["last_mode"] == MODE_AUTO else "exact_topk_kernel",
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": pmc_traffic("coarse_flat_kernel" if stats["last_mode"] == MODE_AUTO else "exact_topk_kernel", nq, n),
                         "traffic_unit": "bytes per launch (rocprofv3 PMC, profiles/r*_pmc_traffic.json)",
                         "flops_per_launch": flops, "launch_ms": dom_ms, "launches_averaged": prof["count"]},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(corpus, levels, queries, k)
        print(json.dumps(line), flush=True)
    index.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
