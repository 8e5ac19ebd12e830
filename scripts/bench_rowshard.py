#!/usr/bin/env python3
"""BASELINE configs[4] (SURVEY.md section 8d "Config 5"): the corpus row-sharded over the GPUs of a node.

    torchrun --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/bench_rowshard.py [--rows-per-gpu 1250000]
    python scripts/bench_rowshard.py            # N = 1: one shard, the all-gather degenerates to a view

Every rank generates its shard ON THE DEVICE (rows_per_gpu x 768 fp32, torch.Generator seeded 1234 + rank, rows
L2-normalised), builds an index with id_base = rank * rows_per_gpu, and the full query batch (100 000 x 768, seed 4321,
replicated) is searched in slices of --slice queries: local top-k of the shard -> ONE all_gather of (score f32, id i64,
level i32) per hit -> merge kernel + level reweight on every rank (rag_project_icd10_amd.sharded.ShardedSearch, ROW).
Rank 0 prints one JSON object: whole-job queries/s over the N-shard corpus, the per-stage split of rank 0 and a
correctness check of a query sample against an exact search of the same shard data (single-GPU EXACT mode).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows-per-gpu", type=int, default=1_250_000)
    ap.add_argument("--queries", type=int, default=100_000)
    ap.add_argument("--slice", type=int, default=16384)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--passes", type=int, default=2)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from rag_project_icd10_amd._native import MODE_EXACT, IcdIndex
    from rag_project_icd10_amd.sharded import ROW_SHARD, ShardedSearch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    n, dim, k = args.rows_per_gpu, 768, args.k

    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    corpus = torch.empty((n, dim), dtype=torch.float32, device=dev)
    for s in range(0, n, 250_000):   # generated in slabs: no 2x transient
        e = min(n, s + 250_000)
        x = torch.randn((e - s, dim), generator=g, device=dev, dtype=torch.float32)
        corpus[s:e] = x / x.norm(dim=1, keepdim=True)
    r = torch.rand(n, generator=g, device=dev)
    levels = torch.where(r < 0.1243, 1, torch.where(r < 0.4234, 2, 3)).to(torch.int32)
    t0 = time.perf_counter()
    index = IcdIndex(corpus, levels, device=local_rank, max_nq=args.slice, max_k=max(k, 10), id_base=rank * n)
    torch.cuda.synchronize(dev)
    t_build = time.perf_counter() - t0
    del corpus
    gq = torch.Generator(device=dev)
    gq.manual_seed(4321)
    queries = torch.randn((args.queries, dim), generator=gq, device=dev, dtype=torch.float32)
    queries /= queries.norm(dim=1, keepdim=True)
    sharded = ShardedSearch.from_index(index, ROW_SHARD)

    def one_pass():
        outs = []
        for s in range(0, args.queries, args.slice):
            outs.append(sharded.search_reweighted(queries[s:s + args.slice], k))
        return outs

    one_pass()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.passes):
        outs = one_pass()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = (time.perf_counter() - t0) / args.passes
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # stage split of one slice on this rank
    q0 = queries[: args.slice]
    torch.cuda.synchronize(dev); t0 = time.perf_counter()
    raw, ids = index.search(q0, k)
    lv = index.lookup_levels(ids)
    torch.cuda.synchronize(dev); t_local = time.perf_counter() - t0
    t0 = time.perf_counter()
    sharded.search_reweighted(q0, k)
    torch.cuda.synchronize(dev); t_all = time.perf_counter() - t0
    # correctness of the local part on a sample: the fp16-certified path against the exact fp32 kernel
    sample = q0[:: max(1, args.slice // 64)][:64].contiguous()
    ra, ia = index.search(sample, k)
    re_, ie = index.search(sample, k, MODE_EXACT)
    ok = bool(torch.equal(ia, ie) and torch.equal(ra.view(torch.int32), re_.view(torch.int32)))
    st = index.stats()
    if rank == 0:
        print(json.dumps({
            "config": f"BASELINE configs[4]: {world} x {n} rows x {dim} row-sharded, {args.queries} queries, top_k={k}, slices of {args.slice}",
            "n_gpus": world, "corpus_rows_total": world * n, "queries": args.queries,
            "queries_per_s": args.queries / elapsed, "s_per_pass": elapsed,
            "rank0_slice_ms": {"local_search_plus_levels": t_local * 1e3, "search_allgather_merge": t_all * 1e3},
            "flops_per_gpu_per_pass": 2.0 * args.queries * n * dim,
            "achieved_tflops_per_gpu": 2.0 * args.queries * n * dim / elapsed / 1e12,
            "local_fast_path_equals_exact_kernel_on_sample": ok, "last_fallback": int(st["last_fallback"]),
            "index_build_s": t_build, "bytes_corpus_f32": int(st["bytes_corpus_f32"]), "bytes_corpus_f16": int(st["bytes_corpus_f16"]),
        }), flush=True)
    index.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
