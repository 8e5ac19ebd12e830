"""bench.py's N-rank control flow on the CPU (gloo, world_size 2): both workloads run with the oracle injected as the
index (the HIP index needs a GPU), rank 0 prints one JSON-able line whose collective really spans 2 ranks, and the
row-sharded leg's own exactness check (per-rank exact hits -> all_gather -> plain-torch merge) passes."""
import argparse
import json
import os
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

from conftest import ROOT


class _OracleIndex:
    """stand-in for _native.IcdIndex with the same methods, computed by oracle/ on CPU tensors (test infrastructure)"""

    def __init__(self, corpus, levels, device=0, max_nq=0, max_k=10, id_base=0):
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as orc
        self.orc = orc
        self.corpus = np.ascontiguousarray(corpus.numpy() if hasattr(corpus, "numpy") else corpus, np.float32)
        self.levels = np.ascontiguousarray(levels.numpy() if hasattr(levels, "numpy") else levels, np.int32)
        self.id_base = int(id_base)
        self.prof = False

    def search(self, q, k, mode=0):
        s, i = self.orc.flat_ip_topk(self.corpus, q.numpy(), k, id_base=self.id_base)
        return torch.from_numpy(s), torch.from_numpy(i)

    def lookup_levels(self, ids):
        i = ids.numpy() - self.id_base
        return torch.from_numpy(np.where(ids.numpy() >= 0, self.levels[np.clip(i, 0, len(self.levels) - 1)], 0).astype(np.int32))

    def search_reweighted(self, q, k, mode=0):
        s, i = self.orc.flat_ip_topk(self.corpus, q.numpy(), k, id_base=self.id_base)
        return tuple(torch.from_numpy(x) for x in self.orc.reweight(s, i, self.levels, id_base=self.id_base))

    def set_profiling(self, on):
        self.prof = on

    def profile_summary(self):
        return {"ms_prep": 0.0, "ms_coarse": 1.0, "ms_finalize": 0.0, "ms_exact": 0.0, "ms_exact_finalize": 0.0, "ms_total": 1.0, "count": 1}

    def stats(self):
        return {"last_mode": 0, "last_fallback": 0, "last_chunks": 1}

    def close(self):
        pass


def _merge_cpu(s, i, l, k):
    """merge_fn of ShardedSearch on CPU tensors: global top-k (score desc, id asc) + level reweight + stable re-sort"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    ms, mi = orc.merge(s.numpy(), i.numpy(), k)
    lv_of = {int(a): int(b) for a, b in zip(i.numpy().ravel(), l.numpy().ravel()) if a >= 0}
    nq = ms.shape[0]
    adj = np.empty((nq, k), np.float64); raw = np.empty((nq, k), np.float32)
    ids = np.empty((nq, k), np.int64); lv = np.empty((nq, k), np.int32)
    for q in range(nq):
        hl = np.asarray([lv_of.get(int(x), 0) for x in mi[q]], np.int32)
        a, r, d, l2 = (np.empty(k, np.float64), np.empty(k, np.float32), np.empty(k, np.int64), np.empty(k, np.int32))
        import ctypes
        vp = ctypes.c_void_p
        msq, miq = np.ascontiguousarray(ms[q]), np.ascontiguousarray(mi[q])
        orc.lib().icd_oracle_reweight_one(vp(msq.ctypes.data), vp(miq.ctypes.data), vp(hl.ctypes.data), ctypes.c_int(k),
                                          vp(a.ctypes.data), vp(r.ctypes.data), vp(d.ctypes.data), vp(l2.ctypes.data))
        adj[q], raw[q], ids[q], lv[q] = a, r, d, l2
    return tuple(torch.from_numpy(x) for x in (adj, raw, ids, lv))


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except Exception:   # surface the failure in the parent instead of a queue timeout
        import traceback
        q.put((rank, "ERROR", traceback.format_exc()))
        raise


def _worker_body(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port), "ICD_BENCH_BACKEND": "gloo", "ICD_BENCH_DEVICE": "cpu"})
    import bench
    from rag_project_icd10_amd.sharded import ROW_SHARD, ShardedSearch
    ctx = bench.Ctx()
    assert ctx.world == world and ctx.dist.get_world_size() == world
    args = argparse.Namespace(steps=2, warmup=1, settle_ms=0.0, nq=40, n=900, k=5, mode="auto", rows_per_gpu=700, rowshard_queries=50,
                              rowshard_slice=32, rowshard_steps=2, no_cpu_baseline=True)

    def index_factory(corpus, levels, device, max_nq, max_k, id_base=0):
        return _OracleIndex(corpus, levels, device, max_nq, max_k, id_base)

    def sharded_factory(index):
        def search_fn(qs, k):
            s, i = index.search(qs, k)
            return s, i, index.lookup_levels(i)
        return ShardedSearch(ROW_SHARD, search_fn=search_fn, merge_fn=_merge_cpu)

    rs = bench.run_rowshard(ctx, args, index_factory=index_factory, sharded_factory=sharded_factory)
    rp = bench.run_replicated(ctx, args, index_factory=index_factory)
    q.put((rank, json.dumps(rs) if rs else None, json.dumps(rp) if rp else None))
    ctx.dist.barrier()
    ctx.dist.destroy_process_group()


def test_bench_two_ranks_on_cpu():
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = 29950 + (os.getpid() % 40)
    procs = [mpctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict((r, (a, b)) for r, a, b in (q.get(timeout=240) for _ in range(2)))
    for r, (a, b) in results.items():
        assert a != "ERROR", b
    for p in procs:
        p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    assert results[1] == (None, None)                      # only rank 0 reports
    rs, rp = json.loads(results[0][0]), json.loads(results[0][1])
    assert rs["n_gpus"] == 2 and rs["config"]["collective_ranks"] == 2 and rs["config"]["corpus_rows_total"] == 1400
    assert rs["ids_exact_on_sample"] and rs["raw_scores_exact_on_sample"] and rs["adjusted_sorted"]
    assert rs["scaling"] == "weak" and rs["metric"] == "queries_per_sec" and rs["value"] > 0 and "roofline" in rs
    assert "configs[4]" in rs["config"]["workload"]
    assert rp["n_gpus"] == 2 and rp["ids_exact"] and rp["recall_at_10"] == 1.0 and rp["adjusted_scores_exact"]
    assert rp["parity_checked_queries"] == 40 and "configs[3]" in rp["config"]["workload"]
    assert abs(rp["value"] - 2 * 40 * 2 / (rp["ms_per_step"] * 2 / 1e3)) / rp["value"] < 1e-6
    for line in (rs, rp):
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                    "vs_baseline", "dtype", "data", "config", "roofline"):
            assert key in line, key
