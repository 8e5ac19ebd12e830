"""CPU engine for the tests of bench.py's N-rank control flow (test infrastructure; bench.py loads this file only when a
test sets ICD_BENCH_DEVICE=cpu and ICD_BENCH_TEST_ENGINE=<this file>): the oracle behind the methods of _native.IcdIndex,
and a ShardedSearch whose merge runs on the CPU."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleIndex:
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

    def set_profiling(self, on, every=1):
        self.prof = on

    def profile_summary(self):
        return {"ms_prep": 0.0, "ms_coarse": 1.0, "ms_finalize": 0.0, "ms_exact": 0.0, "ms_exact_finalize": 0.0, "ms_total": 1.0, "count": 1}

    def stats(self):
        return {"last_mode": 0, "last_fallback": 0, "last_chunks": 1}

    def close(self):
        pass


def merge_cpu(s, i, l, k):
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


def index_factory(corpus, levels, device, max_nq, max_k, id_base=0):
    return OracleIndex(corpus, levels, device, max_nq, max_k, id_base)


def sharded_factory(index):
    sys.path.insert(0, ROOT)
    from rag_project_icd10_amd.sharded import ROW_SHARD, ShardedSearch

    def search_fn(qs, k):
        s, i = index.search(qs, k)
        return s, i, index.lookup_levels(i)
    return ShardedSearch(ROW_SHARD, search_fn=search_fn, merge_fn=merge_cpu)


def on_start(ctx):
    """fault injection for the tests of bench.py's parent / watchdog (tests/test_bench_cpu.py): a rank that dies must fail the
    whole run, a rank that never returns must not hang the parent. Lives HERE, in the test engine, not in bench.py."""
    import time
    if os.environ.get("ICD_BENCH_TEST_FAIL_RANK") == str(ctx.rank):
        raise RuntimeError(f"rank {ctx.rank}: forced failure (ICD_BENCH_TEST_FAIL_RANK)")
    if os.environ.get("ICD_BENCH_TEST_HANG_RANK") == str(ctx.rank):
        time.sleep(10 ** 6)
    if os.environ.get("ICD_BENCH_TEST_TRIAL_HUNG_RANK") == str(ctx.rank):   # as if this rank's C-ABI trial had timed out
        ctx.native_hung = True


def on_exit(ctx):
    """a rank that hangs in the teardown, AFTER rank 0 has printed its line: the parent's watchdog must still relay the line"""
    import time
    if os.environ.get("ICD_BENCH_TEST_HANG_AT_EXIT_RANK") == str(ctx.rank):
        time.sleep(10 ** 6)


def native_trial(ctx, index, queries, sl, k, ref_outs, limit_s, steps=1):
    """stand-in for bench.native_group_trial on gloo ranks (ICD_BENCH_TEST_NATIVE_TRIAL): what the C-ABI group's trial would report,
    so that run_rowshard's ADOPTION logic - every rank must have passed, agreed over the side channel - runs on the CPU.
    "ok": every rank passes; "rank1_differs": rank 1's output differed from the torch engine's; unset: no trial (None)."""
    mode = os.environ.get("ICD_BENCH_TEST_NATIVE_TRIAL")
    if not mode:
        return None
    nq = int(queries.shape[0])
    res = {"status": "ok", "equals_torch_engine": True, "slices_compared": len(ref_outs), "steps": int(steps), "ms_per_step": 2.0,
           "value": nq * 1000.0 / 2.0, "ms_per_slice": 2.0 / max(1, len(ref_outs)), "slice": int(min(sl, nq))}
    if mode == "rank1_differs" and ctx.rank == 1:
        res["equals_torch_engine"] = False
    return res
