"""Multi-process (gloo, world_size 2) coverage of the sharded search: row-sharded all-gather + merge and
query-sharded gather, with the oracle injected as the local engine (no GPU here)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, icd_levels, unit_rows

from rag_project_icd10_amd.sharded import QUERY_SHARD, ROW_SHARD, ShardedSearch, pack_hits, shard_bounds, unpack_hits


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 37000, 40474):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1


def test_pack_roundtrip():
    s = torch.randn(5, 3)
    i = torch.randint(-1, 2 ** 40, (5, 3))
    l = torch.randint(0, 4, (5, 3), dtype=torch.int32)
    s2, i2, l2 = unpack_hits(pack_hits(s, i, l))
    assert torch.equal(s, s2) and torch.equal(i, i2) and torch.equal(l, l2)


def _worker(rank, world, port, mode, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, dim, nq, k = 1003, 64, 21, 7
        corpus, levels, queries = unit_rows(n, dim, 1), icd_levels(n, 2), unit_rows(nq, dim, 3)
        fs, fi = orc.flat_ip_topk(corpus, queries, k)
        want = orc.reweight(fs, fi, levels)

        def merge_fn(s, i, l, kk):
            ms, mi = orc.merge(s.numpy(), i.numpy(), kk)
            lv = np.where(mi >= 0, levels[np.clip(mi, 0, n - 1)], 0).astype(np.int32)
            hit_levels = np.zeros(n + 1, np.int32)
            hit_levels[:n] = levels
            adj, raw, ids, olv = orc.reweight(ms, mi, levels)
            # levels travelled in the payload too: they must agree with the table
            got_lv = {int(a): int(b) for a, b in zip(i.numpy().ravel(), l.numpy().ravel()) if a >= 0}
            assert all(levels[a] == b for a, b in got_lv.items())
            return tuple(torch.from_numpy(x) for x in (adj, raw, ids, olv))

        if mode == ROW_SHARD:
            lo, hi = shard_bounds(n, world, rank)

            def search_fn(qs, kk):
                s, i = orc.flat_ip_topk(corpus[lo:hi], qs.numpy(), kk, id_base=lo)
                lv = np.where(i >= 0, levels[np.clip(i, 0, n - 1)], 0).astype(np.int32)
                return torch.from_numpy(s), torch.from_numpy(i), torch.from_numpy(lv)

            eng = ShardedSearch(ROW_SHARD, search_fn=search_fn, merge_fn=merge_fn)
        else:
            def local_fn(qs, kk):
                s, i = orc.flat_ip_topk(corpus, qs.numpy(), kk)
                return tuple(torch.from_numpy(x) for x in orc.reweight(s, i, levels))

            eng = ShardedSearch(QUERY_SHARD, local_reweighted_fn=local_fn)
        adj, raw, ids, lv = eng.search_reweighted(torch.from_numpy(queries), k)
        ok = (np.array_equal(ids.numpy(), want[2]) and adj.numpy().tobytes() == want[0].tobytes()
              and raw.numpy().tobytes() == want[1].tobytes() and np.array_equal(lv.numpy(), want[3]))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", [ROW_SHARD, QUERY_SHARD])
def test_two_ranks_match_single_process(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400) + (0 if mode == ROW_SHARD else 1)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    results = [q.get(timeout=10) for _ in range(2)]
    assert all(p.exitcode == 0 for p in procs)
    assert sorted(results) == [(0, True), (1, True)]
