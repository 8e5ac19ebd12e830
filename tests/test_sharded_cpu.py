"""Multi-process (gloo, world_size 2, 3 and 8) coverage of the sharded search: row-sharded all-gather + merge and
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


def _worker(rank, world, port, mode, q, sizes=(1003, 64, 21, 7)):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, dim, nq, k = sizes
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
        if mode == ROW_SHARD and world * k > 1024:
            # every rank refuses BEFORE any collective, naming the limit (the merge kernel ranks 1 024 candidates per query)
            try:
                eng.search_reweighted(torch.from_numpy(queries), k)
                ok = False
            except ValueError as exc:
                ok = "1024" in str(exc) and f"{world} * {k}" in str(exc)
            # ... and the group is still usable afterwards (nobody is stuck in a collective the others never entered)
            k2 = 1024 // world
            fs2, fi2 = orc.flat_ip_topk(corpus, queries, k2)
            want2 = orc.reweight(fs2, fi2, levels)
            adj, raw, ids, lv = eng.search_reweighted(torch.from_numpy(queries), k2)
            ok = ok and np.array_equal(ids.numpy(), want2[2]) and adj.numpy().tobytes() == want2[0].tobytes()
            q.put((rank, bool(ok)))
            return
        adj, raw, ids, lv = eng.search_reweighted(torch.from_numpy(queries), k)
        ok = (np.array_equal(ids.numpy(), want[2]) and adj.numpy().tobytes() == want[0].tobytes()
              and raw.numpy().tobytes() == want[1].tobytes() and np.array_equal(lv.numpy(), want[3]))
        if mode == QUERY_SHARD:   # without the gather every rank keeps its own slice (empty when there are fewer queries than ranks)
            lo, hi = shard_bounds(nq, world, rank)
            a2, r2, i2, l2 = eng.search_reweighted(torch.from_numpy(queries), k, gather=False)
            ok = ok and tuple(i2.shape) == (hi - lo, k) and np.array_equal(i2.numpy(), want[2][lo:hi]) \
                and a2.numpy().tobytes() == want[0][lo:hi].tobytes()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _spawn(target, world, args_of_rank, timeout=240):
    """`world` spawned ranks of `target(rank, world, port, *args, q)`; returns the sorted (rank, ok) results; every rank must
    exit 0 inside the limit (nobody hung, nobody died)"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args_of_rank) + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=timeout)
    alive = [p for p in procs if p.is_alive()]
    for p in alive:
        p.kill()
    assert not alive, f"{len(alive)} of {world} ranks still running after {timeout} s"
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return sorted(q.get(timeout=10) for _ in range(world))


def _worker_sized(rank, world, port, mode, sizes, q):
    _worker(rank, world, port, mode, q, sizes)


# configs[3] / configs[4] are 8-rank workloads (the reference itself is one process: main.py:753-758): beyond two ranks the
# split is uneven (n % world != 0), a rank's slice can be empty (nq < world), and world * k approaches the merge's 1 024
MULTI = [
    (3, ROW_SHARD, (1003, 64, 21, 7)),        # 1003 % 3 = 1
    (3, QUERY_SHARD, (1003, 64, 20, 7)),      # 20 % 3 = 2
    (8, ROW_SHARD, (1003, 64, 21, 7)),        # 1003 % 8 = 3: shards of 126 and 125 rows
    (8, QUERY_SHARD, (1003, 64, 5, 7)),       # nq < world: ranks 5, 6, 7 search nothing
    (8, QUERY_SHARD, (1003, 64, 21, 7)),      # 21 % 8 = 5
    (8, ROW_SHARD, (2003, 64, 9, 100)),       # k = 100: world * k = 800 candidates per query
    (8, ROW_SHARD, (1003, 64, 9, 128)),       # world * k = 1 024 exactly (the limit), and k exceeds a shard's 125 rows: padded lists
    (8, ROW_SHARD, (1003, 64, 9, 129)),       # world * k = 1 032: a clean error naming the limit, on every rank, before any collective
]


@pytest.mark.parametrize("world,mode,sizes", MULTI, ids=[f"w{w}-{m}-n{s[0]}-nq{s[2]}-k{s[3]}" for w, m, s in MULTI])
def test_more_than_two_ranks_match_single_process(world, mode, sizes):
    assert _spawn(_worker_sized, world, (mode, sizes)) == [(r, True) for r in range(world)]


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


# ---- the ranks agree on the engine: all native (C-ABI group) or all torch.distributed, never a mix --------------------------
class _FakeGroup:
    """stands in for _native.IcdGroup (the C-ABI engine needs a GPU): prepared without a communicator, connect() is the
    collective step, search() answers through the oracle index and says so"""
    log = []

    def __init__(self, index, rank, fail_connect=False, hang_connect=False):
        self.index, self.rank, self.connected, self.closed, self.fail_connect = index, rank, False, False, fail_connect
        self.hang_connect = hang_connect

    def unique_id(self):
        return bytes(range(1, 129))

    def connect(self, uid):
        assert uid == bytes(range(1, 129))          # rank 0's id reached this rank through the process group
        if self.fail_connect:
            raise RuntimeError("forced failure of icd_group_connect")
        if self.hang_connect:   # what ncclCommInitRank does on the OTHER ranks when one rank's connect failed fast: wait for it
            import time
            time.sleep(10 ** 6)
        self.connected = True

    def search(self, queries, k, gather=True):
        raise AssertionError("the test never searches through the fake group")

    def close(self):
        self.closed = True


def _agree_worker_q_last(rank, world, port, case, q):
    _agree_worker(rank, world, port, case, q)


def _agree_worker(rank, world, port, case, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    import bench_cpu_engine as eng
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    bad = 1 if world == 2 else 5   # the rank that fails
    if case.startswith("connect_fails_alone"):
        os.environ["ICD_GROUP_CONNECT_TIMEOUT_S"] = "4"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, dim, nq, k = 803, 64, 19, 5
        corpus, levels, queries = unit_rows(n, dim, 11), icd_levels(n, 12), unit_rows(nq, dim, 13)
        fs, fi = orc.flat_ip_topk(corpus, queries, k)
        want = orc.reweight(fs, fi, levels)
        lo, hi = shard_bounds(n, world, rank)
        index = eng.OracleIndex(corpus[lo:hi], levels[lo:hi], id_base=lo)
        index.device = 0
        made = []

        def factory():
            if case == "prepare_fails_on_one_rank" and rank == bad:   # this rank's local half fails (icd_group_prepare)
                raise RuntimeError(f"rank {rank}: forced failure of icd_group_prepare")
            # connect_fails_on_rank_0: a symmetric-looking failure (the fake's other ranks return). connect_fails_alone: the
            # REAL asymmetric shape - one rank's ncclCommInitRank fails fast, every other rank waits inside it for ever
            g = _FakeGroup(index, rank, fail_connect=(case == "connect_fails_on_rank_0" and rank == 0) or (case == "connect_fails_alone" and rank == bad),
                           hang_connect=(case == "connect_fails_alone" and rank != bad))
            made.append(g)
            return g

        sh = ShardedSearch.from_index(index, ROW_SHARD, native=True, native_factory=factory)
        sh.merge_fn = eng.merge_cpu
        if case == "all_fine":
            ok = sh.native_group is not None and sh.native_group.connected and not sh.native_group.closed
        else:
            # EVERY rank is on the torch engine: no group left open, none connected on the ranks that could have
            stuck = case == "connect_fails_alone" and rank != bad   # (its connect thread is still inside the fake ncclCommInitRank)
            ok = sh.native_group is None and sh.engine == "torch.distributed" and sh.native_stuck == stuck
            ok = ok and all(g.closed for g in made) != stuck   # a group whose connect is still running is NOT closed under it
            if case == "prepare_fails_on_one_rank":
                ok = ok and not any(g.connected for g in made) and len(made) == (0 if rank == bad else 1)
            adj, raw, ids, lv = sh.search_reweighted(torch.from_numpy(queries), k)   # ... and its results are exact
            ok = ok and np.array_equal(ids.numpy(), want[2]) and adj.numpy().tobytes() == want[0].tobytes() \
                and raw.numpy().tobytes() == want[1].tobytes()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, "prepare_fails_on_one_rank"), (2, "connect_fails_on_rank_0"), (2, "all_fine"),
                                        (8, "prepare_fails_on_one_rank"), (8, "connect_fails_alone"), (2, "connect_fails_alone"), (8, "all_fine")])
def test_ranks_agree_on_the_engine(world, case):
    """ShardedSearch.from_index: a rank whose icd_group_prepare fails must not leave the others waiting in the collective
    ncclCommInitRank, nor run another engine than they do: one all_reduce(MIN) decides for all of them.
    connect_fails_alone: ONE rank's connect fails fast while the others wait inside theirs - the deadline of
    ShardedSearch._connect_with_deadline brings them back, all eight agree on the torch engine and search exactly."""
    assert _spawn(_agree_worker_q_last, world, (case,)) == [(r, True) for r in range(world)]


def test_default_engine_above_one_rank_is_torch_distributed(monkeypatch):
    """from_index: a single rank takes the C-ABI group (needs the library: not here), several ranks take torch.distributed
    collectives unless ICD_SHARDED_ENGINE=native / native=True asks for the group - the C-ABI collective has not run on two
    GPUs yet. Checked on the decision alone: no process group, the factory records whether it was asked."""
    calls = []

    class Idx:
        device, max_nq, max_k = 0, 8, 5
        closed = False

        def search(self, q, k):
            raise AssertionError

        lookup_levels = search_reweighted = search

    def factory():
        calls.append(1)

        class G:
            connected = True

            def close(self):
                pass
        return G()

    monkeypatch.delenv("ICD_SHARDED_ENGINE", raising=False)
    sh = ShardedSearch.from_index(Idx(), ROW_SHARD, native_factory=factory)     # world 1, default: native
    assert sh.world == 1 and sh.native_group is not None and calls == [1]
    monkeypatch.setenv("ICD_SHARDED_ENGINE", "torch")
    sh = ShardedSearch.from_index(Idx(), ROW_SHARD, native_factory=factory)
    assert sh.native_group is None and calls == [1]
    sh = ShardedSearch.from_index(Idx(), ROW_SHARD, native=True, native_factory=factory)   # explicit request wins
    assert sh.native_group is not None and calls == [1, 1]
    sh.close()
    assert sh.native_group is None
