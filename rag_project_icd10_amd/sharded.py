"""Multi-GPU search: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

The reference is single-process (SURVEY.md section 5 "Distributed communication backend: none"); this
is new functionality required by the scaling configs:

  * query-sharded  (config 4): the corpus is replicated, rank r searches its slice of the queries.
    No collective on the data path; `gather=True` adds one all-gather of the (small) results.
  * row-sharded    (config 5): rank r owns rows [base_r, base_r + n_r) with global ids; every rank
    scores ALL queries against its shard, then ONE all-gather moves the per-shard partial top-k
    (16 B per hit: raw score f32 | id i64 | level i32) and every rank merges the G lists per query
    (global top-k by (score desc, id asc)), applies the level reweight and the stable re-sort.
    At Q = 100 000, k = 10 that is 16 MB per rank - microseconds over xGMI next to ~100 ms of MFMA
    work, so no ring/all-reduce is involved.

Two engines behind one class:
  * the C ABI (`from_index(..., native=True)` or ICD_SHARDED_ENGINE=native; the default for a single rank):
    `icd_group_*` of libicdsearch.so - local search, ONE grouped ncclAllGather (RCCL opened by the library itself), merge +
    reweight, all enqueued on one stream inside one call; PyTorch only hands over the 128-byte RCCL unique id of rank 0
    (through the process group that already exists) and the tensors' pointers. The ranks AGREE on the engine: every rank
    does the fallible local half (icd_group_prepare), one all_reduce(MIN) of a success flag decides, and only then does
    every rank enter the collective ncclCommInitRank (icd_group_connect, under a wall-clock limit: a connect that fails on
    one rank alone brings the others back after CONNECT_TIMEOUT_S) - or none does and all of them run the engine below. With more than one rank the default is the torch.distributed engine until the C-ABI collective has run on >= 2
    GPUs (bench.py tries it next to the measurement, under a time limit, and reports the outcome);
  * injected callables + `torch.distributed` collectives (the constructor): the CPU (gloo, world_size 2) tests inject the
    oracle so the sharding + collective logic is covered without a GPU, and a gloo group over GPU tensors (several ranks
    on ONE device, which RCCL refuses) still works.
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

QUERY_SHARD = "query"
ROW_SHARD = "row"
# the row-sharded merge (csrc/icd_search.hip icd_merge_topk, csrc/icd_group.cpp icd_group_prepare) ranks the world * k
# gathered candidates of a query in one work-group's LDS: at most this many
MERGE_MAX_CANDIDATES = 1024
# ncclCommInitRank (icd_group_connect) is collective and has no time limit of its own: a rank whose connect fails FAST (bad id,
# duplicate device, out of memory) would leave the others inside the bootstrap forever. Every rank therefore runs it in a
# worker thread and gives up after this many seconds (ICD_GROUP_CONNECT_TIMEOUT_S); see ShardedSearch._open_native
CONNECT_TIMEOUT_S = 120.0


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous split of n items over `world` ranks (first n % world ranks get one more)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_hits(scores: torch.Tensor, ids: torch.Tensor, levels: torch.Tensor) -> torch.Tensor:
    """[nq,k] f32 / i64 / i32 -> one int32 tensor [nq,k,4] (a single all-gather payload)."""
    out = torch.empty(scores.shape + (4,), dtype=torch.int32, device=scores.device)
    out[..., 0] = scores.contiguous().view(torch.int32)
    out[..., 1:3] = ids.contiguous().view(torch.int32).view(ids.shape + (2,))
    out[..., 3] = levels
    return out


def unpack_hits(buf: torch.Tensor):
    scores = buf[..., 0].contiguous().view(torch.float32)
    ids = buf[..., 1:3].contiguous().view(torch.int64).squeeze(-1)
    levels = buf[..., 3].contiguous()
    return scores, ids, levels


class ShardedSearch:
    """search_fn(queries, k)  -> (raw f32 [nq,k], global ids i64 [nq,k], levels i32 [nq,k])   local shard / replica
       merge_fn(scores[G,nq,k], ids[G,nq,k], levels[G,nq,k], k) -> (adj f64, raw f32, ids i64, levels i32) [nq,k]
       local_reweighted_fn(queries, k) -> (adj, raw, ids, levels)  (query-sharded mode)"""

    def __init__(self, mode: str, search_fn: Optional[Callable] = None, merge_fn: Optional[Callable] = None,
                 local_reweighted_fn: Optional[Callable] = None, group=None):
        if mode not in (QUERY_SHARD, ROW_SHARD):
            raise ValueError(f"mode must be '{QUERY_SHARD}' or '{ROW_SHARD}'")
        self.mode = mode
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.search_fn = search_fn
        self.merge_fn = merge_fn
        self.local_reweighted_fn = local_reweighted_fn
        self.native_group = None   # _native.IcdGroup when the C ABI runs the whole sharded search
        self.native_stuck = False  # a worker thread of THIS process is still inside ncclCommInitRank (see _open_native)
        self.engine = "torch.distributed"   # or "icd_group (C ABI)": which engine search_reweighted runs on (the ranks agree)

    # ---- construction over the HIP index ---------------------------------------------------------------
    @classmethod
    def from_index(cls, index, mode: str, group=None, native: Optional[bool] = None, native_factory: Optional[Callable] = None) -> "ShardedSearch":
        """index: rag_project_icd10_amd._native.IcdIndex over this rank's shard (row mode, created with
        id_base = first global row) or over the full corpus (query mode).
        native (default: a single rank yes; several ranks only with ICD_SHARDED_ENGINE=native on an nccl group): run the
        sharded search through the C ABI's icd_group_* (RCCL inside the library); otherwise torch.distributed collectives
        between the library's kernels. The ranks agree on the engine (_open_native): all native or all torch, never a mix.
        native_factory (tests): a callable that builds this rank's unconnected group object instead of _native.IcdGroup."""
        from . import _native

        def search_fn(q, k):
            raw, ids = index.search(q, k)
            return raw, ids, index.lookup_levels(ids)

        self = cls(mode, search_fn=search_fn, merge_fn=_native.merge_topk,
                   local_reweighted_fn=index.search_reweighted, group=group)
        if native is None:
            env = os.environ.get("ICD_SHARDED_ENGINE", "")
            native = (env == "native" or (env != "torch" and self.world == 1)) and (self.world == 1 or dist.get_backend(group) == "nccl")
        if native:
            self._open_native(index, group, native_factory)
        return self

    def _agree(self, ok: bool, group, device=None) -> bool:
        """True on every rank iff `ok` on every rank (one all_reduce(MIN) over the existing process group)"""
        if self.world == 1:
            return ok
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        if dist.get_backend(group) == "nccl":
            t = t.to(torch.device("cuda", device))
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return bool(int(t.item()) == 1)

    def _open_native(self, index, group, native_factory=None):
        """The C-ABI engine on every rank, or on none. (1) every rank: the LOCAL half (icd_group_prepare: argument checks,
        device buffers, dlopen of librccl) - (2) all_reduce(MIN) of a success flag - (3) every rank: the collective
        ncclCommInitRank (icd_group_connect) with rank 0's unique id, broadcast over the existing group - (4) all_reduce(MIN)
        again. A rank that failed alone would otherwise leave the others waiting inside ncclCommInitRank, or send the ranks
        into different collectives afterwards (VERDICT r3 / ADVICE r3)."""
        import logging
        from . import _native
        log = logging.getLogger(__name__)
        mode = _native.GROUP_ROW_SHARD if self.mode == ROW_SHARD else _native.GROUP_QUERY_SHARD
        make = native_factory or (lambda: _native.IcdGroup(index, mode, rank=self.rank, world=self.world, connect=False, with_comm=self.world > 1))
        grp, err = None, None
        try:
            grp = make()
        except Exception as exc:   # (no librccl, out of memory, bad arguments, ...)
            err = exc
        dev = getattr(index, "device", 0)
        if not self._agree(grp is not None, group, dev):
            if grp is not None:
                grp.close()
            log.warning("rank %d of %d: icd_group_prepare %s: EVERY rank runs the sharded search on torch.distributed collectives",
                        self.rank, self.world, f"failed here ({err})" if err else "failed on another rank")
            return
        if self.world > 1:   # rank 0's RCCL unique id travels through the existing process group (128 bytes)
            t = torch.zeros(_native.GROUP_ID_BYTES, dtype=torch.uint8)
            if self.rank == 0:
                try:
                    t = torch.frombuffer(bytearray(grp.unique_id() if hasattr(grp, "unique_id") else _native.group_unique_id()), dtype=torch.uint8).clone()
                except Exception as exc:   # (an all-zero id: the connect below fails on every rank alike)
                    err = exc
            if dist.get_backend(group) == "nccl":
                t = t.to(torch.device("cuda", dev))
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            uid = bytes(t.cpu().numpy().tobytes())
            ok = any(uid)
            if ok:
                ok, err = self._connect_with_deadline(grp, uid)
            if not self._agree(ok, group, dev):
                if not self.native_stuck:
                    grp.close()   # (a group whose connect is still running is left to process exit: closing it would free what that thread uses)
                log.warning("rank %d of %d: icd_group_connect %s: EVERY rank runs the sharded search on torch.distributed collectives",
                            self.rank, self.world, f"failed here ({err})" if err else "failed on another rank")
                return
        self.native_group = grp
        self.engine = "icd_group (C ABI)"

    def _connect_with_deadline(self, grp, uid):
        """grp.connect(uid) in a worker thread, given up after CONNECT_TIMEOUT_S (env ICD_GROUP_CONNECT_TIMEOUT_S): (ok, error).
        ncclCommInitRank is collective: when it fails fast on ONE rank, the others wait inside its bootstrap with no limit of
        their own while the failed rank waits in the agreement's all_reduce. With the deadline the waiting ranks
        come back, the all_reduce(MIN) completes and every rank runs the torch engine. The worker thread of a rank that gave up
        is STILL inside ncclCommInitRank (ctypes released the GIL; nothing can cancel it): `native_stuck` says so, the group is
        not closed under it, and a long-lived serving process should restart at its next convenience - searches are correct
        meanwhile (the torch engine shares nothing with that communicator). Asymmetric connect failures are only SAFE through
        this deadline; symmetric ones (every rank fails) never needed it."""
        import threading
        limit = float(os.environ.get("ICD_GROUP_CONNECT_TIMEOUT_S", CONNECT_TIMEOUT_S))
        box = {}

        def body():
            try:
                grp.connect(uid)
                box["ok"] = True
            except Exception as exc:
                box["err"] = exc

        th = threading.Thread(target=body, daemon=True, name="icd_group_connect")
        th.start()
        th.join(limit)
        if th.is_alive():
            self.native_stuck = True
            return False, TimeoutError(f"rank {self.rank} of {self.world}: icd_group_connect (ncclCommInitRank) did not return within {limit:.0f} s")
        return bool(box.get("ok")), box.get("err")

    def close(self):
        """the C-ABI group, if any (it borrows the index's handle; IcdIndex.close() would close it too)"""
        if self.native_group is not None:
            self.native_group.close()
            self.native_group = None

    # ---- search ---------------------------------------------------------------------------------------------
    def search_reweighted(self, queries: torch.Tensor, k: int, gather: bool = True):
        """Row mode: `queries` is the full batch on every rank -> identical (adj, raw, ids, levels) on
        every rank. Query mode: `queries` is the full batch on every rank; rank r searches its slice
        and, with gather=True, all ranks receive the full result (else only the local slice)."""
        if self.mode == ROW_SHARD and self.world * int(k) > MERGE_MAX_CANDIDATES:
            # (every rank sees the same world and k: all of them raise here, none enters a collective)
            raise ValueError(f"row-sharded search: world * k = {self.world} * {int(k)} = {self.world * int(k)} candidates per query, "
                             f"the merge ranks at most {MERGE_MAX_CANDIDATES} (k <= {MERGE_MAX_CANDIDATES // self.world} at {self.world} ranks)")
        if self.native_group is not None:
            return self.native_group.search(queries, k, gather=gather)
        if self.mode == ROW_SHARD:
            raw, ids, levels = self.search_fn(queries, k)
            payload = pack_hits(raw, ids, levels)
            if self.world > 1:
                payload = payload.contiguous()
                flat = torch.empty((self.world * payload.shape[0],) + payload.shape[1:], dtype=payload.dtype,
                                   device=payload.device)  # concatenated form: accepted by nccl and gloo
                dist.all_gather_into_tensor(flat, payload, group=self.group)
                gathered = flat.view((self.world,) + payload.shape)
            else:
                gathered = payload.unsqueeze(0)
            s, i, l = unpack_hits(gathered)
            return self.merge_fn(s, i, l, k)
        # query-sharded
        nq = queries.shape[0]
        lo, hi = shard_bounds(nq, self.world, self.rank)
        if hi > lo:
            adj, raw, ids, levels = self.local_reweighted_fn(queries[lo:hi], k)
        else:   # fewer queries than ranks: this rank's slice is empty (the engines are not asked to search nothing)
            qd = queries.device
            adj, raw = torch.empty((0, k), dtype=torch.float64, device=qd), torch.empty((0, k), dtype=torch.float32, device=qd)
            ids, levels = torch.empty((0, k), dtype=torch.int64, device=qd), torch.empty((0, k), dtype=torch.int32, device=qd)
        if not gather or self.world == 1 or nq == 0:
            return adj, raw, ids, levels
        width = -(-nq // self.world)  # pad every slice to the same length for one all-gather
        dev = adj.device
        pay = torch.zeros((width, k, 6), dtype=torch.int32, device=dev)
        m = hi - lo
        if m > 0:
            pay[:m, :, 0:2] = adj.contiguous().view(torch.int32).view(m, k, 2)
            pay[:m, :, 2] = raw.contiguous().view(torch.int32)
            pay[:m, :, 3:5] = ids.contiguous().view(torch.int32).view(m, k, 2)
            pay[:m, :, 5] = levels
        flat = torch.empty((self.world * width, k, 6), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(flat, pay, group=self.group)
        gathered = flat.view(self.world, width, k, 6)
        parts = []
        for r in range(self.world):
            a, b = shard_bounds(nq, self.world, r)
            parts.append(gathered[r, : b - a])
        full = torch.cat(parts, 0)
        return (full[..., 0:2].contiguous().view(torch.float64).squeeze(-1), full[..., 2].contiguous().view(torch.float32),
                full[..., 3:5].contiguous().view(torch.int64).squeeze(-1), full[..., 5].contiguous())
