"""ctypes binding of libicdsearch.so (C ABI: include/icd_search.h).

This is the only place the product touches native code. There is NO CPU fallback: if the shared
library is missing or the device is not an MI355X (gfx950) the constructors raise.

`IcdIndex` is the MI355X replacement for the Milvus Lite collection the reference opens in
services/milvus_service.py:57-206 and searches at :280-285.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libicdsearch.so")

MODE_AUTO = 0   # fp16-MFMA coarse pass + certified exact rescoring (+ exact fallback); same results as EXACT
MODE_EXACT = 1  # fp32-MFMA kernel only
MAX_K = 128
ABI_VERSION = 6   # include/icd_search.h ICD_ABI_VERSION: a stale libicdsearch.so is refused with a clear message

EXPORTED_SYMBOLS = (
    "icd_abi_version", "icd_last_error", "icd_device_count", "icd_index_create", "icd_index_destroy",
    "icd_index_search", "icd_index_search_reweighted", "icd_merge_topk", "icd_index_lookup_levels",
    "icd_index_stats", "icd_index_set_chunks", "icd_index_debug_counters", "icd_index_set_profiling",
    "icd_index_last_profile", "icd_index_profile_summary", "icd_index_set_option", "icd_packed_attention",
    "icd_hier_rescore",
    "icd_score_stats",
    "icd_cosine_rows",
    "icd_index_set_second_pass",
    "icd_group_unique_id", "icd_group_create", "icd_group_prepare", "icd_group_connect", "icd_group_search", "icd_group_destroy",
    "icd_unpack_query_slices", "icd_split_bf16x3", "icd_encoder_create", "icd_encoder_encode", "icd_encoder_encode_many", "icd_encoder_destroy", "icd_pack_winners",
)
# include/icd_search.h: icd_index_create flags and icd_index_set_option ids (A/B and test options of ONE index)
CREATE_CORPUS_ON_DEVICE, CREATE_ROW_ORDER, CREATE_NO_PROBE, CREATE_NO_CENTER = 1, 2, 4, 8
OPTIONS = {"family_order": 1, "stream_one": 2, "host_one": 3, "pacing_shift": 4, "pacing_lead": 5, "exact_narrow": 6, "wide_from": 7}
GROUP_ROW_SHARD = 0
GROUP_QUERY_SHARD = 1
GROUP_ID_BYTES = 128
SPLIT_TAIL = 64   # csrc/attention_kernel.hpp: elements behind [hi | hi | lo] of icd_split_bf16x3's rows


class IcdError(RuntimeError):
    """A libicdsearch call returned a negative icd_status."""

    def __init__(self, code: int, text: str):
        super().__init__(f"libicdsearch error {code}: {text}")
        self.code = code


class _Stats(C.Structure):
    _fields_ = [("n", C.c_int64), ("dim", C.c_int32), ("device", C.c_int32), ("id_base", C.c_int64),
                ("bytes_corpus_f32", C.c_int64), ("bytes_corpus_f16", C.c_int64), ("bytes_workspace", C.c_int64),
                ("max_nq", C.c_int32), ("max_k", C.c_int32), ("fast_path", C.c_int32), ("rmax", C.c_float),
                ("last_nq", C.c_int64), ("last_fallback", C.c_int64), ("last_chunks", C.c_int32),
                ("last_mode", C.c_int32), ("last_second_pass", C.c_int64), ("last_second_pass_lists", C.c_int32),
                ("second_pass_armed", C.c_int32), ("wide_mode", C.c_int32), ("sparse_fallback_armed", C.c_int32),
                ("centered", C.c_int32), ("mean_share", C.c_float)]


_ENC_LAYER_FIELDS = ("w_qkv", "b_qkv", "w_ao", "b_ao", "ln1_g", "ln1_b", "w_up", "b_up", "w_down", "b_down", "ln2_g", "ln2_b")


class _EncoderDesc(C.Structure):   # include/icd_search.h icd_encoder_desc
    _fields_ = ([("layers", C.c_int32), ("hidden", C.c_int32), ("heads", C.c_int32), ("inter", C.c_int32), ("vocab", C.c_int32),
                 ("max_pos", C.c_int32), ("pos_offset", C.c_int32), ("ln_eps", C.c_float),
                 ("word_emb", C.c_void_p), ("pos_emb", C.c_void_p), ("type_emb0", C.c_void_p), ("emb_ln_g", C.c_void_p), ("emb_ln_b", C.c_void_p)]
                + [(name, C.POINTER(C.c_void_p)) for name in _ENC_LAYER_FIELDS] + [("arithmetic", C.c_int32)])


ENCODER_ARITH = {"fp32": 0, "bf16x3": 1}   # include/icd_search.h ICD_ENCODER_ARITH_*
ENCODER_MAX_TOKENS = 512   # include/icd_search.h ICD_ENCODER_MAX_TOKENS
ENCODER_MAX_SEQS = 64      # ... ICD_ENCODER_MAX_SEQS


class _Profile(C.Structure):
    _fields_ = [("ms_prep", C.c_float), ("ms_coarse", C.c_float), ("ms_finalize", C.c_float),
                ("ms_exact", C.c_float), ("ms_exact_finalize", C.c_float), ("ms_total", C.c_float)]


_lib = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """Load libicdsearch.so and declare the prototypes. Raises if the library is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("ICD_SEARCH_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise ImportError(
            f"{p} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C rag_project_icd10_amd/csrc`. There is no CPU fallback for the search path.")
    lib = C.CDLL(p)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.icd_abi_version.restype = C.c_int
    lib.icd_last_error.restype = C.c_char_p
    lib.icd_device_count.restype = C.c_int
    lib.icd_index_create.argtypes = [vp, i64, i32, vp, i64, i32, i32, i32, i32, C.POINTER(vp)]
    lib.icd_index_destroy.argtypes = [vp]
    lib.icd_index_search.argtypes = [vp, vp, i64, i32, i32, i32, vp, vp, i32, vp]
    lib.icd_index_search_reweighted.argtypes = [vp, vp, i64, i32, i32, i32, vp, vp, vp, vp, i32, vp]
    lib.icd_merge_topk.argtypes = [i32, vp, vp, vp, i32, i64, i32, vp, vp, vp, vp, vp]
    lib.icd_index_lookup_levels.argtypes = [vp, vp, i64, vp, vp]
    lib.icd_index_stats.argtypes = [vp, C.POINTER(_Stats)]
    lib.icd_index_set_chunks.argtypes = [vp, i32]
    lib.icd_index_set_second_pass.argtypes = [vp, i32]
    lib.icd_index_set_option.argtypes = [vp, i32, i32]
    lib.icd_group_unique_id.argtypes = [vp]
    lib.icd_group_create.argtypes = [vp, vp, i32, i32, i32, i32, i32, C.POINTER(vp)]
    lib.icd_group_prepare.argtypes = [vp, i32, i32, i32, i32, i32, i32, C.POINTER(vp)]
    lib.icd_group_connect.argtypes = [vp, vp]
    lib.icd_group_search.argtypes = [vp, vp, i64, i32, i32, vp, vp, vp, vp, vp]
    lib.icd_group_destroy.argtypes = [vp]
    lib.icd_split_bf16x3.argtypes = [i32, vp, i64, i32, i64, i32, vp, vp]
    lib.icd_unpack_query_slices.argtypes = [i32, vp, i32, i64, i32, vp, vp, vp, vp, vp]
    lib.icd_packed_attention.argtypes = [i32, vp, i64, vp, i32, i32, i32, i32, vp, i64, vp]
    lib.icd_encoder_create.argtypes = [i32, C.POINTER(_EncoderDesc), C.POINTER(vp)]
    lib.icd_encoder_encode.argtypes = [vp, vp, vp, i32, i32, i32, vp, i32, vp, vp]
    lib.icd_encoder_encode_many.argtypes = [vp, vp, vp, i64, i32, i32, vp, i32, vp]
    lib.icd_encoder_destroy.argtypes = [vp]
    lib.icd_hier_rescore.argtypes = [i32, vp, vp, i64, i32, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.icd_score_stats.argtypes = [i32, vp, vp, i64, i32, i32, vp, vp]
    lib.icd_pack_winners.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp, vp]
    lib.icd_cosine_rows.argtypes = [i32, vp, vp, i64, i64, i32, vp, vp]
    lib.icd_index_debug_counters.argtypes = [vp, vp, i32]
    lib.icd_index_set_profiling.argtypes = [vp, i32]
    lib.icd_index_last_profile.argtypes = [vp, C.POINTER(_Profile)]
    lib.icd_index_profile_summary.argtypes = [vp, C.POINTER(_Profile), C.POINTER(C.c_int32)]
    have = lib.icd_abi_version()
    if have != ABI_VERSION:
        raise ImportError(f"{p} has ABI version {have}, this package needs {ABI_VERSION}: rebuild it (make -C rag_project_icd10_amd/csrc)")
    for name in EXPORTED_SYMBOLS:
        getattr(lib, name)  # AttributeError if the build is stale
    if path is None:
        _lib = lib
    return lib


def _check(lib, rc: int):
    if rc != 0:
        raise IcdError(rc, (lib.icd_last_error() or b"").decode("utf-8", "replace"))


def device_count() -> int:
    lib = load_library()
    n = lib.icd_device_count()
    if n < 0:
        _check(lib, n)
    return n


def _is_torch_tensor(x) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


def _current_stream_ptr(device_index: int) -> int:
    import torch
    return int(torch.cuda.current_stream(device_index).cuda_stream)


class IcdIndex:
    """Exact inner-product index over a dense fp32 corpus resident in one GPU's HBM.

    corpus : (n, dim) float32, C-contiguous numpy array, or a torch CUDA tensor on `device`.
    levels : (n,) ICD hierarchy level per row (1/2/3) or None (all level 1, the reference default
             services/milvus_service.py:247).
    id_base: id of row 0 (row-sharded corpora).
    """

    def __init__(self, corpus, levels=None, *, device: int = 0, max_nq: int = 16384, max_k: int = 100,
                 id_base: int = 0, permute: bool = True, probe: bool = True, center: bool = True):
        """permute / probe / center: A/B and test options of THIS index (icd_index_create flags ICD_CREATE_ROW_ORDER / _NO_PROBE /
        _NO_CENTER): performance decisions only, results are identical either way"""
        self._lib = load_library()
        self._h = C.c_void_p()
        on_dev = 0
        if _is_torch_tensor(corpus):
            if not corpus.is_cuda:
                corpus = corpus.detach().cpu().numpy()
            else:
                import torch
                if corpus.dtype != torch.float32 or not corpus.is_contiguous():
                    corpus = corpus.to(torch.float32).contiguous()
                if corpus.device.index != device:
                    raise ValueError(f"corpus is on cuda:{corpus.device.index}, index on device {device}")
                on_dev = 1
        if on_dev:
            import torch
            n, dim = int(corpus.shape[0]), int(corpus.shape[1])
            cptr = corpus.data_ptr()
            lv = None
            if levels is not None:
                lv = torch.as_tensor(levels).to(device=corpus.device, dtype=torch.int32).contiguous()
            lptr = lv.data_ptr() if lv is not None else None
            torch.cuda.synchronize(device)
        else:
            corpus = np.ascontiguousarray(corpus, dtype=np.float32)
            if corpus.ndim != 2:
                raise ValueError("corpus must be 2-D (n, dim)")
            n, dim = corpus.shape
            cptr = corpus.ctypes.data
            lv = None if levels is None else np.ascontiguousarray(levels, dtype=np.int32)
            if lv is not None and lv.shape != (n,):
                raise ValueError("levels must have shape (n,)")
            lptr = lv.ctypes.data if lv is not None else None
        self.n, self.dim, self.device, self.max_nq, self.max_k = int(n), int(dim), int(device), int(max_nq), int(max_k)
        self.id_base = int(id_base)
        flags = ((CREATE_CORPUS_ON_DEVICE if on_dev else 0) | (0 if permute else CREATE_ROW_ORDER) | (0 if probe else CREATE_NO_PROBE)
                 | (0 if center else CREATE_NO_CENTER))
        _check(self._lib, self._lib.icd_index_create(cptr, n, dim, lptr, id_base, device, max_nq, max_k, flags,
                                                      C.byref(self._h)))

    # -- lifecycle -----------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            for ref in list(getattr(self, "_group_refs", [])):   # (an IcdGroup borrows this handle: it goes first, never dangles)
                grp = ref()
                if grp is not None:
                    grp.close()
            self._lib.icd_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def closed(self) -> bool:
        return not self._h.value

    # -- search --------------------------------------------------------------------------------------
    def _prep_queries(self, queries):
        if _is_torch_tensor(queries) and queries.is_cuda:
            import torch
            q = queries
            if q.dim() == 1:
                q = q.unsqueeze(0)
            if q.dtype != torch.float32 or not q.is_contiguous():
                q = q.to(torch.float32).contiguous()
            if q.device.index != self.device:
                raise ValueError(f"queries on cuda:{q.device.index}, index on device {self.device}")
            return q, True
        if _is_torch_tensor(queries):
            queries = queries.detach().cpu().numpy()
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        return q, False

    def _validate(self, q, k):
        if self.closed:
            raise IcdError(-5, "index is closed")
        if q.shape[-1] != self.dim:
            raise ValueError(f"query dim {q.shape[-1]} != index dim {self.dim}")
        if not (1 <= k <= self.max_k):
            raise ValueError(f"k={k} outside 1..{self.max_k}")

    def search(self, queries, k: int = 10, mode: int = MODE_AUTO):
        """Raw top-k by inner product -> (scores float32 [nq,k], ids int64 [nq,k]), best first.
        Device tensors in -> device tensors out (enqueued on torch's current stream, no sync)."""
        q, on_dev = self._prep_queries(queries)
        self._validate(q, k)
        nq = int(q.shape[0])
        out_s, out_i = [], []
        for s0 in range(0, max(nq, 1), self.max_nq):
            qs = q[s0:s0 + self.max_nq]
            m = int(qs.shape[0])
            if on_dev:
                import torch
                sc = torch.empty((m, k), dtype=torch.float32, device=q.device)
                ids = torch.empty((m, k), dtype=torch.int64, device=q.device)
                if m:
                    _check(self._lib, self._lib.icd_index_search(self._h, qs.data_ptr(), m, k, 1, mode, sc.data_ptr(),
                                                                  ids.data_ptr(), 1, _current_stream_ptr(self.device)))
            else:
                sc = np.empty((m, k), dtype=np.float32)
                ids = np.empty((m, k), dtype=np.int64)
                if m:
                    _check(self._lib, self._lib.icd_index_search(self._h, qs.ctypes.data, m, k, 0, mode, sc.ctypes.data,
                                                                  ids.ctypes.data, 0, None))
            out_s.append(sc)
            out_i.append(ids)
        if len(out_s) == 1:
            return out_s[0], out_i[0]
        if on_dev:
            import torch
            return torch.cat(out_s), torch.cat(out_i)
        return np.concatenate(out_s), np.concatenate(out_i)

    def search_reweighted(self, queries, k: int = 10, mode: int = MODE_AUTO):
        """Raw top-k, then adj = float64(score) * w[level] and a stable descending re-sort of the k hits
        (services/milvus_service.py:290-295,314). Returns (adj f64, raw f32, ids i64, levels i32), each [nq,k]."""
        q, on_dev = self._prep_queries(queries)
        self._validate(q, k)
        nq = int(q.shape[0])
        outs = []
        for s0 in range(0, max(nq, 1), self.max_nq):
            qs = q[s0:s0 + self.max_nq]
            m = int(qs.shape[0])
            if on_dev:
                import torch
                adj = torch.empty((m, k), dtype=torch.float64, device=q.device)
                raw = torch.empty((m, k), dtype=torch.float32, device=q.device)
                ids = torch.empty((m, k), dtype=torch.int64, device=q.device)
                lv = torch.empty((m, k), dtype=torch.int32, device=q.device)
                if m:
                    _check(self._lib, self._lib.icd_index_search_reweighted(
                        self._h, qs.data_ptr(), m, k, 1, mode, adj.data_ptr(), raw.data_ptr(), ids.data_ptr(),
                        lv.data_ptr(), 1, _current_stream_ptr(self.device)))
            else:
                adj = np.empty((m, k), dtype=np.float64)
                raw = np.empty((m, k), dtype=np.float32)
                ids = np.empty((m, k), dtype=np.int64)
                lv = np.empty((m, k), dtype=np.int32)
                if m:
                    _check(self._lib, self._lib.icd_index_search_reweighted(
                        self._h, qs.ctypes.data, m, k, 0, mode, adj.ctypes.data, raw.ctypes.data, ids.ctypes.data,
                        lv.ctypes.data, 0, None))
            outs.append((adj, raw, ids, lv))
        if len(outs) == 1:
            return outs[0]
        if on_dev:
            import torch
            return tuple(torch.cat([o[i] for o in outs]) for i in range(4))
        return tuple(np.concatenate([o[i] for o in outs]) for i in range(4))

    def lookup_levels(self, ids):
        """Levels of hit ids (torch CUDA int64 tensor) -> int32 tensor; ids < 0 give 0."""
        import torch
        ids = ids.contiguous()
        out = torch.empty(ids.shape, dtype=torch.int32, device=ids.device)
        _check(self._lib, self._lib.icd_index_lookup_levels(self._h, ids.data_ptr(), ids.numel(), out.data_ptr(),
                                                            _current_stream_ptr(self.device)))
        return out

    # -- introspection -------------------------------------------------------------------------------
    def stats(self) -> dict:
        st = _Stats()
        _check(self._lib, self._lib.icd_index_stats(self._h, C.byref(st)))
        return {f: getattr(st, f) for f, _ in _Stats._fields_}

    def set_chunks(self, chunks: int):
        _check(self._lib, self._lib.icd_index_set_chunks(self._h, int(chunks)))

    def set_second_pass(self, enabled, adaptive: bool = True):
        """test / A-B switch: the second coarse pass over uncertified queries (default on), and the adaptive list count of
        large batches that follows from its counters"""
        _check(self._lib, self._lib.icd_index_set_second_pass(self._h, 0 if not enabled else (1 if adaptive else 2)))

    def set_option(self, name: str, value: int):
        """A/B and test options of this index (icd_index_set_option; `name` one of OPTIONS): family_order, stream_one, host_one,
        pacing_shift, pacing_lead, exact_narrow - performance decisions only, results are identical whatever they are set to"""
        _check(self._lib, self._lib.icd_index_set_option(self._h, OPTIONS[name], int(value)))

    def set_profiling(self, enabled, every: int = 1):
        """hipEvents around the kernels of every `every`-th search (read back by profile_summary / last_profile)"""
        _check(self._lib, self._lib.icd_index_set_profiling(self._h, max(1, int(every)) if enabled else 0))

    def last_profile(self) -> dict:
        p = _Profile()
        _check(self._lib, self._lib.icd_index_last_profile(self._h, C.byref(p)))
        return {f: float(getattr(p, f)) for f, _ in _Profile._fields_}


def group_unique_id() -> bytes:
    """rank 0: the RCCL unique id every rank passes to IcdGroup (icd_group_unique_id); hand the bytes over any side channel"""
    lib = load_library()
    buf = (C.c_uint8 * GROUP_ID_BYTES)()
    _check(lib, lib.icd_group_unique_id(C.cast(buf, C.c_void_p)))
    return bytes(buf)


class IcdGroup:
    """Multi-GPU search behind the C ABI (icd_group_*): one process per GPU, this rank's IcdIndex (a row shard created with
    id_base = its first global row, or a replica), an RCCL communicator owned by the group. `unique_id`: group_unique_id()
    of rank 0 (None for a single rank). search() takes the FULL query batch as a CUDA tensor on every rank."""

    def __init__(self, index: "IcdIndex", mode: int, rank: int = 0, world: int = 1, unique_id: Optional[bytes] = None,
                 max_nq: Optional[int] = None, max_k: Optional[int] = None, connect: bool = True, with_comm: Optional[bool] = None):
        """connect=True (default): prepare + connect in one go (`unique_id` required for world > 1). connect=False: only the
        LOCAL half (icd_group_prepare: argument checks, buffers, librccl) - the caller lets the ranks agree that every one of
        them got this far and then calls connect(unique_id) on all of them (collective) or close() on all of them."""
        self._lib = load_library()
        self._h = C.c_void_p()
        self.index, self.mode, self.rank, self.world = index, int(mode), int(rank), int(world)
        # (row-sharded: every rank searches the whole slice, so a slice is at most the index's max_nq; query-sharded: a
        #  slice of world x max_nq queries gives every rank max_nq of them)
        self.max_nq = int(max_nq or (index.max_nq if mode == GROUP_ROW_SHARD else index.max_nq * max(1, world)))
        self.max_k = int(max_k or min(index.max_k, 1024 // max(1, world) if mode == GROUP_ROW_SHARD else index.max_k))
        if connect and world > 1 and unique_id is None:
            raise ValueError(f"a group of {world} ranks needs rank 0's {GROUP_ID_BYTES}-byte unique id")
        if with_comm is None:   # (world = 1 with an id: a one-rank communicator, the collective path end to end)
            with_comm = world > 1 or unique_id is not None
        if index.closed:
            raise IcdError(-5, "index is closed")
        _check(self._lib, self._lib.icd_group_prepare(index._h, 1 if with_comm else 0, self.rank, self.world, self.mode, self.max_nq,
                                                       self.max_k, C.byref(self._h)))
        self.connected = not with_comm
        import weakref
        if not hasattr(index, "_group_refs"):
            index._group_refs = []
        index._group_refs.append(weakref.ref(self))   # the group borrows the index's handle: IcdIndex.close() closes it first
        if connect and with_comm:
            try:
                self.connect(unique_id)
            except Exception:
                self.close()
                raise

    def connect(self, unique_id: bytes):
        """COLLECTIVE (ncclCommInitRank): every rank of the group calls it, or none does"""
        if self.connected:
            return
        if unique_id is None or len(unique_id) != GROUP_ID_BYTES:
            raise ValueError(f"unique id must be {GROUP_ID_BYTES} bytes")
        idbuf = (C.c_uint8 * GROUP_ID_BYTES).from_buffer_copy(unique_id)
        _check(self._lib, self._lib.icd_group_connect(self._h, C.cast(idbuf, C.c_void_p)))
        self.connected = True

    def search(self, queries, k: int = 10, gather: bool = True):
        """-> (adj f64, raw f32, ids i64, levels i32), each [nq, k] CUDA tensors (query-sharded with gather=False: only
        this rank's rows). Batches larger than the group's max_nq go through in slices of that many queries."""
        import torch
        if not getattr(self, "_h", None) or not self._h.value:
            raise IcdError(-5, "group is closed")
        q, on_dev = self.index._prep_queries(queries)
        self.index._validate(q, k)   # (icd_group_search has no dim argument: a wrong width would be an out-of-bounds device read)
        if k > self.max_k:
            raise ValueError(f"k={k} outside 1..{self.max_k} (group)")
        if not on_dev:
            q = torch.from_numpy(q).to(torch.device("cuda", self.index.device))
        nq, dev = int(q.shape[0]), q.device
        local_only = self.mode == GROUP_QUERY_SHARD and not gather
        if local_only and nq > self.max_nq:
            # (slices of max_nq would each be split over the ranks: another set of rows than shard_bounds(nq) of the whole batch)
            raise ValueError(f"query-sharded search without gather takes at most the group's max_nq = {self.max_nq} queries per call (got {nq})")
        parts = []
        for s0 in range(0, max(nq, 1), self.max_nq):
            qs = q[s0:s0 + self.max_nq]
            m = int(qs.shape[0])
            outs = (torch.empty((m, k), dtype=torch.float64, device=dev), torch.empty((m, k), dtype=torch.float32, device=dev),
                    torch.empty((m, k), dtype=torch.int64, device=dev), torch.empty((m, k), dtype=torch.int32, device=dev))
            if m:
                _check(self._lib, self._lib.icd_group_search(self._h, qs.data_ptr(), m, k, 1 if gather else 0, *[o.data_ptr() for o in outs],
                                                              _current_stream_ptr(self.index.device)))
            keep = m
            if local_only:
                base, rem = divmod(m, self.world)
                keep = base + (1 if self.rank < rem else 0)
            parts.append(tuple(o[:keep] for o in outs))
        if len(parts) == 1:
            return parts[0]
        return tuple(torch.cat([p[i] for p in parts]) for i in range(4))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.icd_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _profile_summary(self) -> dict:
    """Mean per-kernel ms over the profiled searches since the last summary (events on the search stream)."""
    p, n = _Profile(), C.c_int32(0)
    _check(self._lib, self._lib.icd_index_profile_summary(self._h, C.byref(p), C.byref(n)))
    out = {f: float(getattr(p, f)) for f, _ in _Profile._fields_}
    out["count"] = int(n.value)
    return out


IcdIndex.profile_summary = _profile_summary


def hier_rescore(adj, ids, row_tags, q_params, weights, id_base: int = 0):
    """Device-side hierarchical rescoring of a batch of hit lists (icd_hier_rescore). adj f64 [nq,k], ids i64 [nq,k],
    row_tags u8 [n_rows], q_params f64 [nq,12] are tensors on one GPU; weights: 7 Python floats. Returns
    (order i32, enhanced f64, score f64, vs f64, hb f64, boost f64), each [nq,k], in the final order."""
    import torch
    lib = load_library()
    nq, k = adj.shape
    dev = adj.device
    adj = adj.to(torch.float64).contiguous()
    ids = ids.to(torch.int64).contiguous()
    row_tags = row_tags.to(torch.uint8).contiguous()
    q_params = q_params.to(device=dev, dtype=torch.float64).contiguous()
    assert q_params.shape == (nq, 12) and row_tags.device == dev
    w = (C.c_double * 7)(*[float(x) for x in weights])
    order = torch.empty((nq, k), dtype=torch.int32, device=dev)
    outs = [torch.empty((nq, k), dtype=torch.float64, device=dev) for _ in range(5)]
    _check(lib, lib.icd_hier_rescore(dev.index, adj.data_ptr(), ids.data_ptr(), nq, k, int(id_base), row_tags.numel(),
                                     row_tags.data_ptr(), q_params.data_ptr(), C.cast(w, C.c_void_p), order.data_ptr(),
                                     *[t.data_ptr() for t in outs], _current_stream_ptr(dev.index)))
    return (order, *outs)


def pack_winners(order, ids, raw, adj, enhanced, vs, hb, boost, kk: int):
    """The top kk rescored hits of every query as ONE float64 tensor [8, nq, kk] on the device (icd_pack_winners): id (plane 0
    holds the int64 ids' BIT PATTERNS: read it with .view(torch.int64)), raw score,
    level-reweighted score (of the hit order points at), order, enhanced, vector similarity, hierarchy boost, uncertainty boost."""
    import torch
    lib = load_library()
    nq, k = order.shape
    dev = order.device
    order = order.to(torch.int32).contiguous(); ids = ids.to(torch.int64).contiguous(); raw = raw.to(torch.float32).contiguous()
    f64 = [t.to(torch.float64).contiguous() for t in (adj, enhanced, vs, hb, boost)]
    out = torch.empty((8, nq, kk), dtype=torch.float64, device=dev)
    _check(lib, lib.icd_pack_winners(dev.index, order.data_ptr(), ids.data_ptr(), raw.data_ptr(), *[t.data_ptr() for t in f64],
                                     nq, k, kk, out.data_ptr(), _current_stream_ptr(dev.index)))
    return out


def score_stats(scores, order=None, use=None):
    """Row N3 (icd_score_stats): numpy-identical mean / std / var / max of every query's first `use` scores, plus the
    confidence service's model_uncertainty and prediction_variance. scores f64 [nq,k] on a GPU, order i32 [nq,k] or None
    (entries with order < 0 do not exist). Returns f64 [nq,6]."""
    import torch
    lib = load_library()
    nq, k = scores.shape
    dev = scores.device
    scores = scores.to(torch.float64).contiguous()
    if order is not None:
        order = order.to(device=dev, dtype=torch.int32).contiguous()
        assert order.shape == (nq, k)
    out = torch.empty((nq, 6), dtype=torch.float64, device=dev)
    _check(lib, lib.icd_score_stats(dev.index, scores.data_ptr(), order.data_ptr() if order is not None else None, nq, k,
                                    int(k if use is None else use), out.data_ptr(), _current_stream_ptr(dev.index)))
    return out


def cosine_rows(x, y):
    """Row N3 (icd_cosine_rows): sklearn-style cosine of every row of x (f32 [nq,dim] on a GPU) with the matching row of
    y ([nq,dim]) or with the single row y ([dim]); returns f64 [nq]."""
    import torch
    lib = load_library()
    nq, dim = x.shape
    dev = x.device
    x = x.to(torch.float32).contiguous()
    y = y.to(device=dev, dtype=torch.float32).contiguous()
    if y.dim() == 1:
        assert y.shape[0] == dim
        stride = 0
    else:
        assert y.shape == (nq, dim)
        stride = dim
    out = torch.empty((nq,), dtype=torch.float64, device=dev)
    _check(lib, lib.icd_cosine_rows(dev.index, x.data_ptr(), y.data_ptr(), stride, nq, dim, out.data_ptr(),
                                    _current_stream_ptr(dev.index)))
    return out


def packed_attention(qkv, starts, nseq: int, heads: int, max_len: int, out):
    """icd_packed_attention: softmax(Q K^T / 8) V per sequence and head over packed tokens. qkv f32 [T(+1), 3 * heads * 64]
    (row-contiguous), starts int32 [nseq + 1] on the same GPU, out f32 [T(+1), heads * 64]; sequences of at most 64 tokens.
    Enqueued on the current stream."""
    lib = load_library()
    dev = qkv.device
    assert qkv.is_cuda and qkv.stride(1) == 1 and out.stride(1) == 1 and starts.is_cuda
    _check(lib, lib.icd_packed_attention(dev.index, qkv.data_ptr(), qkv.stride(0), starts.data_ptr(), int(nseq), int(heads), 64,
                                         int(max_len), out.data_ptr(), out.stride(0), _current_stream_ptr(dev.index)))
    return out


class SmallEncoder:
    """The BERT-style sentence encoder for SMALL inputs (include/icd_search.h icd_encoder_*; csrc/encoder_small.hpp): up to
    ENCODER_MAX_SEQS sequences / ENCODER_MAX_TOKENS packed tokens per call, ONE graph launch per forward. Replaces the
    SentenceTransformer.encode call of EmbeddingService.encode_query / encode_single (reference
    services/embedding_service.py:97-102,117-120) for one string or a request's handful.

    Built from a transformers BertModel-like module on a CUDA device with fp32 parameters. The handle COPIES the four Linear
    weights of every layer at creation (into the order its GEMMs read them) and BORROWS embeddings, biases and LayerNorm
    parameters (this object keeps those tensors alive): rebuild it after changing the module's parameters."""

    @staticmethod
    def supported(bert) -> bool:
        try:
            cfg = bert.config
            p = bert.embeddings.word_embeddings.weight
            import torch
            return (type(bert).__name__ in ("BertModel", "XLMRobertaModel", "RobertaModel") and p.is_cuda and p.dtype == torch.float32
                    and getattr(cfg, "position_embedding_type", None) in (None, "absolute") and getattr(cfg, "hidden_act", "gelu") == "gelu"
                    and not getattr(cfg, "is_decoder", False) and int(cfg.hidden_size) in (768, 1024) and int(cfg.hidden_size) // int(cfg.num_attention_heads) == 64
                    and int(cfg.intermediate_size) % int(cfg.hidden_size) == 0 and int(cfg.hidden_size) <= int(cfg.intermediate_size) <= 4 * int(cfg.hidden_size))
        except Exception:
            return False

    def __init__(self, bert, arithmetic: Optional[str] = None):
        """arithmetic: "bf16x3" (the default; env ICD_ENCODER_ARITH) = the split-bf16 GEMMs, "fp32" = fp32-input MFMAs - of BOTH
        forms (encode / encode_many) alike: whichever it is, a list's rows equal the one-string call's bit for bit"""
        import torch
        self._lib = load_library()
        self.arithmetic = (arithmetic or os.environ.get("ICD_ENCODER_ARITH", "bf16x3")).strip().lower()
        if self.arithmetic not in ENCODER_ARITH:
            raise ValueError(f"encoder arithmetic {self.arithmetic!r}: one of {sorted(ENCODER_ARITH)}")
        cfg = bert.config
        emb = bert.embeddings
        self.hidden = int(cfg.hidden_size)
        self.device = emb.word_embeddings.weight.device.index or 0
        keep = []   # every tensor the handle points at

        def ptr(t):
            t = t.detach()
            if not t.is_contiguous():
                t = t.contiguous()
            assert t.dtype == torch.float32 and t.is_cuda
            keep.append(t)
            return t.data_ptr()
        d = _EncoderDesc()
        d.layers, d.hidden, d.heads, d.inter = len(bert.encoder.layer), self.hidden, int(cfg.num_attention_heads), int(cfg.intermediate_size)
        d.vocab, d.max_pos = int(emb.word_embeddings.weight.shape[0]), int(emb.position_embeddings.weight.shape[0])
        d.pos_offset = int(emb.padding_idx) + 1 if type(bert).__name__ != "BertModel" else 0
        d.ln_eps = float(cfg.layer_norm_eps)
        d.arithmetic = ENCODER_ARITH[self.arithmetic]
        d.word_emb, d.pos_emb = ptr(emb.word_embeddings.weight), ptr(emb.position_embeddings.weight)
        d.type_emb0 = ptr(emb.token_type_embeddings.weight[0])
        d.emb_ln_g, d.emb_ln_b = ptr(emb.LayerNorm.weight), ptr(emb.LayerNorm.bias)
        per = {name: [] for name in _ENC_LAYER_FIELDS}
        scratch = []
        for l in bert.encoder.layer:
            a = l.attention.self
            wqkv = torch.cat([a.query.weight, a.key.weight, a.value.weight], 0).detach().contiguous()   # (copied by the handle at create: not kept)
            scratch.append(wqkv)
            per["w_qkv"].append(wqkv.data_ptr())
            per["b_qkv"].append(ptr(torch.cat([a.query.bias, a.key.bias, a.value.bias], 0)))
            per["w_ao"].append(ptr(l.attention.output.dense.weight)); per["b_ao"].append(ptr(l.attention.output.dense.bias))
            per["ln1_g"].append(ptr(l.attention.output.LayerNorm.weight)); per["ln1_b"].append(ptr(l.attention.output.LayerNorm.bias))
            per["w_up"].append(ptr(l.intermediate.dense.weight)); per["b_up"].append(ptr(l.intermediate.dense.bias))
            per["w_down"].append(ptr(l.output.dense.weight)); per["b_down"].append(ptr(l.output.dense.bias))
            per["ln2_g"].append(ptr(l.output.LayerNorm.weight)); per["ln2_b"].append(ptr(l.output.LayerNorm.bias))
        arrays = {}
        for name in _ENC_LAYER_FIELDS:
            arrays[name] = (C.c_void_p * d.layers)(*per[name])
            setattr(d, name, C.cast(arrays[name], C.POINTER(C.c_void_p)))
        self._keep = keep
        h = C.c_void_p()
        _check(self._lib, self._lib.icd_encoder_create(self.device, C.byref(d), C.byref(h)))   # (synchronises: the weight copies are done)
        del scratch
        self._h = h

    def fits(self, lengths) -> bool:
        return 0 < len(lengths) <= ENCODER_MAX_SEQS and sum(lengths) <= ENCODER_MAX_TOKENS and min(lengths) >= 1

    def encode(self, ids, pooling: str = "mean", normalize: bool = True, to_device: bool = False, hidden: bool = False):
        """ids: token id lists (special tokens included). -> float32 [n, hidden] (numpy, or a CUDA tensor with to_device=True:
        enqueued on torch's current stream); with hidden=True also the last hidden state of every token, packed [T, hidden]
        (a CUDA tensor)."""
        import torch
        if not getattr(self, "_h", None) or not self._h.value:
            raise IcdError(-5, "encoder is closed")
        lengths = np.fromiter((len(x) for x in ids), dtype=np.int32, count=len(ids))
        flat = np.fromiter((t for x in ids for t in x), dtype=np.int32, count=int(lengths.sum()))
        n = len(ids)
        dev = torch.device("cuda", self.device)
        hid = torch.empty((int(lengths.sum()), self.hidden), dtype=torch.float32, device=dev) if hidden else None
        if to_device:
            out = torch.empty((n, self.hidden), dtype=torch.float32, device=dev)
            optr = out.data_ptr()
        else:
            out = np.empty((n, self.hidden), dtype=np.float32)
            optr = out.ctypes.data
        _check(self._lib, self._lib.icd_encoder_encode(self._h, flat.ctypes.data, lengths.ctypes.data, n, 1 if pooling == "cls" else 0,
                                                      1 if normalize else 0, optr, 1 if to_device else 0,
                                                      hid.data_ptr() if hidden else None, _current_stream_ptr(self.device)))
        return (out, hid) if hidden else out

    def fits_each(self, lengths) -> bool:
        """every sequence fits a call of its own: encode_many takes the list whatever its size"""
        return len(lengths) > 0 and 1 <= min(lengths) and max(lengths) <= ENCODER_MAX_TOKENS

    def encode_many(self, ids, pooling: str = "mean", normalize: bool = True, to_device: bool = False):
        """ANY number of sequences (each of at most ENCODER_MAX_TOKENS tokens) through the same kernels, cut into calls by the
        library (icd_encoder_encode_many): row i equals encode([ids[i]]) BIT FOR BIT - the arithmetic of a sequence does not
        depend on what shares its call. -> float32 [n, hidden], numpy or (to_device) a CUDA tensor on the current stream."""
        import torch
        if not getattr(self, "_h", None) or not self._h.value:
            raise IcdError(-5, "encoder is closed")
        n = len(ids)
        lengths = np.fromiter((len(x) for x in ids), dtype=np.int32, count=n)
        flat = np.fromiter((t for x in ids for t in x), dtype=np.int32, count=int(lengths.sum()))
        out = torch.empty((n, self.hidden), dtype=torch.float32, device=torch.device("cuda", self.device))
        _check(self._lib, self._lib.icd_encoder_encode_many(self._h, flat.ctypes.data, lengths.ctypes.data, n, 1 if pooling == "cls" else 0,
                                                           1 if normalize else 0, out.data_ptr(), 1, _current_stream_ptr(self.device)))
        return out if to_device else out.cpu().numpy()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.icd_encoder_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def split_bf16x3(x, gelu: bool = False):
    """icd_split_bf16x3: x f32 [rows, cols] on the GPU (row-contiguous) -> bf16 [rows, 3 * cols + 64] = [hi | hi | lo | 1 1 0 ...],
    the A operand of a split-bf16 GEMM (services/embedding_service.py _PackedBert); gelu: through erf-GELU first.
    Enqueued on the current stream."""
    import torch
    lib = load_library()
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and x.shape[1] % 8 == 0 and x.shape[1] >= 64
    out = torch.empty((x.shape[0], 3 * x.shape[1] + SPLIT_TAIL), dtype=torch.bfloat16, device=x.device)
    _check(lib, lib.icd_split_bf16x3(x.device.index, x.data_ptr(), int(x.shape[0]), int(x.shape[1]), int(x.stride(0)), 1 if gelu else 0,
                                     out.data_ptr(), _current_stream_ptr(x.device.index)))
    return out


def merge_topk(scores, ids, levels, k: int):
    """Row-sharded search, step 2 (device tensors): scores/ids/levels are [G, nq, k] gathered from G
    shards; returns (adj f64, raw f32, ids i64, levels i32), each [nq, k], reweighted and re-sorted."""
    import torch
    lib = load_library()
    G, nq, kk = scores.shape
    assert kk == k
    dev = scores.device
    scores = scores.to(torch.float32).contiguous()
    ids = ids.to(torch.int64).contiguous()
    levels = levels.to(torch.int32).contiguous()
    adj = torch.empty((nq, k), dtype=torch.float64, device=dev)
    raw = torch.empty((nq, k), dtype=torch.float32, device=dev)
    oid = torch.empty((nq, k), dtype=torch.int64, device=dev)
    olv = torch.empty((nq, k), dtype=torch.int32, device=dev)
    _check(lib, lib.icd_merge_topk(dev.index, scores.data_ptr(), ids.data_ptr(), levels.data_ptr(), G, nq, k,
                                   adj.data_ptr(), raw.data_ptr(), oid.data_ptr(), olv.data_ptr(),
                                   _current_stream_ptr(dev.index)))
    return adj, raw, oid, olv
