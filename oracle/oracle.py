"""Python face of the CPU oracle — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (rag_project_icd10_amd/) never does.

* `flat_ip_topk`, `reweight`, `merge`: ctypes wrappers of oracle/icd_oracle.c (the canonical fp32
  fmaf-chain restatement of the Milvus FLAT/IP search called at services/milvus_service.py:280-285 and
  of the level reweight at :290-295,314,550-558). Bit-exact target of the HIP kernels.
* `reference_shaped_search`: the reference's CALL SHAPE on the CPU (one query per call: C @ q in
  fp32 BLAS, top-k by raw IP, level weight in float64, stable re-sort) — used as the cpu_baseline
  ("port") in bench.py and as a cross-check of the C oracle (BLAS summation order differs from the
  canonical chain by <= ~2e-7, so ids may differ only for near-ties).
* `level_weight`: services/milvus_service.py:550-558.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libicd_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "icd_oracle.c")):
        subprocess.run(["make", "-C", _HERE, "-B", "libicd_oracle.so"], check=True, capture_output=True)
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        l = C.CDLL(_LIB_PATH)
        vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        l.icd_oracle_flat_ip_topk.argtypes = [vp, i64, i32, vp, i64, i32, i64, i32, vp, vp]
        l.icd_oracle_flat_ip_topk.restype = i32
        l.icd_oracle_scores.argtypes = [vp, vp, i64, i32, vp]
        l.icd_oracle_reweight.argtypes = [vp, vp, vp, i64, i64, i32, vp, vp, vp, vp]
        l.icd_oracle_merge.argtypes = [vp, vp, i32, i64, i32, vp, vp]
        l.icd_oracle_level_weight.argtypes = [i32]
        l.icd_oracle_level_weight.restype = C.c_double
        _lib = l
    return _lib


def scores(query: np.ndarray, corpus: np.ndarray) -> np.ndarray:
    corpus = np.ascontiguousarray(corpus, np.float32)
    query = np.ascontiguousarray(query, np.float32)
    out = np.empty(corpus.shape[0], np.float32)
    lib().icd_oracle_scores(query.ctypes.data, corpus.ctypes.data, corpus.shape[0], corpus.shape[1], out.ctypes.data)
    return out


def flat_ip_topk(corpus: np.ndarray, queries: np.ndarray, k: int, id_base: int = 0, nthreads: int = 0):
    corpus = np.ascontiguousarray(corpus, np.float32)
    queries = np.ascontiguousarray(queries, np.float32)
    if queries.ndim == 1:
        queries = queries[None]
    nq = queries.shape[0]
    s = np.empty((nq, k), np.float32)
    i = np.empty((nq, k), np.int64)
    rc = lib().icd_oracle_flat_ip_topk(corpus.ctypes.data, corpus.shape[0], corpus.shape[1], queries.ctypes.data, nq, k,
                                       id_base, nthreads, s.ctypes.data, i.ctypes.data)
    if rc != 0:
        raise RuntimeError("icd_oracle_flat_ip_topk failed")
    return s, i


def reweight(raw: np.ndarray, ids: np.ndarray, levels: np.ndarray, id_base: int = 0):
    raw = np.ascontiguousarray(raw, np.float32)
    ids = np.ascontiguousarray(ids, np.int64)
    levels = np.ascontiguousarray(levels, np.int32)
    nq, k = raw.shape
    adj = np.empty((nq, k), np.float64)
    oraw = np.empty((nq, k), np.float32)
    oid = np.empty((nq, k), np.int64)
    olv = np.empty((nq, k), np.int32)
    lib().icd_oracle_reweight(raw.ctypes.data, ids.ctypes.data, levels.ctypes.data, id_base, nq, k, adj.ctypes.data,
                              oraw.ctypes.data, oid.ctypes.data, olv.ctypes.data)
    return adj, oraw, oid, olv


def merge(scores_g: np.ndarray, ids_g: np.ndarray, k: int):
    scores_g = np.ascontiguousarray(scores_g, np.float32)
    ids_g = np.ascontiguousarray(ids_g, np.int64)
    G, nq, kk = scores_g.shape
    assert kk == k
    s = np.empty((nq, k), np.float32)
    i = np.empty((nq, k), np.int64)
    lib().icd_oracle_merge(scores_g.ctypes.data, ids_g.ctypes.data, G, nq, k, s.ctypes.data, i.ctypes.data)
    return s, i


def level_weight(level: int) -> float:
    return {1: 1.2, 2: 1.0, 3: 0.8}.get(int(level), 1.0)


def reference_shaped_search(corpus: np.ndarray, levels: np.ndarray, query: np.ndarray, k: int):
    """One query, the way MilvusService.search drives the engine (services/milvus_service.py:280-314)."""
    sc = corpus @ query  # FLAT / IP scan, fp32
    n = sc.shape[0]
    kk = min(k, n)
    part = np.argpartition(-sc, kk - 1)[:kk] if kk < n else np.arange(n)
    order = part[np.lexsort((part, -sc[part]))]  # score desc, id asc
    hits = []
    for i in order:
        base = float(sc[i])
        hits.append((float(base * level_weight(levels[i])), base, int(i)))
    hits.sort(key=lambda h: h[0], reverse=True)  # stable
    return hits
