#!/usr/bin/env python3
"""single_query_probe.py - the reference's own call shape: ONE query per MilvusService.search call
(services/milvus_service.py:280-285, data=[query_vector.tolist()]), k = 5 and 10, 40 474 rows.

For each k, with the single-launch streaming kernel on (default) and off (the per-index option stream_one = 0: memset + stream_topk +
reduce_lists + finalize), three ways of calling:
  device   IcdIndex.search_reweighted on a device-resident query, 400 calls enqueued back to back between two events (what the
           GPU needs per call; bytes / time against 8 TB/s is the HBM-roofline fraction of the call);
  host     the same call with a numpy query and numpy results (H2D + kernels + D2H + stream sync: what icd_index_search costs
           a host caller, one call at a time);
  service  MilvusService.search (numpy vector -> list of hit dicts), one call at a time.
Results of the two kernels are compared bit for bit on 64 queries. Run on the GPU box: python3 scripts/probe/single_query_probe.py"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    from rag_project_icd10_amd import _native
    from rag_project_icd10_amd._native import IcdIndex
    lib = _native.load_library()
    n, dim = 40474, 768
    rng = np.random.default_rng(1234)
    corpus = rng.standard_normal((n, dim), dtype=np.float32)
    corpus /= np.linalg.norm(corpus, axis=1, keepdims=True)
    r = np.random.default_rng(1235).random(n)
    levels = np.where(r < 0.1243, 1, np.where(r < 0.4234, 2, 3)).astype(np.int32)
    queries = rng.standard_normal((64, dim), dtype=np.float32)
    queries /= np.linalg.norm(queries, axis=1, keepdims=True)
    index = IcdIndex(corpus, levels, max_nq=16384, max_k=100)
    dq = torch.from_numpy(queries).cuda()
    bytes_per_call = n * dim * 4
    ref = {}
    for k in (5, 10):
        for one in (1, 0):
            index.set_option("stream_one", one)
            outs = [index.search_reweighted(dq[i:i + 1], k) for i in range(64)]
            torch.cuda.synchronize()
            got = tuple(torch.cat([o[j] for o in outs]).cpu().numpy().tobytes() for j in range(4))
            if one:
                ref[k] = got
            else:
                assert got == ref[k], "the two kernels disagree"
            # device: back-to-back enqueues
            for _ in range(50):
                index.search_reweighted(dq[:1], k)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for i in range(400):
                index.search_reweighted(dq[i & 63:(i & 63) + 1], k)
            e1.record()
            torch.cuda.synchronize()
            wall_us = (time.perf_counter() - t0) / 400 * 1e6
            dev_us = e0.elapsed_time(e1) / 400 * 1e3
            # host: numpy in, numpy out, one call at a time; with the single-launch kernel also the four forms of handing the
            # vector over and of waiting (the per-index option host_one: 0 = copy + stream synchronisation, 1 = vector in the kernel
            # arguments, 2 = polled completion word, 3 = both, the default), interleaved
            forms = (0, 1, 2, 3) if one else (3,)
            lat_by = {b: [] for b in forms}
            for _ in range(20):
                index.search_reweighted(queries[:1], k)
            for i in range(200):
                for b in forms:
                    index.set_option("host_one", b)
                    t0 = time.perf_counter()
                    index.search_reweighted(queries[i & 63:(i & 63) + 1], k)
                    lat_by[b].append((time.perf_counter() - t0) * 1e6)
            index.set_option("host_one", 3)
            for b in forms:
                lat_by[b].sort()
            if one:
                print(f"k={k:2d} host call by form (median us): " + ", ".join(f"host_one={b}: {lat_by[b][100]:.1f}" for b in forms))
            lat = lat_by[3]
            print(f"k={k:2d} single-launch kernel {'on ' if one else 'off'}: device {dev_us:6.1f} us per call (host enqueue {wall_us:5.1f} us; "
                  f"{bytes_per_call / dev_us / 1e6:5.2f} TB/s = {bytes_per_call / dev_us / 1e6 / 8.0:.3f} of 8 TB/s) | "
                  f"host call median {lat[100]:6.1f} us, p10 {lat[20]:6.1f}, p90 {lat[180]:6.1f}")
    index.set_option("stream_one", 1)
    # 3 ... 8 queries per call (a /query request's diagnoses in one search_batch): GPU time by the library's own events
    for nq in (1, 2, 3, 4, 8, 16):
        for one in (1, 0):
            index.set_option("stream_one", one)
            for _ in range(20):
                index.search_reweighted(dq[:nq], 10)
            torch.cuda.synchronize()
            index.set_profiling(True)
            index.profile_summary()
            for i in range(100):
                index.search_reweighted(dq[i & 31:(i & 31) + nq], 10)
            torch.cuda.synchronize()
            prof = index.profile_summary()
            index.set_profiling(False)
            print(f"nq={nq:2d} k=10 single-launch kernel {'on ' if one else 'off'}: GPU {prof['ms_total'] * 1e3:6.1f} us per call (library events, {prof['count']} calls)")
    index.set_option("stream_one", 1)
    index.close()
    # the service: the reference's search(), numpy vector in, hit dicts out
    tmp = tempfile.mkdtemp()
    os.environ.update({"MILVUS_MODE": "local", "MILVUS_DB_PATH": tmp, "MILVUS_COLLECTION_NAME": "icd10"})
    from rag_project_icd10_amd.services.milvus_service import MilvusService

    class Emb:
        def encode_query(self, text):
            return np.zeros(dim, np.float32)
    svc = MilvusService(Emb())
    recs = [{"code": f"X{i:05d}", "preferred_zh": f"t{i}", "level": int(levels[i]), "parent_code": "", "category_path": "", "semantic_text": ""} for i in range(n)]
    for b in range(0, n, 4096):
        svc.insert_records(recs[b:b + 4096], [corpus[i] for i in range(b, min(n, b + 4096))])
    svc.search(queries[0], 5)            # (builds the service's index: the option below is per index)
    for k in (5, 10):
        for one in (1, 0):
            svc._index.set_option("stream_one", one)
            for i in range(20):
                svc.search(queries[i], k)
            lat = []
            for i in range(200):
                t0 = time.perf_counter()
                hits = svc.search(queries[i & 63], k)
                lat.append((time.perf_counter() - t0) * 1e6)
            assert len(hits) == k
            lat.sort()
            print(f"k={k:2d} single-launch kernel {'on ' if one else 'off'}: MilvusService.search median {lat[100]:6.1f} us, p10 {lat[20]:6.1f}, p90 {lat[180]:6.1f}")
    svc._index.set_option("stream_one", 1)


if __name__ == "__main__":
    main()
