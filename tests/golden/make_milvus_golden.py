#!/usr/bin/env python3
"""Golden fixture for MilvusService.search / insert_records, produced by RUNNING THE REFERENCE'S OWN
services/milvus_service.py (unchanged, imported from /root/reference) in the build container:

    python tests/golden/make_milvus_golden.py

Everything around the engine call is the reference's Python: the has_collection guard, the
`data=[query_vector.tolist()]` marshalling, base = float(distance), adjusted = float(base * weight)
in Python doubles, the level-weight table with its default, the stable `list.sort(reverse=True)`,
the result-dict shape, `[]` on a missing collection and on an engine exception, None -> "" and the
defaults of insert_records, ValueError on a length mismatch.

The ENGINE itself (pymilvus==2.5.10 -> Milvus Lite -> knowhere FLAT / IP, requirements.txt:35) is a
native third-party package that is neither vendored in the reference nor installable here. It is
replaced by `StandInFlatIPClient` below - a STAND-IN, labelled as such: the published definition of a
FLAT index with metric IP (exhaustive fp32 inner product of the query against every row, the `limit`
largest, best first), with this build's canonical summation order (fp32 fmaf chain, d ascending) and
tie rule (equal distance: lower primary key first) - both unpinned in Milvus itself (DESIGN.md
section 2). The generator cross-checks the stand-in's distances bit for bit against oracle/icd_oracle.c
and against numpy float64 to 2e-6, so the fixture is consistent with the oracle by construction of
the engine part and pins the oracle's reweight / ordering part against the reference.

Only DATA is written (inputs + the reference's outputs):
  milvus_search_vectors.npz   corpus [384,768] f32, queries [10,768] f32, tiny corpus [3,768]
  milvus_search_cases.json    records handed to insert_records, the rows the reference handed the engine
                              (vectors dropped), search outputs per (query, top_k), the error cases
  milvus_batch_vectors.npz    batch_queries [160,768] f32: a batch large enough to take the MFMA coarse path of the
                              HIP index (more than 16 queries), among them the designed tie / duplicate / zero queries
  milvus_batch_cases.json     the reference's `search(q, 10)` and `search(q, 20)` looped over those 160 queries
"""
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DIM = 768


def chain_f32(C, q):
    """fp32 fmaf chain over d ascending, vectorised over rows. a*b is exact in float64 (24+24 significant bits), the sum
    is rounded to float64 and then to float32: identical to a true fmaf unless the float64 sum is inexact AND lands on a
    float32 rounding tie (probability ~2^-29 per step); the generator asserts equality with the C oracle's fmaf."""
    acc = np.zeros(C.shape[0], np.float32)
    for d in range(C.shape[1]):
        acc = (C[:, d].astype(np.float64) * np.float64(q[d]) + acc.astype(np.float64)).astype(np.float32)
    return acc


class StandInFlatIPClient:
    """STAND-IN for pymilvus.MilvusClient on a FLAT / IP collection (see the module docstring)."""
    instances = []

    def __init__(self, uri=None, **kwargs):
        self.uri = uri
        self.collections = {}
        self.loaded = set()
        self.insert_calls = []
        self.fail_search = False
        StandInFlatIPClient.instances.append(self)

    # -- schema / admin calls the reference makes (services/milvus_service.py:120-206) --
    def has_collection(self, collection_name):
        return collection_name in self.collections

    def create_schema(self, **kwargs):
        fields = []
        return types.SimpleNamespace(fields=fields, add_field=lambda **kw: fields.append(kw))

    def prepare_index_params(self):
        idx = []
        return types.SimpleNamespace(indexes=idx, add_index=lambda **kw: idx.append(kw))

    def create_collection(self, collection_name, schema=None, index_params=None):
        assert index_params.indexes[0]["index_type"] == "FLAT" and index_params.indexes[0]["metric_type"] == "IP"
        self.collections[collection_name] = []

    def get_load_state(self, collection_name):
        return "Loaded" if collection_name in self.loaded else "NotLoad"

    def load_collection(self, collection_name):
        self.loaded.add(collection_name)

    def drop_collection(self, collection_name):
        self.collections.pop(collection_name, None)

    def close(self):
        pass

    def insert(self, collection_name, data):
        self.insert_calls.append(data)
        rows = self.collections[collection_name]
        for row in data:
            row = dict(row)
            row["id"] = len(rows)  # auto_id primary key, insertion order
            rows.append(row)

    def search(self, collection_name, data, limit, output_fields):
        if self.fail_search:
            raise RuntimeError("stand-in engine failure")
        rows = self.collections[collection_name]
        out = []
        for qlist in data:
            q = np.asarray(qlist, dtype=np.float32)
            C = np.asarray([r["vector"] for r in rows], dtype=np.float32).reshape(len(rows), -1)
            dist = chain_f32(C, q)
            order = sorted(range(len(rows)), key=lambda i: (-float(dist[i]), rows[i]["id"]))[:limit]
            hits = []
            for i in order:
                hit = {"id": rows[i]["id"], "distance": float(dist[i])}
                for f in output_fields:
                    hit[f] = rows[i][f]
                hits.append(hit)
            out.append(hits)
        return out


def stub_modules():
    class _Logger:
        def __getattr__(self, name):
            return lambda *a, **k: None

    loguru = types.ModuleType("loguru")
    loguru.logger = _Logger()
    sys.modules["loguru"] = loguru
    pm = types.ModuleType("pymilvus")
    pm.MilvusClient = StandInFlatIPClient

    class _DataType:
        def __getattr__(self, name):
            return name
    pm.DataType = _DataType()
    sys.modules["pymilvus"] = pm


def make_vectors():
    rng = np.random.default_rng(20251003)
    corpus = rng.standard_normal((384, DIM)).astype(np.float32)
    corpus /= np.linalg.norm(corpus, axis=1, keepdims=True)
    levels = rng.choice([1, 2, 3], size=384, p=[0.1243, 0.2991, 0.5766]).astype(np.int64)
    levels[[5, 77, 300]] = [0, 4, 7]                      # outside {1,2,3}: weight 1.0
    # designed rows for the one-hot query e_7 (score = row[7] exactly): equal ADJUSTED scores from different levels
    for row, val, lv in ((10, 0.625, 3), (20, 0.5, 2), (30, 0.5, 7), (40, 0.4375, 1), (50, 0.46875, 2)):
        corpus[row] = 0.0
        corpus[row, 7] = val
        corpus[row, 8] = 0.25 * (row % 3)
        levels[row] = lv
    corpus[200] = corpus[100]                             # duplicate row: equal raw scores, different levels
    levels[100], levels[200] = 3, 1
    corpus[201] = corpus[101]                             # duplicate row, same weight: tie on both scores
    levels[101], levels[201] = 2, 4
    queries = rng.standard_normal((6, DIM)).astype(np.float32)
    queries /= np.linalg.norm(queries, axis=1, keepdims=True)
    onehot = np.zeros((1, DIM), np.float32)
    onehot[0, 7] = 1.0
    queries = np.concatenate([queries, onehot, corpus[100:101], corpus[101:102] * np.float32(0.5), np.zeros((1, DIM), np.float32)])
    tiny = rng.standard_normal((3, DIM)).astype(np.float32)
    return corpus, levels, queries, tiny


def make_records(levels):
    recs = []
    for i, lv in enumerate(levels):
        code = f"T{i // 10:02d}.{i % 10}{'0' if lv == 3 else ''}"
        rec = {"code": code, "preferred_zh": f"测试疾病{i}", "has_complication": i % 11 == 0,
               "main_code": code, "secondary_code": "", "level": int(lv), "parent_code": code.split(".")[0] if lv > 1 else "",
               "category_path": " > ".join([code.split(".")[0], code]) if lv > 1 else code,
               "semantic_text": f"测试疾病{i} | ICD-10: {code}"}
        if i % 7 == 0:
            rec["secondary_code"] = None                  # None -> "" (milvus_service.py:222-228)
        if i % 13 == 0:
            rec["main_code"] = None
        if i == 3:                                        # defaults of .get() (milvus_service.py:230-243)
            rec = {"code": code}
        recs.append(rec)
    return recs


def main():
    stub_modules()
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MILVUS_MODE"] = "local"
    os.environ["MILVUS_DB_PATH"] = "/tmp/_milvus_golden/standin.db"
    os.environ["MILVUS_COLLECTION_NAME"] = "icd10"
    from services.milvus_service import MilvusService  # the REFERENCE's module, unchanged
    import oracle as orc

    corpus, levels, queries, tiny = make_vectors()
    records = make_records(levels)

    class DimProbe:                                       # embedding_service as main.py passes it (dimension probe only)
        def encode_query(self, text):
            return np.zeros(DIM, np.float32)

    svc = MilvusService(DimProbe())
    client = svc.client
    assert isinstance(client, StandInFlatIPClient) and svc.dimension == DIM
    for b in range(0, 384, 128):                          # 128-row insert batches (tools/build_database.py:183-192)
        assert svc.insert_records(records[b:b + 128], [corpus[i] for i in range(b, min(b + 128, 384))]) is True
    inserted = [{k: v for k, v in row.items() if k != "vector"} for call in client.insert_calls for row in call]
    assert len(inserted) == 384

    # engine cross-checks (generation time only): stand-in == C oracle bit for bit; close to float64
    for qi in range(len(queries)):
        mine = chain_f32(corpus, queries[qi])
        assert mine.tobytes() == orc.scores(queries[qi], corpus).tobytes(), qi
        assert np.max(np.abs(mine.astype(np.float64) - corpus.astype(np.float64) @ queries[qi].astype(np.float64))) < 2e-6

    plan = {0: (1, 5, 10, 100), 1: (10,), 2: (5,), 3: (10,), 4: (2,), 5: (50,), 6: (1, 3, 5, 10), 7: (1, 5, 10), 8: (5,), 9: (5,)}
    cases = []
    for qi, ks in plan.items():
        for k in ks:
            cases.append({"query_index": qi, "top_k": k, "out": svc.search(queries[qi], k)})
    default_k = svc.search(queries[0])                    # top_k defaults to 10
    assert default_k == [c for c in cases if c["query_index"] == 0 and c["top_k"] == 10][0]["out"]

    # n < k: a second collection of three rows
    os.environ["MILVUS_COLLECTION_NAME"] = "tiny"
    svc_tiny = MilvusService(DimProbe())
    tiny_recs = [{"code": f"Z0{i}", "preferred_zh": f"小{i}", "level": i + 1} for i in range(3)]
    assert svc_tiny.insert_records(tiny_recs, [tiny[i] for i in range(3)]) is True
    tiny_out = svc_tiny.search(queries[1], 10)
    assert len(tiny_out) == 3

    # missing collection -> [] ; engine exception -> [] ; length mismatch -> ValueError ; list embedding -> False
    svc_tiny.client.drop_collection("tiny")
    missing = svc_tiny.search(queries[1], 5)
    client.fail_search = True
    exc_out = svc.search(queries[0], 5)
    client.fail_search = False
    try:
        svc.insert_records(records[:2], [corpus[0]])
        mismatch = "no error"
    except ValueError as e:
        mismatch = "ValueError: " + str(e)
    list_embedding = svc.insert_records(records[:1], [[0.0] * DIM])   # .tolist() on a list -> caught -> False (:231,266-268)
    weights = {str(lv): svc._calculate_level_weight(lv) for lv in (-1, 0, 1, 2, 3, 4, 7, 100)}

    # a BATCH of queries, the reference's search looped (it has no batch call): the case that puts the reference's own
    # output against the fp16-MFMA coarse kernel + certified rescoring (batches of <= 16 take the streaming kernel)
    rngb = np.random.default_rng(20251004)
    batch = rngb.standard_normal((160, DIM)).astype(np.float32)
    batch /= np.linalg.norm(batch, axis=1, keepdims=True)
    batch[5] = corpus[100]                                # duplicate rows 100 / 200: raw tie, different levels
    batch[6] = queries[6]                                 # the one-hot query: equal adjusted scores from different levels
    batch[7] = 0.0                                        # all scores zero: every hit ties
    batch[8] = corpus[101] * np.float32(0.5)              # duplicate rows 101 / 201, same weight: tie on both scores
    batch[9] = batch[10]                                  # two identical queries in one batch
    for qi in range(len(batch)):
        assert chain_f32(corpus, batch[qi]).tobytes() == orc.scores(batch[qi], corpus).tobytes(), qi
    batch_out = {str(k): [svc.search(batch[i], k) for i in range(len(batch))] for k in (10, 20)}
    np.savez_compressed(os.path.join(HERE, "milvus_batch_vectors.npz"), batch_queries=batch)
    json.dump({"generated_by": "tests/golden/make_milvus_golden.py: /root/reference/services/milvus_service.py search() looped over "
                               "the 160 batch queries (corpus / records of milvus_search_cases.json)",
               "out": batch_out}, open(os.path.join(HERE, "milvus_batch_cases.json"), "w"), ensure_ascii=False, indent=0)

    np.savez_compressed(os.path.join(HERE, "milvus_search_vectors.npz"), corpus=corpus, levels=levels.astype(np.int32),
                        queries=queries, tiny=tiny)
    json.dump({
        "generated_by": "tests/golden/make_milvus_golden.py: /root/reference/services/milvus_service.py over StandInFlatIPClient",
        "dimension": DIM, "collection_name": "icd10",
        "records": records, "inserted_rows": inserted,
        "cases": cases,
        "tiny": {"records": tiny_recs, "query_index": 1, "top_k": 10, "out": tiny_out},
        "missing_collection": missing, "engine_exception": exc_out,
        "insert_length_mismatch": mismatch, "insert_list_embedding": list_embedding,
        "level_weights": weights,
    }, open(os.path.join(HERE, "milvus_search_cases.json"), "w"), ensure_ascii=False, indent=0)
    print("wrote milvus_search_vectors.npz / milvus_search_cases.json:", len(cases), "search cases; milvus_batch_*:", len(batch), "queries x k = 10, 20")


if __name__ == "__main__":
    main()
