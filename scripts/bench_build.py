#!/usr/bin/env python3
"""SURVEY.md row N1, timed end to end: the FULL corpus build on one MI355X.

    CSV (40 474 rows) -> records with the ICD hierarchy -> "query: " + semantic_text, batched BERT-base forward on ROCm
    -> on-disk corpus store (128-row insert batches, the reference's granularity) -> HBM index -> verify (smoke search)

through `DatabaseBuilder.build_full_database(csv, rebuild=True)` - the drop-in for the reference's
tools/build_database.py:194-260,297-337, whose own shape is 40 474 batch-1 forwards + 317 Milvus inserts.

The real CSV cannot travel to the GPU box. The CSV built here has ITS SHAPE (tests/golden/csv_shape.json, statistics of
/root/reference/data/ICD_10v601.csv made by tests/golden/make_csv_shape.py): the same number of codes per hierarchy
level, children per parent drawn from the real histograms, disease names of the real per-level length distribution
(random CJK characters), rows in code order so that `semantic_text` repeats the ancestors' names like the real one. The
length distribution of the resulting semantic_text is reported next to the real one. No model weights are available
offline: the encoder is the seeded random-init BERT-base of text2vec-base-chinese's shape ("synthetic encoder": the
FLOPs and memory traffic of the real model, not its vectors).

Prints one JSON object: `python scripts/bench_build.py > profiles/rNN_build_full.json`.
"""
import csv
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def draw(rng, hist, size):
    keys = np.asarray([int(k) for k in hist], dtype=np.int64)
    p = np.asarray([hist[str(k)] for k in keys], dtype=np.float64)
    return rng.choice(keys, size=size, p=p / p.sum())


def fit_total(rng, counts, total, lo, hi):
    """adjust integer counts (each kept inside [lo, hi]) so that they sum to `total`: bounded passes, no open-ended loop"""
    counts = np.clip(counts.astype(np.int64), lo, hi)
    for _ in range(64):
        diff = int(total - counts.sum())
        if diff == 0:
            break
        room = (counts < hi) if diff > 0 else (counts > lo)
        idx = np.flatnonzero(room)
        if len(idx) == 0:
            break
        pick = rng.choice(idx, size=min(abs(diff), len(idx)), replace=False)
        counts[pick] += 1 if diff > 0 else -1
    return counts


def synth_csv(path, shape, seed=2025):
    """rows in code order: a head, its level-2 children each followed by their level-3 children; the childless
    (morphology-style) heads at the end. Level counts equal the real CSV's; children per parent follow the real
    histograms, adjusted to the totals; name lengths follow the real per-level histograms."""
    rng = np.random.default_rng(seed)
    n1, n2, n3 = (int(shape["level_counts"][l]) for l in ("1", "2", "3"))
    pool = np.asarray([chr(c) for c in range(0x4E00, 0x4E00 + 3000)])
    lens = {l: draw(rng, shape["name_len_hist"][str(l)], n) for l, n in ((1, n1), (2, n2), (3, n3))}
    used = {1: 0, 2: 0, 3: 0}

    def name(level):
        ln = int(lens[level][used[level] % len(lens[level])])
        used[level] += 1
        return "".join(rng.choice(pool, ln))

    heads = n1 - int(shape["childless"]["1"])
    k12 = fit_total(rng, draw(rng, shape["children_hist"]["1->2"], heads), n2, 1, 10)
    parents2 = n2 - int(shape["childless"]["2"])
    k23 = fit_total(rng, draw(rng, shape["children_hist"]["2->3"], parents2), n3, 1, 99)
    has_kids = np.zeros(n2, dtype=bool)
    has_kids[rng.choice(n2, size=parents2, replace=False)] = True
    rows, i2, ip = [], 0, 0
    for h in range(heads):
        head = f"{chr(65 + (h // 100) % 26)}{h % 100:02d}" + ("" if h < 2600 else "X")
        rows.append((head, name(1)))
        for d in range(int(k12[h])):
            c2 = f"{head}.{d}"
            rows.append((c2, name(2)))
            if has_kids[i2]:
                for j in range(int(k23[ip])):
                    rows.append((f"{c2}{j + 1:02d}", name(3)))
                ip += 1
            i2 += 1
    for i in range(n1 - heads):
        rows.append((f"M{800000 + i}/{i % 10}", name(1)))
    with open(path, "w", encoding="utf-8", newline="") as f:
        w = csv.writer(f)
        w.writerow(["code", "disease"])
        w.writerows(rows)
    return len(rows)


class Timer:
    """wraps a bound method: accumulates wall time (the device is synchronised around every call, so that a stage owns
    the GPU work it enqueued) and call count"""

    def __init__(self, obj, name, sync):
        self.fn, self.t, self.n, self.sync = getattr(obj, name), 0.0, 0, sync
        setattr(obj, name, self)

    def __call__(self, *a, **kw):
        self.sync()
        t0 = time.perf_counter()
        out = self.fn(*a, **kw)
        self.sync()
        self.t += time.perf_counter() - t0
        self.n += 1
        return out


def main():
    import torch
    os.environ.setdefault("EMBEDDING_MODEL_NAME", "shibing624/text2vec-base-chinese")
    os.environ.setdefault("ICD_EMBEDDING_ALLOW_SYNTHETIC", "1")
    tmp = tempfile.mkdtemp(prefix="icd_build_")
    os.environ["MILVUS_DB_PATH"] = os.path.join(tmp, "db")
    os.environ["MILVUS_COLLECTION_NAME"] = "icd10_build"
    shape = json.load(open(os.path.join(ROOT, "tests", "golden", "csv_shape.json"), encoding="utf-8"))
    csv_path = os.path.join(tmp, "icd_synthetic_shape.csv")
    nrows = synth_csv(csv_path, shape)

    from rag_project_icd10_amd.tools.build_database import DatabaseBuilder
    sync = (lambda: torch.cuda.synchronize()) if torch.cuda.is_available() else (lambda: None)
    b = DatabaseBuilder()
    t_all0 = time.perf_counter()
    t0 = time.perf_counter()
    b.initialize_services()
    sync()
    t_init = time.perf_counter() - t0
    # (build_full_database calls initialize_services itself: keep the instance we time)
    b.initialize_services = lambda: None
    sync()
    t0 = time.perf_counter()
    ok = b.build_full_database(csv_path, rebuild=True)      # THE measurement: nothing instrumented, encode and append overlap
    sync()
    t_build = time.perf_counter() - t0
    assert ok, "build_full_database failed"
    # the stage split comes from a SECOND build in which every stage is bracketed by device synchronisation (so a stage owns
    # the GPU work it enqueued): the stages then run one after the other and their sum is larger than the build above
    timers = {"csv_to_records": Timer(b, "load_csv_data", sync),
              "tokenise_and_encode": Timer(b.embedding_service, "encode_query_batch", sync),
              "store_append": Timer(b.milvus_service, "insert_records", sync),
              "index_create_and_load": Timer(b.milvus_service, "load_collection", sync),
              "verify": Timer(b, "verify_database", sync)}
    t0 = time.perf_counter()
    ok = b.build_full_database(csv_path, rebuild=True)
    sync()
    t_build_sync = time.perf_counter() - t0
    assert ok, "build_full_database (synchronised run) failed"
    stats = b.milvus_service.get_collection_stats()
    recs = b.milvus_service.client.records
    st = sorted(len(r["semantic_text"]) for r in recs)
    levels = {str(l): sum(1 for r in recs if r["level"] == l) for l in (1, 2, 3)}
    stages = {k: {"s": round(v.t, 4), "calls": v.n} for k, v in timers.items()}
    other = t_build_sync - sum(v.t for v in timers.values())
    out = {
        "what": "DatabaseBuilder.build_full_database(csv, rebuild=True) on one MI355X: CSV -> records -> batched encode on ROCm -> "
                "store -> HBM index -> verify; a synthetic CSV of the real one's shape (tests/golden/csv_shape.json)",
        "rows": int(stats["num_entities"]), "csv_rows": nrows, "levels": levels, "levels_real": shape["level_counts"],
        "semantic_text_len": {"mean": sum(st) / len(st), "p50": st[len(st) // 2], "p90": st[int(len(st) * 0.9)], "p99": st[int(len(st) * 0.99)], "max": st[-1]},
        "semantic_text_len_real": {k: shape["semantic_text_len"][k] for k in ("mean", "p50", "p90", "p99", "max")},
        "build_s": round(t_build, 3), "rows_per_s": round(stats["num_entities"] / t_build, 1),
        "initialize_services_s": round(t_init, 3),
        "synchronised_run": {"what": "a second build with every stage bracketed by device synchronisation (no overlap of encode and append): the stage split",
                             "build_s": round(t_build_sync, 3), "stages": stages, "unattributed_s": round(other, 4)},
        "encoder": b.embedding_service.get_model_info(), "store_bytes": sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(os.environ["MILVUS_DB_PATH"]) for f in fs),
    }
    # the reference's shape for the same work: one batch-1 forward per record, on the host CPU and on the GPU (a sample)
    texts = [r["semantic_text"] for r in recs[:: max(1, len(recs) // 64)]][:64]
    es = b.embedding_service
    es.encode_query_batch = timers["tokenise_and_encode"].fn
    sync()
    t0 = time.perf_counter()
    for t in texts:
        es.encode_query(t)
    sync()
    per_gpu1 = (time.perf_counter() - t0) / len(texts)
    out["reference_shape_on_this_gpu"] = {"s_per_record": round(per_gpu1, 5), "extrapolated_build_s": round(per_gpu1 * len(recs), 1),
                                          "sample": f"{len(texts)} records, encode_query one per call (tools/build_database.py:220-221)"}
    try:
        from rag_project_icd10_amd.services.embedding_service import EmbeddingService
        cpu = EmbeddingService(allow_synthetic=True, device="cpu")
        cpu.encode_query(texts[0])
        t0 = time.perf_counter()
        for t in texts[:24]:
            cpu.encode_query(t)
        per_cpu = (time.perf_counter() - t0) / 24
        out["reference_shape_on_host_cpu"] = {"s_per_record": round(per_cpu, 4), "extrapolated_build_s": round(per_cpu * len(recs), 1),
                                              "cores": os.cpu_count(), "torch_threads": torch.get_num_threads(),
                                              "sample": "24 records, encode_query one per call on the CPU"}
    except Exception as exc:   # a baseline, never fatal
        out["reference_shape_on_host_cpu"] = {"error": str(exc)}
    out["total_wall_s"] = round(time.perf_counter() - t_all0, 2)
    print(json.dumps(out, ensure_ascii=False, indent=1))


if __name__ == "__main__":
    main()
