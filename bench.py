#!/usr/bin/env python3
"""Headline benchmark: batch exact top-k search over the ICD corpus on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload replicated|rowshard]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no torchrun environment starts the N ranks ITSELF: the parent (which never
touches the GPU) runs the second command line above as a child process, relays rank 0's JSON line and exits with the
child's status (spawn_ranks). Under torchrun (WORLD_SIZE set) the process is a rank and runs the workload.

Workload `replicated` (default; BASELINE.json configs[1]), per GPU:
    corpus  37 000 x 768 fp32, iid N(0,1) rows L2-normalised, default_rng(1234); levels default_rng(1235)
            from the real CSV histogram;
    queries 10 000 x 768, default_rng(4321 + rank), resident in HBM before the timed region;
    one STEP = one pass of the hot path over the batch: icd_index_search_reweighted (query fp16 image,
    fp16-MFMA coarse top-k', certification + exact fp32 rescoring, level reweight + stable re-sort,
    exact fallback for uncertified queries), k = 10. Results are bit-identical to the exact kernel.
With N > 1 every rank holds a corpus replica and its own query batch (data-parallel, no collective on
the data path: weak scaling); `value` is the whole-job rate = N * nq * K / max-over-ranks time: the HEADLINE workload
replicated, labelled so. BASELINE configs[3] proper - 1 M queries over the node, 125 000 per GPU - is measured next to it
and reported under "config3" (run_config3: the per-GPU share in slices of the index's batch size, every 100th query
checked against the oracle).

Workload `rowshard` (BASELINE.json configs[4]; SURVEY.md 8d "Config 5"): every rank generates a shard of
1 250 000 x 768 rows ON THE DEVICE (seed 1234 + rank; N = 8 -> the 10 M-row corpus), the query batch (100 000 x 768,
seed 4321, replicated) is searched in slices: local top-k of the shard -> ONE RCCL all_gather of (score f32, id i64,
level i32) per hit -> merge kernel + level reweight on every rank (rag_project_icd10_amd.sharded.ShardedSearch, ROW).
One STEP = one pass of the whole query batch; per-GPU work is fixed as N grows (weak scaling: the corpus grows).
With N > 1 the default run also measures this workload and reports it under "rowshard" in the same JSON line, so the
driver's scaling run (bench.py --gpus 1/2/4/8) yields both curves.

At N = 1 the default line also carries, under "extra" (never the headline): the same step on the real CSV's size
(40 474 rows), on SURVEY 8d's clustered data (512 centroids, x = normalise(c_j + 0.5 eps)), on the headline data at the
serving path's k = 20 (/query searches top_k * 2, services/multi_diagnosis_service.py:153) and on an ICD-SHAPED corpus
("family": 300 families x 124 near-identical rows in code order, mutual cosine 0.99 - the reference's semantic_text repeats
the ancestors' names, tools/build_database.py:156-171 - a fresh index, its first batch and the steady state), each with
ms_per_step, fallback_queries, last_second_pass and ids_exact against the oracle on every query, from the same run; "exact_mode" = the fp32-MFMA kernel alone on the
headline workload against its own 157.3 TFLOP/s peak; "windows" = the K-step window repeated (min / median / max) and
the first window of the process, before the clock has settled.

The JSON line also carries
    roofline      dominant kernel (coarse_flat_kernel): algorithmic FLOPs 2*nq*n*dim per launch / its mean
                  duration over the timed steps (hipEvents recorded by the library on the search
                  stream), against the dense fp16/bf16 MFMA peak (2.5 PFLOP/s) and against the bare MFMA stream
                  of the same loop measured on this chip (clock under load);
    cpu_baseline  the reference's call shape on the host CPU (one query per call: fp32 scan + top-k +
                  level weight + stable sort, oracle/oracle.py reference_shaped_search), rank 0, N = 1
                  only, on a bounded query sample; plus the batched-torch line and the CPU encoder lines of BASELINE.md 3;
    recall_at_10 / ids_exact against the CPU oracle on EVERY query of the batch.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS_F16 = 2500.0  # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_TFLOPS_F32 = 157.3   # fp32-input MFMA = the fp32 vector peak (same guide, "Matrix cores")


def unit_rows(n, dim, seed):
    x = np.random.default_rng(seed).standard_normal((n, dim), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return np.ascontiguousarray(x, dtype=np.float32)


def clustered_rows(n, dim, seed, centroid_seed=77, ncent=512, spread=0.5):
    """SURVEY 8d's clustered variant: 512 unit centroids c_j, x = normalise(c_j + 0.5 eps) with |eps| ~ 1 (eps =
    N(0, I) / sqrt(dim)): rows of one cluster have cosine ~0.8, the shape text embeddings have; corpus and queries share
    the centroids (centroid_seed) and differ in `seed`"""
    cent = unit_rows(ncent, dim, centroid_seed)
    rng = np.random.default_rng(seed)
    x = cent[rng.integers(0, ncent, n)] + (spread / np.sqrt(dim)) * rng.standard_normal((n, dim), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return np.ascontiguousarray(x, dtype=np.float32)


def family_rows(nfam, per, dim, spread, nq, seed):
    """an ICD-shaped corpus: `nfam` families of `per` near-identical rows IN CODE ORDER (a family's rows are neighbours),
    x = normalise(c_f + spread eps); the queries are noisy members of random families. spread 0.10 -> mutual cosine 0.99"""
    rng = np.random.default_rng(seed)
    cent = rng.standard_normal((nfam, dim)).astype(np.float32)
    x = np.repeat(cent, per, axis=0) + spread * rng.standard_normal((nfam * per, dim)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    q = cent[rng.integers(0, nfam, nq)] + spread * rng.standard_normal((nq, dim)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return np.ascontiguousarray(x, dtype=np.float32), np.ascontiguousarray(q, dtype=np.float32)


def icd_levels(n, seed):
    r = np.random.default_rng(seed).random(n)
    return np.where(r < 0.1243, 1, np.where(r < 0.4234, 2, 3)).astype(np.int32)


def cpu_baseline(corpus, levels, queries, k, budget_s=12.0, with_encoder=True):
    """reference-shaped CPU search, one query per call, on a bounded sample of the same workload; the "best CPU"
    batched line and the CPU encoder lines of BASELINE.md section 3 ride along under `extra`"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    import torch
    threads = torch.get_num_threads()
    # `cores` = the threads the timed loop can actually use: numpy's BLAS pool (the matvec of a call), not the host's core count
    np_threads = 0
    try:
        from threadpoolctl import threadpool_info
        np_threads = max([int(p.get("num_threads", 0)) for p in threadpool_info() if p.get("user_api") == "blas"] or [0])
    except Exception:   # pragma: no cover
        pass
    t0 = time.perf_counter()
    done = 0
    for q in queries[:8]:
        orc.reference_shaped_search(corpus, levels, q, k)
        done += 1
    per = (time.perf_counter() - t0) / done
    m = int(max(16, min(len(queries), budget_s / per)))
    t0 = time.perf_counter()
    for q in queries[:m]:
        orc.reference_shaped_search(corpus, levels, q, k)
    dt = time.perf_counter() - t0
    out = {"value": m / dt, "unit": "queries/s", "cores": int(np_threads or os.cpu_count() or threads), "host_cores": int(os.cpu_count() or 0), "torch_threads": int(threads),
           "kind": "port", "sample": f"{m} of the {len(queries)} queries, one query per call (reference call shape), "
                                     f"{corpus.shape[0]}x{corpus.shape[1]} fp32 corpus, numpy/BLAS"}
    extra = {}
    try:   # "CPU-best": batched Q @ C^T + topk in torch fp32
        mb = min(len(queries), 1000)
        tc, tq = torch.from_numpy(corpus), torch.from_numpy(queries[:mb])
        torch.topk(tq[:64] @ tc.T, k, dim=1)
        t0 = time.perf_counter()
        torch.topk(tq @ tc.T, k, dim=1)
        extra["batched_torch_topk"] = {"value": mb / (time.perf_counter() - t0), "unit": "queries/s",
                                       "sample": f"{mb} queries in one torch.topk(Q @ C^T) call, fp32"}
    except Exception as exc:   # a baseline, never fatal
        extra["batched_torch_topk"] = {"error": str(exc)}
    if with_encoder:
        try:   # CPU encoder: the reference's batch-1 encode_query per string, and /embed's batch-32 encode_batch
            os.environ.setdefault("EMBEDDING_MODEL_NAME", "shibing624/text2vec-base-chinese")
            from rag_project_icd10_amd.services.embedding_service import EmbeddingService
            emb = EmbeddingService(allow_synthetic=True, device="cpu")
            texts = [l.strip() for l in open(os.path.join(ROOT, "tests", "golden", "diagnosis_strings.txt"), encoding="utf-8")][:160]
            emb.encode_query(texts[0])
            # BASELINE configs[0] on the host: per string encode_query (batch 1) -> the reference-shaped search, k = 5, over 40 474
            # rows (Gaussian unit rows: the scan's cost does not depend on the data; the GPU's leg, extra.config0, runs over the
            # database it builds). A bounded sample of the 100 strings.
            c40 = unit_rows(40474, corpus.shape[1], 1234)
            l40 = icd_levels(40474, 1235)
            orc.reference_shaped_search(c40, l40, queries[0], 5)
            # (a BOUNDED sample: at most 32 strings and ~30 s - a loaded or oversubscribed host has run this forward at 0.6 s and, once, at
            #  minutes per string; the line must still come out within minutes)
            vecs, d_enc, d_search = [], 0.0, 0.0
            for t in texts[:32]:
                t0 = time.perf_counter()
                v = emb.encode_query(t)
                t1 = time.perf_counter()
                orc.reference_shaped_search(c40, l40, v, 5)
                t2 = time.perf_counter()
                d_enc += t1 - t0
                d_search += t2 - t1
                vecs.append(v)
                if d_enc + d_search > 30.0 and len(vecs) >= 4:
                    break
            ns = len(vecs)
            _CPU_CONFIG0["vectors"] = np.stack(vecs)
            d1 = d_enc
            extra["config0"] = {"value": ns / (d_enc + d_search), "unit": "strings/s", "kind": "port",
                                "sample": f"{ns} of the 100 golden strings: encode_query one per call (HF BertModel fp32 on the CPU, synthetic weights) -> "
                                          "reference-shaped search (oracle/oracle.py), top_k=5, 40474x768 fp32 rows",
                                "encode_ms_per_string": d_enc / ns * 1e3, "search_ms_per_string": d_search / ns * 1e3,
                                "cores": int(np_threads or os.cpu_count() or threads)}
            extra["encoder_reference_shaped"] = {"value": ns / d1, "unit": "strings/s", "sample": f"{ns} strings, encode_query one per call (batch 1)",
                                                 "synthetic_weights": bool(emb.get_model_info().get("synthetic"))}
            nb = 128 if d1 / ns < 0.5 else 32   # (the batch-32 leg too: a quarter of it on a host that slow)
            t0 = time.perf_counter()
            emb.encode_batch(texts[32:32 + nb], show_progress=False)
            d2 = time.perf_counter() - t0
            extra["encoder_embed_shaped"] = {"value": nb / d2, "unit": "strings/s", "sample": f"{nb} strings, encode_batch (batch 32)"}
        except Exception as exc:
            extra["encoder"] = {"error": str(exc)}
    out["extra"] = extra
    return out


def pmc_traffic(kernel, nq, n, suffix=""):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/rNN_pmc_traffic<suffix>.json, made by scripts/gpu_pmc.sh: FETCH_SIZE and WRITE_SIZE collected in their own
    passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950). PMC counters cannot be read from inside
    this process: this is a CITATION of a profile of the same workload on another run (traffic_source names the file),
    reported only for the workload it was collected on."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic{suffix}.json")))
    want = (16384, 1250000) if suffix else (10000, 37000)
    if not files or (nq, n) != want:
        return None, None
    try:
        import hashlib
        raw = open(files[-1], "rb").read()
        d = json.loads(raw)
        for name, v in d.items():
            if kernel in name:
                # (the file's digest rides along: a line names exactly the bytes it cites - scripts/gpu_final.sh makes the PMC
                #  passes BEFORE the bench lines and installs them in profiles/, so a round's lines cite that round's passes)
                return float(v["traffic_bytes"]), f"{os.path.relpath(files[-1], ROOT)} sha256:{hashlib.sha256(raw).hexdigest()[:16]}"
    except Exception:
        return None, None
    return None, None


def reference_gemm():
    """The vendor fp16 GEMM (hipBLASLt through torch.matmul) measured on ONE box in ONE call next to the coarse kernel
    (scripts/probe/gemm_reference.py -> profiles/rNN_gemm_reference.log): TFLOP/s at the coarse pass's own shape
    (10 240 x 37 120 x 768) - the known-good GEMM `roofline.frac_of_reference_gemm` is quoted against (cdna_hip_programming.md
    section 5.4 rule 10). A CITATION of the committed log of the newest round, like `traffic`; None when there is none."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gemm_reference.log")))
    if not files:
        return None, None
    try:
        for ln in open(files[-1]):
            m = re.match(r"torch\.matmul fp16 coarse_shape_10240x37120x768: [0-9.]+ ms, ([0-9.]+) TFLOP/s", ln)
            if m:
                return float(m.group(1)), os.path.relpath(files[-1], ROOT)
    except Exception:
        pass
    return None, None


class Ctx:
    """process-group context of one rank"""

    def __init__(self, backend_env="ICD_BENCH_BACKEND"):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        # test hook (single-GPU boxes / CPU tests): ICD_BENCH_BACKEND=gloo ICD_BENCH_ONE_DEVICE=1 runs the N-rank control
        # flow with every rank on one device and the reductions on the CPU; the driver's runs use nccl (= RCCL)
        self.backend = os.environ.get(backend_env, "nccl")
        if os.environ.get("ICD_BENCH_ONE_DEVICE") == "1":
            self.local_rank = 0
        self.cpu_only = os.environ.get("ICD_BENCH_DEVICE") == "cpu"   # tests inject a CPU search (no HIP index)
        if self.world > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world,
                                        device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)
        if self.cpu_only:
            self.dev = torch.device("cpu")
        else:
            torch.cuda.set_device(self.local_rank)
            self.dev = torch.device("cuda", self.local_rank)
        self.red_dev = self.dev if self.backend == "nccl" else torch.device("cpu")
        # a CPU side channel for decisions that must not depend on the GPU queue (main(): did any rank's C-ABI trial hang?)
        self.side = None
        if self.world > 1 and self.backend == "nccl":
            try:
                self.side = dist.new_group(backend="gloo")
            except Exception as exc:   # (never fatal: the decision then travels over the main group, on the device)
                print(f"bench.py: rank {self.rank}: no gloo side group ({type(exc).__name__}: {exc}); the trial outcome is agreed over the main group", file=sys.stderr)

    def any_rank(self, flag):
        """True on every rank iff `flag` on any rank (all_reduce(MAX) of a CPU tensor over the gloo side group)"""
        if self.world == 1:
            return bool(flag)
        on_cpu = self.side is not None or self.backend != "nccl"
        t = self.torch.tensor([1 if flag else 0], dtype=self.torch.int32, device="cpu" if on_cpu else self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.side)
        return bool(int(t.item()))

    def sync(self):
        if not self.cpu_only:
            self.torch.cuda.synchronize(self.dev)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max_over_ranks(self, x):
        if self.world == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.red_dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def window(self, fn, steps):
        """exactly `steps` steps between barrier + synchronize on both sides; max over ranks"""
        out = None
        self.barrier()
        self.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = fn()
        self.sync()
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0), out

    def timed(self, fn, steps, warmup, before_timing=None, after_timing=None, settle_s=0.0, repeats=0):
        """W untimed steps, then exactly `steps` steps between barrier + synchronize on both sides; max over ranks:
        THE measurement (first return value). Around it, reported next to it and never instead of it:
        settle_s > 0: a fresh process needs ~20 steps until the chip holds its steady clock (scripts/probe/clock_settle.py:
          0.90, 0.77, then 0.745 ms per step in chunks of ten), so before the W warm-up steps of the measurement the device
          is kept busy with the same step for that long (untimed setup, like building the index) - and the window a plain
          "W warm-up steps, K timed steps" gives in a fresh process is measured FIRST and returned as info["first_window_s"];
        repeats: the K-step window is repeated that many times after the measurement (info["repeat_s"])."""
        out = None
        info = {"first_window_s": None, "repeat_s": []}
        if settle_s > 0:
            for _ in range(warmup):
                fn()
            self.sync()
            info["first_window_s"], _ = self.window(fn, steps)
            t_end = time.perf_counter() + settle_s
            while time.perf_counter() < t_end:
                for _ in range(4):
                    fn()
                self.sync()
        for _ in range(warmup):
            out = fn()
        self.sync()
        if before_timing:
            before_timing()
        elapsed, out = self.window(fn, steps)
        if after_timing:
            after_timing()
        for _ in range(repeats):
            info["repeat_s"].append(self.window(fn, steps)[0])
        return elapsed, out, info


def hip_index_factory(corpus, levels, device, max_nq, max_k, id_base=0):
    from rag_project_icd10_amd._native import IcdIndex
    return IcdIndex(corpus, levels, device=device, max_nq=max_nq, max_k=max_k, id_base=id_base)


def oracle_check(corpus, levels, queries, k, out, world=1):
    """parity of a batch against the CPU oracle (the checker, outside any timed region): EVERY query at N = 1; at N > 1 rank 0
    has 1 / N of the host's cores, so it checks every N-th query of its batch (the same ~10 s of oracle time)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    adj, raw, ids, lv = (x.cpu().numpy() for x in out)
    if world > 1:
        sel = np.arange(0, len(queries), world)
        queries, adj, raw, ids, lv = queries[sel], adj[sel], raw[sel], ids[sel], lv[sel]
    t0 = time.perf_counter()
    # (torchrun exports OMP_NUM_THREADS=1 to its ranks: give the checker its share of the host's cores explicitly)
    os_, oi = orc.flat_ip_topk(corpus, queries, k, nthreads=max(1, (os.cpu_count() or 1) // max(1, world)))
    want = orc.reweight(os_, oi, levels)
    return {"recall_at_10": float(np.mean([len(set(a) & set(b)) / k for a, b in zip(ids, oi)])),
            "ids_exact": bool(np.array_equal(ids, want[2])),
            "adjusted_scores_exact": bool(adj.tobytes() == want[0].tobytes()),
            "max_abs_dscore": float(np.max(np.abs(raw.astype(np.float64) - want[1].astype(np.float64)))),
            "parity_checked_queries": int(len(queries)), "parity_check_s": round(time.perf_counter() - t0, 2)}


def side_workload(*a, **kw):
    """an `extra` object never costs the line its headline: a failure is reported in its place"""
    try:
        return _side_workload(*a, **kw)
    except Exception as exc:   # pragma: no cover - reported, the headline above is already measured
        import traceback
        return {"workload": a[3] if len(a) > 3 else "?", "error": f"{type(exc).__name__}: {exc}", "traceback": traceback.format_exc()[-1500:]}


def embed_one_string_extra(ctx, es=None):
    """The embed step at the reference's own call shape: EmbeddingService.encode_query(ONE string) -> SentenceTransformer.encode
    (services/embedding_service.py:97-102,117-120), 100 golden diagnosis strings one call at a time, through the hand-written
    small-input forward (csrc/encoder_small.hpp, icd_encoder_encode: one graph launch) and through the framework's forward
    replayed from a HIP graph (rounds 1-4). Synthetic BERT-base weights (the real model's shapes; no checkpoint offline).
    Checked in the same run: the two paths' embeddings against each other (tolerance 1e-5; both against the CPU fp32 forward of
    the same weights: tests/test_encoder_gpu.py - no CPU forward here, the cpu_baseline leg times one right after)."""
    torch = ctx.torch
    saved = {v: os.environ.get(v) for v in ("EMBEDDING_MODEL_NAME", "ICD_EMBEDDING_ALLOW_SYNTHETIC")}
    os.environ.update({"EMBEDDING_MODEL_NAME": "shibing624/text2vec-base-chinese", "ICD_EMBEDDING_ALLOW_SYNTHETIC": "1"})
    try:
        from rag_project_icd10_amd.services.embedding_service import EmbeddingService
        if es is None:
            es = EmbeddingService(allow_synthetic=True, device=f"cuda:{ctx.local_rank}" if ctx.local_rank else "cuda")
        strings = [l.strip() for l in open(os.path.join(ROOT, "tests", "golden", "diagnosis_strings.txt"), encoding="utf-8") if l.strip()][:100]
        small = getattr(es, "_small", None)
        out = {"workload": "EmbeddingService.encode_query, ONE string per call, 100 golden diagnosis strings, synthetic BERT-base weights",
               "small_input_encoder": small is not None, "synthetic_weights": bool(es.synthetic)}
        vecs = {}
        for mode in (("small", "graph") if small is not None else ("graph",)):
            es._small = small if mode == "small" else None
            for t in strings[:20]:
                es.encode_query(t)
            lat = []
            for i in range(200):
                t0 = time.perf_counter()
                v = es.encode_query(strings[i % len(strings)])
                lat.append((time.perf_counter() - t0) * 1e6)
                if i < len(strings):
                    vecs.setdefault(mode, []).append(v)
            lat.sort()
            out[f"{mode}_us_per_call"] = {"median": lat[100], "p10": lat[20], "p90": lat[180], "calls": len(lat)}
        es._small = small
        if small is not None:
            out["max_abs_d_small_vs_graph"] = float(np.max(np.abs(np.stack(vecs["small"]) - np.stack(vecs["graph"]))))
            out["speedup"] = out["graph_us_per_call"]["median"] / out["small_us_per_call"]["median"]
        if small is not None:
            out["within_1e-5"] = bool(out["max_abs_d_small_vs_graph"] <= 1e-5)
        out["tokens_per_string"] = {"median": int(np.median([len(x) for x in es._tokenize([f"query: {t}" for t in strings])]))}
        return out
    finally:
        for v, val in saved.items():
            if val is None:
                os.environ.pop(v, None)
            else:
                os.environ[v] = val


_CPU_CONFIG0 = {}   # cpu_baseline's leg leaves its vectors of the first golden strings here: config0_extra compares the GPU's with them


def config0_extra(ctx):
    """BASELINE configs[0] as ONE composed workload, the reference's own call shape (services/multi_diagnosis_service.py:152-153,
    services/milvus_service.py:280-285): 100 golden diagnosis strings, per string EmbeddingService.encode_query (one string per
    call) -> MilvusService.search(vector, top_k=5) (one query per call, hit dicts out), over a full-size database - the 40 474
    rows of a CSV of the real one's shape (scripts/bench_build.py synth_csv over tests/golden/csv_shape.json), built in this run
    by DatabaseBuilder.build_full_database (tools/build_database.py:297-337; its seconds are reported, not timed into the loop).
    Synthetic BERT-base weights (no checkpoint offline). Checked in the same run, all 100 strings: the hits' codes and scores
    against the CPU oracle over the stored corpus (bit for bit), a stored row against encode_query of its own text (bit for
    bit), and the GPU's vectors of the first strings against the CPU fp32 forward of the same weights (1e-5; cpu_baseline's leg)."""
    import shutil
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import oracle as orc
    from bench_build import synth_csv
    tmp = tempfile.mkdtemp(prefix="icd_bench_config0_")
    env = {"MILVUS_MODE": "local", "MILVUS_DB_PATH": os.path.join(tmp, "db"), "MILVUS_COLLECTION_NAME": "icd10_config0",
           "EMBEDDING_MODEL_NAME": "shibing624/text2vec-base-chinese", "ICD_EMBEDDING_ALLOW_SYNTHETIC": "1",
           "EMBEDDING_DEVICE": f"cuda:{ctx.local_rank}" if ctx.local_rank else "cuda"}
    saved = {v: os.environ.get(v) for v in env}
    os.environ.update(env)
    try:
        from rag_project_icd10_amd.tools.build_database import DatabaseBuilder
        shape = json.load(open(os.path.join(ROOT, "tests", "golden", "csv_shape.json"), encoding="utf-8"))
        csv_path = os.path.join(tmp, "icd_shape.csv")
        nrows = synth_csv(csv_path, shape)
        b = DatabaseBuilder()
        t0 = time.perf_counter()
        ok = b.build_full_database(csv_path, rebuild=True)
        build_s = time.perf_counter() - t0
        if not ok:
            return {"error": "build_full_database failed"}, None
        es, ms = b.embedding_service, b.milvus_service
        strings = [l.strip() for l in open(os.path.join(ROOT, "tests", "golden", "diagnosis_strings.txt"), encoding="utf-8") if l.strip()][:100]
        k = 5
        for t in strings[:10]:
            ms.search(es.encode_query(t), top_k=k)
        ctx.sync()
        lat, enc_us, vecs, hits_all = [], [], [], []
        t_all = time.perf_counter()
        for t in strings:
            t0 = time.perf_counter()
            v = es.encode_query(t)
            t1 = time.perf_counter()
            hits = ms.search(v, top_k=k)
            t2 = time.perf_counter()
            lat.append((t2 - t0) * 1e6)
            enc_us.append((t1 - t0) * 1e6)
            vecs.append(v)
            hits_all.append(hits)
        total_s = time.perf_counter() - t_all
        corpus, levels = ms.client.matrix(), ms.client.levels()
        codes = [r["code"] for r in ms.client.records]
        V = np.stack(vecs)
        os_, oi = orc.flat_ip_topk(corpus, V, k)
        want = orc.reweight(os_, oi, levels)
        hits_ok = all([h["code"] for h in hits_all[i]] == [codes[j] for j in want[2][i]]
                      and [h["score"] for h in hits_all[i]] == list(want[0][i])
                      and [h["original_score"] for h in hits_all[i]] == [float(x) for x in want[1][i]] for i in range(len(strings)))
        rows = [0, 1, nrows // 2, nrows - 1]
        row_ok = all(np.array_equal(es.encode_query(ms.client.records[i]["semantic_text"]), corpus[i]) for i in rows)
        lat.sort()
        enc_sorted = sorted(enc_us)
        out = {"workload": f"BASELINE configs[0] on the GPU: {len(strings)} golden diagnosis strings, per string encode_query (one string per call) -> "
                           f"MilvusService.search(top_k={k}) (one query per call) over the {nrows}-row database built in this run",
               "strings": len(strings), "top_k": k, "corpus_rows": int(nrows), "synthetic_weights": bool(es.synthetic),
               "strings_per_sec": len(strings) / total_s, "ms_for_all": total_s * 1e3,
               "us_per_string": {"median": lat[len(lat) // 2], "p10": lat[len(lat) // 10], "p90": lat[len(lat) * 9 // 10]},
               "encode_query_us": {"median": enc_sorted[len(enc_sorted) // 2], "p90": enc_sorted[len(enc_sorted) * 9 // 10]},
               "search_us_median": lat[len(lat) // 2] - enc_sorted[len(enc_sorted) // 2],
               "build_full_database_s": round(build_s, 2), "batch_arithmetic": es.batch_arithmetic(),
               "hits_exact": bool(hits_ok), "parity_checked_strings": len(strings),
               "parity": "codes, adjusted and raw scores of every hit == oracle/icd_oracle.c over the stored corpus, bit for bit",
               "stored_row_equals_encode_query_of_its_text": bool(row_ok)}
        if _CPU_CONFIG0.get("vectors") is not None:
            m = min(len(_CPU_CONFIG0["vectors"]), len(V))
            out["max_abs_d_vector_vs_cpu_fp32_forward"] = float(np.max(np.abs(V[:m] - _CPU_CONFIG0["vectors"][:m])))
            out["vectors_checked_against_cpu"] = m
        ms.disconnect()
        return out, es
    finally:
        for v, old in saved.items():
            if old is None:
                os.environ.pop(v, None)
            else:
                os.environ[v] = old
        shutil.rmtree(tmp, ignore_errors=True)


def single_query_extra(ctx, args, index_factory):
    """The reference's own call shape (services/milvus_service.py:280-285: ONE query per MilvusService.search call) at the real
    CSV's size, k = 5 (the /query default top_k) and 10 (search's default): what the GPU needs per call (hipEvents of the
    library around its kernels; bytes of the fp32 corpus / that time against 8 TB/s), what a host caller of icd_index_search
    pays per call (numpy in, numpy out, one call at a time), and the same through MilvusService.search (vector in, hit dicts
    out). Every timed call's result is checked against the oracle afterwards (64 distinct queries)."""
    import tempfile
    torch = ctx.torch
    n, dim = 40474, 768
    corpus, levels = unit_rows(n, dim, 1234), icd_levels(n, 1235)
    queries = unit_rows(64, dim, 777)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    index = index_factory(corpus, levels, ctx.local_rank, 64, 10)
    dq = torch.from_numpy(queries).to(ctx.dev)
    out = {"workload": f"ONE query per call (the reference's call shape), {n}x{dim} fp32 rows", "corpus_rows": n, "by_k": {}}
    bytes_per_call = float(n) * dim * 4
    for k in (5, 10):
        os_, oi = orc.flat_ip_topk(corpus, queries, k)
        want = orc.reweight(os_, oi, levels)
        for _ in range(30):
            index.search_reweighted(dq[:1], k)
        ctx.sync()
        index.set_profiling(True)
        index.profile_summary()
        got = [index.search_reweighted(dq[i:i + 1], k) for i in range(64)]
        ctx.sync()
        prof = index.profile_summary()
        index.set_profiling(False)
        dev_ok = all(np.array_equal(got[i][2].cpu().numpy()[0], want[2][i]) and got[i][0].cpu().numpy().tobytes() == want[0][i].tobytes() for i in range(64))
        # The GPU's own time per call: 64 device-resident calls captured in ONE HIP graph (the call allocates nothing and never
        # synchronises: include/icd_search.h) and replayed - no Python between the launches (33 us per call here), no event pair
        # around every kernel. Never fatal: without it the event figure below is the one priced.
        graph_us = None
        try:
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                index.search_reweighted(dq[:1], k)
            torch.cuda.current_stream().wait_stream(side)
            with torch.cuda.graph(g):
                gouts = [index.search_reweighted(dq[i:i + 1], k) for i in range(64)]
            g.replay()
            ctx.sync()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                g.replay()
            e1.record()
            ctx.sync()
            graph_us = e0.elapsed_time(e1) * 1e3 / 640
            dev_ok = dev_ok and all(np.array_equal(gouts[i][2].cpu().numpy()[0], want[2][i]) and gouts[i][0].cpu().numpy().tobytes() == want[0][i].tobytes() for i in range(64))
            del g, gouts
        except Exception as exc:   # pragma: no cover
            out.setdefault("graph_replay_error", f"{type(exc).__name__}: {exc}"[:300])
        lat, host_ok = [], True
        for i in range(20):
            index.search_reweighted(queries[i:i + 1], k)
        for i in range(192):
            t0 = time.perf_counter()
            r = index.search_reweighted(queries[i & 63:(i & 63) + 1], k)
            lat.append((time.perf_counter() - t0) * 1e6)
            host_ok = host_ok and np.array_equal(r[2][0], want[2][i & 63]) and r[0].tobytes() == want[0][i & 63].tobytes()
        lat.sort()
        # (hipEvents of the library around the call's ONE kernel; an event pair around a 29-us kernel reads ~8 us more than rocprofv3's
        #  kernel duration, profiles/rNN_single_query_kernel_stats.csv - the conservative figure is the one priced here)
        event_us = (float(prof.get("ms_exact", 0.0)) + float(prof.get("ms_exact_finalize", 0.0))) * 1e3
        kernel_us = graph_us if graph_us else event_us
        out["by_k"][str(k)] = {"top_k": k, "gpu_us_per_call_with_event_gaps": float(prof.get("ms_total", 0.0)) * 1e3, "kernel_us_per_call_by_events": event_us,
                               "gpu_us_per_call_graph_replay": graph_us, "priced": "graph replay of 64 calls x 10" if graph_us else "library events",
                               "bytes_per_call": bytes_per_call,
                               "roofline": {"bound": "hbm", "kernel": "stream_topk_kernel", "achieved": bytes_per_call / (kernel_us * 1e-6) / 1e9 if kernel_us > 0 else 0.0,
                                            "peak": 8000.0, "unit": "GB/s", "frac": bytes_per_call / (kernel_us * 1e-6) / 8e12 if kernel_us > 0 else 0.0,
                                            "calls_averaged": int(prof.get("count", 0))},
                               "icd_index_search_host_call_us": {"median": lat[len(lat) // 2], "p10": lat[len(lat) // 10], "p90": lat[len(lat) * 9 // 10], "calls": len(lat)},
                               "ids_exact": bool(dev_ok and host_ok), "parity_checked_calls": 64 + len(lat)}
    index.close()
    try:   # the service on top: MilvusService.search(vector, top_k) -> list of hit dicts
        tmp = tempfile.mkdtemp(prefix="icd_bench_store_")
        saved = {v: os.environ.get(v) for v in ("MILVUS_MODE", "MILVUS_DB_PATH", "MILVUS_COLLECTION_NAME")}
        os.environ.update({"MILVUS_MODE": "local", "MILVUS_DB_PATH": tmp, "MILVUS_COLLECTION_NAME": "icd10_bench"})
        from rag_project_icd10_amd.services.milvus_service import MilvusService

        class _Dim:
            def encode_query(self, text):
                return np.zeros(dim, np.float32)
        svc = MilvusService(_Dim())
        recs = [{"code": f"X{i:05d}", "preferred_zh": "", "level": int(levels[i])} for i in range(n)]
        for b in range(0, n, 8192):
            svc.insert_records(recs[b:b + 8192], [corpus[i] for i in range(b, min(n, b + 8192))])
        for k in (5, 10):
            os_, oi = orc.flat_ip_topk(corpus, queries, k)
            want = orc.reweight(os_, oi, levels)
            for i in range(20):
                svc.search(queries[i], k)
            lat, ok = [], True
            for i in range(192):
                t0 = time.perf_counter()
                hits = svc.search(queries[i & 63], k)
                lat.append((time.perf_counter() - t0) * 1e6)
                ok = ok and [h["code"] for h in hits] == [f"X{j:05d}" for j in want[2][i & 63]] and [h["score"] for h in hits] == list(want[0][i & 63])
            lat.sort()
            out["by_k"][str(k)]["milvus_service_search_us"] = {"median": lat[len(lat) // 2], "p10": lat[len(lat) // 10], "p90": lat[len(lat) * 9 // 10], "calls": len(lat)}
            out["by_k"][str(k)]["service_hits_exact"] = bool(ok)
        svc.disconnect()
        for v, old in saved.items():
            if old is None:
                os.environ.pop(v, None)
            else:
                os.environ[v] = old
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    except Exception as exc:   # pragma: no cover - reported in place
        out["milvus_service_error"] = f"{type(exc).__name__}: {exc}"
    return out


def two_streams_extra(ctx, args, index_factory, corpus, levels, queries, k, steps):
    """The headline step PIPELINED: two index handles over the same corpus on two streams, consecutive steps alternating between
    them, so that one step's finalize (gather- and latency-bound, no MFMA) and its gated launches run under the next step's coarse
    sweep. What a server with back-to-back batches can do with the C ABI as it is (one stream per handle); never the headline:
    `value` times the steps one after the other on one stream. VERDICT r5 asked for this overlap INSIDE one step (two half
    batches): measured slower (profiles/r06_two_streams_overlap.log: +8 % on one stream and on two - a 5 000-query sweep
    reloads its queries twice as often, and finalize cannot share a CU with the coarse kernel's 452 registers per lane)."""
    from rag_project_icd10_amd._native import MODE_AUTO
    torch = ctx.torch
    ia = index_factory(corpus, levels, ctx.local_rank, len(queries), max(k, 10))
    ib = index_factory(corpus, levels, ctx.local_rank, len(queries), max(k, 10))
    dq = torch.from_numpy(queries).to(ctx.dev)
    sa, sb = torch.cuda.Stream(device=ctx.dev), torch.cuda.Stream(device=ctx.dev)
    outs = [None, None]

    def pair():
        with torch.cuda.stream(sa):
            outs[0] = ia.search_reweighted(dq, k, MODE_AUTO)
        with torch.cuda.stream(sb):
            outs[1] = ib.search_reweighted(dq, k, MODE_AUTO)

    ctx.sync()
    t_end = time.perf_counter() + args.settle_ms / 1e3
    while time.perf_counter() < t_end:
        pair()
        ctx.sync()
    for _ in range(max(3, args.warmup)):
        pair()
    ctx.sync()
    pairs = max(1, steps // 2)
    t0 = time.perf_counter()
    for _ in range(pairs):
        pair()
    ctx.sync()
    elapsed = time.perf_counter() - t0
    obj = {"workload": f"the headline step pipelined: two index handles on two streams, {2 * pairs} steps alternating between them",
           "steps": 2 * pairs, "ms_per_step": elapsed / (2 * pairs) * 1e3, "queries_per_sec": len(queries) * 2 * pairs / elapsed,
           "what": "throughput of back-to-back batches when a step's finalize runs under the next step's coarse sweep; the headline `value` is NOT this"}
    pa, pb = oracle_check(corpus, levels, queries, k, outs[0], ctx.world), oracle_check(corpus, levels, queries, k, outs[1], ctx.world)
    obj.update({"ids_exact": bool(pa["ids_exact"] and pb["ids_exact"]), "adjusted_scores_exact": bool(pa["adjusted_scores_exact"] and pb["adjusted_scores_exact"]),
                "parity_checked_queries": pa["parity_checked_queries"] + pb["parity_checked_queries"]})
    ia.close()
    ib.close()
    return obj


def host_buffers_extra(ctx, args, index_factory, corpus, levels, queries, k, steps):
    """The PCIe-inclusive rates (never `value`, which starts with its inputs resident in HBM): the headline batch handed over in HOST
    memory through the C ABI's host pointers - pageable numpy arrays, then the same arrays in pinned memory - and what a host
    application reaches with the DEVICE-pointer ABI when it pipelines for itself: batch i + 1's queries on their way over PCIe (a
    copy stream, pinned staging) and batch i - 1's results on their way back while batch i searches. Results of every form against
    the device-resident call's (which the line checks against the oracle), bit for bit."""
    from rag_project_icd10_amd._native import MODE_AUTO
    torch = ctx.torch
    index = index_factory(corpus, levels, ctx.local_rank, len(queries), max(k, 10))
    dq = torch.from_numpy(queries).to(ctx.dev)
    ref = [t.cpu().numpy() for t in index.search_reweighted(dq, k, MODE_AUTO)]
    obj = {"workload": f"the headline batch ({len(queries)} x {queries.shape[1]} fp32 queries, {queries.nbytes / 1e6:.1f} MB in, "
                       f"{sum(r.nbytes for r in ref) / 1e6:.2f} MB out per step) from and to HOST memory"}

    def timed(fn, n):
        for _ in range(3):
            out = fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        ctx.sync()
        return (time.perf_counter() - t0) / n * 1e3, out

    same = lambda out: bool(all(np.array_equal(np.asarray(a), b) for a, b in zip(out, ref)))
    ms, out = timed(lambda: index.search_reweighted(queries, k, MODE_AUTO), steps)
    obj["pageable"] = {"ms_per_step": ms, "queries_per_sec": len(queries) / ms * 1e3, "equal_to_device_call": same(out),
                       "what": "icd_index_search_reweighted with host pointers (numpy arrays): H2D, search, D2H, one after the other"}
    pinned_q = torch.from_numpy(queries).pin_memory()
    ms, out = timed(lambda: index.search_reweighted(pinned_q.numpy(), k, MODE_AUTO), steps)
    obj["pinned"] = {"ms_per_step": ms, "queries_per_sec": len(queries) / ms * 1e3, "equal_to_device_call": same(out),
                     "what": "the same call, the query array in pinned host memory"}
    # pipelined by the caller over the device-pointer ABI: two device query buffers, a copy stream each way, pinned result blocks
    cs_in, cs_out, main = torch.cuda.Stream(device=ctx.dev), torch.cuda.Stream(device=ctx.dev), torch.cuda.current_stream(ctx.dev)
    dbuf = [torch.empty_like(dq), torch.empty_like(dq)]
    hres = [[torch.empty(r.shape, dtype=torch.from_numpy(r).dtype).pin_memory() for r in ref] for _ in range(2)]
    ev_in = [torch.cuda.Event(), torch.cuda.Event()]
    ev_done = [torch.cuda.Event(), torch.cuda.Event()]
    ev_out = [torch.cuda.Event(), torch.cuda.Event()]

    def pipeline(n):
        with torch.cuda.stream(cs_in):
            dbuf[0].copy_(pinned_q, non_blocking=True)
            ev_in[0].record(cs_in)
        for i in range(n):
            b = i & 1
            if i + 1 < n:
                with torch.cuda.stream(cs_in):
                    if i >= 1:
                        cs_in.wait_event(ev_done[1 - b])          # (the search that read this buffer two steps ago has finished)
                    dbuf[1 - b].copy_(pinned_q, non_blocking=True)
                    ev_in[1 - b].record(cs_in)
            main.wait_event(ev_in[b])
            outs = index.search_reweighted(dbuf[b], k, MODE_AUTO)
            ev_done[b].record(main)
            with torch.cuda.stream(cs_out):
                cs_out.wait_event(ev_done[b])
                if i >= 2:
                    ev_out[b].synchronize()                        # (the host has had this pinned block since two steps ago)
                for h, o in zip(hres[b], outs):
                    h.copy_(o, non_blocking=True)
                    o.record_stream(cs_out)
                ev_out[b].record(cs_out)
        ctx.sync()
        return [h.numpy() for h in hres[(n - 1) & 1]]

    pipeline(4)
    t0 = time.perf_counter()
    out = pipeline(steps)
    ms = (time.perf_counter() - t0) / steps * 1e3
    obj["pipelined_by_the_caller"] = {"ms_per_step": ms, "queries_per_sec": len(queries) / ms * 1e3, "equal_to_device_call": same(out), "steps": steps,
                                      "what": "device-pointer calls; batch i + 1's H2D (pinned, a copy stream) and batch i - 1's D2H under batch i's search"}
    index.close()
    return obj


def _side_workload(ctx, args, index_factory, name, corpus, levels, queries, k, mode, steps, first_batch=False):
    """one of the line's `extra` objects: the same step on other data / another size / the exact kernel alone, timed
    over `steps` steps after a short warm-up and checked against the oracle on every query, in this run.
    first_batch: also time the FIRST search of the fresh index on its own (what a new index costs its first user batch)"""
    from rag_project_icd10_amd._native import MODE_AUTO
    torch = ctx.torch
    index = index_factory(corpus, levels, ctx.local_rank, len(queries), max(k, 10))
    dq = torch.from_numpy(queries).to(ctx.dev)
    fn = lambda: index.search_reweighted(dq, k, mode)
    first_ms = first_stats = None
    if first_batch:
        ctx.sync()
        t0 = time.perf_counter()
        fn()
        ctx.sync()
        first_ms = (time.perf_counter() - t0) * 1e3
        first_stats = index.stats()
    # the same untimed settle + warm-up the headline gets (a fresh index and a fresh clock ramp otherwise cost these
    # lines ~15 %: 40 474 rows measured 0.81 ms per step without it, 0.69 as a main workload)
    t_end = time.perf_counter() + args.settle_ms / 1e3
    while time.perf_counter() < t_end:
        for _ in range(4):
            fn()
        ctx.sync()
    for _ in range(max(3, args.warmup)):
        fn()
    ctx.sync()
    index.set_profiling(True, every=args.profile_every)
    index.profile_summary()
    elapsed, out = ctx.window(fn, steps)
    prof = index.profile_summary()
    index.set_profiling(False)
    stats = index.stats()
    fast = mode == MODE_AUTO and stats["last_mode"] == MODE_AUTO
    dom_ms = prof["ms_coarse"] if fast else prof["ms_exact"]
    flops = 2.0 * len(queries) * corpus.shape[0] * corpus.shape[1]
    achieved = flops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    peak = PEAK_TFLOPS_F16 if fast else PEAK_TFLOPS_F32
    obj = {"workload": name, "corpus_rows": int(corpus.shape[0]), "nq": int(len(queries)), "top_k": k, "steps": steps,
           "ms_per_step": elapsed / steps * 1e3, "queries_per_sec": len(queries) * steps / elapsed,
           "fallback_queries": int(stats["last_fallback"]), "coarse_chunks": int(stats["last_chunks"]),
           "last_second_pass": int(stats.get("last_second_pass", 0)), "wide_mode": int(stats.get("wide_mode", 0)),
           "kernel_ms": {kname: round(v, 5) for kname, v in prof.items() if kname != "count"},
           "roofline": {"bound": "mfma", "kernel": "coarse_flat_kernel" if fast else "exact_topk_kernel", "achieved": achieved,
                        "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "launch_ms": dom_ms,
                        "launches_averaged": prof["count"]}}
    if first_batch:
        obj["first_batch_ms"] = first_ms
        obj["first_batch"] = {"ms": first_ms, "fallback_queries": int(first_stats["last_fallback"]),
                              "last_second_pass": int(first_stats.get("last_second_pass", 0)), "wide_mode": int(first_stats.get("wide_mode", 0))}
    obj.update(oracle_check(corpus, levels, queries, k, out, ctx.world))
    index.close()
    return obj


def run_replicated(ctx, args, index_factory=hip_index_factory):
    """BASELINE configs[1] (N = 1) / configs[3] (N > 1): corpus replicated, every rank its own query batch"""
    from rag_project_icd10_amd._native import MODE_AUTO, MODE_EXACT
    torch = ctx.torch
    dim, nq, n, k = 768, args.nq, args.n, args.k
    mode = MODE_AUTO if args.mode == "auto" else MODE_EXACT
    corpus, levels = unit_rows(n, dim, 1234), icd_levels(n, 1235)
    queries = unit_rows(nq, dim, 4321 + ctx.rank)
    index = index_factory(corpus, levels, ctx.local_rank, nq, max(k, 10))
    dq = torch.from_numpy(queries).to(ctx.dev)
    ctx.sync()
    prof = {}

    def before():
        # hipEvents around the kernels of every 5th step of the timed region (an event between two kernels keeps the
        # second from starting under the first one's tail: ~4 % on this step if every step carried them)
        index.set_profiling(True, every=args.profile_every)
        index.profile_summary()  # reset the event window

    def after():   # the kernel times of the measured window only (the repeats that follow are not averaged in)
        prof.update(index.profile_summary())
        index.set_profiling(False)

    elapsed, out, info = ctx.timed(lambda: index.search_reweighted(dq, k, mode), args.steps, args.warmup, before, after,
                                   settle_s=args.settle_ms / 1e3, repeats=args.repeats)
    stats = index.stats()
    line = None
    if ctx.rank == 0:
        parity = oracle_check(corpus, levels, queries, k, out, ctx.world)
        fast = mode == MODE_AUTO and stats["last_mode"] == MODE_AUTO
        dom_ms = prof["ms_coarse"] if fast else prof["ms_exact"]
        flops = 2.0 * nq * n * dim
        achieved = flops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        peak = PEAK_TFLOPS_F16 if fast else PEAK_TFLOPS_F32
        kern = "coarse_flat_kernel" if fast else "exact_topk_kernel"
        traffic, traffic_src = pmc_traffic(kern, nq, n)
        ref_tf, ref_src = reference_gemm() if fast else (None, None)
        per_step = sorted(x / args.steps * 1e3 for x in [elapsed] + info["repeat_s"])
        line = {
            "metric": "queries_per_sec", "value": ctx.world * nq * args.steps / elapsed, "unit": "queries/s",
            "n_gpus": ctx.world, "steps": args.steps, "warmup": args.warmup, "settle_ms_before_warmup": args.settle_ms,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16" if fast else "f32",
            "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[1]: {nq} random fp32 768-d queries x {n}x768 corpus, " if ctx.world == 1 else
                                    f"BASELINE configs[1] replicated on every GPU (the headline workload, {nq} random fp32 768-d queries per GPU x {n}x768 corpus; "
                                    f"configs[3] at its stated size is the `config3` object), ")
                                   + f"top_k={k}, search only (pre-embedded, HBM-resident inputs), level reweight fused",
                       "nq_per_gpu": nq, "corpus_rows": n, "dim": dim, "top_k": k, "mode": args.mode,
                       "parallelism": f"query-sharded x{ctx.world}, corpus replicated" if ctx.world > 1 else "single GPU",
                       "collective_ranks": ctx.dist.get_world_size() if ctx.world > 1 else 1,
                       "result_arithmetic": "fp32 canonical chain (bit-identical to the CPU oracle)"},
            **parity,
            "fallback_queries": int(stats["last_fallback"]), "coarse_chunks": int(stats["last_chunks"]),
            "kernel_ms": {kname: round(v, 5) for kname, v in prof.items() if kname != "count"},
            "roofline": {"bound": "mfma", "kernel": kern,
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "frac_of_reference_gemm": (achieved / ref_tf) if ref_tf else None,
                         "reference_gemm": {"tflops": ref_tf, "what": "vendor fp16 GEMM (hipBLASLt through torch.matmul) at 10240 x 37120 x 768 on random unit rows, "
                                            "the same box and call as the coarse kernel's own line in that log; a plain GEMM: it selects nothing and writes 760 MB",
                                            "source": ref_src} if ref_tf else None,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_unit": "bytes per launch (rocprofv3 PMC passes of this kernel, 2 x FETCH_SIZE + WRITE_SIZE); a citation "
                                         "of the committed profile of this workload, not measured in this run",
                         "flops_per_launch": flops, "launch_ms": dom_ms, "launches_averaged": prof["count"],
                         "launches_sampled": f"every {args.profile_every}th step of the timed region carries hipEvents",
                         #  the WHOLE step priced the same way: the kernel's algorithmic flops over ms_per_step - what the
                         # prep, finalize and gated launches around the dominant kernel cost the step
                         "whole_step": {"achieved": flops / (elapsed / args.steps) / 1e12, "frac": flops / (elapsed / args.steps) / 1e12 / peak,
                                        "what": "flops_per_launch / ms_per_step: the dominant kernel's algorithmic work over the whole step's wall time"}},
            "extra": {"windows": {"what": f"the {args.steps}-step window repeated {1 + len(info['repeat_s'])} times back to back (the first is `value`), ms per step",
                                  "min": per_step[0], "median": per_step[len(per_step) // 2], "max": per_step[-1],
                                  "first_window_of_the_process": (info["first_window_s"] / args.steps * 1e3) if info["first_window_s"] else None}},
        }
    index.close()
    del index, dq
    if ctx.world == 1 and not args.no_extras:
        # (every rank would run these; they are N = 1 lines: other data and sizes next to the Gaussian headline)
        steps_x = max(5, min(args.steps, 20))
        ex = line["extra"]
        n_real = 40474
        ex["real_size"] = side_workload(ctx, args, index_factory, f"the real CSV's size: {nq} Gaussian queries x {n_real}x768 Gaussian rows",
                                        unit_rows(n_real, dim, 1234), icd_levels(n_real, 1235), queries, k, mode, steps_x)
        ex["clustered"] = side_workload(ctx, args, index_factory, f"SURVEY 8d clustered: 512 centroids, x = normalise(c_j + 0.5 eps), {nq} queries x {n}x768",
                                        clustered_rows(n, dim, 2234), levels, clustered_rows(nq, dim, 5321), k, mode, steps_x)
        if mode == MODE_AUTO and not args.no_family:
            ex["k20"] = side_workload(ctx, args, index_factory, f"the headline data at the serving path's k = 20 (/query searches top_k * 2): {nq} x {n}x768",
                                      corpus, levels, queries, 20, mode, steps_x)
            fam_c, fam_q = family_rows(args.family_families, args.family_rows, dim, 0.10, nq, 7)
            ex["family"] = side_workload(ctx, args, index_factory, f"ICD-shaped corpus: {args.family_families} families x {args.family_rows} near-identical rows in code order "
                                         f"(mutual cosine 0.99), {nq} queries, k = {k}; a fresh index: first batch, then the steady state",
                                         fam_c, icd_levels(len(fam_c), 8), fam_q, k, mode, steps_x, first_batch=True)
        if mode == MODE_AUTO and not args.no_family:
            # the rest of the reference's parameter range: /query searches top_k * 2 with top_k <= 50 (models/icd_models.py:138,
            # services/multi_diagnosis_service.py:153) -> k = 100; the code-default encoder is 1024-d (services/embedding_service.py:26)
            ex["k100"] = side_workload(ctx, args, index_factory, f"the headline data at k = 100 (the largest k /query can ask for): {nq} x {n}x768",
                                       corpus, levels, queries, 100, mode, max(3, steps_x // 2))
            ex["dim1024"] = side_workload(ctx, args, index_factory, f"1024-d (the reference's code-default encoder): {nq} x {n}x1024, k = {k}",
                                          unit_rows(n, 1024, 2234), levels, unit_rows(nq, 1024, 6321), k, mode, max(3, steps_x // 2))
            if not ctx.cpu_only:   # (the host-call and service latencies are the library's: nothing to measure on the CPU test engine)
                try:
                    ex["single_query"] = single_query_extra(ctx, args, index_factory)
                except Exception as exc:   # pragma: no cover - reported, never fatal
                    ex["single_query"] = {"error": f"{type(exc).__name__}: {exc}"}
        if mode == MODE_AUTO and not ctx.cpu_only:
            try:
                ex["two_streams"] = two_streams_extra(ctx, args, index_factory, corpus, levels, queries, k, steps_x)
            except Exception as exc:   # pragma: no cover - reported, never fatal
                ex["two_streams"] = {"error": f"{type(exc).__name__}: {exc}"}
        if mode == MODE_AUTO:
            ex["exact_mode"] = side_workload(ctx, args, index_factory, f"--mode exact: the fp32-MFMA kernel alone, {nq} x {n}x768",
                                             corpus, levels, queries, k, MODE_EXACT, max(3, steps_x // 4))
    if line is not None and ctx.world == 1 and not args.no_extras and not args.no_rowshard and not ctx.cpu_only and mode == MODE_AUTO:
        # BASELINE configs[4]'s per-GPU share - one 1.25 M-row shard, the 100 000-query batch in slices - in the N = 1 line too
        # (reduced passes), through the C-ABI group (one rank: its all-gather step degenerates, everything else is the N > 1 path)
        try:
            rs_args = argparse.Namespace(**vars(args))
            rs_args.rowshard_steps = max(1, min(args.rowshard_steps, args.rowshard_steps_n1))
            line["extra"]["rowshard"] = run_rowshard(ctx, rs_args, index_factory=index_factory)
        except Exception as exc:   # pragma: no cover - reported, never fatal
            import traceback
            line["extra"]["rowshard"] = {"error": f"{type(exc).__name__}: {exc}", "traceback": traceback.format_exc()[-1500:]}
    if line is not None and ctx.world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(corpus, levels, queries, k)
    # LAST, behind the CPU baseline: with the GPU encoder built BEFORE it in this process the baseline's CPU forward ran 60 x slower
    # on the GPU box (0.09 against 5.2 strings/s, twice; a 16-CPU share under three thread pools) - not reproduced outside bench.py
    # (GPU service first, then a new CPU service: 100-150 ms per string either way), so the order is what guards the baseline.
    if (line is not None and ctx.world == 1 and not ctx.cpu_only and mode == MODE_AUTO and not args.no_extras and not args.no_family
            and not getattr(args, "no_embed_extra", False)):
        es0 = None
        try:
            line.setdefault("extra", {})["config0"], es0 = config0_extra(ctx)
        except Exception as exc:   # pragma: no cover - reported, never fatal
            import traceback
            line.setdefault("extra", {})["config0"] = {"error": f"{type(exc).__name__}: {exc}", "traceback": traceback.format_exc()[-1500:]}
        try:
            line.setdefault("extra", {})["embed_one_string"] = embed_one_string_extra(ctx, es0)
        except Exception as exc:   # pragma: no cover - reported, never fatal
            line.setdefault("extra", {})["embed_one_string"] = {"error": f"{type(exc).__name__}: {exc}"}
    # (the very last: pinned host memory and two copy streams - in front of the CPU baseline this extra left the baseline's CPU forward
    #  minutes per batch on the GPU box, like the GPU encoder above: the order is what guards the baseline)
    if line is not None and ctx.world == 1 and not ctx.cpu_only and mode == MODE_AUTO and not args.no_extras:
        try:
            line.setdefault("extra", {})["host_buffers"] = host_buffers_extra(ctx, args, index_factory, corpus, levels, queries, k, max(5, min(args.steps, 20)))
        except Exception as exc:   # pragma: no cover - reported, never fatal
            line.setdefault("extra", {})["host_buffers"] = {"error": f"{type(exc).__name__}: {exc}"}
    return line


def run_config3(ctx, args, index_factory=hip_index_factory):
    """BASELINE configs[3] at its stated size: the 37k corpus replicated, 1 M queries over an 8-GPU node = 125 000 per GPU
    (args.config3_queries), every rank its own batch (seed 9000 + rank), searched in slices of the index's batch size with no
    collective on the data path. One STEP = one pass over the rank's whole share; every 100th query is checked against the
    oracle after the timed region."""
    from rag_project_icd10_amd._native import MODE_AUTO
    torch = ctx.torch
    dim, n, k = 768, args.n, args.k
    nq = int(args.config3_queries)
    sl = min(nq, max(1, int(args.rowshard_slice)))
    corpus, levels = unit_rows(n, dim, 1234), icd_levels(n, 1235)
    queries = unit_rows(nq, dim, 9000 + ctx.rank)
    index = index_factory(corpus, levels, ctx.local_rank, sl, max(k, 10))
    dq = torch.from_numpy(queries).to(ctx.dev)
    ctx.sync()

    def one_pass():
        return [index.search_reweighted(dq[s:s + sl], k, MODE_AUTO) for s in range(0, nq, sl)]

    steps = max(1, min(args.steps, args.config3_steps))
    elapsed, outs, _ = ctx.timed(one_pass, steps, 1)
    stats = index.stats()
    line = None
    if ctx.rank == 0:
        sel = np.arange(0, nq, 100 if nq >= 1000 else 1)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as orc
        adj, raw, ids, lv = (torch.cat([o[j] for o in outs]).cpu().numpy()[sel] for j in range(4))
        t0 = time.perf_counter()
        os_, oi = orc.flat_ip_topk(corpus, queries[sel], k, nthreads=max(1, (os.cpu_count() or 1) // max(1, ctx.world)))
        want = orc.reweight(os_, oi, levels)
        line = {"metric": "queries_per_sec", "value": ctx.world * nq * steps / elapsed, "unit": "queries/s", "n_gpus": ctx.world,
                "steps": steps, "warmup": 1, "ms_per_step": elapsed / steps * 1e3, "scaling": "weak", "dtype": "f16", "data": "synthetic",
                "config": {"workload": f"BASELINE configs[3]: {n}x{dim} corpus replicated, {ctx.world * nq} synthetic 768-d queries sharded per GPU "
                                       f"({nq} per GPU, slices of {sl}), top_k={k}, search + level reweight, no data-path collective",
                           "nq_per_gpu": nq, "queries_total": ctx.world * nq, "corpus_rows": n, "slice": sl, "top_k": k,
                           "parallelism": f"query-sharded x{ctx.world}, corpus replicated"},
                "ids_exact": bool(np.array_equal(ids, want[2])), "adjusted_scores_exact": bool(adj.tobytes() == want[0].tobytes()),
                "max_abs_dscore": float(np.max(np.abs(raw.astype(np.float64) - want[1].astype(np.float64)))),
                "parity_checked_queries": int(len(sel)), "parity_check_s": round(time.perf_counter() - t0, 2),
                "fallback_queries_last_slice": int(stats.get("last_fallback", 0))}
    index.close()
    return line


def native_group_trial(ctx, index, queries, sl, k, ref_outs, limit_s, steps=1):
    """N > 1 only, AFTER the row-sharded measurement on the torch.distributed engine: the same workload through the C-ABI group
    (icd_group_*: RCCL opened by the library, ncclCommInitRank + one grouped ncclAllGather + merge on one stream). It has never
    run on more than one GPU, so it runs in a side thread under a time limit: the ranks agree on prepare / connect
    (ShardedSearch._open_native), EVERY slice must equal the torch engine's output bit for bit, and then the whole query batch
    is timed under the same contract as the measurement before it (warm-up pass, barrier + synchronize on both sides, max over
    ranks). When all of that holds, run_rowshard reports THIS engine's number as the leg's `value` (config.engine says so) and
    keeps the torch engine's next to it. A hang costs `limit_s` seconds and is reported; the measurement above is untouched."""
    import threading
    from rag_project_icd10_amd.sharded import ROW_SHARD, ShardedSearch
    torch = ctx.torch
    res = {"status": "started"}
    nq = int(queries.shape[0])

    def body():
        try:
            if not ctx.cpu_only:
                torch.cuda.set_device(ctx.dev)   # (a new thread starts on device 0: the barrier and the reductions of ctx.timed below must run on this rank's device)
            sh = ShardedSearch.from_index(index, ROW_SHARD, native=True)
            if sh.native_group is None:
                res["status"] = "not opened: the ranks agreed on the torch.distributed engine (see the warning on stderr)"
                return
            outs = [sh.search_reweighted(queries[s:s + sl], k) for s in range(0, nq, sl)]
            ctx.sync()
            res["equals_torch_engine"] = bool(all(torch.equal(a, b) for o, r in zip(outs, ref_outs) for a, b in zip(o, r)))
            res["slices_compared"] = len(outs)
            elapsed, _, _ = ctx.timed(lambda: [sh.search_reweighted(queries[s:s + sl], k) for s in range(0, nq, sl)], steps, 1)
            res["steps"] = int(steps)
            res["ms_per_step"] = elapsed / steps * 1e3
            res["value"] = nq * steps / elapsed
            res["ms_per_slice"] = elapsed / steps / len(outs) * 1e3
            res["slice"] = int(min(sl, nq))
            sh.close()
            res["status"] = "ok"
        except Exception as exc:   # reported, never fatal: the line's numbers do not depend on this engine
            res["status"] = f"error: {exc}"

    th = threading.Thread(target=body, daemon=True)
    th.start()
    th.join(limit_s)
    if th.is_alive():
        res["status"] = f"timed out after {limit_s:.0f} s (a rank stuck in the C-ABI collective)"
        ctx.native_hung = True
    return res


def device_shard(torch, dev, n, dim, seed):
    """one shard generated on the device, rows L2-normalised, levels from the real CSV histogram"""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    corpus = torch.empty((n, dim), dtype=torch.float32, device=dev)
    for s in range(0, n, 250_000):   # generated in slabs: no 2x transient
        e = min(n, s + 250_000)
        x = torch.randn((e - s, dim), generator=g, device=dev, dtype=torch.float32)
        corpus[s:e] = x / x.norm(dim=1, keepdim=True)
    r = torch.rand(n, generator=g, device=dev)
    levels = torch.where(r < 0.1243, 1, torch.where(r < 0.4234, 2, 3)).to(torch.int32)
    return corpus, levels


def exact_merge_torch(torch, scores, ids, k):
    """[G, m, k] best-first lists -> global top-k by (score desc, id asc); plain torch, the checker of the sample"""
    G, m, _ = scores.shape
    s = scores.permute(1, 0, 2).reshape(m, G * k).to(torch.float64)
    i = ids.permute(1, 0, 2).reshape(m, G * k)
    o1 = torch.argsort(i, dim=1, stable=True)
    s, i = torch.gather(s, 1, o1), torch.gather(i, 1, o1)
    o2 = torch.argsort(-s, dim=1, stable=True)
    return torch.gather(s, 1, o2)[:, :k].to(torch.float32), torch.gather(i, 1, o2)[:, :k]


def run_rowshard(ctx, args, index_factory=hip_index_factory, sharded_factory=None, native_trial=None):
    """BASELINE configs[4]: the corpus row-sharded over the ranks (1.25 M rows per GPU), the query batch replicated"""
    from rag_project_icd10_amd.sharded import ROW_SHARD, ShardedSearch
    torch, dist = ctx.torch, ctx.dist
    n, dim, k = args.rows_per_gpu, 768, args.k
    nq, sl = args.rowshard_queries, args.rowshard_slice
    corpus, levels = device_shard(torch, ctx.dev, n, dim, 1234 + ctx.rank)
    t0 = time.perf_counter()
    index = index_factory(corpus, levels, ctx.local_rank, sl, max(k, 10), ctx.rank * n)
    ctx.sync()
    t_build = time.perf_counter() - t0
    shard_host = corpus.cpu().numpy()   # the oracle's copy of this rank's shard (the checker below; outside the timed region)
    levels_host = levels.cpu().numpy()
    del corpus
    gq = torch.Generator(device=ctx.dev)
    gq.manual_seed(4321)
    queries = torch.randn((nq, dim), generator=gq, device=ctx.dev, dtype=torch.float32)
    queries /= queries.norm(dim=1, keepdim=True)
    sharded = sharded_factory(index) if sharded_factory else ShardedSearch.from_index(index, ROW_SHARD)

    def one_pass():
        return [sharded.search_reweighted(queries[s:s + sl], k) for s in range(0, nq, sl)]

    steps, warmup = max(1, min(args.steps, args.rowshard_steps)), 1

    def before():
        if hasattr(index, "set_profiling"):
            index.set_profiling(True)
            index.profile_summary()

    elapsed, outs, _ = ctx.timed(one_pass, steps, warmup, before)
    prof = index.profile_summary() if hasattr(index, "profile_summary") else {"ms_coarse": 0.0, "count": 0}
    if hasattr(index, "set_profiling"):
        index.set_profiling(False)
    # correctness on a query sample, against the CPU ORACLE: every rank runs oracle/icd_oracle.c over its own shard (host
    # copy), the per-shard exact hits are gathered and merged by plain torch (score desc, id asc), and the ids / raw scores
    # the row-sharded HIP path returned must equal that merge bit for bit
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    # eight queries of EVERY slice, spread by stride - the short last slice (another launch geometry) included
    starts = list(range(0, nq, sl))
    per = 8 if ctx.world <= 2 else 4
    pick = [(j, (i * min(sl, nq - s0)) // per) for j, s0 in enumerate(starts) for i in range(min(per, min(sl, nq - s0)))]
    sel = torch.tensor([starts[j] + o for j, o in pick], device=queries.device, dtype=torch.long)
    m = int(len(sel))
    sample = queries[sel].contiguous()
    t0 = time.perf_counter()
    os_, oi_ = orc.flat_ip_topk(shard_host, sample.cpu().numpy(), k, id_base=ctx.rank * n,
                                nthreads=max(1, (os.cpu_count() or 1) // max(1, ctx.world)))
    check_s = time.perf_counter() - t0
    es, ei = torch.from_numpy(os_).to(ctx.dev), torch.from_numpy(oi_).to(ctx.dev)
    if ctx.world > 1:
        gs = [torch.empty_like(es) for _ in range(ctx.world)]
        gi = [torch.empty_like(ei) for _ in range(ctx.world)]
        dist.all_gather(gs, es.contiguous())
        dist.all_gather(gi, ei.contiguous())
        es, ei = torch.stack(gs), torch.stack(gi)
    else:
        es, ei = es.unsqueeze(0), ei.unsqueeze(0)
    ws, wi = exact_merge_torch(torch, es, ei, k)
    adj0, raw0, ids0, lv0 = (torch.stack([outs[j][c][o] for j, o in pick]) for c in range(4))   # the sampled rows of every slice
    # the path's output is in reweighted order: compare as sets of (id, raw score) per query, and its adjusted order
    got = torch.argsort(ids0, dim=1)
    want = torch.argsort(wi, dim=1)
    ids_ok = bool(torch.equal(torch.gather(ids0, 1, got), torch.gather(wi, 1, want)))
    raw_ok = bool(torch.equal(torch.gather(raw0, 1, got).view(torch.int32), torch.gather(ws, 1, want).view(torch.int32)))
    sorted_ok = all(bool((o[0][:, 1:] <= o[0][:, :-1]).all()) for o in outs)
    # rank 0 also checks the reweighted ORDER of the sample: adjusted score = raw x weight of the level the hit carries
    # (levels of its own shard from the host copy; hits of other shards carry their level in the payload)
    adj_ok = True
    if ctx.rank == 0:
        a0, r0, i0, l0 = (x.cpu().numpy() for x in (adj0, raw0, ids0, lv0))
        w = np.where(l0 == 1, 1.2, np.where(l0 == 3, 0.8, 1.0))
        adj_ok = bool(np.array_equal(a0, r0.astype(np.float64) * w))
        mine = (i0 >= 0) & (i0 < n)
        adj_ok = adj_ok and bool(np.array_equal(l0[mine], levels_host[i0[mine]]))
    stats = index.stats() if hasattr(index, "stats") else {}
    trial = None
    if ctx.world > 1 and native_trial is not None and not args.no_native_trial:   # (the CPU test engine's stand-in: the adoption logic below on gloo ranks)
        trial = native_trial(ctx, index, queries, sl, k, outs, args.native_trial_limit, steps)
    elif ctx.world > 1 and sharded_factory is None and getattr(sharded, "native_group", None) is None and not args.no_native_trial:
        trial = native_group_trial(ctx, index, queries, sl, k, outs, args.native_trial_limit, steps)
    # the C-ABI group becomes the leg's engine when its trial passed on EVERY rank: opened, every slice bit-identical to the torch
    # engine's, the timed passes finished (the ranks agree over the CPU side channel: a rank-0-only decision would be a lie)
    trial_ok = bool(trial) and trial.get("status") == "ok" and bool(trial.get("equals_torch_engine"))
    native_default = ctx.world > 1 and trial is not None and not ctx.any_rank(not trial_ok)
    if trial is not None:
        trial["adopted_as_the_legs_engine"] = bool(native_default)   # (false with status ok here: another rank's trial did not pass)
    line = None
    if ctx.rank == 0:
        flops_gpu = 2.0 * nq * n * dim
        launches = -(-nq // sl)
        dom_ms = float(prof.get("ms_coarse", 0.0))
        # (slices differ in size only in the last one: price the mean launch against the mean slice)
        achieved = (flops_gpu / launches) / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        traffic, traffic_src = pmc_traffic("coarse_flat_kernel", sl, n, "_rowshard")
        native_engine = getattr(sharded, "native_group", None) is not None
        value, ms_step = nq * steps / elapsed, elapsed / steps * 1e3
        torch_leg = None
        if native_default:   # (the trial's number: the same passes, the same timing contract, the C-ABI collective)
            torch_leg = {"value": value, "ms_per_step": ms_step, "what": "the same passes on all_gather_into_tensor (torch.distributed), measured first"}
            value, ms_step, native_engine = float(trial["value"]), float(trial["ms_per_step"]), True
        line = {
            "metric": "queries_per_sec", "value": value, "unit": "queries/s",
            "n_gpus": ctx.world, "steps": steps, "warmup": warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[4]: {ctx.world} x {n} rows x {dim} corpus row-sharded (generated on device), "
                                   f"{nq} queries replicated, top_k={k}, slices of {sl}: local top-k -> all_gather -> merge + reweight",
                       "rows_per_gpu": n, "corpus_rows_total": ctx.world * n, "queries": nq, "slice": sl, "dim": dim, "top_k": k,
                       "parallelism": f"row-sharded x{ctx.world}", "collective": ("ncclAllGather inside icd_group_search (C ABI, RCCL opened by the library)" if native_engine
                                      else "all_gather_into_tensor (torch.distributed)") if ctx.world > 1 else "none (one shard)",
                       "collective_ranks": dist.get_world_size() if ctx.world > 1 else 1,
                       "engine": "icd_group (C ABI)" if native_engine or ctx.world == 1 else getattr(sharded, "engine", "torch.distributed")},
            "torch_distributed_engine": torch_leg,
            "whole_job_tflops": ctx.world * flops_gpu / (ms_step * 1e-3) / 1e12,
            "ids_exact_on_sample": ids_ok, "raw_scores_exact_on_sample": raw_ok, "adjusted_sorted": sorted_ok,
            "adjusted_scores_exact_on_sample": adj_ok, "sample_queries": int(m), "sample_slices": int(len(starts)),
            "sample_rule": f"{per} queries of every slice of {sl} (the last slice has {nq - starts[-1]}), spread by stride",
            "sample_checked_against": "oracle/icd_oracle.c over every rank's shard (host copy), merged by plain torch", "parity_check_s": round(check_s, 2),
            "fallback_queries_last_slice": int(stats.get("last_fallback", 0)),
            "index_build_s": round(t_build, 3),
            "native_group_trial": trial,
            "roofline": {"bound": "mfma", "kernel": "coarse_flat_kernel", "achieved": achieved, "peak": PEAK_TFLOPS_F16,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_TFLOPS_F16, "traffic": traffic, "traffic_source": traffic_src,
                         "flops_per_launch": flops_gpu / launches, "launch_ms": dom_ms, "launches_averaged": int(prof.get("count", 0))},
        }
    if not getattr(ctx, "native_hung", False):   # (a rank stuck inside the C-ABI collective still owns the group: leave it to process exit)
        if hasattr(sharded, "close"):
            sharded.close()
        index.close()
    return line


def test_engine():
    """CPU tests of the N-rank control flow (ICD_BENCH_DEVICE=cpu) name a Python file that provides `index_factory` and
    `sharded_factory` (tests/bench_cpu_engine.py); the product path has no such hook: without it the HIP index is the
    only engine and a box without a GPU fails loudly"""
    path = os.environ.get("ICD_BENCH_TEST_ENGINE")
    if not path or os.environ.get("ICD_BENCH_DEVICE") != "cpu":
        return {}
    import importlib.util
    spec = importlib.util.spec_from_file_location("icd_bench_test_engine", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {"index_factory": mod.index_factory, "sharded_factory": mod.sharded_factory}
    for hook in ("on_start", "on_exit", "native_trial"):
        if hasattr(mod, hook):
            out[hook] = getattr(mod, hook)
    return out


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def visible_gpus():
    """GPUs this process may use, counted WITHOUT touching the HIP runtime (torch.cuda.device_count() falls through to
    hipGetDeviceCount when amdsmi is absent, which opens /dev/kfd). The visibility variables when set - the MINIMUM over
    all that are set: HIP_/CUDA_VISIBLE_DEVICES index INTO the subset ROCR_VISIBLE_DEVICES leaves, so the smallest list
    bounds the count - else the KFD topology nodes that have SIMDs (CPU nodes have none) and whose render device this
    process can open (a container may see every node of the host and own only some). None when neither source exists."""
    counts = []
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            counts.append(len([t for t in v.split(",") if t.strip() != ""]))
    if counts:
        return min(counts)
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        count = 0
        for node in os.listdir(base):
            simd, minor = 0, None
            with open(os.path.join(base, node, "properties")) as fh:
                for ln in fh:
                    if ln.startswith("simd_count"):
                        simd = int(ln.split()[1])
                    elif ln.startswith("drm_render_minor"):
                        minor = int(ln.split()[1])
            if simd <= 0:
                continue
            dev = f"/dev/dri/renderD{minor}" if minor is not None and minor > 0 else None
            if dev is not None and os.path.exists("/dev/dri") and not os.access(dev, os.R_OK | os.W_OK):
                continue   # (a node of the host this container was not given)
            count += 1
        return count
    except OSError:
        return None


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as ONE child (python -m torch.distributed.run, one
    rank per GPU), relay rank 0's JSON line, exit with the child's status. The parent never initialises HIP (a process that
    has must not exec or fork GPU work): it counts the devices from the environment / sysfs (visible_gpus). The child runs in
    its own process group under a wall-clock limit (--rank-timeout): a rank that hangs - one stuck in a collective the others
    never entered - gets the whole group killed and the parent exits 124, instead of hanging until the caller's own limit."""
    import signal
    import subprocess
    import threading
    one_device = os.environ.get("ICD_BENCH_ONE_DEVICE") == "1" or os.environ.get("ICD_BENCH_DEVICE") == "cpu"
    have = visible_gpus()
    if have is None:   # no sysfs topology (not a ROCm box): ask torch, which may initialise the runtime - nothing is started after a refusal
        import torch
        have = torch.cuda.device_count()
    if not one_device and have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible", file=sys.stderr)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), "--", os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (dmabuf IPC: RCCL between processes needs it on this driver)
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, start_new_session=True)
    timed_out = []

    def descendants(root):
        """every live process below `root` (torchrun starts each rank in a session of its own, so the launcher's process group
        does not contain them): the /proc parent links, followed from the child this function started - exact PIDs, never a
        name pattern"""
        kids = {}
        for ent in os.listdir("/proc"):
            if ent.isdigit():
                try:
                    with open(f"/proc/{ent}/stat") as fh:
                        st = fh.read()
                    ppid = int(st[st.rindex(")") + 2:].split()[1])
                    kids.setdefault(ppid, []).append(int(ent))
                except (OSError, ValueError):
                    pass
        out, todo = [], [root]
        while todo:
            for c in kids.get(todo.pop(), []):
                out.append(c)
                todo.append(c)
        return out

    def kill_group():
        timed_out.append(True)
        victims = descendants(p.pid)   # (listed BEFORE the launcher dies: its orphans would be re-parented to init)
        for pid in [p.pid] + victims:
            try:
                os.kill(pid, signal.SIGKILL)
            except OSError:
                pass
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass

    watchdog = threading.Timer(max(1.0, float(args.rank_timeout)), kill_group)
    watchdog.daemon = True
    watchdog.start()
    line = None
    for raw in p.stdout:
        t = raw.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        elif t:
            print(t, file=sys.stderr)   # anything else a rank printed
    rc = p.wait()
    watchdog.cancel()
    if timed_out:
        print(f"bench.py: the {args.gpus}-rank child did not finish within --rank-timeout {args.rank_timeout:.0f} s: the ranks (its process tree) were killed", file=sys.stderr)
        if line is not None:   # rank 0 had printed its line before a rank hung (e.g. in the teardown): the measurement is relayed, the status says what happened
            print(line, flush=True)
        return 124
    if rc != 0 or line is None:
        print(f"bench.py: the {args.gpus}-rank child exited with status {rc}" + ("" if line else " and printed no result line"), file=sys.stderr)
        return rc if rc != 0 else 3
    print(line, flush=True)
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--settle-ms", type=float, default=60.0,
                    help="untimed: keep the device busy with the step for this long before the warm-up steps (clock ramp of a fresh process)")
    ap.add_argument("--profile-every", type=int, default=5, help="hipEvents around the kernels of every N-th step of the timed region")
    ap.add_argument("--repeats", type=int, default=5, help="replicated workload: repeat the K-step window this many times after the measurement (reported under extra.windows)")
    ap.add_argument("--workload", choices=["replicated", "rowshard"], default="replicated")
    ap.add_argument("--nq", type=int, default=10000)
    ap.add_argument("--n", "--corpus-rows", dest="n", type=int, default=37000)   # (--corpus-rows: `--n` is an ambiguous prefix for torch.distributed.run's own parser)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--mode", choices=["auto", "exact"], default="auto")
    ap.add_argument("--rows-per-gpu", type=int, default=1_250_000)
    ap.add_argument("--rowshard-queries", type=int, default=100_000)
    ap.add_argument("--rowshard-slice", type=int, default=16384)
    ap.add_argument("--rowshard-steps", type=int, default=3, help="passes of the row-sharded workload (each ~0.2 s per GPU)")
    ap.add_argument("--no-rowshard", action="store_true", help="skip the row-sharded leg of the default run (N > 1: `rowshard`; N = 1: `extra.rowshard`)")
    ap.add_argument("--rowshard-steps-n1", type=int, default=2, help="N = 1: passes of the row-sharded leg inside the default line (extra.rowshard)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="N = 1: skip the real-size / clustered / k = 20 / family / exact-mode side workloads")
    ap.add_argument("--no-family", action="store_true", help="N = 1: skip the k = 20 and ICD-shaped (family) side workloads")
    ap.add_argument("--family-families", type=int, default=300)
    ap.add_argument("--family-rows", type=int, default=124)
    ap.add_argument("--config3-queries", type=int, default=125_000, help="N > 1: queries per GPU of the BASELINE configs[3] leg (1 M over 8 GPUs)")
    ap.add_argument("--config3-steps", type=int, default=3)
    ap.add_argument("--no-config3", action="store_true", help="N > 1: skip the configs[3] leg of the default run")
    ap.add_argument("--no-native-trial", action="store_true", help="N > 1: skip the time-limited trial of the C-ABI RCCL group behind the row-sharded leg")
    ap.add_argument("--native-trial-limit", type=float, default=90.0)
    ap.add_argument("--no-embed-extra", action="store_true", help="leave extra.embed_one_string out (the encode_query latency of the embed step)")
    ap.add_argument("--rank-timeout", type=float, default=1500.0, help="--gpus N > 1 started by this process: wall-clock limit of the ranks in seconds")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    in_group = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus > 1 and not in_group:
        sys.exit(spawn_ranks(args, argv))
    if in_group and int(os.environ["WORLD_SIZE"]) != args.gpus and os.environ.get("RANK") == "0":
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks: reporting the latter", file=sys.stderr)

    ctx = Ctx()
    eng = test_engine()
    on_exit = eng.pop("on_exit", None)
    if "on_start" in eng:   # (the CPU test engine's fault injection: a rank that dies / never returns; nothing of the kind lives here)
        eng.pop("on_start")(ctx)
    rs_eng = dict(eng)
    if "native_trial" in eng:
        eng = {k: v for k, v in eng.items() if k != "native_trial"}
    if args.workload == "rowshard":
        line = run_rowshard(ctx, args, **rs_eng)
    else:
        line = run_replicated(ctx, args, **({"index_factory": eng["index_factory"]} if eng else {}))
        if ctx.world > 1 and not args.no_config3:
            c3 = run_config3(ctx, args, **({"index_factory": eng["index_factory"]} if eng else {}))
            if line is not None:
                line["config3"] = c3
        if ctx.world > 1 and not args.no_rowshard:
            rs = run_rowshard(ctx, args, **rs_eng)
            if line is not None:
                line["rowshard"] = rs
    if ctx.rank == 0:
        print(json.dumps(line), flush=True)
    # A rank whose C-ABI trial timed out has a thread stuck in a GPU collective: it cannot enter the teardown's barrier on the
    # same device queue, and the ranks whose trial finished would wait there for it until --rank-timeout. The ranks
    # AGREE on the outcome over the gloo side group (CPU tensors; made before the trial): if any rank hung, every rank leaves
    # without the barrier - the line is out, nothing else of value can happen.
    if on_exit:   # (test engine: a rank that hangs in the teardown, after rank 0's line is out)
        on_exit(ctx)
    hung = ctx.any_rank(getattr(ctx, "native_hung", False))
    if hung:
        if ctx.rank == 0:
            print("bench.py: a rank's C-ABI group trial timed out (native_group_trial.status): every rank exits without the teardown barrier", file=sys.stderr)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
    if ctx.world > 1:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
