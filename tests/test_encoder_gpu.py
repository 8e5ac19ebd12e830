"""Encoder on ROCm against the same seeded weights on the CPU in fp32 (rows a3-a5 of SURVEY.md section 8): the three
ways a string reaches the GPU forward - encode_query (one string, the replayed HIP-graph bucket), encode_batch (batch 32,
`/embed`), encode_query_batch (batch 256, length-sorted, the corpus build) - must give the embedding the plain eager CPU
forward gives: max |delta| <= 1e-5 per component, unit norms. Weights are the seeded synthetic BERT-base (no checkpoint
offline: numerical parity with sentence-transformers itself stays unpinned)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(scope="module")
def services():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    os.environ["EMBEDDING_MODEL_NAME"] = "shibing624/text2vec-base-chinese"
    os.environ.pop("ICD_EMBEDDING_DTYPE", None)
    from rag_project_icd10_amd.services.embedding_service import EmbeddingService
    gpu = EmbeddingService(allow_synthetic=True, device="cuda")
    cpu = EmbeddingService(allow_synthetic=True, device="cpu")
    for (n1, p1), (n2, p2) in zip(gpu.model.state_dict().items(), cpu.model.state_dict().items()):
        assert n1 == n2 and torch.equal(p1.cpu(), p2)                    # same seeded weights on both devices
    return gpu, cpu


def _strings():
    texts = [l.rstrip("\n") for l in open(os.path.join(GOLDEN, "diagnosis_strings.txt"), encoding="utf-8")][:300]
    return texts + ["", " ", "肺", "x" * 128, "高血压" * 60, "Possible 肺炎?"]   # empty, one token, > max_seq_length


def test_encode_query_graph_bucket_matches_cpu(services):
    gpu, cpu = services
    for t in _strings()[:40] + _strings()[-6:]:
        a, b = gpu.encode_query(t), cpu.encode_query(t)
        assert a.shape == (768,) and a.dtype == np.float32
        assert np.max(np.abs(a - b)) <= TOL, t
        assert abs(np.linalg.norm(a) - 1.0) <= 1e-5
    assert gpu._small is not None, "one string per call should take the small-input encoder (csrc/encoder_small.hpp)"
    again = gpu.encode_query(_strings()[3])                                # a replay gives the same bits as the first replay
    assert np.array_equal(again, gpu.encode_query(_strings()[3]))


def test_encode_query_through_the_replayed_framework_graph_matches_cpu(services, monkeypatch):
    """ICD_EMBEDDING_SMALL=0 (or a model the small-input encoder is not instantiated for): the framework's forward, replayed
    from a HIP graph per (batch, width) bucket"""
    gpu, cpu = services
    monkeypatch.setattr(gpu, "_small", None)
    for t in _strings()[:12] + _strings()[-6:]:
        a, b = gpu.encode_query(t), cpu.encode_query(t)
        assert np.max(np.abs(a - b)) <= TOL, t
    assert gpu._graphs, "the one-string path should have captured a HIP graph"
    texts = _strings()[:24]                                                 # 24 strings through the (batch, width) graph buckets
    a, b = gpu.encode_query_batch(texts, batch_size=32), cpu.encode_query_batch(texts, batch_size=32)
    assert np.max(np.abs(a - b)) <= TOL


def test_small_input_encoder_matches_the_framework_forward(services):
    """csrc/encoder_small.hpp through icd_encoder_encode: 1 ... 64 sequences, 1 ... 512 packed tokens, every token bucket and
    its edges - pooled rows (mean and [CLS], normalised or not) and the last hidden state of every token against
    transformers' padded BertModel forward of the same weights on the GPU (1e-5); host and device outputs identical; the
    descriptor's limits refused with a clear error."""
    import torch
    from rag_project_icd10_amd import _native
    gpu, _ = services
    enc = gpu._small
    assert enc is not None
    rng = np.random.default_rng(5)
    vocab = gpu.model.bert.config.vocab_size

    def make(lengths):
        return [[101] + [int(v) for v in rng.integers(1000, vocab, size=n - 2)] + [102] if n >= 2 else [101] for n in lengths]

    def reference(ids, pooling):
        width = max(len(x) for x in ids)
        tok = torch.zeros((len(ids), width), dtype=torch.long)
        mask = torch.zeros((len(ids), width), dtype=torch.long)
        for r, x in enumerate(ids):
            tok[r, :len(x)] = torch.tensor(x)
            mask[r, :len(x)] = 1
        tok, mask = tok.cuda(), mask.cuda()
        with torch.no_grad():
            hidden = gpu.model.bert(input_ids=tok, attention_mask=mask).last_hidden_state
        m = mask.unsqueeze(-1).float()
        pooled = hidden[:, 0] if pooling == "cls" else (hidden * m).sum(1) / m.sum(1)
        rows = torch.cat([hidden[r, :len(x)] for r, x in enumerate(ids)], 0)
        return pooled.cpu().numpy(), torch.nn.functional.normalize(pooled, p=2, dim=1).cpu().numpy(), rows.cpu().numpy()

    cases = [[1], [2], [3], [15], [16], [17], [31], [32], [33], [64], [65], [100], [128], [5, 9, 12, 30], [4] * 32, [1] * 32, [64, 64],
             [17, 1, 40, 2, 23], [16, 16], [7] * 18, [100, 28], [129], [200], [130, 100], [8] * 32, [256], [90, 3, 128, 35],
             [257], [300, 200], [512], [8] * 64, [1] * 64, [400, 100, 12], [128] * 4]
    for lengths in cases:
        ids = make(lengths)
        for pooling in ("mean", "cls"):
            want_raw, want_unit, want_rows = reference(ids, pooling)
            got_unit, rows = enc.encode(ids, pooling=pooling, normalize=True, hidden=True)
            got_raw = enc.encode(ids, pooling=pooling, normalize=False)
            assert got_unit.shape == (len(ids), 768) and got_unit.dtype == np.float32
            assert np.max(np.abs(got_unit - want_unit)) <= TOL, (lengths, pooling)
            # (un-normalised rows have norm ~ 10-20: 2e-5 is 1-2e-6 relative with the fp32 MFMAs; the split-bf16 arithmetic, the
            #  default, sits ~1e-6 off the fp32 forward on UNIT rows - tests above and below - which is 4e-5 here)
            assert np.max(np.abs(got_raw - want_raw)) <= (2e-5 if enc.arithmetic == "fp32" else 8e-5), (lengths, pooling)
            assert np.max(np.abs(rows.cpu().numpy() - want_rows)) <= 5e-5, (lengths, pooling)
            dev = enc.encode(ids, pooling=pooling, normalize=True, to_device=True)
            torch.cuda.synchronize()
            assert dev.is_cuda and np.array_equal(dev.cpu().numpy(), got_unit)
    with pytest.raises(_native.IcdError, match="512 tokens"):
        enc.encode(make([400, 113]))
    with pytest.raises(_native.IcdError, match="sequences per call"):
        enc.encode(make([2] * 65))
    with pytest.raises(_native.IcdError, match="vocabulary"):
        enc.encode([[101, vocab, 102]])
    assert not enc.fits([400, 113]) and not enc.fits([2] * 65) and enc.fits([512]) and not enc.fits([])
    assert enc.fits_each([512] * 100) and not enc.fits_each([513]) and not enc.fits_each([])


def test_encode_batch_32_matches_cpu(services):
    gpu, cpu = services
    texts = _strings()[:100] + _strings()[-6:]
    a = np.asarray(gpu.encode_batch(texts, show_progress=False), dtype=np.float64)
    b = np.asarray(cpu.encode_batch(texts, show_progress=False), dtype=np.float64)
    assert a.shape == (len(texts), 768)
    assert np.max(np.abs(a - b)) <= TOL
    assert np.max(np.abs(np.linalg.norm(a, axis=1) - 1.0)) <= 1e-5
    for i in (0, 7, 50, len(texts) - 6, len(texts) - 1):                   # batching does not change a row: not one bit
        assert np.array_equal(gpu.encode_single(texts[i]).astype(np.float64), a[i]), i


def test_encode_query_batch_256_matches_cpu_and_stays_on_device(services):
    import torch
    gpu, cpu = services
    texts = _strings()
    dev = gpu.encode_query_batch(texts, batch_size=256, to_device=True)
    assert dev.is_cuda and dev.dtype == torch.float32 and dev.shape == (len(texts), 768)
    a = dev.cpu().numpy()
    b = cpu.encode_query_batch(texts, batch_size=256)
    assert np.max(np.abs(a - b)) <= TOL
    assert np.max(np.abs(np.linalg.norm(a, axis=1) - 1.0)) <= 1e-5
    assert gpu.batch_arithmetic() == "canonical"
    for i in (0, 17, len(texts) - 6, len(texts) - 3):                      # the batch row == the one-at-a-time call, bit for bit
        assert np.array_equal(a[i], gpu.encode_query(texts[i])), i


def test_packed_forward_matches_the_padded_hf_forward(services, monkeypatch):
    """large batches run the encoder over packed tokens (one GEMM per Linear over all tokens, padding only inside the
    attention): the same embeddings as transformers' padded BertModel forward, on the CPU and on ROCm, with mean and CLS
    pooling, and whatever the chunking"""
    gpu, cpu = services
    texts = _strings()
    assert gpu._packed is not None and cpu._packed is not None
    monkeypatch.setattr(gpu, "_batch_fast", True)                          # (ICD_EMBEDDING_BATCH=fast: the packed forward; the default is the canonical path)
    packed = gpu.encode_query_batch(texts, batch_size=256)
    monkeypatch.setattr(cpu, "_packed", None)
    monkeypatch.setattr(gpu, "_packed", None)
    padded_cpu = cpu.encode_query_batch(texts, batch_size=256)            # HF BertModel, padded, fp32 on the CPU
    monkeypatch.setattr(gpu, "_small", None)                               # (the padded HF forward on the GPU: no small-input encoder, no packed one)
    padded_gpu = gpu.encode_query_batch(texts, batch_size=256)
    monkeypatch.undo()
    assert np.max(np.abs(packed - padded_cpu)) <= TOL and np.max(np.abs(packed - padded_gpu)) <= TOL
    monkeypatch.setattr(gpu, "_batch_fast", True)
    monkeypatch.setattr(gpu, "PACK_TOKENS", 700)                           # many chunks
    assert np.max(np.abs(gpu.encode_query_batch(texts, batch_size=256) - packed)) <= TOL
    monkeypatch.undo()
    monkeypatch.setattr(gpu, "_batch_fast", True)
    monkeypatch.setattr(gpu, "pooling", "cls")
    cls_packed = gpu.encode_query_batch(texts, batch_size=256)
    monkeypatch.setattr(gpu, "_packed", None)
    monkeypatch.setattr(gpu, "_small", None)
    monkeypatch.setattr(gpu.model, "pooling", "cls")
    cls_padded = gpu.encode_query_batch(texts, batch_size=256)
    assert np.max(np.abs(cls_packed - cls_padded)) <= TOL and np.max(np.abs(cls_packed - packed)) > 1e-3


def test_split_bf16_gemms_of_the_packed_forward_stay_within_the_fp32_tolerance(services, monkeypatch):
    """The packed encoder's Linear layers run as split-bf16 GEMMs on the bf16 MFMA (x_hi W_hi + x_hi W_lo + x_lo W_hi, fp32
    accumulation and output, one GEMM over the concatenated K; ICD_EMBEDDING_GEMM=bf16x3, the default): the embeddings stay
    within the tolerance of the fp32 forward (the arithmetic the reference reaches through SentenceTransformer.encode,
    services/embedding_service.py:97-102) - against the SAME service with fp32 GEMMs and against the CPU fp32 forward - and
    the cosines of all pairs move by less than 1e-5 (north_star's score tolerance)"""
    gpu, cpu = services
    texts = _strings()
    assert gpu._packed is not None
    monkeypatch.setattr(gpu, "_batch_fast", True)
    got = gpu.encode_query_batch(texts, batch_size=256)
    if not gpu._packed.split_gemm:
        pytest.skip("this torch has no mm(out_dtype=...): the packed forward runs fp32 GEMMs")
    monkeypatch.setattr(gpu._packed, "split_gemm", False)
    fp32 = gpu.encode_query_batch(texts, batch_size=256)
    monkeypatch.undo()
    canon = gpu.encode_query_batch(texts, batch_size=256)                  # the canonical path (the default): what the fast one is "within" of
    print(f"split-bf16 GEMMs: max |d embedding| vs the canonical small-input forward {float(np.max(np.abs(got - canon))):.2e}")
    assert np.max(np.abs(got - canon)) <= TOL
    ref = cpu.encode_query_batch(texts, batch_size=256)
    d_gpu, d_cpu = float(np.max(np.abs(got - fp32))), float(np.max(np.abs(got - ref)))
    d_cos = float(np.max(np.abs(got.astype(np.float64) @ got.astype(np.float64).T - ref.astype(np.float64) @ ref.astype(np.float64).T)))
    print(f"split-bf16 GEMMs: max |d embedding| vs the fp32 GEMMs on the GPU {d_gpu:.2e}, vs the CPU fp32 forward {d_cpu:.2e}; max |d cosine| {d_cos:.2e}")
    assert d_gpu <= TOL and d_cpu <= TOL and d_cos <= 1e-5


def test_native_packed_attention_kernel_matches_sdpa():
    """icd_packed_attention alone (through the C ABI): softmax(Q K^T / 8) V per sequence and head over packed tokens against
    a float64 softmax attention (and torch's fp32 scaled_dot_product_attention for scale) on every sequence separately - lengths 1 .. 512
    (one pass up to 64 keys, the chunked recurrence beyond), 12 heads, rows of the packed QKV
    that do not start at row 0, an output with a wider row stride"""
    import torch
    from rag_project_icd10_amd import _native
    torch.manual_seed(5)
    rng = np.random.default_rng(6)
    lengths = [512, 511, 200, 129, 128, 127, 65, 64, 64, 63, 33, 32, 31, 17, 16, 15, 8, 2, 1] + [int(x) for x in rng.integers(1, 65, 300)] + [int(x) for x in rng.integers(65, 260, 12)]
    heads, dh = 12, 64
    H = heads * dh
    T = sum(lengths)
    qkv = torch.randn(T + 3, 3 * H, device="cuda") * 2.0
    starts = np.concatenate([[0], np.cumsum(lengths)]) + 2                      # the packed rows start at row 2
    out = torch.full((T + 3, H + 64), float("nan"), device="cuda")
    _native.packed_attention(qkv, torch.from_numpy(starts.astype(np.int32)).cuda(), len(lengths), heads, max(lengths), out)
    torch.cuda.synchronize()
    worst, worst_sdpa = 0.0, 0.0
    for s, L in enumerate(lengths):
        a = int(starts[s])
        blk = qkv[a:a + L].view(L, 3, heads, dh)
        q, k, v = (blk[:, i].transpose(0, 1) for i in range(3))                 # [heads, L, dh]
        # float64 reference (the inputs are scaled up: scores of +-100, outputs of a few units); SDPA in fp32 for scale
        p64 = torch.softmax(q.double() @ k.double().transpose(1, 2) / 8.0, dim=-1) @ v.double()
        want = p64.transpose(0, 1).reshape(L, H)
        sdpa = torch.nn.functional.scaled_dot_product_attention(q[None], k[None], v[None])[0].transpose(0, 1).reshape(L, H)
        got = out[a:a + L, :H]
        worst = max(worst, float((got.double() - want).abs().max()))
        worst_sdpa = max(worst_sdpa, float((sdpa.double() - want).abs().max()))
    assert worst <= 5e-5 and worst <= 1.5 * worst_sdpa + 1e-6, (worst, worst_sdpa)   # as close to float64 as SDPA's own fp32 (1.46e-5 / 1.45e-5)
    assert bool(torch.isnan(out[:2]).all()) and bool(torch.isnan(out[:, H:]).all())   # nothing written outside its rows / columns
    with pytest.raises(_native.IcdError):
        _native.packed_attention(qkv, torch.zeros(2, dtype=torch.int32, device="cuda"), 1, heads, 513, out)


def test_packed_forward_with_native_attention_matches_sdpa_groups(services, monkeypatch):
    """the encoder with the hand-written attention kernel against the same packed forward with SDPA groups, on strings on
    both sides of the kernel's 64-key chunk and up to max_seq_length"""
    gpu, _ = services
    texts = _strings() + ["肺" * n for n in (54, 55, 56, 57, 58, 90, 119, 120, 121)]   # "query: " + CLS/SEP: 62 .. 66, 98, 127, 128 (and truncated to 128) tokens
    assert gpu._packed.native_attention is not None
    monkeypatch.setattr(gpu, "_batch_fast", True)
    native = gpu.encode_query_batch(texts, batch_size=256)
    monkeypatch.setattr(gpu._packed, "native_attention", None)
    sdpa = gpu.encode_query_batch(texts, batch_size=256)
    assert np.max(np.abs(native - sdpa)) <= 2e-6
    monkeypatch.setattr(gpu, "pooling", "cls")
    cls_sdpa = gpu.encode_query_batch(texts, batch_size=256)
    monkeypatch.undo()
    monkeypatch.setattr(gpu, "_batch_fast", True)
    monkeypatch.setattr(gpu, "pooling", "cls")
    assert np.max(np.abs(gpu.encode_query_batch(texts, batch_size=256) - cls_sdpa)) <= 2e-6


@pytest.fixture(scope="module")
def encoder_corpus(tmp_path_factory):
    """The reference searches rows produced by the SAME encoder call that encodes the query (tools/build_database.py:217-222:
    encode_query(semantic_text) per record): anisotropic, family-shaped embeddings, not Gaussian rows. The full-size corpus
    (40 474 rows of the real CSV's shape: scripts/bench_build.py synth_csv over tests/golden/csv_shape.json) built on the GPU
    by DatabaseBuilder, once for the module; and the 1 000 golden diagnosis strings."""
    import json
    import sys
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from bench_build import synth_csv
    tmp = tmp_path_factory.mktemp("enc_corpus")
    env = {"MILVUS_DB_PATH": str(tmp / "db"), "MILVUS_COLLECTION_NAME": "icd10_enc_test", "MILVUS_MODE": "local",
           "EMBEDDING_MODEL_NAME": "shibing624/text2vec-base-chinese", "ICD_EMBEDDING_ALLOW_SYNTHETIC": "1"}
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    os.environ.pop("ICD_EMBEDDING_BATCH", None)
    try:
        shape = json.load(open(os.path.join(GOLDEN, "csv_shape.json"), encoding="utf-8"))
        csv_path = str(tmp / "shape.csv")
        nrows = synth_csv(csv_path, shape)
        from rag_project_icd10_amd.tools.build_database import DatabaseBuilder
        b = DatabaseBuilder()
        assert b.build_full_database(csv_path, rebuild=True)
        ms, es = b.milvus_service, b.embedding_service
        assert es.batch_arithmetic() == "canonical"
        corpus, levels = ms.client.matrix(), ms.client.levels()
        assert corpus.shape == (nrows, 768) and nrows == 40474
        strings = [l.strip() for l in open(os.path.join(GOLDEN, "diagnosis_strings.txt"), encoding="utf-8") if l.strip()][:1000]
        yield b, ms, es, corpus, levels, strings
        ms.disconnect()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("k", [10, 20])
def test_search_on_an_encoder_made_corpus_is_bit_exact(k, encoder_corpus):
    """1 000 golden diagnosis strings are encoded by the service that built the corpus, and the batch search - whatever mix of
    certified, second-pass and exact-re-search queries this data produces - must equal the oracle bit for bit."""
    import sys
    import torch
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    b, ms, es, corpus, levels, strings = encoder_corpus
    dq = es.encode_query_batch(strings, batch_size=256, to_device=True)
    queries = dq.cpu().numpy()
    os_, oi = orc.flat_ip_topk(corpus, queries, k)
    want = orc.reweight(os_, oi, levels)
    for rep in range(2):   # the first batch of the fresh index, then one planned from its counters
        adj, raw, ids, lv = ms.search_batch(dq, top_k=k)
        torch.cuda.synchronize()
        st = ms._ready_index().stats()
        assert np.array_equal(ids.cpu().numpy(), want[2]), (rep, st)
        assert adj.cpu().numpy().tobytes() == want[0].tobytes() and raw.cpu().numpy().tobytes() == want[1].tobytes(), (rep, st)
        assert np.array_equal(lv.cpu().numpy(), want[3])
    print(f"encoder-made corpus k={k}: second pass {st['last_second_pass']} (x {st['last_second_pass_lists']} lists), "
          f"exact re-search {st['last_fallback']} of {len(strings)}, wide_mode {st['wide_mode']}")


def test_a_corpus_row_is_the_vector_its_text_gets_as_a_query(encoder_corpus):
    """the reference's property (tools/build_database.py:217-222 and services/embedding_service.py:117-120 are the same call):
    the vector stored for a record IS encode_query(its semantic_text) - bit for bit, although the build encodes thousands of
    records per call; so a record searched with its own text scores itself first with a raw score equal to its own norm chain"""
    b, ms, es, corpus, levels, strings = encoder_corpus
    recs = ms.client.records
    rng = np.random.default_rng(12)
    for i in [0, 1, 2, len(recs) - 1] + [int(x) for x in rng.integers(0, len(recs), 60)]:
        v = es.encode_query(recs[i]["semantic_text"])
        assert np.array_equal(v, corpus[i]), (i, recs[i]["code"])
    hits = ms.search(es.encode_query(recs[777]["semantic_text"]), top_k=5)
    assert hits and hits[0]["code"] == recs[777]["code"]


def test_batch_rows_equal_one_string_per_call_bit_for_bit(services):
    """ONE embedding arithmetic whatever the call shape: encode_query_batch / encode_batch (thousands of tokens, cut into calls
    by icd_encoder_encode_many) return for every string the bits encode_query / encode_single return for it alone - in any
    order, next to any neighbours, host or device output"""
    import torch
    gpu, _ = services
    assert gpu.batch_arithmetic() == "canonical"
    texts = _strings() + ["肺" * n for n in (54, 55, 56, 57, 58, 90, 119, 120, 121)]
    one = np.stack([gpu.encode_query(t) for t in texts])
    got = gpu.encode_query_batch(texts)
    assert got.dtype == np.float32 and np.array_equal(got, one)
    dev = gpu.encode_query_batch(texts, to_device=True)
    torch.cuda.synchronize()
    assert dev.is_cuda and np.array_equal(dev.cpu().numpy(), one)
    perm = np.random.default_rng(3).permutation(len(texts))
    assert np.array_equal(gpu.encode_query_batch([texts[i] for i in perm]), one[perm])          # other neighbours, other calls
    assert np.array_equal(gpu.encode_query_batch(texts[:70]), one[:70])                         # 65+ strings: two calls of the library
    emb = np.asarray(gpu.encode_batch(texts[:100], show_progress=False), dtype=np.float32)      # /embed: "passage: " prefixes
    assert np.array_equal(emb, np.stack([gpu.encode_single(t) for t in texts[:100]]))
    # the library entry point itself: sequences of 1 ... 512 tokens in one list
    enc = gpu._small
    rng = np.random.default_rng(9)
    vocab = gpu.model.bert.config.vocab_size
    lengths = [1, 2, 512, 3, 500, 13, 100, 412, 16, 17] + [int(x) for x in rng.integers(1, 130, 150)]
    ids = [[int(v) for v in rng.integers(1000, vocab, size=n)] for n in lengths]
    many = enc.encode_many(ids, pooling="mean", normalize=True)
    for i in range(len(ids)):
        assert np.array_equal(many[i], enc.encode([ids[i]])[0]), (i, lengths[i])
    # passes cut by the SEQUENCE limit (2 048 per pass) and a list that ends exactly on a pass's token limit
    tiny_lengths = [int(x) for x in rng.integers(1, 5, 4500)]
    tiny = [[int(v) for v in rng.integers(1000, vocab, size=n)] for n in tiny_lengths]
    tiny_out = enc.encode_many(tiny)
    for i in (0, 1, 2047, 2048, 2049, 4095, 4096, 4499):
        assert np.array_equal(tiny_out[i], enc.encode([tiny[i]])[0]), (i, tiny_lengths[i])
    full = [[int(v) for v in rng.integers(1000, vocab, size=128)] for _ in range(64 + 3)]          # 64 x 128 = 8 192 tokens, then 3 more
    full_out = enc.encode_many(full)
    for i in (0, 63, 64, 66):
        assert np.array_equal(full_out[i], enc.encode([full[i]])[0]), i
    # the C entry point with a HOST output buffer (the binding always hands it a device one): both forms, returns when filled
    import ctypes
    for sub in (ids[:5], ids):                                           # one small call; the batch form
        lens_ = np.asarray([len(x) for x in sub], dtype=np.int32)
        flat_ = np.asarray([t for x in sub for t in x], dtype=np.int32)
        host = np.full((len(sub), 768), np.nan, dtype=np.float32)
        rc = enc._lib.icd_encoder_encode_many(enc._h, flat_.ctypes.data, lens_.ctypes.data, len(sub), 0, 1, host.ctypes.data, 0, None)
        assert rc == 0 and np.array_equal(host, many[:len(sub)])
    many_cls = enc.encode_many(ids[:40], pooling="cls", normalize=False)
    assert np.array_equal(many_cls, np.concatenate([enc.encode([x], pooling="cls", normalize=False) for x in ids[:40]]))
    from rag_project_icd10_amd import _native
    with pytest.raises(_native.IcdError, match="tokens"):
        enc.encode_many([[101] * 513])
    assert enc.encode_many([], to_device=False).shape == (0, 768)


@pytest.mark.parametrize("k", [10, 20])
def test_both_encoder_paths_give_the_same_code_lists(k, encoder_corpus, monkeypatch):
    """the 1 000 golden strings through BOTH query paths - encode_query one string per call (the reference's
    shape, services/multi_diagnosis_service.py:152) and encode_query_batch - searched over the encoder-made 40 474-row corpus.
    Canonical batch path (the default): the vectors are bit-identical, so codes AND scores of all 1 000 strings are. Fast path
    (ICD_EMBEDDING_BATCH=fast, packed split-bf16 GEMMs): vectors within 1e-5; every string whose code list differs is
    explained by a near-tie - the canonical scores of the two hits that swapped differ by less than BOUND - and counted."""
    import sys
    import torch
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    b, ms, es, corpus, levels, strings = encoder_corpus
    BOUND = 1e-5   # = north_star's score tolerance: two hits closer than this are a tie at the precision the scores are specified to
    single = np.stack([es.encode_query(t) for t in strings])
    batched = es.encode_query_batch(strings, to_device=True)
    assert np.array_equal(batched.cpu().numpy(), single)
    codes = [r["code"] for r in ms.client.records]
    adj_b, raw_b, ids_b, _ = (x.cpu().numpy() for x in ms.search_batch(batched, top_k=k))
    one_by_one = [ms.search(single[i], top_k=k) for i in range(len(strings))]
    for i, hits in enumerate(one_by_one):   # canonical: codes and scores, every string
        assert [h["code"] for h in hits] == [codes[j] for j in ids_b[i]], strings[i]
        assert [h["score"] for h in hits] == list(adj_b[i]) and [h["original_score"] for h in hits] == [float(x) for x in raw_b[i]]
    # the fast path against it
    fast = es.encode_query_batch(strings, to_device=True, fast=True)
    d_vec = float(np.max(np.abs(fast.cpu().numpy() - single)))
    assert 0.0 < d_vec <= 1e-5, d_vec                                   # (another arithmetic: not bit-identical, inside the tolerance)
    adj_f, raw_f, ids_f, _ = (x.cpu().numpy() for x in ms.search_batch(fast, top_k=k))
    assert float(np.max(np.abs(raw_f.astype(np.float64) - raw_b.astype(np.float64))[ids_f == ids_b], initial=0.0)) <= 1e-5
    differ = [i for i in range(len(strings)) if not np.array_equal(ids_f[i], ids_b[i])]
    w = np.where(levels == 1, 1.2, np.where(levels == 3, 0.8, 1.0))
    if differ:
        os_, oi = orc.flat_ip_topk(corpus, single[differ], k + 1)       # canonical raw scores of the top k + 1, descending
        for r, i in enumerate(differ):
            raw_gaps = os_[r][:-1].astype(np.float64) - os_[r][1:].astype(np.float64)
            adj_sorted = np.sort(os_[r][:k].astype(np.float64) * w[oi[r][:k]])[::-1]
            adj_gaps = adj_sorted[:-1] - adj_sorted[1:]
            # a different list needs two hits to change places: in the raw order (membership: the k-th against the k + 1-th, or any
            # pair inside) or in the reweighted order of the k members - and those two were closer than the arithmetic's reach
            assert min(float(raw_gaps.min()), float(adj_gaps.min())) <= BOUND, (strings[i], raw_gaps.min(), adj_gaps.min())
    print(f"k={k}: canonical batch == one string per call on {len(strings)}/{len(strings)} strings (codes and scores, bit for bit); "
          f"fast (split-bf16) path: max |d vector| {d_vec:.2e}, {len(strings) - len(differ)}/{len(strings)} identical code lists, "
          f"{len(differ)} near-tie swaps (canonical gap <= {BOUND:g})")
    assert len(differ) <= 150   # (measured: 11 at k = 10, 50 at k = 20 on the synthetic-weight corpus, whose families are tight)


def test_split_bf16x3_kernel_makes_the_documented_operand():
    """icd_split_bf16x3 alone (through the C ABI): [hi | hi | lo | 1 1 0 ... 0] with hi = bf16(x) rounded to nearest even and
    lo = bf16(x - hi), bit for bit what torch's casts give; with act = 1 the input goes through erf-GELU first (BertIntermediate,
    the activation inside SentenceTransformer.encode: reference services/embedding_service.py:97-102) within one bf16 ulp of
    torch's; hi + lo reproduces x to 2^-16 relative; a strided input view works"""
    import torch
    from rag_project_icd10_amd import _native
    torch.manual_seed(11)
    for rows, cols in ((1, 64), (37, 768), (1000, 3072)):
        big = torch.randn((rows, cols + 16), device="cuda") * torch.logspace(-3, 2, cols + 16, device="cuda")
        x = big[:, :cols]                                      # a view with a wider row stride
        out = _native.split_bf16x3(x)
        assert out.shape == (rows, 3 * cols + _native.SPLIT_TAIL) and out.dtype == torch.bfloat16
        hi = x.to(torch.bfloat16)
        lo = (x - hi.float()).to(torch.bfloat16)
        assert torch.equal(out[:, :cols].view(torch.int16), hi.view(torch.int16))
        assert torch.equal(out[:, cols:2 * cols].view(torch.int16), hi.view(torch.int16))
        assert torch.equal(out[:, 2 * cols:3 * cols].view(torch.int16), lo.view(torch.int16))
        tail = out[:, 3 * cols:].float()
        assert bool((tail[:, :2] == 1).all()) and bool((tail[:, 2:] == 0).all())
        rel = ((hi.float() + lo.float() - x).abs() / x.abs().clamp(min=1e-30)).max().item()
        assert rel <= 2.0 ** -16
        g = _native.split_bf16x3(x, gelu=True)
        want = torch.nn.functional.gelu(x)
        got = g[:, :cols].float() + g[:, 2 * cols:3 * cols].float()
        assert ((got - want).abs() <= 2.0 ** -15 * want.abs() + 1e-30).all()


@pytest.mark.parametrize("family", ["bert", "xlm-roberta"])
def test_small_input_encoder_at_hidden_1024(family):
    """the reference's CODE-default encoder is 1024-d (multilingual-e5-large: XLM-R large, services/embedding_service.py:26): the
    small-input forward at hidden 1024 / 16 heads / inter 4096 (256 columns of K per wave), BERT and XLM-R flavours (positions
    from padding_idx + 1, one token type), two seeded layers, against transformers' forward on the GPU"""
    import torch
    from transformers import BertConfig, BertModel, XLMRobertaConfig, XLMRobertaModel
    from rag_project_icd10_amd import _native
    torch.manual_seed(11)
    if family == "bert":
        model = BertModel(BertConfig(vocab_size=3000, hidden_size=1024, num_hidden_layers=2, num_attention_heads=16, intermediate_size=4096,
                                     max_position_embeddings=512), add_pooling_layer=False)
    else:
        model = XLMRobertaModel(XLMRobertaConfig(vocab_size=3000, hidden_size=1024, num_hidden_layers=2, num_attention_heads=16, intermediate_size=4096,
                                                 max_position_embeddings=514, type_vocab_size=1, pad_token_id=1, layer_norm_eps=1e-5), add_pooling_layer=False)
    model = model.eval().cuda()
    assert _native.SmallEncoder.supported(model)
    enc = _native.SmallEncoder(model)
    rng = np.random.default_rng(3)
    try:
        for lengths in ([1], [9], [16], [17], [40], [5, 11, 30], [3] * 20, [100, 60, 90], [256]):
            ids = [[int(v) for v in rng.integers(5, 3000, size=n)] for n in lengths]
            width = max(lengths)
            tok = torch.full((len(ids), width), 1, dtype=torch.long)
            mask = torch.zeros((len(ids), width), dtype=torch.long)
            for r, x in enumerate(ids):
                tok[r, :len(x)] = torch.tensor(x)
                mask[r, :len(x)] = 1
            with torch.no_grad():
                hidden = model(input_ids=tok.cuda(), attention_mask=mask.cuda()).last_hidden_state
            m = mask.cuda().unsqueeze(-1).float()
            want = torch.nn.functional.normalize((hidden * m).sum(1) / m.sum(1), p=2, dim=1).cpu().numpy()
            want_rows = torch.cat([hidden[r, :len(x)] for r, x in enumerate(ids)], 0).cpu().numpy()
            got, rows = enc.encode(ids, pooling="mean", normalize=True, hidden=True)
            assert got.shape == (len(ids), 1024)
            assert np.max(np.abs(got - want)) <= TOL, (family, lengths)
            assert np.max(np.abs(rows.cpu().numpy() - want_rows)) <= 5e-5, (family, lengths)
        # the batch form at hidden 1024 (csrc/encoder_big.hpp, 256-column K pieces): 90 sequences = 2 000+ tokens in one list, every
        # row bit for bit what the sequence gets alone
        lengths = [int(x) for x in rng.integers(1, 60, 88)] + [300, 1]
        ids = [[int(v) for v in rng.integers(5, 3000, size=n)] for n in lengths]
        many = enc.encode_many(ids)
        for i in (0, 1, 17, 45, 87, 88, 89):
            assert np.array_equal(many[i], enc.encode([ids[i]])[0]), (family, i, lengths[i])
    finally:
        enc.close()


def test_small_input_encoder_from_several_threads(services):
    """FastAPI serves /embed and /query from a thread pool: calls on one handle are serialised by the library (its descriptor and
    result blocks are per handle); four threads x 60 calls of different strings, host and device outputs mixed, give what one
    thread gives"""
    import threading
    import torch
    gpu, _ = services
    texts = _strings()[:24]
    want = [gpu.encode_query(t) for t in texts]
    errors = []

    def work(seed):
        try:
            rng = np.random.default_rng(seed)
            for _ in range(60):
                j = int(rng.integers(0, len(texts)))
                if rng.random() < 0.5:
                    got = gpu.encode_query(texts[j])
                else:
                    got = gpu.encode_query_batch([texts[j]], to_device=True)
                    torch.cuda.synchronize()
                    got = got[0].cpu().numpy()
                if not np.array_equal(got, want[j]):
                    errors.append((seed, j, float(np.max(np.abs(got - want[j])))))
        except Exception as exc:   # pragma: no cover
            errors.append(repr(exc))
    threads = [threading.Thread(target=work, args=(s,)) for s in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors[:5]


@pytest.mark.parametrize("arithmetic", ["fp32", "bf16x3"])
def test_both_arithmetics_of_the_canonical_encoder(arithmetic):
    """The hand-written encoder's GEMMs in either arithmetic (icd_encoder_desc.arithmetic; ICD_ENCODER_ARITH): fp32-input MFMAs
    (exact products: ~1e-7 off the framework's fp32 forward) or the split-bf16 form, the default (three bf16 MFMAs per 32
    k-values: ~1e-6 off, 1 / 5 of the matrix time). Under EACH the batch form returns the one-string call's bits - the property
    the search relies on (reference: tools/build_database.py:217-222 = services/embedding_service.py:117-120) - and both stay
    inside the 1e-5 tolerance against the framework's forward of the same weights."""
    import torch
    from transformers import BertConfig, BertModel
    from rag_project_icd10_amd import _native
    torch.manual_seed(21)
    model = BertModel(BertConfig(vocab_size=3000, hidden_size=768, num_hidden_layers=3, num_attention_heads=12, intermediate_size=3072,
                                 max_position_embeddings=512), add_pooling_layer=False).eval().cuda()
    enc = _native.SmallEncoder(model, arithmetic=arithmetic)
    assert enc.arithmetic == arithmetic
    rng = np.random.default_rng(8)
    lengths = [int(x) for x in rng.integers(1, 70, 120)] + [512, 300, 1]
    ids = [[int(v) for v in rng.integers(5, 3000, size=n)] for n in lengths]
    try:
        many = enc.encode_many(ids)                                     # ~4 500 tokens: the batch form
        worst = 0.0
        for i in list(range(0, len(ids), 7)) + [len(ids) - 3, len(ids) - 2, len(ids) - 1]:
            one = enc.encode([ids[i]])[0]
            assert np.array_equal(many[i], one), (arithmetic, i, lengths[i])
            tok = torch.tensor([ids[i]], dtype=torch.long).cuda()
            with torch.no_grad():
                hidden = model(input_ids=tok, attention_mask=torch.ones_like(tok)).last_hidden_state
            want = torch.nn.functional.normalize(hidden.mean(1), p=2, dim=1)[0].cpu().numpy()
            worst = max(worst, float(np.max(np.abs(one - want))))
        print(f"arithmetic {arithmetic}: max |d vector| vs the framework's fp32 forward {worst:.2e} (three layers)")
        assert worst <= (5e-7 if arithmetic == "fp32" else TOL)
    finally:
        enc.close()


@pytest.mark.gpu
def test_calls_on_different_streams_share_the_handles_workspace_in_order():
    """The activations belong to the handle, not to a stream. A device-output call returns without waiting; the next call may
    come on ANOTHER stream (a serving process: the build thread's batches beside a request's query) and must not start on the
    workspace before the first has left it (csrc/icd_encoder.hpp ev_tail: a device-side wait, the host never blocks). Every
    combination of the two call forms on two streams, results against the same calls made one stream, one at a time."""
    import torch
    from transformers import BertConfig, BertModel
    from rag_project_icd10_amd import _native
    torch.manual_seed(5)
    model = BertModel(BertConfig(vocab_size=2000, hidden_size=768, num_hidden_layers=4, num_attention_heads=12, intermediate_size=3072,
                                 max_position_embeddings=512), add_pooling_layer=False).eval().cuda()
    enc = _native.SmallEncoder(model)
    rng = np.random.default_rng(2)
    big = [[int(v) for v in rng.integers(5, 2000, size=int(n))] for n in rng.integers(20, 120, 200)]     # ~14 000 tokens: two passes of the batch form
    mid = [[int(v) for v in rng.integers(5, 2000, size=int(n))] for n in rng.integers(5, 40, 30)]        # several small calls from the ring
    one = [[int(v) for v in rng.integers(5, 2000, size=33)]]
    try:
        want = {"big": enc.encode_many(big), "mid": enc.encode_many(mid), "one": enc.encode(one)}
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        calls = {"big": lambda: enc.encode_many(big, to_device=True), "mid": lambda: enc.encode_many(mid, to_device=True),
                 "one": lambda: enc.encode(one, to_device=True)}
        for first in ("big", "mid", "one"):
            for second in ("big", "mid", "one"):
                for _ in range(3):
                    with torch.cuda.stream(s1):
                        a = calls[first]()
                    with torch.cuda.stream(s2):
                        b = calls[second]()
                    torch.cuda.synchronize()
                    assert np.array_equal(a.cpu().numpy(), want[first]), (first, second, "first call")
                    assert np.array_equal(b.cpu().numpy(), want[second]), (first, second, "second call")
        # ... and a host-output call behind a device-output one on another stream
        with torch.cuda.stream(s1):
            a = enc.encode_many(big, to_device=True)
        with torch.cuda.stream(s2):
            b = enc.encode(one)
        torch.cuda.synchronize()
        assert np.array_equal(a.cpu().numpy(), want["big"]) and np.array_equal(b, want["one"])
    finally:
        enc.close()
