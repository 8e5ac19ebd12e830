#!/usr/bin/env python3
"""Randomised shape sweep of the HIP search path against the CPU oracle (GPU box only; not part of pytest).

    python scripts/gpu_fuzz.py [--cases 40] [--seed 7]

Every case draws (rows, queries, dim, k, data kind, mode, id_base), searches through the C ABI and compares ids (exact),
scores (bit-identical) and the reweighted outputs with oracle/ on a query sample. Data kinds: gaussian unit rows,
clustered rows (near ties: forces the exact fallback), rows with exact duplicates, a batch whose fp16 image overflows,
corpora and batches scaled far below fp16's normal range ("tiny"), corpora of tight families of near-identical rows in code
order with large batches ("family": the first coarse pass certifies little, the second coarse pass / the wide-window retry
must; the searches of one index alternate so that the armed, disarmed and wide-mode states all occur), anisotropic rows
("aniso": a large common component at several strengths - the fp16 image is centred above a share of 0.25 - also scaled as a
whole, with rows of other magnitudes mixed in, and with duplicates).
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def rows(rng, n, dim, kind):
    if kind == "clustered":
        cent = rng.standard_normal((32, dim)).astype(np.float32)
        x = cent[rng.integers(0, 32, n)] + 0.05 * rng.standard_normal((n, dim)).astype(np.float32)
    else:
        x = rng.standard_normal((n, dim)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    if kind == "family":      # families of ~124 near-identical rows, in code order (cosine ~0.99 within a family)
        per = 124
        cent = rng.standard_normal(((n + per - 1) // per, dim)).astype(np.float32)
        x = np.repeat(cent, per, axis=0)[:n] + 0.1 * rng.standard_normal((n, dim)).astype(np.float32)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
    if kind == "aniso":       # normalise(mu0 + s e / sqrt(dim)): mean pairwise cosine 1 / (1 + s^2)
        mu0 = np.random.default_rng(5).standard_normal(dim).astype(np.float32)
        mu0 /= np.linalg.norm(mu0)
        sdev = float(rng.choice([0.07, 0.2, 0.5, 1.0, 2.5]))
        x = mu0[None, :] + (sdev / np.sqrt(dim)) * rng.standard_normal((n, dim)).astype(np.float32)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
    if kind == "dups" and n > 20:
        src = rng.integers(0, n, n // 10)
        dst = rng.integers(0, n, n // 10)
        x[dst] = x[src]
    return np.ascontiguousarray(x, dtype=np.float32)


def fuzz_encoder(args):
    """--focus encoder: the small-input sentence encoder (csrc/encoder_small.hpp through icd_encoder_encode) against
    transformers' padded fp32 forward of the same seeded BERT-base weights on the GPU: random numbers of sequences (1 ... 64) and
    lengths (1 ... 512 packed tokens: every token bucket, sequences that straddle the 16-token tiles), mean and [CLS] pooling,
    normalised or not, pooled rows and the last hidden state of every token; tolerance 1e-5 on unit rows. Every fourth case also
    runs a LIST of a few thousand tokens through icd_encoder_encode_many (the batch form, csrc/encoder_big.hpp) with the case's
    sequences scattered in it: their rows must be bit for bit what the small-input form gave them."""
    import torch
    os.environ.setdefault("EMBEDDING_MODEL_NAME", "shibing624/text2vec-base-chinese")
    from rag_project_icd10_amd.services.embedding_service import EmbeddingService
    es = EmbeddingService(allow_synthetic=True, device="cuda")
    enc = es._small
    assert enc is not None, "the small-input encoder is not available"
    rng = np.random.default_rng(args.seed)
    vocab = es.model.bert.config.vocab_size
    bad, t0 = 0, time.time()
    for case in range(args.cases):
        nseq = int(rng.choice([1, 1, 1, 2, 3, 5, 8, 13, 32, 47, 64]))
        budget = int(rng.choice([16, 17, 32, 33, 64, 100, 128, 129, 200, 256, 257, 400, 512]))
        budget = max(budget, nseq)
        cuts = np.sort(rng.choice(np.arange(1, budget), size=nseq - 1, replace=False)) if nseq > 1 else np.array([], dtype=np.int64)
        lengths = np.diff(np.concatenate([[0], cuts, [budget]])).astype(int).tolist()
        ids = [[101] + [int(v) for v in rng.integers(1000, vocab, size=n - 2)] + [102] if n >= 2 else [101] for n in lengths]
        pooling = str(rng.choice(["mean", "cls"]))
        width = max(lengths)
        tok = torch.zeros((nseq, width), dtype=torch.long)
        mask = torch.zeros((nseq, width), dtype=torch.long)
        for r, x in enumerate(ids):
            tok[r, :len(x)] = torch.tensor(x)
            mask[r, :len(x)] = 1
        tok, mask = tok.cuda(), mask.cuda()
        with torch.no_grad():
            hidden = es.model.bert(input_ids=tok, attention_mask=mask).last_hidden_state
        m = mask.unsqueeze(-1).float()
        pooled = hidden[:, 0] if pooling == "cls" else (hidden * m).sum(1) / m.sum(1)
        want = torch.nn.functional.normalize(pooled, p=2, dim=1).cpu().numpy()
        want_rows = torch.cat([hidden[r, :len(x)] for r, x in enumerate(ids)], 0).cpu().numpy()
        got, rows_ = enc.encode(ids, pooling=pooling, normalize=True, hidden=True)
        d1, d2 = float(np.max(np.abs(got - want))), float(np.max(np.abs(rows_.cpu().numpy() - want_rows)))
        again = enc.encode(ids, pooling=pooling, normalize=True)
        ok = d1 <= 1e-5 and d2 <= 5e-5 and np.array_equal(again, got)
        if case % 4 == 0:   # the batch form: the case's sequences scattered among fillers, 1 000 ... 20 000 tokens in all
            nfill = int(rng.choice([40, 300, 1200]))
            fill = [[int(v) for v in rng.integers(1000, vocab, size=int(n))] for n in rng.integers(1, 40, nfill)]
            where = np.sort(rng.choice(np.arange(nfill + nseq), size=nseq, replace=False))
            mixed, it, k_ = [], iter(fill), 0
            for pos in range(nfill + nseq):
                if k_ < nseq and pos == where[k_]:
                    mixed.append(ids[k_]); k_ += 1
                else:
                    mixed.append(next(it))
            many = enc.encode_many(mixed, pooling=pooling, normalize=True)
            same = np.array_equal(many[where], got)
            if not same:
                print(f"FAIL case {case}: the batch form differs from the small-input form (lengths {lengths}, {nfill} fillers)", flush=True)
            ok = ok and same
        if not ok:
            bad += 1
            print(f"FAIL case {case}: lengths {lengths} pooling {pooling}: max |d pooled| {d1:.2e}, max |d hidden| {d2:.2e}, replay equal {np.array_equal(again, got)}", flush=True)
    print(f"gpu_fuzz --focus encoder: {args.cases - bad} ok, {bad} failed in {time.time() - t0:.1f} s (seed {args.seed})", flush=True)
    return 1 if bad else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--focus", choices=["", "exact_k", "one_query", "encoder"], default="",
                    help="exact_k: mostly ICD_MODE_EXACT at k = 33 ... 100 on batches the narrow certified lists take (round 5); "
                         "one_query: mostly one or two queries per call at k <= 16 (the single-launch streaming kernel)")
    args = ap.parse_args()
    if args.focus == "encoder":
        sys.exit(fuzz_encoder(args))
    import oracle as orc
    from rag_project_icd10_amd import _native
    from rag_project_icd10_amd._native import MODE_AUTO, MODE_EXACT, IcdIndex
    rng = np.random.default_rng(args.seed)
    bad = 0
    t0 = time.time()
    for case in range(args.cases):
        dim = int(rng.choice([768, 768, 768, 1024, 96, 256]))
        n = int(rng.choice([1, 7, 127, 128, 129, 1000, 4097, 20000, 37000, 100003]))
        nq = int(rng.choice([1, 2, 9, 16, 17, 64, 65, 128, 129, 1000, 3000, 6000]))
        k = int(rng.choice([1, 5, 10, 10, 10, 12, 13, 17, 20, 20, 32, 33, 50, 100]))
        kind = str(rng.choice(["gauss", "gauss", "clustered", "dups", "overflow", "zeros", "scaled", "tiny", "tiny", "family", "family", "aniso", "aniso", "aniso"]))
        if kind == "family":
            dim, n, nq = int(rng.choice([768, 768, 1024])), int(rng.choice([4097, 20000, 37000])), int(rng.choice([1000, 3000, 6000]))
        if kind == "aniso" and rng.random() < 0.7:   # mostly the shapes where the coarse pass and its certificate do the work
            dim, n, nq = int(rng.choice([768, 768, 1024])), int(rng.choice([4097, 20000, 37000])), int(rng.choice([129, 1000, 3000, 6000]))
        mode = MODE_AUTO if rng.random() < 0.8 else MODE_EXACT
        id_base = int(rng.choice([0, 0, 5_000_000_000]))
        if args.focus == "exact_k" and rng.random() < 0.85:
            k, mode = int(rng.integers(33, 101)), MODE_EXACT
            n, nq = int(rng.choice([8192, 8200, 20000, 37000, 100003])), int(rng.choice([65, 129, 1000, 3000]))
        if args.focus == "one_query" and rng.random() < 0.85:
            k, nq = int(rng.integers(1, 17)), int(rng.choice([1, 1, 2]))
        if n * dim > 100003 * 768 or (n >= 100000 and nq > 1000):
            nq = min(nq, 1000)
        corpus = rows(rng, n, dim, "gauss" if kind in ("overflow", "zeros", "scaled", "tiny") else kind)
        queries = rows(rng, nq, dim, "gauss" if kind in ("overflow", "dups", "zeros", "scaled", "tiny") else kind)
        if kind == "family":      # queries near rows of the corpus
            queries = corpus[rng.integers(0, n, nq)] + 0.02 * rng.standard_normal((nq, dim)).astype(np.float32)
            queries = np.ascontiguousarray(queries / np.linalg.norm(queries, axis=1, keepdims=True), dtype=np.float32)
        if kind == "aniso":
            twist = str(rng.choice(["plain", "plain", "scaled", "mixed", "dups"]))
            if twist == "scaled":     # the whole corpus / batch at another magnitude (the mean scales with it)
                corpus *= np.float32(10.0 ** rng.uniform(-20, 3))
                queries *= (10.0 ** rng.uniform(-6, 2, (nq, 1))).astype(np.float32)
            if twist == "mixed":      # a few rows of very different norm or direction among the anisotropic ones
                m = rng.random(n) < 0.02
                corpus[m] *= rng.choice([-3.0, 1e-3, 25.0], (int(m.sum()), 1)).astype(np.float32)
            if twist == "dups" and n > 20:
                src, dst = rng.integers(0, n, n // 10), rng.integers(0, n, n // 10)
                corpus[dst] = corpus[src]
        if kind == "tiny":        # whole corpus / batch far below fp16's normal range (2^-14), or only parts of them
            corpus *= np.float32(10.0 ** rng.uniform(-30, -4))
            if rng.random() < 0.5:
                corpus[rng.random(n) < 0.5] *= np.float32(1e-5)
            queries *= (10.0 ** rng.uniform(-8, 0, (nq, 1))).astype(np.float32)
        if kind == "zeros":       # all-zero queries and rows: every score ties at 0, ids must come out row-ascending
            queries[rng.random(nq) < 0.3] = 0.0
            corpus[rng.random(n) < 0.2] = 0.0
        if kind == "scaled":      # wildly different norms, negative correlations, subnormal-sized components
            corpus *= rng.choice([1e-3, 1.0, 40.0], (n, 1)).astype(np.float32)
            queries *= rng.choice([-7.0, 1e-2, 1.0], (nq, 1)).astype(np.float32)
            corpus[:, : dim // 8] *= 1e-6
        if kind == "overflow":
            queries[rng.random(nq) < 0.5, rng.integers(0, dim)] = 3e5
        r = rng.random(n)
        levels = np.where(r < 0.1243, 1, np.where(r < 0.4234, 2, 3)).astype(np.int32)
        probe = bool(rng.random() < 0.75)                       # the corpus-shape probe of icd_index_create, mostly on
        idx = IcdIndex(corpus, levels, max_nq=nq, max_k=k, id_base=id_base, probe=probe)
        sp = int(rng.choice([1, 1, 1, 2, 0]))                   # adaptive (default) / second pass without the adaptive parts / off
        if sp != 1:
            idx.set_second_pass(sp != 0, adaptive=False)
        s, i = idx.search(queries, k, mode)
        long_run = bool(rng.random() < 0.25) and nq <= 3000 and n <= 37000
        if long_run:              # a long history: the streaming fallback's launches are dropped after 96 clean searches
            for _ in range(110):
                idx.search_reweighted(queries, k, mode)
        if kind == "family":      # more searches on the same index: counters arrive, the second pass disarms / wide mode switches on
            for _ in range(int(rng.integers(0, 6))):
                idx.search(queries, k, mode)
                idx.stats()
        adj, raw, ids, lv = idx.search_reweighted(queries, k, mode)
        st = idx.stats()
        idx.close()
        sample = np.unique(np.concatenate([np.arange(0, nq, max(1, nq // 24)), [nq - 1]]))
        os_, oi = orc.flat_ip_topk(corpus, queries[sample], k, id_base=id_base)
        want = orc.reweight(os_, oi, levels, id_base=id_base)
        ok = (np.array_equal(i[sample], oi) and s[sample].tobytes() == os_.tobytes()
              and np.array_equal(ids[sample], want[2]) and adj[sample].tobytes() == want[0].tobytes()
              and np.array_equal(lv[sample], want[3]))
        bad += 0 if ok else 1
        print(f"{'ok  ' if ok else 'FAIL'} case {case:3d}: n={n} nq={nq} dim={dim} k={k} kind={kind} mode={'auto' if mode == MODE_AUTO else 'exact'} "
              f"id_base={id_base} probe={int(probe)} sp={sp} long={int(long_run)} -> sparse_armed={st['sparse_fallback_armed']} last_mode={st['last_mode']} lists={st['last_chunks']} second_pass={st['last_second_pass']} wide={st['wide_mode']} centered={st['centered']} fallback={st['last_fallback']}", flush=True)
    print(f"gpu_fuzz: {args.cases - bad} ok, {bad} failed in {time.time() - t0:.1f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
