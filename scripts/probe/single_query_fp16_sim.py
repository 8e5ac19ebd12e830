#!/usr/bin/env python3
"""VERDICT r5 item 5, priced before building it: ONE query per call through the 57 MB centred fp16 image instead of the 124 MB fp32
rows. The sweep would halve its bytes (29 -> ~15 us at the measured 4.3 TB/s); what it adds is the certificate's tail in the LAST
work-group of the single launch: every candidate within 2 eps of the k-th best coarse score must be rescored with the canonical
fp32 chain (a 3-KB row gather + a 768-step chain each, 64 rows per pass of its four waves, ~3 us per pass + ranking) before
the outputs can be written. This probe counts those survivors with the library's own error bound (DESIGN.md section 4.2:
eps = 1.2e-3 |q'| rmax' + 2 D 2^-24 |q'| rmax_unc' for a centred image), on
  (a) the Gaussian benchmark corpus (40 474 unit rows), and
  (b) an ENCODER-MADE corpus of the real CSV's shape (DatabaseBuilder over tests/golden/csv_shape.json, synthetic weights),
      queries = golden diagnosis strings through the same encoder,
for k = 5 and 10, 200 queries each. Prints the distribution of survivors and the passes of 64 rows they cost."""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def survivors(corpus, queries, k, torch):
    c = torch.from_numpy(corpus).cuda()
    mu = c.mean(0)
    share = float((mu @ mu) / (c * c).sum(1).mean())
    centred = share >= 0.25
    img = (c - mu) if centred else c
    rmax = float(img.norm(dim=1).max())
    rmax_unc = float(c.norm(dim=1).max())
    c16 = img.half()
    q = torch.from_numpy(queries).cuda()
    coarse = (q.half().float() @ c16.float().T)                      # exact fp16 products, fp32 sums
    qn = q.norm(dim=1)
    eps = 1.2e-3 * qn * rmax + (2 * corpus.shape[1] * 2.0 ** -24 * qn * rmax_unc if centred else 0.0)
    kth = coarse.topk(k, dim=1).values[:, -1]
    cnt = (coarse >= (kth - 2 * eps)[:, None]).sum(1).cpu().numpy()
    return centred, share, cnt


def main():
    import torch
    from bench_build import synth_csv
    rng = np.random.default_rng(1234)
    g = rng.standard_normal((40474, 768), dtype=np.float32)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    gq = np.random.default_rng(4321).standard_normal((200, 768), dtype=np.float32)
    gq /= np.linalg.norm(gq, axis=1, keepdims=True)
    tmp = tempfile.mkdtemp(prefix="icd_sq_")
    os.environ.update({"MILVUS_MODE": "local", "MILVUS_DB_PATH": os.path.join(tmp, "db"), "MILVUS_COLLECTION_NAME": "icd10_sq",
                       "EMBEDDING_MODEL_NAME": "shibing624/text2vec-base-chinese", "ICD_EMBEDDING_ALLOW_SYNTHETIC": "1"})
    from rag_project_icd10_amd.tools.build_database import DatabaseBuilder
    shape = json.load(open(os.path.join(ROOT, "tests", "golden", "csv_shape.json"), encoding="utf-8"))
    csv_path = os.path.join(tmp, "shape.csv")
    synth_csv(csv_path, shape)
    b = DatabaseBuilder()
    assert b.build_full_database(csv_path, rebuild=True)
    enc_corpus = b.milvus_service.client.matrix()
    strings = [l.strip() for l in open(os.path.join(ROOT, "tests", "golden", "diagnosis_strings.txt"), encoding="utf-8") if l.strip()][:200]
    eq = b.embedding_service.encode_query_batch(strings)
    for name, corpus, queries in (("Gaussian unit rows", g, gq), ("encoder-made corpus (synthetic weights), golden diagnosis strings", enc_corpus, eq)):
        for k in (5, 10):
            centred, share, cnt = survivors(corpus, queries, k, torch)
            passes = np.ceil(cnt / 64)
            print(f"{name}, k = {k}: image {'centred' if centred else 'not centred'} (mean share {share:.2f}); candidates within 2 eps of the k-th best: "
                  f"median {int(np.median(cnt))}, p90 {int(np.percentile(cnt, 90))}, max {int(cnt.max())}; passes of 64 rows: median {int(np.median(passes))}, "
                  f"p90 {int(np.percentile(passes, 90))}; queries with more than 256 survivors (beyond a last work-group's reach): {int((cnt > 256).sum())} of {len(cnt)}", flush=True)


if __name__ == "__main__":
    main()
