/*
 * icd_oracle.c — TEST INFRASTRUCTURE ONLY (CPU restatement of the reference's search arithmetic).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product path (rag_project_icd10_amd/ + libicdsearch.so) never does.
 *
 * What it restates (reference = /root/reference, read-only, never copied):
 *   - services/milvus_service.py:280-285   client.search(data=[q], limit=top_k) on a FLAT / IP index
 *     (index_type/metric_type hard-coded at services/milvus_service.py:33-34,189-194): exact inner
 *     product of one query against every row, k largest, descending.
 *   - services/milvus_service.py:290-295   adjusted = float(base_score * level_weight)   (Python double)
 *   - services/milvus_service.py:550-558   level weights {1:1.2, 2:1.0, 3:0.8}, default 1.0
 *   - services/milvus_service.py:314       candidates.sort(key=score, reverse=True)      (stable)
 *
 * The engine below the call site (pymilvus==2.5.10 -> Milvus Lite -> knowhere FLAT) is a third-party
 * dependency that is not vendored in the reference and not installable here; its published algorithm
 * is "brute-force inner product in fp32, return the k best". Its fp32 summation ORDER is unspecified,
 * so this oracle pins one canonical order:
 *       score = fmaf chain over d = 0..dim-1, starting from +0.0f, one rounding per step
 * and the tie-break (score desc, row id asc). Parity of the summation order against Milvus itself is
 * UNPINNED (difference <= ~2e-7 for unit vectors); everything else is pinned by definition.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ICD_ORACLE_ABI 1

int icd_oracle_abi(void) { return ICD_ORACLE_ABI; }

/* canonical score of one (query,row) pair */
static inline float chain_score(const float *q, const float *c, int dim) {
    float acc = 0.0f;
    for (int d = 0; d < dim; ++d) acc = __builtin_fmaf(q[d], c[d], acc);
    return acc;
}

/* scores of one query against rows [0,n): 8 independent chains in flight for ILP; each chain is
 * still the strict d-ascending fmaf chain. */
void icd_oracle_scores(const float *q, const float *corpus, int64_t n, int dim, float *out) {
    int64_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const float *c0 = corpus + (size_t)i * dim;
        float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
        for (int d = 0; d < dim; ++d) {
            const float qd = q[d];
            a0 = __builtin_fmaf(qd, c0[d], a0);
            a1 = __builtin_fmaf(qd, c0[(size_t)dim + d], a1);
            a2 = __builtin_fmaf(qd, c0[(size_t)2 * dim + d], a2);
            a3 = __builtin_fmaf(qd, c0[(size_t)3 * dim + d], a3);
            a4 = __builtin_fmaf(qd, c0[(size_t)4 * dim + d], a4);
            a5 = __builtin_fmaf(qd, c0[(size_t)5 * dim + d], a5);
            a6 = __builtin_fmaf(qd, c0[(size_t)6 * dim + d], a6);
            a7 = __builtin_fmaf(qd, c0[(size_t)7 * dim + d], a7);
        }
        out[i] = a0; out[i + 1] = a1; out[i + 2] = a2; out[i + 3] = a3;
        out[i + 4] = a4; out[i + 5] = a5; out[i + 6] = a6; out[i + 7] = a7;
    }
    for (; i < n; ++i) out[i] = chain_score(q, corpus + (size_t)i * dim, dim);
}

/* a is strictly better than b: higher score first, then lower row id */
static inline int better(float sa, int64_t ia, float sb, int64_t ib) {
    return (sa > sb) || (sa == sb && ia < ib);
}

/* top-k of a score vector. NaN scores are skipped. Output sorted best-first; unused slots get
 * score = -INFINITY, id = -1. Returns the number of valid hits. */
int icd_oracle_select_topk(const float *scores, int64_t n, int64_t id_base, int k,
                           float *out_scores, int64_t *out_ids) {
    int cnt = 0;
    for (int j = 0; j < k; ++j) { out_scores[j] = -INFINITY; out_ids[j] = -1; }
    for (int64_t i = 0; i < n; ++i) {
        const float s = scores[i];
        if (s != s) continue;
        const int64_t id = id_base + i;
        if (cnt == k && !better(s, id, out_scores[k - 1], out_ids[k - 1])) continue;
        int pos = (cnt < k) ? cnt : k - 1;
        while (pos > 0 && better(s, id, out_scores[pos - 1], out_ids[pos - 1])) {
            out_scores[pos] = out_scores[pos - 1];
            out_ids[pos] = out_ids[pos - 1];
            --pos;
        }
        out_scores[pos] = s;
        out_ids[pos] = id;
        if (cnt < k) ++cnt;
    }
    return cnt;
}

/* FLAT / IP search: nq queries, each against all n rows; out_* are [nq][k]. id_base is added to the
 * row index (row-sharded corpora). nthreads <= 0 -> all cores. Returns 0, or -1 on bad arguments /
 * allocation failure. */
int icd_oracle_flat_ip_topk(const float *corpus, int64_t n, int dim, const float *queries, int64_t nq,
                            int k, int64_t id_base, int nthreads, float *out_scores, int64_t *out_ids) {
    if (!corpus || !queries || !out_scores || !out_ids || n < 0 || nq < 0 || dim <= 0 || k <= 0) return -1;
    int fail = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel
    {
        float *buf = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
        if (!buf) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for schedule(dynamic, 1)
            for (int64_t qi = 0; qi < nq; ++qi) {
                icd_oracle_scores(queries + (size_t)qi * dim, corpus, n, dim, buf);
                icd_oracle_select_topk(buf, n, id_base, k, out_scores + (size_t)qi * k, out_ids + (size_t)qi * k);
            }
            free(buf);
        }
    }
    return fail ? -1 : 0;
}

/* services/milvus_service.py:550-558 */
double icd_oracle_level_weight(int32_t level) {
    switch (level) {
        case 1: return 1.2;
        case 2: return 1.0;
        case 3: return 0.8;
        default: return 1.0;
    }
}

/* services/milvus_service.py:290-295,314 applied to one query's raw hits (best-first):
 * adj = (double)raw * w[level]; stable sort by adj descending. hit_levels[j] is the level of hit j.
 * Invalid hits (id < 0) stay at the end. Outputs may not alias inputs. */
void icd_oracle_reweight_one(const float *raw, const int64_t *ids, const int32_t *hit_levels, int k,
                             double *out_adj, float *out_raw, int64_t *out_ids, int32_t *out_levels) {
    int m = 0;
    for (int j = 0; j < k; ++j) {
        if (ids[j] < 0) continue;
        const double adj = (double)raw[j] * icd_oracle_level_weight(hit_levels[j]);
        int pos = m;
        /* stable: move past strictly smaller keys only */
        while (pos > 0 && out_adj[pos - 1] < adj) {
            out_adj[pos] = out_adj[pos - 1]; out_raw[pos] = out_raw[pos - 1];
            out_ids[pos] = out_ids[pos - 1]; out_levels[pos] = out_levels[pos - 1];
            --pos;
        }
        out_adj[pos] = adj; out_raw[pos] = raw[j]; out_ids[pos] = ids[j]; out_levels[pos] = hit_levels[j];
        ++m;
    }
    for (int j = m; j < k; ++j) { out_adj[j] = -INFINITY; out_raw[j] = -INFINITY; out_ids[j] = -1; out_levels[j] = 0; }
}

/* batch form: levels is the per-row level table of the (shard of the) corpus, id_base its first id */
void icd_oracle_reweight(const float *raw, const int64_t *ids, const int32_t *levels, int64_t id_base,
                         int64_t nq, int k, double *out_adj, float *out_raw, int64_t *out_ids,
                         int32_t *out_levels) {
    int32_t *hl = (int32_t *)malloc(sizeof(int32_t) * (size_t)k);
    for (int64_t qi = 0; qi < nq; ++qi) {
        const size_t o = (size_t)qi * k;
        for (int j = 0; j < k; ++j) hl[j] = ids[o + j] >= 0 ? levels[ids[o + j] - id_base] : 0;
        icd_oracle_reweight_one(raw + o, ids + o, hl, k, out_adj + o, out_raw + o, out_ids + o, out_levels + o);
    }
    free(hl);
}

/* merge G best-first lists per query (row-sharded search): [G][nq][k] -> [nq][k] */
void icd_oracle_merge(const float *scores, const int64_t *ids, int G, int64_t nq, int k,
                      float *out_scores, int64_t *out_ids) {
    for (int64_t qi = 0; qi < nq; ++qi) {
        float *os = out_scores + (size_t)qi * k;
        int64_t *oi = out_ids + (size_t)qi * k;
        int cnt = 0;
        for (int j = 0; j < k; ++j) { os[j] = -INFINITY; oi[j] = -1; }
        for (int g = 0; g < G; ++g) {
            const size_t o = ((size_t)g * nq + qi) * k;
            for (int j = 0; j < k; ++j) {
                const float s = scores[o + j]; const int64_t id = ids[o + j];
                if (id < 0 || s != s) continue;
                if (cnt == k && !better(s, id, os[k - 1], oi[k - 1])) continue;
                int pos = (cnt < k) ? cnt : k - 1;
                while (pos > 0 && better(s, id, os[pos - 1], oi[pos - 1])) { os[pos] = os[pos - 1]; oi[pos] = oi[pos - 1]; --pos; }
                os[pos] = s; oi[pos] = id;
                if (cnt < k) ++cnt;
            }
        }
    }
}
