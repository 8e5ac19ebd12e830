/*
 * icd_search.h — C ABI of libicdsearch.so: exact inner-product (FLAT / IP) top-k search over a dense
 * fp32 corpus on one MI355X (gfx950), with the ICD level reweight of the reference fused in.
 *
 * This is the drop-in boundary for the reference's vector-engine calls. The reference has no FFI of
 * its own (pure Python); each entry point below names the reference call it
 * replaces (paths relative to the reference repository root):
 *
 *   icd_index_create            <- MilvusClient.create_collection + insert + load_collection
 *                                  services/milvus_service.py:163-206 (schema, FLAT/IP index),
 *                                  :208-269 (insert_records), :137-161 (load to memory)
 *   icd_index_search            <- MilvusClient.search(data=[q], limit=top_k)
 *                                  services/milvus_service.py:280-285
 *   icd_index_search_reweighted <- the same call plus the per-hit level weight and stable re-sort
 *                                  services/milvus_service.py:290-295,314,550-558
 *   icd_index_stats             <- get_collection_stats / get_memory_usage
 *                                  services/milvus_service.py:322-341,498-522
 *   icd_index_destroy           <- release_collection / disconnect
 *                                  services/milvus_service.py:400-424,452-496
 *   icd_merge_topk              <- no counterpart (the reference is single-process); merges the
 *                                  per-shard partial top-k lists of a row-sharded corpus after the
 *                                  RCCL all-gather.
 *
 * Conventions
 *   - All matrices are dense row-major (exactly numpy's C order): corpus [n][dim], queries [nq][dim],
 *     outputs [nq][k].
 *   - Pointers are borrowed for the duration of the call. `*_on_device` says whether a pointer is a
 *     HIP device pointer (1) or host memory (0). Outputs are caller-allocated.
 *   - `stream` is a hipStream_t (NULL = the default stream). Calls with device inputs AND device
 *     outputs only enqueue work (no allocation, no synchronisation: graph-capturable); calls that
 *     touch host buffers synchronise the stream before returning.
 *   - Result order: best first by (score descending, row id ascending). Scores are the canonical
 *     fp32 fmaf chain over d = 0..dim-1, bit-identical to oracle/icd_oracle.c.
 *     Slots beyond the number of valid hits (n < k, NaN scores) hold id = -1, score = -inf.
 *   - Row ids are `id_base + row index` (id_base != 0 for a shard of a larger corpus).
 *   - Every function returns ICD_OK (0) or a negative icd_status; icd_last_error() gives the text.
 *   - A handle is thread-compatible: one search at a time per handle.
 */
#ifndef ICD_SEARCH_H
#define ICD_SEARCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICD_ABI_VERSION 1
#define ICD_MAX_K 128

typedef struct icd_index icd_index;

typedef enum icd_status {
    ICD_OK = 0,
    ICD_ERR_INVALID = -1,     /* bad argument */
    ICD_ERR_HIP = -2,         /* a HIP runtime call failed */
    ICD_ERR_NOMEM = -3,       /* device or host allocation failed */
    ICD_ERR_UNSUPPORTED = -4, /* shape / device not supported (not gfx950, dim, k) */
    ICD_ERR_STATE = -5        /* handle destroyed or not initialised */
} icd_status;

typedef enum icd_mode {
    ICD_MODE_AUTO = 0,  /* fp16-MFMA coarse pass + exact fp32 rescoring, certified per query;
                           uncertified queries re-run on the exact kernel. Results identical to EXACT. */
    ICD_MODE_EXACT = 1  /* fp32-MFMA kernel only */
} icd_mode;

typedef struct icd_stats {
    int64_t n;                  /* rows in this index / shard */
    int32_t dim;
    int32_t device;
    int64_t id_base;
    int64_t bytes_corpus_f32;   /* HBM bytes held */
    int64_t bytes_corpus_f16;
    int64_t bytes_workspace;
    int32_t max_nq;
    int32_t max_k;
    int32_t fast_path;          /* 1 if the fp16 coarse path is usable for this index */
    float   rmax;               /* max row L2 norm */
    /* counters of the most recent search (valid after the stream is synchronised) */
    int64_t last_nq;
    int64_t last_fallback;      /* queries that took the exact fallback in AUTO mode */
    int32_t last_chunks;        /* candidate lists per query (P) of the last call's scoring pass */
    int32_t last_mode;
} icd_stats;

/* per-kernel device time of the most recent search, measured with hipEvents on the search stream.
 * Only recorded while profiling is enabled (icd_index_set_profiling). */
typedef struct icd_profile {
    float ms_prep;      /* query fp32->fp16 + norms */
    float ms_coarse;    /* fp16 MFMA + fused top-k' (dominant kernel) */
    float ms_finalize;  /* merge + certify + exact rescoring + reweight */
    float ms_exact;     /* fp32 MFMA kernel (EXACT mode, or the AUTO fallback) */
    float ms_exact_finalize;
    float ms_total;     /* first event to last event */
} icd_profile;

int icd_abi_version(void);

/* Text of the most recent error on this thread (never NULL). */
const char *icd_last_error(void);

/* Number of HIP devices visible, or a negative icd_status. */
int icd_device_count(void);

/*
 * Build an index over `n` rows of `dim` floats. `levels` (int32[n], ICD hierarchy level of each row:
 * 1, 2 or 3) may be NULL (all rows level 1 -> weight 1.2, the reference default level,
 * services/milvus_service.py:247). The corpus is copied to HBM; the caller's buffer is not retained.
 * max_nq: largest query batch a single search call will receive (workspace is sized for it).
 * max_k:  largest k (<= ICD_MAX_K).
 */
int icd_index_create(const float *corpus, int64_t n, int32_t dim, const int32_t *levels,
                     int64_t id_base, int32_t device, int32_t max_nq, int32_t max_k,
                     int32_t corpus_on_device, icd_index **out);

int icd_index_destroy(icd_index *idx);

/* Raw top-k by inner product. out_scores float[nq][k], out_ids int64[nq][k]. */
int icd_index_search(icd_index *idx, const float *queries, int64_t nq, int32_t k,
                     int32_t queries_on_device, int32_t mode,
                     float *out_scores, int64_t *out_ids, int32_t out_on_device, void *stream);

/*
 * Raw top-k, then adj = (double)score * w[level] and a stable descending re-sort of the k hits
 * (the order MilvusService.search returns). All four outputs are [nq][k] in the re-sorted order:
 * out_adj (double), out_raw (float, the inner product), out_ids, out_levels. Any of out_raw /
 * out_levels may be NULL.
 */
int icd_index_search_reweighted(icd_index *idx, const float *queries, int64_t nq, int32_t k,
                                int32_t queries_on_device, int32_t mode,
                                double *out_adj, float *out_raw, int64_t *out_ids, int32_t *out_levels,
                                int32_t out_on_device, void *stream);

/*
 * Row-sharded search, step 2: merge `G` best-first lists per query (layout [G][nq][k], as produced by
 * all-gathering the outputs of icd_index_search on every shard, together with the level of every
 * hit) into the global top-k, then reweight + stable re-sort as above. Device pointers only;
 * enqueues on `stream`. `device` selects the GPU.
 */
int icd_merge_topk(int32_t device, const float *scores, const int64_t *ids, const int32_t *levels,
                   int32_t G, int64_t nq, int32_t k,
                   double *out_adj, float *out_raw, int64_t *out_ids, int32_t *out_levels, void *stream);

/* Level of each hit id (device pointers, [count]); ids < 0 give level 0. Used by the row-sharded
 * path to attach levels to the raw hits before the all-gather. */
int icd_index_lookup_levels(icd_index *idx, const int64_t *ids, int64_t count, int32_t *out_levels,
                            void *stream);

/*
 * Hierarchical rescoring of a batch of hit lists, device side: the string-free arithmetic of
 * HierarchicalSimilarityService.batch_calculate_similarities (services/hierarchical_similarity_service.py:475-518,
 * 243-291, 575) and of the uncertainty boost (services/uncertainty_diagnosis_service.py:190-238) for hits shaped as
 * MilvusService.search returns them, for nq queries at once. IEEE double in the reference's evaluation order:
 * bit-identical to the Python doubles.
 *   adj, ids        [nq][k] outputs of icd_index_search_reweighted (device pointers), k <= 128
 *   row_tags        [n_rows] one byte per corpus row: bits 0-3 index of the code's first letter in the chapter table
 *                   A,B,C,E,I,J,K,N,S (15 = none), bit 7 = the code matches \.9\d*$
 *   q_params        [nq][12] per query, computed from its text on the host: uncertainty weight (0 = none), context
 *                   relevance, exact-match flag, the nine chapter boosts
 *   weights         [7] factor weights hierarchy/entity/coherence/category/context, the coherence value, the level term
 *   outputs         [nq][k] in the final order (enhanced score descending, stable): index of the hit in search order,
 *                   enhanced score, the record's score after the uncertainty boost, the vector_similarity and
 *                   hierarchy_boost factors, the applied uncertainty boost
 */
int icd_hier_rescore(int32_t device, const double *adj, const int64_t *ids, int64_t nq, int32_t k, int64_t id_base,
                     int64_t n_rows, const uint8_t *row_tags, const double *q_params, const double *weights,
                     int32_t *out_order, double *out_enhanced, double *out_score, double *out_vs, double *out_hb,
                     double *out_boost, void *stream);

/* SURVEY.md row N3, score statistics. Replaces np.mean / np.std / np.var / max over the candidates' scores in
 * MultiDimensionalConfidenceService._assess_model_uncertainty (services/multidimensional_confidence_service.py:936-963)
 * and ._calculate_prediction_variance (:1087-1099), for nq queries at once, bit-identical to numpy's float64 results.
 *   scores   [nq][k] device doubles (icd_hier_rescore's out_enhanced, or out_adj of a search), k <= 128
 *   order    nullable [nq][k] device: entry j of a query exists iff order[j] >= 0 (icd_hier_rescore's out_order)
 *   use      statistics over the first min(use, existing) entries of every query (the request's top_k)
 *   out      [nq][6] device doubles: mean, std, var, max, model_uncertainty, prediction_variance
 */
int icd_score_stats(int32_t device, const double *scores, const int32_t *order, int64_t nq, int32_t k, int32_t use,
                    double *out, void *stream);

/* SURVEY.md row N3, semantic coherence. Replaces sklearn cosine_similarity([q], [c])[0][0] in
 * MultiDimensionalConfidenceService._calculate_semantic_factors (:273-280) for nq query vectors at once: rows
 * normalised in double, then their dot product (equal to sklearn's to 1e-14; the summation order differs).
 *   x        [nq][dim] device fp32 query vectors
 *   y        device fp32: [nq][dim] with y_stride = dim, or ONE row shared by every query with y_stride = 0 (the live
 *            /query path embeds the empty string as "the candidate": services/multi_diagnosis_service.py:178-186 hands
 *            the confidence service records without 'preferred_zh')
 *   out      [nq] device doubles
 */
int icd_cosine_rows(int32_t device, const float *x, const float *y, int64_t y_stride, int64_t nq, int32_t dim,
                    double *out, void *stream);

/* (waits for the device: last_fallback is read from the device when asked for, not copied back by every search) */
int icd_index_stats(icd_index *idx, icd_stats *out);

/* Tuning knob / test hook (0 = automatic): aim for about `chunks` candidate lists per query in the coarse pass
 * (the automatic choice gives every CU the same number of 128 x 128 tiles). */
int icd_index_set_chunks(icd_index *idx, int32_t chunks);

/* Test switch, process-wide, read by icd_index_create: 0 keeps the fp16 corpus copy in row order instead of the
 * golden-ratio permutation (results are identical; only the share of certified queries changes). Default 1. */
int icd_debug_set_permute(int32_t enabled);

/* Diagnostic builds only (make ABLATE=1, env ICD_FLAT_VAR with bit 1024): per-wave cycle sums of the coarse kernel,
 * [work-group][wave][8] = {LDS-DMA wait, barrier, stage body, fused select, tiles, ...}. */
int icd_index_debug_counters(icd_index *idx, unsigned long long *out, int32_t count);

int icd_index_set_profiling(icd_index *idx, int32_t enabled);
/* Synchronises the events of the most recent search and fills `out`. */
int icd_index_last_profile(icd_index *idx, icd_profile *out);
/* Mean per-kernel times over the profiled searches since the previous summary (at most the last 128),
 * and how many were averaged; resets the window. The events are recorded on the search stream, so this
 * measures the kernels inside a timed region without a synchronisation per search. */
int icd_index_profile_summary(icd_index *idx, icd_profile *out_mean, int32_t *out_count);

#ifdef __cplusplus
}
#endif
#endif /* ICD_SEARCH_H */
