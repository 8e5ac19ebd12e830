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
 *   icd_group_*                 <- no counterpart: one process per GPU, an RCCL communicator owned by the group; the
 *                                  row-sharded search (local top-k -> ncclAllGather -> merge + reweight) and the
 *                                  query-sharded one (SURVEY.md section 8b "suggested C ABI", 8e) in one call, on one stream.
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
 *   - Threads: calls on one handle are serialised by the library (a mutex guards the handle's host-side state: plans,
 *     adaptive counters, profiling ring), so concurrent callers cannot corrupt it. The handle's DEVICE workspace is
 *     reused by every search: enqueue the searches of one handle on ONE stream (or order the streams yourself); two
 *     searches of one handle executing concurrently on different streams is a data race on the device. Different
 *     handles are independent.
 */
#ifndef ICD_SEARCH_H
#define ICD_SEARCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICD_ABI_VERSION 6   /* 6: icd_encoder_encode_many; the process-wide icd_debug_set_* switches became per-index options (icd_index_create flags, icd_index_set_option); ICD_ENCODER_MAX_TOKENS 512, ICD_ENCODER_MAX_SEQS 64 (round 6). 5: icd_unpack_query_slices, icd_debug_set_stream_one, icd_debug_set_pacing, icd_debug_set_exact_narrow, icd_debug_set_host_one, icd_split_bf16x3, icd_encoder_create / _encode / _destroy, icd_pack_winners (round 5). 4: icd_debug_set_family_order, icd_debug_set_center, icd_stats.centered / mean_share appended, icd_group_prepare / icd_group_connect (round 4). 3: icd_stats.sparse_fallback_armed appended, icd_debug_set_create_probe, icd_packed_attention (round 3). 2: + icd_hier_rescore, icd_score_stats, icd_cosine_rows, icd_debug_set_permute (round 2), the group entry points (round 3) */
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
    int64_t last_second_pass;   /* queries the first coarse pass could not certify and the second one took (AUTO mode) */
    int32_t last_second_pass_lists; /* candidate lists per query of that second pass (0: it was not part of the call) */
    int32_t second_pass_armed;  /* 1: the next large search carries the second pass's launches (armed for the first searches of
                                   an index and after any search that flagged more than the streaming kernel takes cheaply;
                                   dropped after a few consecutive searches that flagged fewer) */
    int32_t wide_mode;          /* 1: large batches are planned with the second pass's list count from the start (the
                                   previous large batch needed the second pass for most of its queries) */
    int32_t sparse_fallback_armed; /* 1: the exact re-search behind the last AUTO search carried the streaming kernel's two
                                   launches (few flagged queries: ~0.05 ms); 0: they were left out after a long run of
                                   searches with nothing flagged, and the fp32-MFMA kernel takes any count (~0.4 ms, once:
                                   the run restarts and the required length doubles). Results are exact either way. */
    int32_t centered;           /* 1: the fp16 image of the corpus holds the rows minus their column mean (the rows share a large
                                   common component, as sentence embeddings do: the coarse pass ranks by q.(c - mu), the same
                                   order as q.c, with an fp16 error relative to the centred rows). Results are exact either way. */
    float   mean_share;         /* |mu|^2 / mean |row|^2: the mean pairwise cosine of unit rows (centred when >= 0.25) */
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
 * flags:  ICD_CREATE_CORPUS_ON_DEVICE - `corpus` (and `levels`) are device pointers; the rest are A/B and test options of
 *         THIS index (results are identical with or without them): ICD_CREATE_ROW_ORDER keeps the fp16 corpus copy in row
 *         order instead of the golden-ratio permutation (only the share of certified queries changes); ICD_CREATE_NO_PROBE
 *         skips the corpus-shape probe (with it, an index whose max_nq allows batches of >= 2 048 queries searches 2 048
 *         evenly spaced rows of its own corpus once and starts large batches on the wide partition, icd_stats.wide_mode, when
 *         most of them could not be certified from the narrow plan's lists - a corpus of tight families of near-identical
 *         rows; later searches keep deciding from their own counters); ICD_CREATE_NO_CENTER keeps the fp16 image uncentred
 *         whatever the rows look like (icd_stats.centered).
 */
enum { ICD_CREATE_CORPUS_ON_DEVICE = 1, ICD_CREATE_ROW_ORDER = 2, ICD_CREATE_NO_PROBE = 4, ICD_CREATE_NO_CENTER = 8 };
int icd_index_create(const float *corpus, int64_t n, int32_t dim, const int32_t *levels,
                     int64_t id_base, int32_t device, int32_t max_nq, int32_t max_k,
                     int32_t flags, icd_index **out);

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

/*
 * Multi-GPU search, one process per GPU. Every rank creates its icd_index first (ROW_SHARD: over its rows, with id_base =
 * index of its first row in the whole corpus; QUERY_SHARD: over the whole corpus), then the group:
 *     rank 0:   icd_group_unique_id(id)   -> hand the ICD_GROUP_ID_BYTES bytes to every rank (file, socket, MPI, ...)
 *     all:      icd_group_create(index, id, rank, world, mode, max_nq, max_k, &group)     (collective: ncclCommInitRank)
 *     all:      icd_group_search(group, queries, nq, k, gather, out_adj, out_raw, out_ids, out_levels, stream)
 * `queries` is the FULL batch [nq][dim] on every rank (device pointer); outputs are device pointers [nq][k] in the order
 * icd_index_search_reweighted gives. ROW_SHARD: identical results on every rank (local raw top-k with global ids and
 * levels -> one grouped ncclAllGather of 16 bytes per hit -> icd_merge_topk). QUERY_SHARD: rank r searches rows
 * [lo_r, hi_r) of the batch (contiguous split, the first nq % world ranks one row more); gather = 1: one grouped
 * ncclAllGather gives every rank all nq rows; gather = 0: no collective at all, rows 0 .. hi_r - lo_r of the outputs hold
 * this rank's slice. Everything is enqueued on `stream`; nothing synchronises. world = 1 needs no id and no RCCL (given
 * an id all the same, a one-rank communicator is created and the collective path runs end to end: tests).
 * librccl.so.1 is opened (dlopen) by the first call that needs it.
 * icd_group_create = icd_group_prepare (everything that can fail on ONE rank alone: argument checks, the device buffers,
 * opening librccl) + icd_group_connect (ncclCommInitRank: COLLECTIVE - a rank that fails before it leaves the others
 * waiting inside it). A host whose ranks can fail locally calls the two itself and lets the ranks agree on the outcome of
 * prepare (any side channel) before any of them connects; rag_project_icd10_amd/sharded.py does exactly that over the
 * torch.distributed group that already exists. Errors of a group call name the rank and the world size.
 */
#define ICD_GROUP_ID_BYTES 128
typedef struct icd_group icd_group;
typedef enum icd_group_mode { ICD_GROUP_ROW_SHARD = 0, ICD_GROUP_QUERY_SHARD = 1 } icd_group_mode;
int icd_group_unique_id(uint8_t *out_id /* [ICD_GROUP_ID_BYTES] */);
int icd_group_create(icd_index *local, const uint8_t *id, int32_t rank, int32_t world, int32_t mode, int32_t max_nq,
                     int32_t max_k, icd_group **out);
/* with_comm: 1 = a communicator will follow (always the case for world > 1; a one-rank group may ask for one: tests) */
int icd_group_prepare(icd_index *local, int32_t with_comm, int32_t rank, int32_t world, int32_t mode, int32_t max_nq,
                      int32_t max_k, icd_group **out);
int icd_group_connect(icd_group *group, const uint8_t *id);
int icd_group_search(icd_group *group, const float *queries, int64_t nq, int32_t k, int32_t gather, double *out_adj,
                     float *out_raw, int64_t *out_ids, int32_t *out_levels, void *stream);
int icd_group_destroy(icd_group *group);   /* (the index stays with the caller) */

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

/* The winners of a rescored batch for the host in ONE array (row N2: what MultiDiagnosisService.match_diagnoses_batch turns into
 * Candidate objects - reference services/multi_diagnosis_service.py:147-176 builds them from the rescored hit dicts): for the
 * top kk <= k positions of every query, out[c][q][j] (8-byte slots, c = 0 .. 7; doubles except c = 0, which holds the int64 id's
 * BIT PATTERN: reinterpret that plane as int64) = id, raw score and level-reweighted score of the hit
 * order[q][j] points at, order[q][j] itself, and enhanced / vector-similarity / hierarchy-boost / uncertainty-boost [q][j] of
 * icd_hier_rescore's outputs. One launch and one device-to-host copy instead of three gathers, five slices and eight copies.
 * All pointers device; order int32 [nq][k], ids int64 [nq][k], raw float32 [nq][k], the rest float64 [nq][k]; out [8][nq][kk]. */
int icd_pack_winners(int32_t device, const int32_t *order, const int64_t *ids, const float *raw, const double *adj, const double *enhanced,
                     const double *vs, const double *hb, const double *boost, int64_t nq, int32_t k, int32_t kk, double *out, void *stream);

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

/* The self-attention of the packed BERT encoder (row a3-a5 / N1 of SURVEY.md section 8: the text -> vector forward the
 * reference reaches through sentence-transformers; rag_project_icd10_amd/services/embedding_service.py _PackedBert runs
 * every Linear of the encoder over packed tokens and calls this for the attention): softmax(Q K^T / sqrt(64)) V per
 * sequence and head, fp32.
 *   qkv      [T][ld] device fp32: Q | K | V of a token side by side (ld >= 3 * heads * 64, a multiple of 4)
 *   starts   [nseq + 1] device int32: first packed row of every sequence (sequence s is rows starts[s] .. starts[s + 1])
 *   max_len  the longest sequence (host-known): 1 .. 512 (up to 64 tokens one pass; longer ones in chunks of 64 keys)
 *   out      [T][out_ld] device fp32: heads side by side
 * Enqueues one launch on `stream`. */
int icd_packed_attention(int32_t device, const float *qkv, int64_t ld, const int32_t *starts, int32_t nseq, int32_t heads,
                         int32_t head_dim, int32_t max_len, float *out, int64_t out_ld, void *stream);

/* The A operand of a split-bf16 GEMM, in one pass (the Linear layers inside SentenceTransformer.encode, reference
 * services/embedding_service.py:97-102, run as x_hi W_hi + x_hi W_lo + x_lo W_hi + b on the bf16 MFMA with fp32 accumulation
 * and output): x fp32 [rows][cols] (row stride ld elements; device, 16-byte aligned), optionally through erf-GELU (act = 1:
 * BertIntermediate), -> out bf16 [rows][3 cols + 64] = [hi | hi | lo | 1 1 0 ... 0], hi = bf16(x) rounded to nearest
 * even, lo = bf16(x - hi). The weight operand is [W_hi^T; W_lo^T; W_hi^T; b_hi; b_lo; 0 x 62] (3 cols + 64 rows: K stays a
 * multiple of the GEMM's 64-deep step). cols must be a multiple of 8 and at least 64. Enqueued on `stream`. */
int icd_split_bf16x3(int32_t device, const float *x, int64_t rows, int32_t cols, int64_t ld, int32_t act, void *out, void *stream);

/* (last_fallback: every search copies its counters to pinned host memory behind itself, on its stream; this call waits
 *  for the last search of this handle - an event on that stream - and for nothing else on the device) */
int icd_index_stats(icd_index *idx, icd_stats *out);

/* Tuning knob / test hook (0 = automatic): aim for about `chunks` candidate lists per query in the coarse pass
 * (the automatic choice gives every CU the same number of 128 x 128 tiles). */
int icd_index_set_chunks(icd_index *idx, int32_t chunks);

/* Test / A-B switch (default 1): 0 turns the second coarse pass (and the adaptive list count that follows from its counters)
 * off - queries the first pass cannot certify then go straight to the exact re-search, as before round 3; 2 keeps the
 * second pass but never switches to the wide partition (every large batch runs narrow plan + second pass). */
int icd_index_set_second_pass(icd_index *idx, int32_t enabled);

/* A/B and test options of ONE index (round 6: these were process-wide icd_debug_set_* switches; a serving process with two
 * indexes must not have one caller's test hook change the other's behaviour). Every one is a performance decision only:
 * results are identical whatever it is set to. Defaults in brackets. Calls on one handle are serialised with its searches.
 *   ICD_OPT_FAMILY_ORDER [1]  0 keeps the wide-window finalize of a family-shaped corpus (icd_stats.wide_mode) in batch order
 *                             instead of visiting the queries family by family, XCD by XCD (finalize.hpp, order_*_kernel).
 *   ICD_OPT_STREAM_ONE   [1]  0 sends calls of one or two queries (the reference's own call shape,
 *                             services/milvus_service.py:280-285) through the general streaming path - memset, stream_topk,
 *                             reduce_lists, finalize: four operations - instead of the single-launch kernel that folds all of it.
 *   ICD_OPT_HOST_ONE     [3]  how a HOST caller's ONE query reaches and leaves the single-launch kernel. Bit 1: the vector
 *                             travels in the kernel's arguments (no host-to-device copy command in front of the launch). Bit 2:
 *                             the call returns when the kernel's last work-group has stored the call's sequence number behind
 *                             the outputs in the index's mapped host block (polled for a bounded time, then the stream
 *                             synchronisation as before). 0 = the copy + hipStreamSynchronize form.
 *   ICD_OPT_PACING_SHIFT [3], ICD_OPT_PACING_LEAD [2]  the coarse sweep over an fp16 image that does not stay in the Infinity
 *                             Cache (a row shard) paces the work-groups that sweep the same corpus tiles: epochs of 2^shift
 *                             tiles, a class stays within `lead` epochs, so that a tile is fetched once per XCD
 *                             (csrc/coarse_flat_kernel.hpp). shift < 0 turns pacing off.
 *   ICD_OPT_EXACT_NARROW [1]  0 makes ICD_MODE_EXACT at k > 32 keep lists of KP >= k (64- / 128-entry candidate buffers, one
 *                             work-group per CU) instead of certified lists of 32 over row-strided chunks with a re-search of
 *                             the queries the certificate cannot clear (k range: /query searches top_k * 2 with top_k <= 50,
 *                             models/icd_models.py:138, services/multi_diagnosis_service.py:153).
 *   ICD_OPT_WIDE_FROM   [48]  searches with k above this keep 24 candidates per coarse list (about k / 6 lists per query) instead of
 *                             16 (about k / 4 lists): a slower sweep, but no query whose top-k crowd one list is left for the
 *                             exact re-search (profiles/r06_k100_lists.log). >= ICD_MAX_K: lists of 16 at every k.
 * No reference counterpart (the reference delegates the search to Milvus Lite). */
enum { ICD_OPT_FAMILY_ORDER = 1, ICD_OPT_STREAM_ONE = 2, ICD_OPT_HOST_ONE = 3, ICD_OPT_PACING_SHIFT = 4, ICD_OPT_PACING_LEAD = 5, ICD_OPT_EXACT_NARROW = 6,
       ICD_OPT_WIDE_FROM = 7 };
int icd_index_set_option(icd_index *idx, int32_t option, int32_t value);

/* Test entry: the unpack step of a query-sharded icd_group_search (one kernel: the all-gathered PADDED slices -> the
 * contiguous [nq][k] outputs) on a caller-made receive buffer, so that its index arithmetic can be checked for any world
 * size on ONE GPU. `gathered` (device) is laid out as the group's receive buffer: with width = ceil(nq / world) and
 * per = width * k, four arrays back to back - adj f64 [world][per] | ids i64 [world][per] | raw f32 [world][per] |
 * levels i32 [world][per]; rank r's slice holds queries [lo_r, hi_r) (contiguous split, the first nq % world ranks one
 * more) in its first hi_r - lo_r rows. Outputs: device pointers, [nq][k]. No reference counterpart (single process). */
int icd_unpack_query_slices(int32_t device, const void *gathered, int32_t world, int64_t nq, int32_t k, double *out_adj,
                                  float *out_raw, int64_t *out_ids, int32_t *out_levels, void *stream);

/* ---- the sentence encoder for SMALL inputs ------------------------------------------------
 * The reference embeds ONE string per call - EmbeddingService.encode_query / encode_single -> SentenceTransformer.encode
 * (services/embedding_service.py:97-102, 117-120) - and a /query request a handful (services/multi_diagnosis_service.py:97,
 * 137). At 10-100 tokens the framework's forward is ~220 kernels of ~5 us (1.1 ms replayed from a graph); this is the same
 * BERT forward (post-LN encoder, absolute positions, erf-GELU, 64-wide heads, fp32) as 7 hand-written launches per layer
 * (csrc/encoder_small.hpp), replayed as ONE graph per token bucket. All pointers are device fp32 (e.g. the torch module's
 * parameters). icd_encoder_create COPIES the four Linear weights of every layer (into the order its GEMMs read them: the
 * caller's may be freed or changed afterwards without effect) and BORROWS the embeddings, biases and LayerNorm parameters:
 * the caller keeps those alive and unchanged while the handle lives. */
typedef struct icd_encoder icd_encoder;
typedef struct {
    int32_t layers, hidden, heads, inter;   /* hidden = heads * 64 = 768 or 1024; inter a multiple of hidden, at most 4 hidden */
    int32_t vocab, max_pos;                 /* rows of word_emb / pos_emb */
    int32_t pos_offset;                     /* position of a sequence's first token: 0 (BERT), padding_idx + 1 (RoBERTa / XLM-R) */
    float ln_eps;
    const float *word_emb, *pos_emb, *type_emb0, *emb_ln_g, *emb_ln_b;
    /* [layers] host arrays of device pointers; Linear weights as torch keeps them, [out][in] row-major */
    const float *const *w_qkv;  const float *const *b_qkv;    /* [3 hidden][hidden]: query | key | value rows stacked */
    const float *const *w_ao;   const float *const *b_ao;     /* attention output dense [hidden][hidden] */
    const float *const *ln1_g;  const float *const *ln1_b;
    const float *const *w_up;   const float *const *b_up;     /* intermediate dense [inter][hidden] */
    const float *const *w_down; const float *const *b_down;   /* output dense [hidden][inter] */
    const float *const *ln2_g;  const float *const *ln2_b;
    int32_t arithmetic;                     /* of the four Linear layers' GEMMs, in BOTH forms (one call / icd_encoder_encode_many) alike:
                                               ICD_ENCODER_ARITH_FP32 (0) fp32-input MFMA, exact products; ICD_ENCODER_ARITH_BF16X3 (1) the
                                               split-bf16 form - x = x_hi + x_lo, w = w_hi + w_lo in bf16, three bf16 MFMAs per 32 k-values
                                               (hi hi, hi lo, lo hi) into fp32: 1 / 5 of the matrix time, ~1e-6 off the fp32 forward */
} icd_encoder_desc;
enum { ICD_ENCODER_ARITH_FP32 = 0, ICD_ENCODER_ARITH_BF16X3 = 1 };
#define ICD_ENCODER_MAX_TOKENS 512   /* packed tokens per call: any ONE sequence a BERT-style encoder takes fits */
#define ICD_ENCODER_MAX_SEQS 64      /* sequences per call */
int icd_encoder_create(int32_t device, const icd_encoder_desc *desc, icd_encoder **out);
/* ids: HOST int32, the sequences' token ids back to back (special tokens included); lengths: HOST int32 [nseq], each >= 1,
 * their sum <= ICD_ENCODER_MAX_TOKENS. pooling 0 = mean over a sequence's tokens, 1 = its first token; normalize 1 =
 * torch.nn.functional.normalize(p = 2). out: [nseq][hidden] fp32, host (the call returns when it is filled) or device
 * (out_on_device = 1: enqueued on `stream`, no synchronisation). hidden_out: NULL, or a DEVICE buffer [sum lengths][hidden]
 * that receives the last hidden state of every token (token classification heads). Calls on one handle are serialised; the
 * handle's activations belong to the handle, not to a stream: a call on ANOTHER stream than the one before waits on the device
 * (an event, no host block) until that call's work has left them - icd_encoder_encode and icd_encoder_encode_many alike. */
int icd_encoder_encode(icd_encoder *e, const int32_t *ids, const int32_t *lengths, int32_t nseq, int32_t pooling, int32_t normalize,
                       float *out, int32_t out_on_device, float *hidden_out, void *stream);
/* ANY number of sequences through the same kernels, in calls of at most ICD_ENCODER_MAX_TOKENS tokens / ICD_ENCODER_MAX_SEQS
 * sequences cut greedily in the given order. The arithmetic of a sequence does not depend on what else shares its call
 * (every reduction runs over the sequence's own tokens in an order fixed by the kernels), so out[i] equals BIT FOR BIT what
 * icd_encoder_encode returns for sequence i alone: the reference embeds corpus rows and queries through the same one-string
 * call (tools/build_database.py:217-222, services/embedding_service.py:117-120) - identical text, identical vector - and this
 * is how a corpus build or a batch of queries keeps that property. lengths[i] in 1 .. min(ICD_ENCODER_MAX_TOKENS, max_pos -
 * pos_offset). out: [nseq][hidden] fp32, device (enqueued on `stream`, no synchronisation) or host (returns when filled). */
int icd_encoder_encode_many(icd_encoder *e, const int32_t *ids, const int32_t *lengths, int64_t nseq, int32_t pooling, int32_t normalize,
                            float *out, int32_t out_on_device, void *stream);
int icd_encoder_destroy(icd_encoder *e);

/* Diagnostic builds only (make ABLATE=1, env ICD_FLAT_VAR with bit 1024): per-wave cycle sums of the coarse kernel,
 * [work-group][wave][8] = {LDS-DMA wait, barrier, stage body, fused select, tiles, ...}. */
int icd_index_debug_counters(icd_index *idx, unsigned long long *out, int32_t count);

/* enabled: 0 off; 1 events around every kernel of every search; N > 1 of every N-th search (an event between two kernels
 * keeps the second from starting under the first one's tail: a timed region samples instead of paying that on every step) */
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
