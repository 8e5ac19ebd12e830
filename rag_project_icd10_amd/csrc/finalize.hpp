// finalize.hpp — one wave per query: merge the per-chunk partial lists, (fast path) certify and
// rescore with the canonical fp32 chain, select the top-k by (score desc, row asc), then apply the
// ICD level reweight and the stable re-sort.
//
// Reference call sites restated here:
//   services/milvus_service.py:280-285  (top_k hits, best first)
//   services/milvus_service.py:290-295  adjusted = float(base_score * level_weight)  [Python double]
//   services/milvus_service.py:550-558  level weights {1:1.2, 2:1.0, 3:0.8}, default 1.0
//   services/milvus_service.py:314      stable sort by adjusted score, descending
#pragma once
#include "exact_kernel.hpp"
#include "topk_select.hpp"

namespace icd {

constexpr int FIN_MAX_CAND = 512;  // P * KP
constexpr int FIN_MAX_CAND_X = 2048;   // ... of the exact kernel's lists at k <= 64 (finalize<false, ., 4> only: 32 keys per lane; a short
                                       // flagged list is cut into up to 2048 / KP row chunks so that every CU has a work-group)
constexpr int FIN_MAX_K = 128;     // largest k of a search
constexpr int FIN_MAX_T = 256;     // largest rescoring window (RESCORE): the best T coarse candidates, T/64 per lane
constexpr int FIN_EF = FIN_MAX_CAND / 64;
constexpr int FIN_EF_X = FIN_MAX_CAND_X / 64;
#ifndef ICD_FIN_WALK
#define ICD_FIN_WALK 8
#endif
constexpr int FIN_WALK = ICD_FIN_WALK;   // 64-B blocks of a row a lane keeps in flight per trip of the rescoring walk (divides 48 and 64)
#ifndef ICD_FIN_OCC
#define ICD_FIN_OCC 7
#endif

struct FinArgs {
    const float *part_scores;  // [slot][P][KP]
    const int *part_rows;
    const float *bounds;       // fast path: [slot][P] largest score each list may have dropped (-inf: none)
    long long perm_mul;        // fast path: list rows are positions of the permuted fp16 corpus; original row =
    int perm_mod;              //            (position * perm_mul) mod perm_mod (perm_mod = 0: identity)
    double perm_inv;           //            1 / perm_mod when perm_mod^2 < 2^53 (the modulo then runs in f64), else 0
    int P, KP;
    int P_dense;        // with nq_ptr: lists per slot when more than sparse_max slots are active (0: always P)
    int sparse_max;
    int dense_one_per_cu;   // (exact_adaptive_chunks)
    int dense_grid, dense_bmq, dense_max_p, n_rows;   // dense_max_p > 0: exact_topk chose its chunk count on the device
    int lds_cand;       // candidate slots the launch's LDS was sized for (0: P * KP)
    int lists_by_query; // with qlist: the lists (and bounds) of slot s belong to query qlist[s] and sit at that query's index
    int skip_below;     // with qlist + nq_ptr: at most this many slots -> this stage is skipped; the slot list is handed on
                        // unchanged to the stage behind it (flagged / nflag), which is cheaper for a handful of queries
    int narrow_check;   // exact path (no rescoring): the lists are NARROWER than k may need (KP < the members of the top-k one list could
                        // hold). A list that is full may have dropped rows; if its worst kept key beats the k-th best of the merge
                        // (or the merge found fewer than k), a dropped row may belong to the top-k: the query is appended to
                        // flagged / nflag and re-searched with lists of KP >= k. Otherwise everything any list dropped ranks below
                        // the k-th best and the merge IS the exact top-k.
    int skip_walk;      // ablation builds only: skip the rescoring walk (timing; results are wrong)
    int wide_window;    // fast path: rescoring window as wide as the instantiation allows (second chance of a query whose
                        // first window overflowed with near-ties), not the one sized for k
    int nq;             // slots (upper bound if nq_ptr)
    const int *nq_ptr;  // nullable
    const int *qlist;   // nullable: slot -> query index
    int k;
    // fast path only
    const float *queries;  // fp32 [*][dim]
    const float *corpus;   // fp32 [n][dim]
    int dim;
    const float *qnorm;          // [query] norm of the query's SCALED fp16 image (coarse_common.hpp)
    const int *qexp;             // [query] its power-of-two scale exponent
    const unsigned char *qbad;   // [query] 1 = fp16 image unusable (non-finite input)
    float rmax, eps_rel;         // rmax: largest norm of the scaled corpus rows (of the CENTRED rows when the index centres its image)
    float rmax_unc, eps_f32;     // centred image only (else 0): largest norm of the uncentred rows in the image's scale, and the fp32
                                 // chain's relative error bound on them (coarse_common.hpp, "centred")
    int cexp;                    // the corpus's scale exponent
    int *nflag;    // fallback counter
    int *flagged;  // fallback list
    float *thr0;   // nullable, fast path: [query] an uncertified query leaves the smallest canonical score of its k best
                   // coarse candidates here - k distinct rows reach it, so no row below it is in the top-k and the exact
                   // re-search starts its lists at that threshold instead of -inf (exact_kernel.hpp)
    int *counters;         // nullable (last launch of a search): the search's four fallback counters (+ [4], the run length below) ...
    int track_run;         // finalize<false> of a fast-path search: counters[4] = consecutive searches whose re-search had nothing to do
    int *host_counters;    // ... are copied to this pinned host array by block 0 (icd_index_stats reads them after the stream's event)
    // level table
    const int *levels;  // [n] or null
    long long id_base;
    // outputs, any may be null
    float *out_scores;      // raw order [query][k]
    long long *out_ids;
    double *out_adj;        // re-sorted order [query][k]
    float *out_adj_raw;
    long long *out_adj_ids;
    int *out_adj_levels;
};

__device__ __forceinline__ double level_weight(int level) {
    return level == 1 ? 1.2 : (level == 3 ? 0.8 : 1.0);
}

__device__ __forceinline__ float wave_max_f32(float v) {
    // DPP reduction (no LDS crossbar round trips): row scans, then the two cross-row broadcasts; lane 63 holds the max
    const int ninf = (int)0xff800000u;
#define ICD_DPP_MAX(ctrl, rmask) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(ninf, __float_as_int(v), ctrl, rmask, 0xf, false)))
    ICD_DPP_MAX(0x111, 0xf);   // row_shr:1
    ICD_DPP_MAX(0x112, 0xf);   // row_shr:2
    ICD_DPP_MAX(0x114, 0xf);   // row_shr:4
    ICD_DPP_MAX(0x118, 0xf);   // row_shr:8
    ICD_DPP_MAX(0x142, 0xa);   // row_bcast:15
    ICD_DPP_MAX(0x143, 0xc);   // row_bcast:31
#undef ICD_DPP_MAX
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Writes the final raw + re-sorted outputs for one query from sorted[0..nres) (best first).
__device__ __forceinline__ void emit_outputs(const FinArgs &a, int qidx, const u64 *sorted, int nres,
                                             double *adjbuf, int lane) {
    const int k = a.k;
    const size_t o = (size_t)qidx * k;
    float sc[2];
    int row[2], lvl[2];
    double adj[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int j = lane + 64 * e;
        sc[e] = -INFINITY; row[e] = -1; lvl[e] = 0; adj[e] = -INFINITY;
        if (j < nres) {
            const u64 key = sorted[j];
            sc[e] = key_score(key);
            row[e] = (int)key_row(key);
            lvl[e] = a.levels ? a.levels[row[e]] : 1;
            adj[e] = (double)sc[e] * level_weight(lvl[e]);
            adjbuf[j] = adj[e];
        }
        if (j < k) {
            if (a.out_scores) a.out_scores[o + j] = sc[e];
            if (a.out_ids) a.out_ids[o + j] = row[e] >= 0 ? a.id_base + row[e] : -1ll;
        }
    }
    if (!a.out_adj && !a.out_adj_ids && !a.out_adj_raw && !a.out_adj_levels) return;
    // stable descending rank of adj among the nres hits
    int pos[2] = {0, 0};
    for (int i = 0; i < nres; ++i) {
        const double ai = adjbuf[i];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int j = lane + 64 * e;
            pos[e] += (ai > adj[e] || (ai == adj[e] && i < j)) ? 1 : 0;
        }
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int j = lane + 64 * e;
        if (j < nres) {
            const size_t w = o + pos[e];
            if (a.out_adj) a.out_adj[w] = adj[e];
            if (a.out_adj_raw) a.out_adj_raw[w] = sc[e];
            if (a.out_adj_ids) a.out_adj_ids[w] = a.id_base + row[e];
            if (a.out_adj_levels) a.out_adj_levels[w] = lvl[e];
        } else if (j < k) {
            const size_t w = o + j;
            if (a.out_adj) a.out_adj[w] = -INFINITY;
            if (a.out_adj_raw) a.out_adj_raw[w] = -INFINITY;
            if (a.out_adj_ids) a.out_adj_ids[w] = -1ll;
            if (a.out_adj_levels) a.out_adj_levels[w] = 0;
        }
    }
}

// LDS per wave: keys[max(ncand, 64) rounded to 64] u64 | sorted[64 ewm] u64 | adjbuf[64 or 128] double | qvec[dim] float (RESCORE)
__host__ __device__ inline int fin_key_slots(int ncand) { return ncand <= 64 ? 64 : ((ncand + 63) & ~63); }
// ewm = rescoring candidates per lane of the instantiation (1: k <= 32; 4: larger k and the exact path)
__host__ __device__ inline int fin_sorted_slots(int ewm) { return 64 * ewm; }
__host__ __device__ inline int fin_adj_slots(int ewm) { return ewm == 1 ? 64 : FIN_MAX_K; }
__host__ __device__ inline size_t fin_wave_lds_bytes(bool rescore, int dim, int ncand, int ewm) {
    return (size_t)fin_key_slots(ncand) * 8 + (size_t)fin_sorted_slots(ewm) * 8 + (size_t)fin_adj_slots(ewm) * 8 + (rescore ? (size_t)dim * 4 : 0);
}

// (p * mul) mod n. A 64-bit integer modulo is ~150 VALU instructions on this machine and the kernel is VALU-issue
// bound; with n^2 < 2^53 the product, the quotient estimate and the remainder are exact in f64 (6 instructions).
__device__ __forceinline__ int perm_row(int p, long long mul, int n, double inv) {
    if (inv > 0.0) {
        const double x = (double)p * (double)mul;          // < n^2 < 2^53: exact
        double q = floor(x * inv);                         // within 1 of the true quotient
        double r = __builtin_fma(-q, (double)n, x);        // exact
        if (r < 0.0) r += (double)n;
        if (r >= (double)n) r -= (double)n;
        return (int)r;
    }
    return (int)(((long long)p * mul) % n);
}

// the certification bound of one query (derivation: step 3 of finalize_kernel)
__device__ __forceinline__ float fin_eps(const FinArgs &a, float qn, int qexp) {
    return a.eps_rel * qn * a.rmax + a.eps_f32 * qn * a.rmax_unc + ldexpf((float)a.dim, qexp + a.cexp - 148);
}

// Canonical score of one corpus row, four lanes per row (the quad walk of step 4 of finalize_kernel): lane 4 g + qi loads
// piece 4 b + qi of every 64-byte block b, all four lanes run their four fmaf from the same running value and adopt lane
// 0's, 1's, 2's, 3's result in turn. Every lane of the quad returns the chain value (d ascending: the oracle's bits).
__device__ __forceinline__ float fin_quad_row(const float *corpus, uint32_t row, int dim, const float *qvec, int qi) {
    const int nblk = dim >> 4;
    const f32x4 *c4 = reinterpret_cast<const f32x4 *>(corpus + (size_t)row * dim) + qi;
    const f32x4 *q4q = reinterpret_cast<const f32x4 *>(qvec) + qi;
    float acc = 0.0f;
#define ICD_QUAD_STEP(ctrl)                                                                          \
    {                                                                                                \
        float t = acc;                                                                               \
        t = __builtin_fmaf(qv.x, cv.x, t);                                                           \
        t = __builtin_fmaf(qv.y, cv.y, t);                                                           \
        t = __builtin_fmaf(qv.z, cv.z, t);                                                           \
        t = __builtin_fmaf(qv.w, cv.w, t);                                                           \
        acc = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), ctrl, 0xf, 0xf, false)); \
    }
    // (eight blocks = eight independent 16-B loads per lane in flight per trip; the inner trip count is a constant
    //  because hipcc does not unroll a runtime-count loop around the convergent DPP move)
    for (int b0 = 0; b0 < nblk; b0 += FIN_WALK) {
        f32x4 cvv[FIN_WALK];
#pragma unroll
        for (int u = 0; u < FIN_WALK; ++u) cvv[u] = c4[4 * (b0 + u)];
#pragma unroll
        for (int u = 0; u < FIN_WALK; ++u) {
            const f32x4 cv = cvv[u];
            const f32x4 qv = q4q[4 * (b0 + u)];
            ICD_QUAD_STEP(0x00)   // quad_perm [0,0,0,0]: everyone continues from lane 0's four steps
            ICD_QUAD_STEP(0x55)   // [1,1,1,1]
            ICD_QUAD_STEP(0xAA)   // [2,2,2,2]
            ICD_QUAD_STEP(0xFF)   // [3,3,3,3]
        }
    }
#undef ICD_QUAD_STEP
    return acc;
}

// Load one query's candidates (NE per lane), rank them (rank_top) and scatter the best T, sorted, into sorted[].
// Returns the number of valid candidates; tau picks up the score of the candidate of rank T (the best one left out).
template <bool RESCORE, int NE>
__device__ __forceinline__ int fin_merge(const FinArgs &a, size_t pbase, int ncand, int T, u64 *keys, u64 *sorted, int lane, float &tau) {
    u64 key[NE];
    int nvalid = 0;
    {
        int crow[NE];
        float cscore[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) {   // rows and scores as independent loads: one round trip, not two per element
            const int i = lane + 64 * e;
            crow[e] = -1;
            cscore[e] = 0.0f;
            if (i < ncand) {
                crow[e] = a.part_rows[pbase + i];
                cscore[e] = a.part_scores[pbase + i];
            }
        }
        if (RESCORE && a.perm_mod > 0) {   // list positions of the permuted fp16 corpus -> original rows
#pragma unroll
            for (int e = 0; e < NE; ++e)
                if (crow[e] >= 0) crow[e] = perm_row(crow[e], a.perm_mul, a.perm_mod, a.perm_inv);
        }
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int i = lane + 64 * e;
            key[e] = crow[e] >= 0 ? make_key(cscore[e], (uint32_t)crow[e]) : 0ull;
            if (i < ncand) keys[i] = key[e];
            nvalid += __popcll(__ballot(key[e] != 0ull));
        }
    }
    int rank[NE];
    rank_top<NE>(key, rank, ncand, T, keys, lane);   // (may replace keys[] by the survivor list)
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        if (key[e] != 0ull) {
            if (rank[e] < T) sorted[rank[e]] = key[e];
            else if (RESCORE && rank[e] == T) tau = fmaxf(tau, key_score(key[e]));
        }
    }
    return nvalid;
}

// Rank ns keys staged at keys[0..ns) among themselves (E2 per lane) and write them, best first, to sorted[].
template <int E2>
__device__ __forceinline__ void fin_rank_staged(const u64 *keys, int ns, u64 *sorted, int lane) {
    u64 mine[E2];
    int rk[E2];
#pragma unroll
    for (int e = 0; e < E2; ++e) {
        const int i = lane + 64 * e;
        mine[e] = i < ns ? keys[i] : 0ull;
        rk[e] = 0;
    }
    for (int j = 0; j < ns; ++j) {
        const u64 kj = keys[j];   // (wave-uniform address: an LDS broadcast)
#pragma unroll
        for (int e = 0; e < E2; ++e) rk[e] += (kj > mine[e]) ? 1 : 0;
    }
#pragma unroll
    for (int e = 0; e < E2; ++e)
        if (mine[e] != 0ull) sorted[rk[e]] = mine[e];
}

// The merge of a WIDE rescoring window (T = 128 / 256: a family of near-identical rows, 128-512 candidates in 8-32 lists).
// Ranking every candidate against every other (rank_top) is ncand^2 / 64 64-bit compares per lane: 384 candidates cost
// 9 000 VALU instructions per query, 0.26 of the 0.32 ms the wide-window finalize took per 10 000 family queries (the
// rescoring walk: 0.07). But the window is only the candidates within 2 eps of the k-th best, and a lower bound of that
// score is cheap: the k-th largest of the 64 lane maxima (k distinct candidates reach it). Everything below
// `that - 2 eps` is below the certification limit L whatever the k-th best turns out to be: it can neither enter the
// window nor the result, and - being a KNOWN candidate below L - it does not raise tau. The survivors (the family: ~124)
// are compacted and ranked among themselves; a survivor's rank among them IS its rank among all candidates. More
// survivors than the window holds: the general path (exact top-T, the rest raises tau).
// Returns the number of keys in sorted[] (the caller's nres).
template <int NE>
__device__ __forceinline__ int fin_merge_window(const FinArgs &a, size_t pbase, int ncand, int T, int k, float two_eps,
                                                u64 *keys, u64 *sorted, int lane, float &tau) {
    u64 key[NE];
    int nvalid = 0;
    {
        int crow[NE];
        float cscore[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int i = lane + 64 * e;
            crow[e] = -1;
            cscore[e] = 0.0f;
            if (i < ncand) {
                crow[e] = a.part_rows[pbase + i];
                cscore[e] = a.part_scores[pbase + i];
            }
        }
        if (a.perm_mod > 0) {
#pragma unroll
            for (int e = 0; e < NE; ++e)
                if (crow[e] >= 0) crow[e] = perm_row(crow[e], a.perm_mul, a.perm_mod, a.perm_inv);
        }
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            key[e] = crow[e] >= 0 ? make_key(cscore[e], (uint32_t)crow[e]) : 0ull;
            nvalid += __popcll(__ballot(key[e] != 0ull));
        }
    }
    u64 lmax = 0ull;
#pragma unroll
    for (int e = 0; e < NE; ++e) lmax = key[e] > lmax ? key[e] : lmax;
    int above = 0;   // lanes whose maximum beats this lane's (keys are unique; empty lanes hold 0)
    for (int l = 0; l < 64; ++l) above += (readlane_u64(lmax, l) > lmax) ? 1 : 0;
    const u64 hit = __ballot(lmax != 0ull && above == k - 1);
    float lb = -INFINITY;   // fewer than k non-empty lanes: keep everything
    if (hit) {
        lb = key_score(readlane_u64(lmax, __ffsll((long long)hit) - 1)) - two_eps;
        lb = lb - fabsf(lb) * 1.2e-7f;   // (one ulp down, as for L)
    }
    const u64 lt = (1ull << lane) - 1ull;
    int ns = 0;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const bool keep = key[e] != 0ull && key_score(key[e]) >= lb;
        const u64 m = __ballot(keep);
        if (keep) keys[ns + __popcll(m & lt)] = key[e];
        ns += __popcll(m);
    }
    if (ns > T) {   // more near-ties than the window holds
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int i = lane + 64 * e;
            if (i < ncand) keys[i] = key[e];
        }
        int rank[NE];
        rank_top<NE>(key, rank, ncand, T, keys, lane);
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            if (key[e] != 0ull) {
                if (rank[e] < T) sorted[rank[e]] = key[e];
                else if (rank[e] == T) tau = fmaxf(tau, key_score(key[e]));
            }
        }
        return min(nvalid, T);
    }
    if (ns <= 128) fin_rank_staged<2>(keys, ns, sorted, lane);
    else fin_rank_staged<4>(keys, ns, sorted, lane);
    return ns;
}

// DEEP: (round 1-2: a deeper per-lane prefetch of the rescoring rows for small launches; the quad-cooperative walk below
// made it moot - the parameter is kept so that the host's instantiation table stays as it is)
// EWM: rescoring candidates per lane the instantiation can hold (1 for k <= 32: the common case keeps its registers and
// resident waves; 4 for k up to 100).
template <bool RESCORE, bool DEEP = false, int EWM = 4>
__global__ __launch_bounds__(256, EWM == 1 ? ICD_FIN_OCC : 1) void finalize_kernel(FinArgs a) {   // (7 waves per SIMD: <= 72 VGPRs)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (!RESCORE && a.host_counters && blockIdx.x == 0) {
        if (threadIdx.x < 4) a.host_counters[threadIdx.x] = a.counters[threadIdx.x];
        if (threadIdx.x == 4) {   // (one launch per search, block 0 only: no race on the run length)
            int run = a.counters[4];
            if (a.track_run) {
                run = (a.nq_ptr && *a.nq_ptr == 0) ? min(run + 1, 1 << 30) : 0;
                a.counters[4] = run;
            }
            a.host_counters[4] = run;
        }
    }
    const int nq = a.nq_ptr ? min(*a.nq_ptr, a.nq) : a.nq;
    if (RESCORE && a.skip_below > 0 && nq <= a.skip_below) {   // (kernel-uniform) hand the slot list on as it is
        if (blockIdx.x == 0 && a.qlist) {
            for (int i = threadIdx.x; i < nq; i += blockDim.x) a.flagged[i] = a.qlist[i];
            if (threadIdx.x == 0) *a.nflag = nq;
        }
        return;
    }
    const int slot = blockIdx.x * 4 + wave;
    if (slot >= nq) return;  // wave-uniform; no work-group barriers below
    const int qidx = a.qlist ? a.qlist[slot] : slot;
    if (qidx < 0) return;   // (a padding entry of the family order, order_scatter_kernel)

    // the fallback's two producers leave different numbers of lists per slot (stream_topk: P, exact_topk: P_dense)
    int P = a.P;
    if (a.nq_ptr && a.P_dense > 0 && nq > a.sparse_max)
        P = a.dense_max_p > 0 ? exact_adaptive_chunks(nq, a.dense_bmq, a.dense_grid, a.dense_max_p, a.n_rows, a.dense_one_per_cu) : a.P_dense;
    const int ncand = P * a.KP;
    const int lds_cand = a.lds_cand > 0 ? a.lds_cand : ncand;
    char *wbase = smem + (size_t)wave * fin_wave_lds_bytes(RESCORE, a.dim, lds_cand, EWM);
    u64 *keys = reinterpret_cast<u64 *>(wbase);
    u64 *sorted = keys + fin_key_slots(lds_cand);
    double *adjbuf = reinterpret_cast<double *>(sorted + fin_sorted_slots(EWM));
    float *qvec = reinterpret_cast<float *>(adjbuf + fin_adj_slots(EWM));
    // issued first, consumed last: the query (rescoring operand) and its norm / usability flag travel while the
    // candidates are merged (a wave's life is a chain of dependent round trips; these need not be part of it)
    float qn_early = 0.0f;
    int qexp_early = 0;
    unsigned char qbad_early = 0;
    if (RESCORE) {
        qn_early = a.qnorm[qidx];
        qexp_early = a.qexp[qidx];
        qbad_early = a.qbad[qidx];
        const float *qsrc0 = a.queries + (size_t)qidx * a.dim;
        for (int d = lane * 4; d < a.dim; d += 256)
            *reinterpret_cast<float4 *>(qvec + d) = *reinterpret_cast<const float4 *>(qsrc0 + d);
    }

    const int k = a.k;
    const size_t lslot = a.lists_by_query ? (size_t)qidx : (size_t)slot;
    const size_t pbase = lslot * ncand;

    // 1-2. load the candidates, rank them, keep the best T sorted. Instantiated per candidates-per-lane (2 covers the
    // common P * KP <= 128; the loops of an 8-per-lane instance would run six dead elements through ~12 VALU per
    // candidate: the kernel is VALU-issue bound, 4 000 VALU per wave)
    // RESCORE keeps the best T coarse candidates as the rescoring window; everything past T counts as dropped (it raises
    // tau). 32 is plenty at k <= 12 (the window holds ~12 rows) and, with many lists, lets rank_top's prefilter cut the
    // ranking loop from ~300 survivors to ~50.
    const int T = !RESCORE ? k
                  : (a.wide_window && EWM >= 4) ? (ncand >= 256 ? 256 : (ncand >= 128 ? 128 : 64))
                  : ((EWM >= 4 && k > 64 && ncand >= 256) ? 256 : (EWM >= 2 && k > 32 && ncand >= 128) ? 128 : ((ncand > 128 && k <= 12) ? 32 : 64));
    float tau = -INFINITY;  // largest coarse score that may have been dropped anywhere
    if (RESCORE) {
        // every list comes with the threshold it ended on: nothing it dropped scores above that. The lists
        // are ranked by a key with 6 low score bits dropped when scores tie (Sel2), hence the relative slack.
        if (lane < P) {
            const float bd = a.bounds[lslot * P + lane];
            if (bd > -INFINITY) tau = bd + fabsf(bd) * COARSE_KEY_SLACK;
        }
    }
    int nres = 0;
    bool merged = false;
    if constexpr (RESCORE && EWM >= 4) {
        if (T > 64) {   // (wave-uniform) a wide window: only what can be within 2 eps of the k-th best is ranked
            const float two_eps = 2.0f * fin_eps(a, qn_early, qexp_early);
            nres = ncand <= 128 ? fin_merge_window<2>(a, pbase, ncand, T, k, two_eps, keys, sorted, lane, tau)
                                : fin_merge_window<FIN_EF>(a, pbase, ncand, T, k, two_eps, keys, sorted, lane, tau);
            merged = true;
        }
    }
    if (!merged) {
        int nvalid;
        if (ncand <= 128) nvalid = fin_merge<RESCORE, 2>(a, pbase, ncand, T, keys, sorted, lane, tau);
        else if (!RESCORE && EWM >= 4 && ncand > FIN_MAX_CAND) {
            if constexpr (!RESCORE && EWM >= 4) nvalid = fin_merge<false, FIN_EF_X>(a, pbase, ncand, T, keys, sorted, lane, tau);
            else nvalid = 0;
        }
        else nvalid = fin_merge<RESCORE, FIN_EF>(a, pbase, ncand, T, keys, sorted, lane, tau);
        nres = min(nvalid, T);
    }

    if (RESCORE) {
        tau = wave_max_f32(tau);
        // 3. certification window: candidate of coarse rank lane + 64 e (one per lane for k <= 32, up to four for larger k)
        const int EW = T / 64 > 1 ? (T / 64 < EWM ? T / 64 : EWM) : 1;
        u64 mine[EWM];
        float coarse[EWM];
#pragma unroll
        for (int e = 0; e < EWM; ++e) {
            const int idx = lane + 64 * e;
            mine[e] = (e < EW && idx < nres) ? sorted[idx] : 0ull;
            coarse[e] = mine[e] != 0ull ? key_score(mine[e]) : -INFINITY;
        }
        // |coarse - canonical| <= eps for every row, in the coarse pass's scaled units 2^(qexp + cexp):
        //   relative part: fp16 rounding of both operands (their images sit in fp16's normal range by construction; the
        //   components that still underflow add at most 2^-25 sqrt(dim) (|q| + |c|), which both norms >= 1 turn into a
        //   relative 3.3e-6: inside EPS_REL's margin) + fp32 accumulation of both chains;
        //   absolute part: the canonical chain runs on the UNSCALED fp32 data, where a result in fp32's subnormal range
        //   rounds to a multiple of 2^-149: <= dim * 2^-150 over the chain (2^-148 here), times the scale. For data
        //   anywhere near unit norm this term is 0; for data scaled by ~1e-38 it takes over and nothing is certified.
        //   centred image (coarse = q.(c - mu) in scaled units): the rounding terms above are relative to the CENTRED rows
        //   (rmax); the canonical chain still runs on the uncentred rows, and its own accumulation error - relative to THEIR
        //   norm - is the second term (eps_f32 = 2 dim 2^-24 >= the chain's gamma_dim). q.mu is one constant per query: it
        //   shifts every coarse score of the query alike and never enters a comparison.
        const float eps = fin_eps(a, qn_early, qexp_early);
        bool certified;
        float L;
        if (nres >= k) {
            const float sk = key_score(sorted[k - 1]);
            L = sk - 2.0f * eps;
            L = L - fabsf(L) * 1.2e-7f;  // one ulp down: the subtraction itself rounds
            certified = tau < L;
        } else {
            L = -INFINITY;
            certified = (tau == -INFINITY);
        }
        if (qbad_early) certified = false;
        if (!certified) {
            float t0 = -INFINITY;
            if (a.thr0 && nres >= k) {   // the canonical scores of the k best coarse candidates: k distinct rows reach their minimum
                const int gi = lane >> 2, qi = lane & 3;
                float mn = INFINITY;
                for (int g0 = 0; g0 < k; g0 += 16) {
                    const int r = g0 + gi;
                    const float sc = fin_quad_row(a.corpus, key_row(sorted[r < k ? r : g0]), a.dim, qvec, qi);
                    if (r < k) mn = (sc == sc) ? fminf(mn, sc) : -INFINITY;
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) mn = fminf(mn, __shfl_xor(mn, off));
                t0 = mn;
            }
            if (lane == 0) {
                const int i = atomicAdd(a.nflag, 1);
                a.flagged[i] = qidx;
                if (a.thr0) a.thr0[qidx] = t0;
            }
            return;
        }
        // 4. canonical rescoring of the window (scalar fmaf chain, d ascending).
        // The window is a PREFIX of the coarse ranking (rank = lane + 64 e): W rows. One lane walking its own 3 KB row costs 24
        // dependent round trips (8 pieces of 16 B in flight) with a dozen lanes active and a dozen cache lines touched per
        // load instruction. Here FOUR lanes share a row, 16 rows at a time: lane 4 g + i loads piece 4 b + i of every 64-B
        // block b of row g - four times the bytes in flight per row, a quarter of the round trips and of the load
        // instructions (finalize 0.098 -> 0.072 ms per 10 000 queries at k = 10). The chain stays strictly sequential in d:
        // within a block all four lanes run their four fmaf from the SAME running value, then the quad adopts lane 0's
        // result, repeats and adopts lane 1's, ... (one DPP quad broadcast per step; three of the four lanes' arithmetic is
        // discarded each time, VALU slots this kernel has to spare). Same fmaf order, d ascending: the same bits.
        // (Round 4, WIDE windows - a 124-row family: one lane per row, 64 rows per pass, has a fifth of the VALU work and was
        //  measured SLOWER, 0.51 against 0.32 ms per 10 000 family queries - 64 different rows per load instruction are 64
        //  cache lines per instruction and the CU's waves evict each other's lines from its 32 KB L1; two lanes per row
        //  gained 4 % while the all-pairs merge was the larger cost and loses 12 % since it is not (0.27 against 0.24 ms);
        //  walking the window in ROW order so that the four waves of a work-group touch the same rows together: 0.38.)
        u64 xkey[EWM];
#pragma unroll
        for (int e = 0; e < EWM; ++e) xkey[e] = 0ull;
        int W = 0;
        {
#pragma unroll
            for (int e = 0; e < EWM; ++e) W += __popcll(__ballot(e < EW && mine[e] != 0ull && coarse[e] >= L));
#ifdef ICD_ABLATE
            if (a.skip_walk) W = 0;   // (timing only: the kernel without its rescoring walk; results are wrong)
#endif
            const int gi = lane >> 2, qi = lane & 3;
            float *hand = reinterpret_cast<float *>(adjbuf);   // (free until emit_outputs)
            for (int g0 = 0; g0 < W; g0 += 16) {
                const int r = g0 + gi;
                const uint32_t row = key_row(sorted[r < W ? r : g0]);   // (past the window: a valid row, result unused)
                const float acc = fin_quad_row(a.corpus, row, a.dim, qvec, qi);
                // every lane of quad g holds the score of rank g0 + g: hand it to the lane (and element) that owns that rank
                if (qi == 0) hand[gi] = acc;   // (same wave: LDS serves its operations in order)
                const int e_g = g0 >> 6, l0 = g0 & 63;
                if (lane >= l0 && lane < l0 + 16 && g0 + (lane - l0) < W) {
                    const float sc = hand[lane - l0];
#pragma unroll
                    for (int e = 0; e < EWM; ++e)
                        if (e == e_g && sc == sc && sc != -INFINITY) xkey[e] = make_key(sc, key_row(mine[e]));
                }
            }
        }
        // 5. rank the rescored candidates and keep the best k
#pragma unroll
        for (int e = 0; e < EWM; ++e)
            if (e < EW) keys[lane + 64 * e] = xkey[e];
        int xr[EWM];
#pragma unroll
        for (int e = 0; e < EWM; ++e) xr[e] = 0;
        if constexpr (EWM == 1) {
            // only the window lanes hold a rescored key: visit those (about a dozen), not all 64 slots
            u64 live = __ballot(xkey[0] != 0ull);
            while (live) {
                const int j = __ffsll((long long)live) - 1;
                live &= live - 1;
                xr[0] += (readlane_u64(xkey[0], j) > xkey[0]) ? 1 : 0;
            }
        } else {
            for (int j = 0; j < W; ++j) {   // (the window is the prefix [0, W) of the coarse ranking: only those slots hold a key)
                const u64 kj = keys[j];
#pragma unroll
                for (int e = 0; e < EWM; ++e) xr[e] += (kj > xkey[e]) ? 1 : 0;
            }
        }
        int nx = 0;
#pragma unroll
        for (int e = 0; e < EWM; ++e) {
            nx += __popcll(__ballot(xkey[e] != 0ull));
            if (xkey[e] != 0ull && xr[e] < k) sorted[xr[e]] = xkey[e];
        }
        nres = min(nx, k);
    }
    if constexpr (!RESCORE) {
        if (a.narrow_check) {   // (kernel-uniform)
            const u64 kth = nres >= k ? sorted[k - 1] : 0ull;
            bool may_have_dropped = false;
            for (int p = lane; p < P; p += 64) {
                const size_t o = pbase + (size_t)p * a.KP + (a.KP - 1);
                const int r = a.part_rows[o];
                if (r >= 0) may_have_dropped = may_have_dropped || make_key(a.part_scores[o], (uint32_t)r) > kth;
            }
            if (__ballot(may_have_dropped) != 0ull && lane == 0) a.flagged[atomicAdd(a.nflag, 1)] = qidx;
        }
    }
    // 6-7. outputs (raw order + level reweight / stable re-sort)
    emit_outputs(a, qidx, sorted, nres, adjbuf, lane);
}

// ---- query order of the wide-window finalize on a family-shaped corpus --------------------------------------------
// The reference's corpus rows repeat their ancestors' names (tools/build_database.py:156-171): sibling codes are near-
// identical rows that sit next to each other in code order, and a query's rescoring window is its whole family - 124
// rows x 3 KB = 381 KB per query, the SAME rows for every query of that family. In batch order those windows are fetched
// from the fabric once per query (3.7 GB per 10 000 queries); with the queries ordered by the original row of their best
// coarse candidate, and consecutive positions of that order mapped to ONE XCD (blockIdx is dealt round-robin over the
// eight), a family's rows are served to its later queries by that XCD's L2.
// Two launches: (1) one wave per query - best candidate, its original row, a histogram of row buckets; (2) prefix sum
// over the buckets, counting-sort scatter into `order` (slot of finalize -> query; -1 = padding). Order within a
// bucket is whatever the atomics give: it decides which wave rescales which query first, never a result.
constexpr int ORDER_BUCKETS = 1024;

struct OrderArgs {
    const float *part_scores;  // [query][P][KP]
    const int *part_rows;      // positions of the permuted fp16 corpus
    int P, KP, nq;
    long long perm_mul; int perm_mod; double perm_inv;
    int shift;                 // bucket = original row >> shift (< ORDER_BUCKETS)
    int *key;                  // [nq] out of (1): the query's bucket
    unsigned int *hist;        // [2 ORDER_BUCKETS] zero on entry of (1): the histogram, then (2)'s cursors
    int *order;                // [4 * ceil(nq / 4)] out of (2)
};

__global__ __launch_bounds__(256) void order_keys_kernel(OrderArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave;
    if (q >= a.nq) return;
    const int ncand = a.P * a.KP;
    const size_t pbase = (size_t)q * ncand;
    u64 best = 0ull;
    for (int i = lane; i < ncand; i += 64) {
        const int r = a.part_rows[pbase + i];
        const float sc = a.part_scores[pbase + i];
        if (r >= 0 && sc == sc) {
            const u64 kx = make_key(sc, (uint32_t)r);
            best = kx > best ? kx : best;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const u64 o = ((u64)(uint32_t)__shfl_xor((int)(best >> 32), off) << 32) | (uint32_t)__shfl_xor((int)best, off);
        best = o > best ? o : best;
    }
    if (lane == 0) {
        int b = 0;
        if (best != 0ull) {
            int row = (int)key_row(best);
            if (a.perm_mod > 0) row = perm_row(row, a.perm_mul, a.perm_mod, a.perm_inv);
            b = min(row >> a.shift, ORDER_BUCKETS - 1);
        }
        a.key[q] = b;
        atomicAdd(&a.hist[b], 1u);
    }
}

// slot of finalize (block b = slot / 4 runs on XCD b % 8) for position p of the sorted order: consecutive chunks of four
// positions go to blocks of ONE XCD (the bijective form of cdna_hip_programming.md T1)
__device__ __forceinline__ int order_slot(int p, int nblk) {
    const int c = p >> 2, w = p & 3;
    const int qn = nblk >> 3, rn = nblk & 7;
    int xcd, j;
    if (c < rn * (qn + 1)) { xcd = c / (qn + 1); j = c - xcd * (qn + 1); }
    else { const int c2 = c - rn * (qn + 1); xcd = rn + c2 / max(qn, 1); j = c2 - (xcd - rn) * max(qn, 1); }
    return (j * 8 + xcd) * 4 + w;
}

// Every work-group of 1 024 threads takes 1 024 queries: it rebuilds the exclusive prefix over the buckets from the finished
// histogram (4 KB, one bucket per thread) and places its queries with an atomic cursor per bucket (hist[ORDER_BUCKETS + b],
// cleared with the histogram before launch (1)). (Round 4, first form: ONE work-group looping over the whole batch, 18 us per
// 10 000 queries of pure latency.)
__global__ __launch_bounds__(1024) void order_scatter_kernel(OrderArgs a) {
    __shared__ unsigned int offs[ORDER_BUCKETS];
    __shared__ unsigned int wsum[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int nblk = (a.nq + 3) >> 2;
    if (blockIdx.x == gridDim.x - 1)
        for (int i = a.nq + t; i < 4 * nblk; i += 1024) a.order[order_slot(i, nblk)] = -1;   // (padding: positions are dense below nq)
    // exclusive prefix over the buckets (one bucket per thread)
    const unsigned int h = a.hist[t];
    unsigned int v = h;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned int o = (unsigned int)__shfl_up((int)v, off);
        if (lane >= off) v += o;
    }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    unsigned int base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    offs[t] = base + v - h;
    __syncthreads();
    const int q = blockIdx.x * 1024 + t;
    if (q < a.nq) {
        const int b = a.key[q];
        const int p = (int)(offs[b] + atomicAdd(&a.hist[ORDER_BUCKETS + b], 1u));
        if (p < 4 * nblk) a.order[order_slot(p, nblk)] = q;   // (always true for a consistent histogram)
    }
}

// ---- row-sharded merge: G gathered best-first lists per query -> global top-k + reweight ---------
struct MergeArgs {
    const float *scores;     // [G][nq][k]
    const long long *ids;    // global ids
    const int *levels;       // level of every hit
    int G, nq, k;
    double *out_adj;
    float *out_raw;
    long long *out_ids;
    int *out_levels;
};

// One wave per query; G*k <= 1024 candidates handled with 16 per lane.
__global__ __launch_bounds__(256) void merge_topk_kernel(MergeArgs a) {
    constexpr int EM = 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave;
    if (q >= a.nq) return;
    // per wave: sc[1024] f32 | id[1024] i64 | lv[1024] i32 | sel (k<=128): adj f64, raw f32, id i64, lvl i32
    char *wb = smem + (size_t)wave * (1024 * 16 + 128 * 24);
    float *csc = reinterpret_cast<float *>(wb);
    long long *cid = reinterpret_cast<long long *>(wb + 4096);
    int *clv = reinterpret_cast<int *>(wb + 4096 + 8192);
    double *sadj = reinterpret_cast<double *>(wb + 16384);
    float *sraw = reinterpret_cast<float *>(wb + 16384 + 1024);
    long long *sid = reinterpret_cast<long long *>(wb + 16384 + 1024 + 512);
    int *slv = reinterpret_cast<int *>(wb + 16384 + 1024 + 512 + 1024);
    const int k = a.k, ncand = a.G * k;
    float sc[EM];
    long long id[EM];
    int nvalid = 0;
#pragma unroll
    for (int e = 0; e < EM; ++e) {
        const int i = lane + 64 * e;
        sc[e] = -INFINITY;
        id[e] = -1;
        if (i < ncand) {
            const int g = i / k, j = i - g * k;
            const size_t src = ((size_t)g * a.nq + q) * k + j;
            const long long v = a.ids[src];
            const float s = a.scores[src];
            if (v >= 0 && s == s) {
                sc[e] = s;
                id[e] = v;
                clv[i] = a.levels[src];
            }
            csc[i] = sc[e];
            cid[i] = id[e];
        }
        nvalid += __popcll(__ballot(id[e] >= 0));
    }
    int rank[EM];
#pragma unroll
    for (int e = 0; e < EM; ++e) rank[e] = 0;
    const int ef = (ncand + 63) >> 6;
    for (int j = 0; j < ncand; ++j) {
        const float sj = csc[j];
        const long long ij = cid[j];
        if (ij < 0) continue;
#pragma unroll
        for (int e = 0; e < EM; ++e)
            if (e < ef) rank[e] += (sj > sc[e] || (sj == sc[e] && ij < id[e])) ? 1 : 0;
    }
    const int nres = min(nvalid, k);
#pragma unroll
    for (int e = 0; e < EM; ++e) {
        const int i = lane + 64 * e;
        if (i < ncand && id[e] >= 0 && rank[e] < k) {
            const int lv = clv[i];
            sraw[rank[e]] = sc[e];
            sid[rank[e]] = id[e];
            slv[rank[e]] = lv;
            sadj[rank[e]] = (double)sc[e] * level_weight(lv);
        }
    }
    // stable re-sort by adjusted score
    const size_t o = (size_t)q * k;
    for (int j = lane; j < k; j += 64) {
        if (j < nres) {
            const double aj = sadj[j];
            int pos = 0;
            for (int i = 0; i < nres; ++i) {
                const double ai = sadj[i];
                pos += (ai > aj || (ai == aj && i < j)) ? 1 : 0;
            }
            if (a.out_adj) a.out_adj[o + pos] = aj;
            if (a.out_raw) a.out_raw[o + pos] = sraw[j];
            if (a.out_ids) a.out_ids[o + pos] = sid[j];
            if (a.out_levels) a.out_levels[o + pos] = slv[j];
        } else {
            if (a.out_adj) a.out_adj[o + j] = -INFINITY;
            if (a.out_raw) a.out_raw[o + j] = -INFINITY;
            if (a.out_ids) a.out_ids[o + j] = -1ll;
            if (a.out_levels) a.out_levels[o + j] = 0;
        }
    }
}

__global__ void lookup_levels_kernel(const long long *ids, long long count, const int *levels,
                                     long long id_base, long long n, int *out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const long long r = ids[i] - id_base;
    out[i] = (ids[i] >= 0 && r >= 0 && r < n) ? (levels ? levels[r] : 1) : 0;
}

}  // namespace icd
