// hier_kernel.hpp — the string-free arithmetic of the hierarchical rescoring, on the device, for a whole batch of
// queries at once.
//
// Restates, for hits shaped as MilvusService.search returns them (the live /query path: level, parent_code, preferred_zh
// and semantic_text sit under "metadata"/"title", so the rescoring's top-level reads see their defaults):
//   services/uncertainty_diagnosis_service.py:190-238  score += boost * weight for ".9"-coded candidates, stable re-sort
//   services/hierarchical_similarity_service.py:143-219 calculate_enhanced_similarity (exact-match rule on an empty title)
//   services/hierarchical_similarity_service.py:243-291 hierarchy boost = min(level term + chapter boost * 0.4, 0.3)
//   services/hierarchical_similarity_service.py:475-518 weighted score: the /0.2, /0.15 ... divisions, the > 0.95 halving,
//                                                       the + 0.15 bonus, the 1.8 clamp
//   services/hierarchical_similarity_service.py:575     stable sort by the enhanced score
// Everything that reads strings (uncertainty markers, chapter keywords in the query, context relevance) is computed once
// per QUERY on the host and arrives in q_params; per-ROW string facts (chapter letter, ".9" code) arrive as one tag byte.
// The arithmetic is IEEE double, one rounding per Python operator (no contraction), in the reference's evaluation order:
// results are bit-identical to the Python doubles (tests/test_gpu_parity.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace icd {

constexpr int HIER_QP = 12;        // per query: [0] uncertainty weight (0: none) [1] context relevance [2] exact-match flag [3..11] chapter boosts
constexpr int HIER_NCHAP = 9;
constexpr int HIER_MAX_K = 128;

struct HierArgs {
    const double *adj;       // [nq][k] level-reweighted scores in search order (icd_index_search_reweighted)
    const long long *ids;    // [nq][k], < 0 = no hit
    int nq, k;
    long long id_base, n_rows;
    const unsigned char *row_tags;   // [n_rows]: bits 0-3 chapter index (15 = none), bit 7 = code matches \.9\d*$
    const double *q_params;          // [nq][HIER_QP]
    double w_hb, w_em, w_sc, w_ca, w_cr, sc_value, level_term;
    int *out_order;          // [nq][k] position -> index of the hit in search order (-1 past the hits)
    double *out_enhanced;    // [nq][k] final score, best first
    double *out_score;       // [nq][k] the record's score after the uncertainty boost
    double *out_vs;          // [nq][k] vector_similarity factor
    double *out_hb;          // [nq][k] hierarchy_boost factor
    double *out_boost;       // [nq][k] uncertainty boost applied (0 = none)
};

#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void hier_rescore_kernel(HierArgs a) {
    __shared__ double sh_key[4][HIER_MAX_K];
    __shared__ double sh_val[4][HIER_MAX_K][4];
    __shared__ int sh_idx[4][HIER_MAX_K];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave;
    if (q >= a.nq) return;
    const int k = a.k;
    const double *qp = a.q_params + (size_t)q * HIER_QP;
    const double uw = qp[0], cr = qp[1];
    const bool exact = qp[2] != 0.0;
    double *key = sh_key[wave];
    int *idx = sh_idx[wave];
    // ---- uncertainty pre-pass: boosted score, stable descending re-sort (only when the query carries a marker) ----
    double score[2], boost[2];
    int src[2];
    int nvalid = 0;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int j = lane + 64 * e;
        score[e] = 0.0; boost[e] = 0.0; src[e] = -1;
        if (j < k) {
            const long long id = a.ids[(size_t)q * k + j];
            if (id >= 0) {
                const double base = a.adj[(size_t)q * k + j];
                const long long r = id - a.id_base;
                const unsigned char tag = (r >= 0 && r < a.n_rows) ? a.row_tags[r] : 15;
                double b = (tag & 0x80) ? 0.15 : 0.0;
                double s = base;
                if (uw > 0.0 && b > 0.0) s = base + b * uw; else b = 0.0;
                score[e] = s; boost[e] = b; src[e] = j;
            }
        }
        nvalid += __popcll(__ballot(src[e] >= 0));
    }
    // hits are a prefix of the k slots (invalid ids only at the end)
    auto stable_rank = [&](const double (&v)[2]) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int j = lane + 64 * e;
            if (j < nvalid) key[j] = v[e];
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        int rank[2] = {0, 0};
        for (int i = 0; i < nvalid; ++i) {
            const double ki = key[i];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int j = lane + 64 * e;
                rank[e] += (ki > v[e] || (ki == v[e] && i < j)) ? 1 : 0;
            }
        }
        __builtin_amdgcn_wave_barrier();
        return (rank[0] & 0xffff) | (rank[1] << 16);
    };
    double (*val)[4] = sh_val[wave];
    if (uw > 0.0) {
        const int pk = stable_rank(score);
        const int rk[2] = {pk & 0xffff, pk >> 16};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int j = lane + 64 * e;
            if (j < nvalid) { val[rk[e]][0] = score[e]; val[rk[e]][1] = boost[e]; idx[rk[e]] = src[e]; }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int j = lane + 64 * e;
            if (j < nvalid) { score[e] = val[j][0]; boost[e] = val[j][1]; src[e] = idx[j]; }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // ---- factors and the weighted score, candidate j of the (re-sorted) list ----
    double enh[2], vs[2], hb[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int j = lane + 64 * e;
        enh[e] = 0.0; vs[e] = 0.0; hb[e] = 0.0;
        if (j < nvalid) {
            const long long r = a.ids[(size_t)q * k + src[e]] - a.id_base;
            const unsigned char tag = (r >= 0 && r < a.n_rows) ? a.row_tags[r] : 15;
            double v = score[e];
            if (exact && v < 0.9) v = 1.0;
            double b = a.level_term;
            const int c = tag & 15;
            if (c < HIER_NCHAP) b = b + qp[3 + c] * 0.4;
            const double h = b < 0.3 ? b : 0.3;                         // min(boost, 0.3)
            const bool hp = v > 0.95;
            double extra = 0.0;
            extra = extra + h * a.w_hb / 0.2 * (hp ? 0.5 : 1.0);
            extra = extra + 0.0 * a.w_em / 0.15;                        // entity_match_score is 0 without entities
            if (a.sc_value > v) extra = extra + (a.sc_value - v) * a.w_sc / 0.08;
            extra = extra + 0.0 * a.w_ca / 0.04;                        // category_alignment is 0 without entities
            extra = extra + cr * a.w_cr / 0.03;
            if (hp) extra = extra + 0.15;
            double s = v + extra;
            s = s < 1.8 ? s : 1.8;                                      // min(base + extra, 1.8)
            if (exact) s = s > 1.5 ? s : 1.5;                           // max(score, 1.5)
            enh[e] = s; vs[e] = v; hb[e] = h;
        }
    }
    // ---- final stable descending sort by the enhanced score ----
    const int pk = stable_rank(enh);
    const int rk[2] = {pk & 0xffff, pk >> 16};
    const size_t o = (size_t)q * k;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int j = lane + 64 * e;
        if (j < nvalid) {
            const size_t w = o + rk[e];
            a.out_order[w] = src[e];
            a.out_enhanced[w] = enh[e];
            a.out_score[w] = score[e];
            a.out_vs[w] = vs[e];
            a.out_hb[w] = hb[e];
            a.out_boost[w] = boost[e];
        } else if (j < k) {
            const size_t w = o + j;
            a.out_order[w] = -1;
            a.out_enhanced[w] = -INFINITY; a.out_score[w] = -INFINITY; a.out_vs[w] = 0.0; a.out_hb[w] = 0.0; a.out_boost[w] = 0.0;
        }
    }
}
#pragma clang fp contract(fast)

// The winners of a batch for the host: what MultiDiagnosisService.match_diagnoses_batch needs of the top `kk` rescored hits of
// every query, as ONE array of doubles [8][nq][kk] (one device-to-host copy; the int64 ids travel as bit patterns in their slots - read them back as int64 -, exact for the int32 order and the
// float32 raw scores alike): 0 id, 1 raw score, 2 level-reweighted score (both of the hit the order points at), 3 order,
// 4 enhanced, 5 vector similarity, 6 hierarchy boost, 7 uncertainty boost (the last four already in final order).
struct PackWinnersArgs {
    const int *order; const long long *ids; const float *raw; const double *adj, *enh, *vs, *hb, *boost;
    int nq, k, kk;
    double *out;
};
__global__ __launch_bounds__(256) void pack_winners_kernel(PackWinnersArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long per = (long long)a.nq * a.kk;
    if (i >= per) return;
    const int q = (int)(i / a.kk), j = (int)(i - (long long)q * a.kk);
    const size_t src = (size_t)q * a.k + j;
    const int o = a.order[src];
    const size_t hit = (size_t)q * a.k + (o < 0 ? 0 : o);   // (past the hits: a valid slot, the host stops at order < 0)
    a.out[0 * per + i] = __longlong_as_double(a.ids[hit]);   // the id's BIT PATTERN in the slot (ids are arbitrary int64: id_base of a shard; a double holds integers to 2^53 only)
    a.out[1 * per + i] = (double)a.raw[hit];
    a.out[2 * per + i] = a.adj[hit];
    a.out[3 * per + i] = (double)o;
    a.out[4 * per + i] = a.enh[src];
    a.out[5 * per + i] = a.vs[src];
    a.out[6 * per + i] = a.hb[src];
    a.out[7 * per + i] = a.boost[src];
}

}  // namespace icd
