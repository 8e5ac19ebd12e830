// stream_kernel.hpp — exact FLAT / IP search for SPARSE query sets: the corpus is streamed once per
// pass of ST_QB queries and every lane walks the canonical fp32 fmaf chain of its own row.
//
// Used where the MFMA kernels are the wrong tool (bound: HBM / Infinity-Cache bandwidth, not MFMA):
//   * the AUTO fallback when only a few queries failed certification (the fp32-MFMA exact kernel
//     would sweep the whole corpus with 128-query tiles for them: 70 ms for 5 queries on a 1.25 M-row
//     shard, profiles/r01_sizes_before_pmin2.log);
//   * the reference's own call shape, one query per call (services/milvus_service.py:280-285),
//     and any small batch / large k.
//
// Work-group = 4 waves over a contiguous row range; a wave takes 64 rows at a time (lane = row),
// 32-float slices are loaded as full 128-B lines (8 lanes per row), staged in a wave-private LDS
// tile (row stride 36 floats: conflict-free ds_read_b128) and consumed d-ascending, so each
// (row, query) score is bit-identical to oracle/icd_oracle.c chain_score(). Selection is the exact
// rule (score desc, row asc) with a per-query threshold and the u64-key compaction of topk_select.hpp.
// The candidate buffer of a (wave, query) holds 64 E entries with E one larger than the exact MFMA
// kernel uses for the same KP: a 64-row step can append 64 entries at once.
// Output: one best-first list per (query slot, wave) -> reduce_lists_kernel -> finalize_kernel<false>.
#pragma once
#include "topk_select.hpp"

namespace icd {

constexpr int ST_QB = 8;          // queries per pass (template QB <= ST_QB: fewer for tiny batches)
constexpr int ST_TS = 36;         // tile row stride in floats (32 + 4)
constexpr int ST_PF = 1;          // 32-float slices prefetched per lane
constexpr int ST_MAX_ACTIVE = 64; // the sparse path is taken for at most this many queries

struct StreamArgs {
    const float *corpus;
    const float *queries;
    const int *qlist;     // nullable: slot -> query index
    const int *nq_ptr;    // nullable: device-side number of slots
    int nq;               // slots (upper bound when nq_ptr is given)
    int max_active;       // run only if the actual slot count is <= this (else the MFMA kernel runs)
    int n, dim;           // dim multiple of 32
    int rows_per_wg;      // multiple of 256
    int nwg;              // work-groups = row ranges
    float *list_scores;   // [slot][4 * nwg][KP]
    int *list_rows;
};

template <int KP, int E, int QB>
__host__ __device__ constexpr size_t stream_lds_bytes(int dim) {
    return (size_t)QB * dim * 4 + (size_t)4 * 64 * ST_TS * 4 + (size_t)4 * QB * 64 * E * 8;
}

template <int KP, int E, int QB>
__global__ __launch_bounds__(256) void stream_topk_kernel(StreamArgs a) {
    constexpr int ST_QB = QB;   // shadows the namespace constant inside the kernel
    constexpr int CAP = 64 * E;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nq = a.nq_ptr ? min(*a.nq_ptr, a.nq) : a.nq;
    if (nq <= 0 || nq > a.max_active) return;   // work-group-uniform
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dim = a.dim, nsl = dim >> 5;
    float *qs = reinterpret_cast<float *>(smem);                                   // [ST_QB][dim]
    float *tile = qs + (size_t)ST_QB * dim + (size_t)wave * 64 * ST_TS;            // wave-private
    u64 *bufs = reinterpret_cast<u64 *>(smem + (size_t)ST_QB * dim * 4 + (size_t)4 * 64 * ST_TS * 4) +
                (size_t)wave * ST_QB * CAP;                                        // [ST_QB][CAP]
    const int row_begin = blockIdx.x * a.rows_per_wg;
    const int row_end = min(a.n, row_begin + a.rows_per_wg);
    const int nlists = 4 * a.nwg;

    for (int q0 = 0; q0 < nq; q0 += ST_QB) {
        const int nqp = min(ST_QB, nq - q0);
        __syncthreads();   // previous pass is done with qs
        for (int i = tid * 4; i < ST_QB * dim; i += 1024) {
            const int qi = i / dim, d = i - qi * dim;
            const int slot = q0 + min(qi, nqp - 1);   // unused slots repeat the last query (never emitted)
            const int gq = a.qlist ? a.qlist[slot] : slot;
            *reinterpret_cast<float4 *>(qs + i) = *reinterpret_cast<const float4 *>(a.queries + (size_t)gq * dim + d);
        }
        __syncthreads();
        float thr[ST_QB];
        uint32_t thr_row[ST_QB];
        int cnt[ST_QB];
#pragma unroll
        for (int qi = 0; qi < ST_QB; ++qi) { thr[qi] = -INFINITY; thr_row[qi] = 0u; cnt[qi] = 0; }

        for (int r0 = row_begin + wave * 64; r0 < row_end; r0 += 256) {
            float acc[ST_QB];
#pragma unroll
            for (int qi = 0; qi < ST_QB; ++qi) acc[qi] = 0.0f;
            // this lane fetches piece (lane & 7) of rows r0 + 8 i + (lane >> 3), i = 0..7. Named scalars,
            // not arrays: hipcc keeps arrays of float4 / pointers in scratch here, and a scratch round
            // trip per slice is slower than the HBM stream this kernel is meant to be bound by.
#define ICD_ROWPTR(i) (a.corpus + (size_t)min(r0 + (i) * 8 + (lane >> 3), a.n - 1) * dim + (lane & 7) * 4)
            const float *s0 = ICD_ROWPTR(0), *s1 = ICD_ROWPTR(1), *s2 = ICD_ROWPTR(2), *s3 = ICD_ROWPTR(3);
            const float *s4 = ICD_ROWPTR(4), *s5 = ICD_ROWPTR(5), *s6 = ICD_ROWPTR(6), *s7 = ICD_ROWPTR(7);
#undef ICD_ROWPTR
#define ICD_LD(p, o) (*reinterpret_cast<const float4 *>((p) + (o)))
            float4 p0 = ICD_LD(s0, 0), p1 = ICD_LD(s1, 0), p2 = ICD_LD(s2, 0), p3 = ICD_LD(s3, 0);
            float4 p4 = ICD_LD(s4, 0), p5 = ICD_LD(s5, 0), p6 = ICD_LD(s6, 0), p7 = ICD_LD(s7, 0);
            float *wdst = tile + (lane >> 3) * ST_TS + (lane & 7) * 4;
            for (int s = 0; s < nsl; ++s) {
                *reinterpret_cast<float4 *>(wdst + 0 * 8 * ST_TS) = p0;
                *reinterpret_cast<float4 *>(wdst + 1 * 8 * ST_TS) = p1;
                *reinterpret_cast<float4 *>(wdst + 2 * 8 * ST_TS) = p2;
                *reinterpret_cast<float4 *>(wdst + 3 * 8 * ST_TS) = p3;
                *reinterpret_cast<float4 *>(wdst + 4 * 8 * ST_TS) = p4;
                *reinterpret_cast<float4 *>(wdst + 5 * 8 * ST_TS) = p5;
                *reinterpret_cast<float4 *>(wdst + 6 * 8 * ST_TS) = p6;
                *reinterpret_cast<float4 *>(wdst + 7 * 8 * ST_TS) = p7;
                if (s + 1 < nsl) {
                    const int o = (s + 1) * 32;
                    p0 = ICD_LD(s0, o); p1 = ICD_LD(s1, o); p2 = ICD_LD(s2, o); p3 = ICD_LD(s3, o);
                    p4 = ICD_LD(s4, o); p5 = ICD_LD(s5, o); p6 = ICD_LD(s6, o); p7 = ICD_LD(s7, o);
                }
                const float4 *c4 = reinterpret_cast<const float4 *>(tile + lane * ST_TS);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4 cv = c4[j];
#pragma unroll
                    for (int qi = 0; qi < ST_QB; ++qi) {
                        const float4 qv = *reinterpret_cast<const float4 *>(qs + qi * dim + s * 32 + j * 4);  // broadcast
                        acc[qi] = __builtin_fmaf(qv.x, cv.x, acc[qi]);
                        acc[qi] = __builtin_fmaf(qv.y, cv.y, acc[qi]);
                        acc[qi] = __builtin_fmaf(qv.z, cv.z, acc[qi]);
                        acc[qi] = __builtin_fmaf(qv.w, cv.w, acc[qi]);
                    }
                }
            }
#undef ICD_LD
            // exact select: (score desc, row asc); NaN and -inf never enter
            const uint32_t row = (uint32_t)(r0 + lane);
            const bool rvalid = (int)row < row_end;
#pragma unroll
            for (int qi = 0; qi < ST_QB; ++qi) {
                const float v = acc[qi];
                const bool pass = rvalid && (v > thr[qi] || (v == thr[qi] && row < thr_row[qi])) && v != -INFINITY;
                const u64 m = __ballot(pass);
                if (m) {
                    u64 *qb = bufs + (size_t)qi * CAP;
                    if (pass) qb[cnt[qi] + __popcll(m & ((1ull << lane) - 1ull))] = make_key(v, row);
                    cnt[qi] += __popcll(m);
                    if (cnt[qi] > CAP - 64) {
                        u64 kth;
                        compact_one<KP, E>(qb, cnt[qi], lane, kth);
                        if (cnt[qi] >= KP) {
                            thr[qi] = key_score(kth);
                            thr_row[qi] = key_row(kth);
                            cnt[qi] = KP;
                        }
                    }
                }
            }
        }
        // this wave's best-first list of every query of the pass
#pragma unroll
        for (int qi = 0; qi < ST_QB; ++qi) {
            if (qi < nqp) {
                u64 *qb = bufs + (size_t)qi * CAP;
                u64 kth;
                if (cnt[qi] > 0) compact_one<KP, E>(qb, cnt[qi], lane, kth);
                const int nb = min(cnt[qi], KP);
                const size_t o = ((size_t)(q0 + qi) * nlists + (size_t)blockIdx.x * 4 + wave) * KP;
                for (int j = lane; j < KP; j += 64) {
                    float s = -INFINITY;
                    int r = -1;
                    if (j < nb) { const u64 k = qb[j]; s = key_score(k); r = (int)key_row(k); }
                    a.list_scores[o + j] = s;
                    a.list_rows[o + j] = r;
                }
            }
        }
    }
}

// One wave per (slot, output list g): merge lists g, g + P_out, g + 2 P_out, ... of the slot's nlists
// best-first lists (up to 512 candidates per wave) into the best KP. Output layout [slot][P_out][KP]
// = what finalize_kernel<false> reads.
struct ReduceArgs {
    const float *list_scores;
    const int *list_rows;
    int nlists, KP, P_out;
    const int *nq_ptr;
    int nq, max_active;
    float *part_scores;
    int *part_rows;
};

__global__ __launch_bounds__(256) void reduce_lists_kernel(ReduceArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nq = a.nq_ptr ? min(*a.nq_ptr, a.nq) : a.nq;
    if (nq <= 0 || nq > a.max_active) return;
    const int w = blockIdx.x * 4 + wave;
    const int slot = w / a.P_out, g = w - slot * a.P_out;
    if (slot >= nq) return;
    u64 *keys = reinterpret_cast<u64 *>(smem) + (size_t)wave * 512;
    const int per = (a.nlists + a.P_out - 1) / a.P_out;   // lists merged by this wave
    const int ncand = per * a.KP;                          // <= 512 (checked on the host)
    u64 key[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int i = lane + 64 * e;
        key[e] = 0ull;
        if (i < ncand) {
            const int li = g + (i / a.KP) * a.P_out, j = i % a.KP;
            if (li < a.nlists) {
                const size_t src = ((size_t)slot * a.nlists + li) * a.KP + j;
                const int row = a.list_rows[src];
                if (row >= 0) key[e] = make_key(a.list_scores[src], (uint32_t)row);
            }
            keys[i] = key[e];
        }
    }
    int rank[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int ef = (ncand + 63) >> 6;
    for (int j = 0; j < ncand; ++j) {
        const u64 kj = keys[j];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (e < ef) rank[e] += (kj > key[e]) ? 1 : 0;
    }
    int nvalid = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) nvalid += __popcll(__ballot(key[e] != 0ull));
    const size_t o = ((size_t)slot * a.P_out + g) * a.KP;
    for (int j = lane; j < a.KP; j += 64)
        if (j >= nvalid) { a.part_scores[o + j] = -INFINITY; a.part_rows[o + j] = -1; }
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if (key[e] != 0ull && rank[e] < a.KP) {
            a.part_scores[o + rank[e]] = key_score(key[e]);
            a.part_rows[o + rank[e]] = (int)key_row(key[e]);
        }
}

}  // namespace icd
