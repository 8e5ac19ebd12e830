// exact_kernel.hpp — exact FLAT / IP scoring on the fp32-input MFMA with a fused per-query top-KP.
//
// Replaces the engine behind MilvusClient.search on the FLAT/IP index
// (services/milvus_service.py:33-34,189-194,280-285). v_mfma_f32_32x32x2_f32 is bit-for-bit the
// k-ordered fmaf chain, and k = 2s + (lane>>5) is mapped to d ascending, so every score equals
// oracle/icd_oracle.c's chain_score() bit for bit.
//
// Work-group = NW waves; tile = NW*32 queries x 128 corpus rows; stages of BK = 32 (16) floats of K; the corpus rows of
// a stage go through LDS (register-staged double buffering, rows padded to BK + 1 floats: conflict-free ds_read_b32), the
// queries straight from global memory into the MFMA's operand registers.
#pragma once
#include "topk_select.hpp"

namespace icd {

struct ExactArgs {
    const float *corpus;   // [n][dim]
    const float *queries;  // [*][dim]
    const int *qlist;      // nullable: slot -> query index
    const int *nq_ptr;     // nullable: device-side query count (fallback list length)
    int nq;                // query slots (upper bound when nq_ptr is given)
    int min_active;        // with nq_ptr: run only if the device-side count exceeds this (else the streaming kernel runs)
    int n;                 // rows
    int dim;               // multiple of 32
    int P;                 // corpus chunks
    int rows_per_chunk;    // multiple of 128
    int adaptive_one_per_cu;   // with adaptive_max_p: the CU count when a CU holds two work-groups of this kernel, else 0 (exact_adaptive_chunks)
    int adaptive_max_p;    // > 0 (with nq_ptr): choose the chunk count on the device from the actual slot count:
                           // S = min(adaptive_max_p, gridDim.x / active query tiles), lists laid out [slot][S][KP]
                           // (finalize.hpp recomputes S the same way). A short flagged list then spreads over many
                           // more work-groups than the worst-case (every query flagged) chunk count allows.
    const float *thr0;     // nullable: [query] a score that k distinct rows of the corpus are known to reach (finalize.hpp: the
                           // canonical scores of the query's k best coarse candidates). The lists then start at that threshold
                           // instead of -inf: a chunk keeps only rows that can be in the top-k (ties with thr0 pass), and the
                           // compaction storms of a list's first tiles - every row passes an empty list's threshold, 32 queries
                           // are compacted every other 32-row group: more than the tiles' MFMA time - do not happen.
    int strided;           // 1: chunk c holds the rows c, c + P, c + 2 P, ... instead of a contiguous range (P = this->P, not
                           // adaptive). Neighbouring rows - an ICD family sits in code order, tools/build_database.py:156-171 - then
                           // spread evenly over a query's lists, which is what lets lists NARROWER than k be certified
                           // (finalize.hpp, narrow_check). Lists carry global row ids either way.
    float *part_scores;    // [slot][P][KP]
    int *part_rows;        // [slot][P][KP]
};

// chunk count of the adaptive layout (device: exact_topk_kernel; the same call in finalize.hpp)
// one_per_cu > 0 (kernels of which a CU holds two work-groups): when the list limit max_p caps the count somewhere between
// one and two work-groups per CU, take exactly one per CU - 320 work-groups on 256 CUs leave 64 CUs with two that share
// their MFMA pipes and finish last (measured: 0.55 against 0.45 ms for a 568-query re-search)
__host__ __device__ inline int exact_adaptive_chunks(int nq_active, int bmq, int grid, int max_p, int n_rows, int one_per_cu = 0) {
    const int mtiles = (nq_active + bmq - 1) / bmq;
    const int row_tiles = (n_rows + 127) / 128;
    int p = mtiles > 0 ? grid / mtiles : 1;
    if (p > max_p) {
        p = max_p;
        if (one_per_cu > 0 && mtiles * p > one_per_cu && mtiles <= one_per_cu) p = one_per_cu / mtiles;
    }
    p = p < row_tiles ? p : row_tiles;
    return p > 1 ? p : 1;
}

// CAPV: entries of one query's candidate buffer (<= 64 E; a compaction is due above CAPV - 32); BK: floats of K per stage;
// OCC: waves per SIMD the register budget is sized for (two work-groups of four waves per CU when their LDS fits twice).
//
// The QUERY operand never touches LDS: a wave's 32 queries are its own (no other wave reads them), so lane (c, h) loads
// the BK / 2 consecutive floats [h BK / 2, (h + 1) BK / 2) of query c for the stage straight into registers (a full
// 128-byte line per query and stage at BK = 32) and one v_permlane32_swap per register pair turns them into the MFMA's
// k-step pairs: swap(r[2 i], r[2 i + 1]) leaves (k = 2 i | 2 i + 1) in r[2 i] - pair i - and
// (k = BK / 2 + 2 i | BK / 2 + 2 i + 1) in r[2 i + 1] - pair BK / 4 + i. (Round 4; before, the queries were staged like the
// corpus rows: 34 KB of LDS, half of the scalar LDS writes of a stage, and with the 64-KB candidate buffers of k > 16 only
// two waves fitted a CU: `--mode exact` at k = 20 ran at 0.27 of the fp32 MFMA peak against 0.59 at k = 10.)
// GROUP: rows of a 32-row block tested between two overflow checks (32, or 16: two half blocks, for buffers with little room
// above KP)
template <int KP, int E, int NW, int CAPV = 64 * E, int BK = 32, int OCC = 1, int GROUP = 32>
__global__ __launch_bounds__(NW * 64, OCC) void exact_topk_kernel(ExactArgs a) {
    constexpr int BMQ = NW * 32, BN = 128, LDT = BK + 1, NT = NW * 64;
    constexpr int CAP = CAPV, LIMIT = CAP - GROUP;
    static_assert(GROUP == 32 || GROUP == 16, "a whole 32-row block or half of one per overflow check");
    static_assert(CAP <= 64 * E && LIMIT >= KP && CAP % 2 == 0, "candidate buffer: KP kept + GROUP appended per check, E keys per lane");
    static_assert(BK == 16 || BK == 32, "stage depth");
    constexpr int C4 = BK / 4;             // float4 per corpus row and stage
    constexpr int CL = (BN * C4) / NT;     // corpus float4 loads per thread per stage
    constexpr int QH = BK / 2;             // floats of a query a lane holds per stage = k-step pairs per stage
    constexpr int Q4 = QH / 4;
    static_assert(CL >= 1 && (BN * C4) % NT == 0, "stage loads divide over the threads");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *Cs = reinterpret_cast<float *>(smem);
    u64 *bufs = reinterpret_cast<u64 *>(smem + (((size_t)(2 * BN * LDT) * 4 + 15) & ~(size_t)15));

    const int nq = a.nq_ptr ? min(*a.nq_ptr, a.nq) : a.nq;
    if (a.nq_ptr && nq <= a.min_active) return;
    int P = a.P, rows_per_chunk = a.rows_per_chunk;
    if (a.nq_ptr && a.adaptive_max_p > 0) {
        P = exact_adaptive_chunks(nq, BMQ, (int)gridDim.x, a.adaptive_max_p, a.n, a.adaptive_one_per_cu);
        rows_per_chunk = (((a.n + BN - 1) / BN + P - 1) / P) * BN;
    }
    const int mtile = blockIdx.x / P, chunk = blockIdx.x % P;
    const int slot0 = mtile * BMQ;
    if (slot0 >= nq) return;
    // (strided: the tile loop, the select and the keys run over LOCAL row numbers 0 .. nloc - 1 of the chunk; local order is global order)
    const bool strided = a.strided != 0;
    const int row_begin = strided ? 0 : chunk * rows_per_chunk;
    const int row_end = strided ? (chunk < a.n ? (a.n - chunk + P - 1) / P : 0) : min(a.n, row_begin + rows_per_chunk);
    if (row_begin >= row_end) {   // (more chunks than row tiles: this work-group's lists stay empty)
        for (int b = 0; b < BMQ; ++b) {
            const int slot = slot0 + b;
            if (slot >= nq) break;
            const size_t o = ((size_t)slot * P + chunk) * KP;
            for (int j = threadIdx.x; j < KP; j += NT) { a.part_scores[o + j] = -INFINITY; a.part_rows[o + j] = -1; }
        }
        return;
    }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dim = a.dim, nks = dim / BK;
    const int h = lane >> 5, c = lane & 31;

    int cdst[CL], crow_off[CL], ccol[CL];
#pragma unroll
    for (int i = 0; i < CL; ++i) {
        const int idx = tid + NT * i;
        crow_off[i] = idx / C4;
        ccol[i] = (idx % C4) * 4;
        cdst[i] = crow_off[i] * LDT + ccol[i];
    }

    SelState st;
    const int my_slot = slot0 + wave * 32 + c;
    const bool my_valid = my_slot < nq;
    st.thr = my_valid ? -INFINITY : INFINITY;
    st.thr_row = 0u;
    st.cnt = 0;
    u64 *wbuf = bufs + (size_t)(wave * 32) * CAP;
    u64 *qbuf = wbuf + (size_t)c * CAP;

    // this lane's half of its query's stage (slots past the end read the last query; their lists are never written)
    const float *qsrc;
    {
        const int sq = min(my_slot, nq - 1);
        const int gq = a.qlist ? a.qlist[sq] : sq;
        qsrc = a.queries + (size_t)gq * dim + h * QH;
        if (a.thr0 && my_valid) {
            const float t0 = a.thr0[gq];
            if (t0 > -INFINITY) { st.thr = t0; st.thr_row = 0xFFFFFFFFu; }   // (rows that tie with it pass)
        }
    }
    float bq[QH];   // the stage's k-step pairs of the query operand: bq[s] = (k = 2 s | 2 s + 1)
    auto q_pairs = [&](const float4 (&qreg)[Q4]) {
        float r[QH];
#pragma unroll
        for (int i = 0; i < Q4; ++i) { r[4 * i] = qreg[i].x; r[4 * i + 1] = qreg[i].y; r[4 * i + 2] = qreg[i].z; r[4 * i + 3] = qreg[i].w; }
#pragma unroll
        for (int i = 0; i < QH / 2; ++i) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(r[2 * i]), __float_as_uint(r[2 * i + 1]), false, false);
            bq[i] = __uint_as_float(sw[0]);
            bq[QH / 2 + i] = __uint_as_float(sw[1]);
        }
    };

    for (int tile_row0 = row_begin; tile_row0 < row_end; tile_row0 += BN) {
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

        const float *csrc[CL];
#pragma unroll
        for (int i = 0; i < CL; ++i) {
            const int row = strided ? chunk + min(tile_row0 + crow_off[i], row_end - 1) * P : min(tile_row0 + crow_off[i], a.n - 1);
            csrc[i] = a.corpus + (size_t)row * dim + ccol[i];
        }
        float4 qreg[Q4], creg[CL];
#pragma unroll
        for (int i = 0; i < Q4; ++i) qreg[i] = *reinterpret_cast<const float4 *>(qsrc + 4 * i);
#pragma unroll
        for (int i = 0; i < CL; ++i) creg[i] = *reinterpret_cast<const float4 *>(csrc[i]);
        __syncthreads();  // previous tile's readers are done with buffer 0
#pragma unroll
        for (int i = 0; i < CL; ++i) {
            float *d = Cs + cdst[i];
            d[0] = creg[i].x; d[1] = creg[i].y; d[2] = creg[i].z; d[3] = creg[i].w;
        }
        q_pairs(qreg);
        __syncthreads();

        for (int ks = 0; ks < nks; ++ks) {
            const int cur = ks & 1;
            const bool more = ks + 1 < nks;
            if (more) {
                const int k0 = (ks + 1) * BK;
#pragma unroll
                for (int i = 0; i < Q4; ++i) qreg[i] = *reinterpret_cast<const float4 *>(qsrc + k0 + 4 * i);
#pragma unroll
                for (int i = 0; i < CL; ++i) creg[i] = *reinterpret_cast<const float4 *>(csrc[i] + k0);
            }
            const float *crow = Cs + cur * BN * LDT + c * LDT + h;
            // The corpus operands of k-step pair g + 1 are read BEFORE the eight MFMAs of pair g are issued: read right in
            // front of their MFMAs (what hipcc makes of the plain loop) every group waits out an LDS round trip with the pipe
            // draining. Two register sets, constant indices.
            float av[2][2][4];
            auto load_pair = [&](int g, int set) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int t = 0; t < 4; ++t) av[set][j][t] = crow[t * 32 * LDT + 2 * (2 * g + j)];
            };
            load_pair(0, 0);
#pragma unroll
            for (int g = 0; g < BK / 4; ++g) {
                if (g + 1 < BK / 4) load_pair(g + 1, (g + 1) & 1);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][j][t], bq[2 * g + j], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);    // the next pair's eight reads first ...
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);    // ... then this pair's MFMAs
            }
            if (more) {
                float *cd = Cs + (cur ^ 1) * BN * LDT;
#pragma unroll
                for (int i = 0; i < CL; ++i) {
                    float *d = cd + cdst[i];
                    d[0] = creg[i].x; d[1] = creg[i].y; d[2] = creg[i].z; d[3] = creg[i].w;
                }
                q_pairs(qreg);   // (the MFMAs above have been issued: in-order issue, their operands are read)
            }
            __syncthreads();
        }

        // fused select
        const bool partial = tile_row0 + BN > row_end;
        if (KP <= 32 && tile_row0 == row_begin && !partial) {
            // Threshold bootstrap of a chunk's first tile. Starting at -inf every row of the first tiles is appended and
            // each 16-register group ends in a compaction of all 32 queries (~1 ms per chunk: more than a 9-tile
            // chunk's MFMAs). Exact lists need a threshold that at least KP rows reach: every lane keeps the KP / 2 best
            // of its 64 scores (branch-free insertion, 2 VALU per kept score and score, once per chunk) and the smaller of the
            // two lanes' (KP / 2)-th best is reached by KP distinct rows of this tile. Ties with it pass (thr_row = max).
            // (KP = 32, round 5: the narrow lists of k > 32 run 22-tile chunks: -inf starts cost them a fifth of their time)
            constexpr int BD = KP / 2;
            float m[BD];
#pragma unroll
            for (int j = 0; j < BD; ++j) m[j] = -INFINITY;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[t][r];
#pragma unroll
                    for (int j = 0; j < BD; ++j) {
                        const float lo = fminf(m[j], v);
                        m[j] = fmaxf(m[j], v);
                        v = lo;
                    }
                }
            const float other = __shfl_xor(m[BD - 1], 32);
            const float t0 = fminf(m[BD - 1], other);
            if (my_valid && t0 > st.thr) { st.thr = t0; st.thr_row = 0xFFFFFFFFu; }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t row0 = (uint32_t)(tile_row0 + t * 32);
            if (partial) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (int)row0 + 4 * h + (r & 3) + 8 * (r >> 2);
                    if (row >= row_end) acc[t][r] = __builtin_nanf("");
                }
            }
            if constexpr (GROUP == 32) {
                filter16<true>(acc[t], row0, st, qbuf, lane);
                if (__any(st.cnt > LIMIT)) compact_wave<KP, E, CAP, GROUP>(wbuf, st, lane, false);
            } else {
                filter16<true, 0, 8>(acc[t], row0, st, qbuf, lane);
                if (__any(st.cnt > LIMIT)) compact_wave<KP, E, CAP, GROUP>(wbuf, st, lane, false);
                filter16<true, 8, 16>(acc[t], row0, st, qbuf, lane);
                if (__any(st.cnt > LIMIT)) compact_wave<KP, E, CAP, GROUP>(wbuf, st, lane, false);
            }
        }
    }

    // final: sorted top-KP of every query of this wave -> partial list
    compact_wave<KP, E, CAP, GROUP>(wbuf, st, lane, true);
    for (int b = 0; b < 32; ++b) {
        const int slot = slot0 + wave * 32 + b;
        if (slot >= nq) break;
        const int nb = min(readlane<int>(st.cnt, b), KP);
        const u64 *qb = wbuf + (size_t)b * CAP;
        const size_t o = ((size_t)slot * P + chunk) * KP;
        for (int j = lane; j < KP; j += 64) {
            float s = -INFINITY;
            int row = -1;
            if (j < nb) {
                const u64 k = qb[j];
                s = key_score(k);
                row = (int)key_row(k);
                if (strided) row = chunk + row * P;
            }
            a.part_scores[o + j] = s;
            a.part_rows[o + j] = row;
        }
    }
}

template <int KP, int E, int NW, int CAPV = 64 * E, int BK = 32>
inline size_t exact_lds_bytes() {
    constexpr int BMQ = NW * 32, BN = 128, LDT = BK + 1;
    size_t stage = (((size_t)(2 * BN * LDT) * 4 + 15) & ~(size_t)15);
    return stage + (size_t)BMQ * CAPV * 8;
}

}  // namespace icd
