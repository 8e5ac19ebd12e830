// coarse_kernel.hpp — fp16 MFMA scoring (v_mfma_f32_32x32x16_f16) with a fused per-query top-KP.
//
// The dominant kernel of the hot path: replaces the FLAT/IP scan behind MilvusClient.search
// (services/milvus_service.py:280-285) for query batches. Its output is a candidate list per
// (query, corpus chunk); finalize.hpp certifies and rescoring restores exact fp32 results.
//
// Geometry
//   work-group  = 4 waves, one per SIMD, 128 queries (32 per wave, one query column per lane pair)
//   queries     = B operand, held in registers for the whole sweep (D/16 fragments of 8 halves)
//   corpus rows = A operand, 128-row tiles streamed through an LDS ring by LDS-DMA:
//                 stage = 128 rows x 64 halves (16 KiB), 4 ring slots, 16 one-KiB pieces per stage
//                 (4 per wave), each piece = 8 rows x one full 128-B line
//   swizzle     = LDS slot (row, p) holds 16-B piece p ^ ((row>>1)&7) of the row's 128-B segment:
//                 applied on the DMA SOURCE address and on the ds_read_b128 address (the LDS
//                 destination of an LDS-DMA is lane-linear); A-fragment reads are conflict-free.
#pragma once
#include "topk_select.hpp"

namespace icd {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int CO_KP = 16;
constexpr int CO_BM = 128;
constexpr int CO_BN = 128;
constexpr int CO_BK = 64;
constexpr int CO_S = 4;
constexpr int CO_STAGE_BYTES = CO_BN * CO_BK * 2;  // 16384
constexpr int CO_RING_BYTES = CO_S * CO_STAGE_BYTES;
constexpr int CO_CAP = 64;
constexpr int CO_LDS_BYTES = CO_RING_BYTES + CO_BM * CO_CAP * 8;

struct CoarseArgs {
    const _Float16 *q16;     // [nq_pad][D], rows >= nq are zero
    const _Float16 *c16;     // [n_pad][D], rows >= n are zero
    int nq;
    int n;                   // valid rows
    int n_pad;               // multiple of 128
    int P;                   // corpus chunks
    int rows_per_chunk;      // multiple of 128
    float *part_scores;      // [nq][P][KP]
    int *part_rows;
};

#define ICD_GLDS16(gptr, lptr)                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),       \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

template <int D>
__global__ __launch_bounds__(256, 1) void coarse_topk_kernel(CoarseArgs a) {
    constexpr int KS = D / CO_BK;      // stages per tile
    constexpr int NF = D / 16;         // query fragments per lane
    static_assert(KS % CO_S == 0, "ring slot must be a compile-time function of the stage");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64 *bufs = reinterpret_cast<u64 *>(smem + CO_RING_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int mtile = blockIdx.x / a.P, chunk = blockIdx.x % a.P;
    const int slot0 = mtile * CO_BM;
    const int row_begin = chunk * a.rows_per_chunk;
    const int row_end = min(a.n_pad, row_begin + a.rows_per_chunk);  // multiple of 128
    const int ntiles = (row_end - row_begin) / CO_BN;
    if (ntiles <= 0) return;

    // ---- query fragments -> registers (B operand: lane holds Q[query c][16 s + 8 h + j]) ----------
    half8 qf[NF];
    {
        const _Float16 *qrow = a.q16 + (size_t)(slot0 + wave * 32 + c) * D + 8 * h;
#pragma unroll
        for (int s = 0; s < NF; ++s) qf[s] = *reinterpret_cast<const half8 *>(qrow + 16 * s);
    }

    // ---- DMA source offsets (bytes from the tile's first row, k = 0) --------------------------------
    uint32_t src_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int blk = wave * 4 + i;
        const int row_local = blk * 8 + (lane >> 3);
        const int piece = (lane & 7) ^ ((row_local >> 1) & 7);
        src_off[i] = (uint32_t)row_local * (uint32_t)(D * 2) + (uint32_t)piece * 16u;
    }
    const char *cbase = reinterpret_cast<const char *>(a.c16);
    const int last_tile_row0 = a.n_pad - CO_BN;

    // ---- A-fragment LDS read offsets ----------------------------------------------------------------
    uint32_t rd_off[4];
    {
        const int sw = (c >> 1) & 7;
#pragma unroll
        for (int s = 0; s < 4; ++s) rd_off[s] = (uint32_t)c * 128u + (uint32_t)(((2 * s + h) ^ sw) * 16);
    }

    SelState st;
    st.thr = (slot0 + wave * 32 + c) < a.nq ? -INFINITY : INFINITY;
    st.thr_row = 0u;
    st.cnt = 0;
    u64 *wbuf = bufs + (size_t)(wave * 32) * CO_CAP;
    u64 *qbuf = wbuf + (size_t)c * CO_CAP;

    // issue the 4 pieces of global stage g (g counts stages over the whole sweep)
    auto issue_stage = [&](int g_tile, int g_ks, int ring_slot) {
        int trow0 = row_begin + g_tile * CO_BN;
        trow0 = min(trow0, last_tile_row0);  // stages past the sweep re-read valid memory, never consumed
        const char *src = cbase + (size_t)trow0 * (size_t)(D * 2) + (size_t)g_ks * (CO_BK * 2);
        char *dst = smem + ring_slot * CO_STAGE_BYTES + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) ICD_GLDS16(src + src_off[i], dst + i * 1024);
    };

    // prologue: stages 0..S-2
#pragma unroll
    for (int p = 0; p < CO_S - 1; ++p) issue_stage(p / KS, p % KS, p % CO_S);

    for (int tile = 0; tile < ntiles; ++tile) {
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            constexpr int AHEAD = CO_S - 1;
            // stage (tile,ks) landed for this wave when all but the youngest 2 stages are done
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            {
                const int nks = ks + AHEAD;
                const int ntile = tile + (nks >= KS ? 1 : 0);
                issue_stage(ntile, nks % KS, nks % CO_S);
            }
            const char *sbase = smem + (ks % CO_S) * CO_STAGE_BYTES;
            // fragment reads run one k-step ahead of the MFMAs that consume them
            half8 af[2][4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                af[0][t] = *reinterpret_cast<const half8 *>(sbase + t * 4096 + rd_off[0]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s + 1 < 4) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        af[(s + 1) & 1][t] = *reinterpret_cast<const half8 *>(sbase + t * 4096 + rd_off[s + 1]);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s & 1][t], qf[ks * 4 + s], acc[t], 0, 0, 0);
            }
        }

        // ---- fused select on the finished 128-row tile ---------------------------------------------
        const int tile_row0 = row_begin + tile * CO_BN;
        const bool partial = tile_row0 + CO_BN > a.n;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t row0 = (uint32_t)(tile_row0 + t * 32);
            if (partial) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (int)row0 + 4 * h + (r & 3) + 8 * (r >> 2);
                    if (row >= a.n) acc[t][r] = __builtin_nanf("");
                }
            }
            filter16<false>(acc[t], row0, st, qbuf, lane);
            if (__any(st.cnt > CO_CAP - 32)) compact_wave<CO_KP, 1>(wbuf, st, lane, false);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the run-ahead stages

    compact_wave<CO_KP, 1>(wbuf, st, lane, true);
    for (int b = 0; b < 32; ++b) {
        const int slot = slot0 + wave * 32 + b;
        if (slot >= a.nq) break;
        const int nb = min(readlane<int>(st.cnt, b), CO_KP);
        const u64 *qb = wbuf + (size_t)b * CO_CAP;
        const size_t o = ((size_t)slot * a.P + chunk) * CO_KP;
        if (lane < CO_KP) {
            float s = -INFINITY;
            int row = -1;
            if (lane < nb) {
                const u64 k = qb[lane];
                s = key_score(k);
                row = (int)key_row(k);
            }
            a.part_scores[o + lane] = s;
            a.part_rows[o + lane] = row;
        }
    }
}

// ---- fp32 -> fp16 images ---------------------------------------------------------------------------
// One wave per row: converts with round-to-nearest-even, accumulates the row's squared norm in fp32,
// flags rows whose fp16 image is unusable (non-finite input or |x| > 65504).
struct ConvertArgs {
    const float *src;        // [rows][dim]
    _Float16 *dst;           // [rows_pad][dim]
    int rows, rows_pad, dim;
    float *norm;             // nullable [rows]: L2 norm rounded up
    unsigned char *bad;      // nullable [rows]
    unsigned int *rmax_bits; // nullable: atomicMax of norm bits (norm >= 0)
    unsigned int *any_bad;   // nullable: set to 1 if any row is bad
};

__global__ __launch_bounds__(256) void convert_rows_kernel(ConvertArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= a.rows_pad) return;
    _Float16 *d = a.dst + (size_t)row * a.dim;
    if (row >= a.rows) {
        for (int i = lane * 4; i < a.dim; i += 256) {
            d[i] = (_Float16)0.0f; d[i + 1] = (_Float16)0.0f; d[i + 2] = (_Float16)0.0f; d[i + 3] = (_Float16)0.0f;
        }
        return;
    }
    const float *s = a.src + (size_t)row * a.dim;
    float ss = 0.0f;
    bool bad = false;
    for (int i = lane * 4; i < a.dim; i += 256) {
        const float4 v = *reinterpret_cast<const float4 *>(s + i);
        const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bad |= !(fabsf(f[j]) <= 65504.0f);
            ss = __builtin_fmaf(f[j], f[j], ss);
            d[i + j] = (_Float16)f[j];
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off);
    const bool anybad = __any(bad);
    if (lane == 0) {
        float nrm = sqrtf(ss) * 1.000001f;  // round up: it multiplies an error bound
        if (!(nrm == nrm)) nrm = INFINITY;
        if (a.norm) a.norm[row] = nrm;
        if (a.bad) a.bad[row] = anybad ? 1 : 0;
        if (a.rmax_bits) atomicMax(a.rmax_bits, __float_as_uint(nrm));
        if (a.any_bad && anybad) atomicOr(a.any_bad, 1u);
    }
}

}  // namespace icd
