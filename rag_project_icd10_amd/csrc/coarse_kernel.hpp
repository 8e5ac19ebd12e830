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
#include <type_traits>

#include "topk_select.hpp"

namespace icd {

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E) — keeps every accumulator
// index a constant so the arrays stay in registers
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int CO_KP = 16;
constexpr int CO_BM = 128;
constexpr int CO_BN = 128;
constexpr int CO_BK = 64;
constexpr int CO_S = 4;
constexpr int CO_STAGE_BYTES = CO_BN * CO_BK * 2;  // 16384
constexpr int CO_RING_BYTES = CO_S * CO_STAGE_BYTES;
constexpr int CO_CAP = 64;
constexpr int CO_LDS_BYTES = CO_RING_BYTES + CO_BM * CO_CAP * 8 + 4 * 256;

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
    unsigned long long *dbg; // diagnostic builds only: [block][wave][4] cycle sums
};

// VAR (diagnostic ablations, timing only; VAR = 0 is the product kernel):
//   1 skip the fused select   2 skip the MFMAs   4 no vmcnt wait before the barrier   8 s_memtime stamps
#define ICD_STAMP(t) do { __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); \
    __builtin_amdgcn_sched_barrier(0); } while (0)

#define ICD_GLDS16(gptr, lptr)                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),       \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

template <int D, int VAR>
__global__ __launch_bounds__(256, 1) void coarse_topk_kernel(CoarseArgs a) {
    constexpr int KS = D / CO_BK;      // stages per tile
    constexpr int NF = D / 16;         // query fragments per lane
    constexpr int NKSTEP = KS * 4;     // k16-steps per tile
    constexpr bool OVERLAP = (VAR & 16) != 0;  // filter tile t inside the MFMA stream of tile t+1
    static_assert(KS % CO_S == 0, "ring slot must be a compile-time function of the stage");
    using Ops = Sel2Ops<CO_KP>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

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

    // ---- select state ---------------------------------------------------------------------------------
    const uint32_t wave_qbase = (uint32_t)CO_RING_BYTES + (uint32_t)(wave * 32) * Ops::QBYTES;
    const uint32_t wave_scratch = (uint32_t)CO_RING_BYTES + (uint32_t)CO_BM * Ops::QBYTES + (uint32_t)wave * 256u;
    Sel2 st;
    Ops::init(st, wave_qbase + (uint32_t)c * Ops::QBYTES, h, (slot0 + wave * 32 + c) < a.nq);

    // issue the 4 pieces of one stage
    // VAR & 256: buffer_load ... lds with the per-lane part in voffset and the tile/stage part in a
    // scalar soffset (no per-piece 64-bit VALU address arithmetic)
    const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(cbase) + (size_t)row_begin * (size_t)(D * 2), 0,
        (int)min((size_t)(a.n_pad - row_begin) * (size_t)(D * 2), (size_t)0x7FFFFFFF), 0x00020000);
    auto issue_stage = [&](int g_tile, int g_ks, int ring_slot) {
        if constexpr ((VAR & 128) != 0) return;  // timing-only ablation: no DMA at all
        int trow0 = row_begin + g_tile * CO_BN;
        trow0 = min(trow0, last_tile_row0);  // stages past the sweep re-read valid memory, never consumed
        char *dst = smem + ring_slot * CO_STAGE_BYTES + wave * 4096;
        if constexpr ((VAR & 256) != 0) {
            const uint32_t soff = (uint32_t)(trow0 - row_begin) * (uint32_t)(D * 2) + (uint32_t)g_ks * (CO_BK * 2);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, (__attribute__((address_space(3))) void *)(dst + i * 1024),
                                                         16, src_off[i], soff, 0, 0);
        } else {
            const char *src = cbase + (size_t)trow0 * (size_t)(D * 2) + (size_t)g_ks * (CO_BK * 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) ICD_GLDS16(src + src_off[i], dst + i * 1024);
        }
    };

    // filter one register of a finished tile (flat index F = 16 t + r). GUARD: rows >= n never pass
    // (only the last tile of the corpus has such rows).
    auto filter_reg = [&](const f32x16 (&pa)[4], auto F, uint32_t rowbase, auto GUARD) {
        constexpr int f = decltype(F)::value;
        constexpr int t = f >> 4, r = f & 15;
        constexpr uint32_t roff = (uint32_t)(t * 32 + (r & 3) + 8 * (r >> 2));
        float v = pa[t][r];
        if constexpr (decltype(GUARD)::value) {
            if ((int)(rowbase + roff) >= a.n) v = -INFINITY;
        }
        if (v > st.thr) {
            *reinterpret_cast<float *>(smem + st.aw) = v;
            *reinterpret_cast<uint32_t *>(smem + st.aw + Ops::ROW_OFF) = rowbase + roff;
            st.aw += st.inc;
        }
        if constexpr (r == 15) Ops::check(st, lane, smem, wave_qbase, wave_scratch, false);
    };
    auto filter_tile = [&](const f32x16 (&pa)[4], int tile_row0) {
        const uint32_t rowbase = (uint32_t)(tile_row0 + 4 * h);
        if (tile_row0 + CO_BN > a.n) static_for<0, 64>([&](auto F) { filter_reg(pa, F, rowbase, std::true_type{}); });
        else static_for<0, 64>([&](auto F) { filter_reg(pa, F, rowbase, std::false_type{}); });
    };

    // prologue: stages 0..S-2
#pragma unroll
    for (int p = 0; p < CO_S - 1; ++p) issue_stage(p / KS, p % KS, p % CO_S);

    unsigned long long t_wait = 0, t_body = 0, t_epi = 0, t0 = 0, t1 = 0, t2 = 0;
    f32x16 pacc[4];  // previous tile's scores (OVERLAP)
    if (OVERLAP) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) pacc[t][r] = -INFINITY;
    }
    for (int tile = 0; tile < ntiles; ++tile) {
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        const int tile_row0 = row_begin + tile * CO_BN;

        static_for<0, KS>([&](auto KSI) {
            constexpr int ks = decltype(KSI)::value;
            constexpr int AHEAD = CO_S - 1;
            if (VAR & 8) ICD_STAMP(t0);
            // stage (tile,ks) landed for this wave when all but the youngest 2 stages are done
            if (VAR & 4) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (VAR & 8) ICD_STAMP(t1);
            {
                constexpr int nks = ks + AHEAD;
                const int ntile = tile + (nks >= KS ? 1 : 0);
                issue_stage(ntile, nks % KS, nks % CO_S);
            }
            const char *sbase = smem + (ks % CO_S) * CO_STAGE_BYTES;
            if constexpr ((VAR & 64) != 0) {
                // whole-stage prefetch: all 16 A fragments in flight before the first MFMA
                half8 af[16];
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    af[i] = *reinterpret_cast<const half8 *>(sbase + (i & 3) * 4096 + rd_off[i >> 2]);
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], qf[ks * 4 + (i >> 2)], acc[i & 3], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            } else {
            // fragment reads run one k-step ahead of the MFMAs that consume them
            half8 af[2][4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                af[0][t] = *reinterpret_cast<const half8 *>(sbase + t * 4096 + rd_off[0]);
            static_for<0, 4>([&](auto SI) {
                constexpr int s = decltype(SI)::value;
                if constexpr (s + 1 < 4) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        af[(s + 1) & 1][t] = *reinterpret_cast<const half8 *>(sbase + t * 4096 + rd_off[s + 1]);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (VAR & 2) asm volatile("" ::"v"(af[s & 1][t]), "v"(qf[ks * 4 + s]));
                    else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s & 1][t], qf[ks * 4 + s], acc[t], 0, 0, 0);
                }
                if constexpr (OVERLAP && !(VAR & 1)) {
                    // the previous tile's select rides in the gaps of this tile's MFMAs
                    constexpr int j = ks * 4 + s;
                    constexpr int f0 = (j * 64) / NKSTEP, f1 = ((j + 1) * 64) / NKSTEP;
                    static_for<f0, f1>([&](auto F) { filter_reg(pacc, F, (uint32_t)(tile_row0 - CO_BN + 4 * h), std::true_type{}); });
                }
            });
            if constexpr ((VAR & 32) != 0) {
                // pin the order: reads of k-step s+1 are issued before the MFMAs of k-step s
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            }
            if (VAR & 8) { ICD_STAMP(t2); t_wait += t1 - t0; t_body += t2 - t1; }
        });
        if (VAR & 8) ICD_STAMP(t0);

        if (VAR & 1) {
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[t][r];
            if (sum == 1.2345e30f) st.kept = 1;  // keeps the MFMAs live, never true
        } else if (OVERLAP) {
#pragma unroll
            for (int t = 0; t < 4; ++t) pacc[t] = acc[t];
        } else {
            filter_tile(acc, tile_row0);
        }
        if (VAR & 8) { ICD_STAMP(t1); t_epi += t1 - t0; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the run-ahead stages
    if (OVERLAP && !(VAR & 1)) filter_tile(pacc, row_begin + (ntiles - 1) * CO_BN);
    if ((VAR & 8) && a.dbg && lane == 0) {
        unsigned long long *d = a.dbg + ((size_t)blockIdx.x * 4 + wave) * 4;
        d[0] = t_wait; d[1] = t_body; d[2] = t_epi; d[3] = (unsigned long long)ntiles;
    }

    // ---- final: sorted top-KP of every query of this wave -> partial list ---------------------------
    Ops::check(st, lane, smem, wave_qbase, wave_scratch, true);
    for (int b = 0; b < 32; ++b) {
        const int slot = slot0 + wave * 32 + b;
        if (slot >= a.nq) break;
        const int nb = readlane<int>(st.kept, b);
        const uint32_t qb = wave_qbase + (uint32_t)b * Ops::QBYTES;
        const size_t o = ((size_t)slot * a.P + chunk) * CO_KP;
        if (lane < CO_KP) {
            float s = -INFINITY;
            int row = -1;
            if (lane < nb) {
                s = *reinterpret_cast<const float *>(smem + qb + lane * 4);
                row = (int)*reinterpret_cast<const uint32_t *>(smem + qb + Ops::ROW_OFF + lane * 4);
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
