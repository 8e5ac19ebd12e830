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

#ifndef ICD_CO_KP
#define ICD_CO_KP 16
#endif
constexpr int CO_KP = ICD_CO_KP;   // candidates kept per (query, list)
constexpr int CO_BM = 128;
constexpr int CO_BN = 128;
constexpr int CO_BK = 64;
constexpr int CO_S = 4;
constexpr int CO_STAGE_BYTES = CO_BN * CO_BK * 2;  // 16384
constexpr int CO_RING_BYTES = CO_S * CO_STAGE_BYTES;
constexpr int CO_CAP = 64;
constexpr int CO_CHECK_EVERY = 8;                       // registers between overflow checks (2 lanes append per register)
constexpr int CO_LIMIT = CO_CAP - 2 * CO_CHECK_EVERY;    // compact a query once it holds more entries than this
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

// VAR: build options of the kernel. Product = CO_PRODUCT_VAR; the ablation bits are timing-only.
//   1    ablation: skip the fused select            8    diagnostic: s_memtime stamps
//   128  ablation: no LDS-DMA at all                16   select of tile t inside the MFMA stream of t+1
//   512  software-pipelined stage: the barrier that publishes stage g+1 sits in the middle of stage
//        g and the first fragments of g+1 are read before g ends (no read-latency bubble per stage)
//   1024 compact early at tile ends (all waves in step) in addition to the overflow guard
//   2048 manually ordered stream: every MFMA is a one-instruction asm volatile and the select of the
//        previous tile is a branch-free 6-instruction asm step placed between them (implies 512 + 16);
//        the compiler still allocates registers and inserts the LDS waits
//   4096 (with 2048) split select step: v_cmp into an SGPR mask in one MFMA gap, masked append one
//        k-step later (no VALU -> SALU -> VALU dependency chain in the stream)
#define ICD_STAMP(t) do { __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); \
    __builtin_amdgcn_sched_barrier(0); } while (0)

// ---- one-instruction asm pieces of the manually ordered stream (VAR & 2048) --------------------------
template <bool FIRST>
__device__ __forceinline__ void mfma_asm(f32x16 &acc, const half8 &fa, const half8 &fb) {
    // the query fragment is read straight from the AGPR half (acc 64 + Q 192 = 256 AGPRs): no
    // v_accvgpr_read copy in front of the MFMA, which would be a VALU -> MFMA-operand hazard the
    // compiler cannot see inside an asm statement
    if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(acc) : "v"(fa), "a"(fb) : "memory");
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(fa), "a"(fb) : "memory");
}
// branch-free select step: if (v > thr) { score[aw] = v; row[aw] = rowbase + ROFF; aw += inc; }
template <int ROFF>
__device__ __forceinline__ void filter_asm(uint32_t &aw, float v, float thr, uint32_t rowbase, uint32_t inc) {
    uint32_t row;
    unsigned long long sv;
    asm volatile("v_cmp_gt_f32 vcc, %[v], %[thr]\n\t"
                 "s_and_saveexec_b64 %[sv], vcc\n\t"
                 "v_add_u32 %[row], %[ro], %[rb]\n\t"
                 "ds_write2st64_b32 %[aw], %[v], %[row] offset1:1\n\t"
                 "v_add_u32 %[aw], %[aw], %[inc]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [aw] "+v"(aw), [row] "=&v"(row), [sv] "=&s"(sv)
                 : [v] "v"(v), [thr] "v"(thr), [rb] "v"(rowbase), [ro] "i"(ROFF), [inc] "v"(inc)
                 : "vcc", "memory");
}

// split select step (VAR & 4096): the compare writes a wave mask into an SGPR pair ...
__device__ __forceinline__ unsigned long long fcmp_asm(float v, float thr) {
    unsigned long long m;
    asm volatile("v_cmp_gt_f32_e64 %[m], %[v], %[thr]" : [m] "=s"(m) : [v] "v"(v), [thr] "v"(thr) : "memory");
    return m;
}
// ... and a later MFMA gap consumes it: nothing but a scalar test when no lane passed
template <int ROFF>
__device__ __forceinline__ void fapp_asm(uint32_t &aw, float v, unsigned long long m, uint32_t rowbase, uint32_t inc) {
    uint32_t row;
    asm volatile("s_cmp_lg_u64 %[m], 0\n\t"
                 "s_cbranch_scc0 1f\n\t"
                 "s_mov_b64 exec, %[m]\n\t"
                 "v_add_u32 %[row], %[ro], %[rb]\n\t"
                 "ds_write2st64_b32 %[aw], %[v], %[row] offset1:1\n\t"
                 "v_add_u32 %[aw], %[aw], %[inc]\n\t"
                 "s_mov_b64 exec, -1\n"
                 "1:"
                 : [aw] "+v"(aw), [row] "=&v"(row)
                 : [m] "s"(m), [v] "v"(v), [rb] "v"(rowbase), [ro] "i"(ROFF), [inc] "v"(inc)
                 : "scc", "memory");
}

template <int D, int VAR>
__global__ __launch_bounds__(256, 1) void coarse_topk_kernel(CoarseArgs a) {
    constexpr int KS = D / CO_BK;      // stages per tile
    constexpr int NF = D / 16;         // query fragments per lane
    constexpr int NKSTEP = KS * 4;     // k16-steps per tile
    constexpr bool NOSELECT = (VAR & 1) != 0, STAMPS = (VAR & 8) != 0, OVERLAP = (VAR & 16) != 0 || (VAR & 2048) != 0;
    constexpr bool MANUAL = (VAR & 2048) != 0, SPLITSEL = (VAR & 4096) != 0;
    constexpr int CHK = (VAR & 32) ? 4 : ((VAR & 64) ? 16 : CO_CHECK_EVERY);  // registers between overflow checks (A/B: 4, 16)
    constexpr int LIM = CO_CAP - 2 * CHK;                                        // compact above this many entries
    constexpr bool NOBAR = (VAR & 8192) != 0, NOREAD = (VAR & 16384) != 0;   // timing ablations only (results invalid)
    constexpr bool NODMA = (VAR & 128) != 0, PIPE = MANUAL || (VAR & 512) != 0, TILE_END_COMPACT = (VAR & 1024) != 0;
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

    // ---- LDS-DMA: per-lane source offsets (bytes from the tile's first row, k = 0) -------------------
    // piece i of this wave = rows 8*(4*wave+i)..+7, one full 128-B line each; 16-B pieces XOR-swizzled
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
    // buffer_load ... lds: per-lane part in voffset, tile/stage part in a scalar soffset
    const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(cbase) + (size_t)row_begin * (size_t)(D * 2), 0,
        (int)min((size_t)(a.n_pad - row_begin) * (size_t)(D * 2), (size_t)0x7FFFFFFF), 0x00020000);
    auto issue_stage = [&](int g_tile, int g_ks, int ring_slot) {
        if constexpr (NODMA) return;
        int trow0 = row_begin + g_tile * CO_BN;
        trow0 = min(trow0, last_tile_row0);  // stages past the sweep re-read valid memory, never consumed
        char *dst = smem + ring_slot * CO_STAGE_BYTES + wave * 4096;
        const uint32_t soff = (uint32_t)(trow0 - row_begin) * (uint32_t)(D * 2) + (uint32_t)g_ks * (CO_BK * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, (__attribute__((address_space(3))) void *)(dst + i * 1024),
                                                     16, src_off[i], soff, 0, 0);
    };

    // ---- A-fragment LDS read offsets ----------------------------------------------------------------
    uint32_t rd_off[4];
    {
        const int sw = (c >> 1) & 7;
#pragma unroll
        for (int s = 0; s < 4; ++s) rd_off[s] = (uint32_t)c * 128u + (uint32_t)(((2 * s + h) ^ sw) * 16);
    }
    auto read_frags = [&](half8 (&f)[4], int ring_slot, int s) {
        if constexpr (NOREAD) {
#pragma unroll
            for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(f[t]));
            return;
        }
        const char *sb = smem + ring_slot * CO_STAGE_BYTES + rd_off[s];
#pragma unroll
        for (int t = 0; t < 4; ++t) f[t] = *reinterpret_cast<const half8 *>(sb + t * 4096);
    };

    // ---- select state ---------------------------------------------------------------------------------
    const uint32_t wave_qbase = (uint32_t)CO_RING_BYTES + (uint32_t)(wave * 32) * Ops::QBYTES;
    const uint32_t wave_scratch = (uint32_t)CO_RING_BYTES + (uint32_t)CO_BM * Ops::QBYTES + (uint32_t)wave * 256u;
    Sel2 st;
    Ops::init(st, wave_qbase + (uint32_t)c * Ops::QBYTES, h, (slot0 + wave * 32 + c) < a.nq);

    unsigned long long cprof[2] = {0, 0};  // STAMPS: compactions, cycles inside them
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
        // wave-uniform skip first (v_cmp + one scalar branch when no lane passes: the common case once the
        // thresholds have warmed up), the per-lane append only behind it
        const bool pass = v > st.thr;
        if (__builtin_amdgcn_ballot_w64(pass) != 0ull) {
            asm volatile("" ::: "memory");   // keeps the scalar branch: without it the two conditions are merged into a predicate
            if (pass) {
                *reinterpret_cast<float *>(smem + st.aw) = v;
                *reinterpret_cast<uint32_t *>(smem + st.aw + Ops::ROW_OFF) = rowbase + roff;
                st.aw += st.inc;
            }
        }
        if constexpr (r % CHK == CHK - 1) Ops::check(st, lane, smem, wave_qbase, wave_scratch, false, LIM, STAMPS ? cprof : nullptr);
    };
    auto filter_tile = [&](const f32x16 (&pa)[4], int tile_row0) {
        const uint32_t rowbase = (uint32_t)(tile_row0 + 4 * h);
        if (tile_row0 + CO_BN > a.n) static_for<0, 64>([&](auto F) { filter_reg(pa, F, rowbase, std::true_type{}); });
        else static_for<0, 64>([&](auto F) { filter_reg(pa, F, rowbase, std::false_type{}); });
        if constexpr (TILE_END_COMPACT) Ops::check(st, lane, smem, wave_qbase, wave_scratch, false, CO_KP + 8);
    };

    // prologue: stages 0..S-2 in flight
#pragma unroll
    for (int p = 0; p < CO_S - 1; ++p) issue_stage(p / KS, p % KS, p % CO_S);

    unsigned long long t_wait = 0, t_body = 0, t_epi = 0, t0 = 0, t1 = 0, t2 = 0;
    f32x16 pacc[4];   // previous tile's scores (OVERLAP)
    if (OVERLAP) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) pacc[t][r] = -INFINITY;
    }
    unsigned long long msk[4] = {0, 0, 0, 0};   // SPLITSEL: wave masks of the registers compared last
    half8 afn[4];     // PIPE: fragments of (next stage, k-step 0), read before the stage begins
    if constexpr (PIPE) {
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // stage 0 published
        read_frags(afn, 0, 0);
    }

    for (int tile = 0; tile < ntiles; ++tile) {
        f32x16 acc[4];
        if constexpr (!MANUAL) {   // (the manual stream's first MFMA of a tile takes C = 0 instead)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        }
        const int tile_row0 = row_begin + tile * CO_BN;

        static_for<0, KS>([&](auto KSI) {
            constexpr int ks = decltype(KSI)::value;
            constexpr int slot = ks % CO_S;
            auto mfma4 = [&](const half8 (&f)[4], int qi) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[t], qf[qi], acc[t], 0, 0, 0);
            };
            auto overlap_filter = [&](auto SI) {
                if constexpr (OVERLAP && !NOSELECT) {
                    constexpr int j = ks * 4 + decltype(SI)::value;
                    constexpr int f0 = (j * 64) / NKSTEP, f1 = ((j + 1) * 64) / NKSTEP;
                    static_for<f0, f1>([&](auto F) { filter_reg(pacc, F, (uint32_t)(tile_row0 - CO_BN + 4 * h), std::true_type{}); });
                }
            };
            if constexpr (MANUAL) {
                // ---- manually ordered stream -------------------------------------------------------
                // asm volatile statements keep their program order; "memory" clobbers pin the LDS reads
                // and the DMA between them. One k-step = 4 MFMAs (128 cycles of matrix pipe) with the
                // next k-step's 4 fragment reads ahead of it and 1-2 filter steps of the previous tile.
                const uint32_t prev_rowbase = (uint32_t)(tile_row0 - CO_BN + 4 * h);
                // the MFMA stays a compiler-visible builtin (the compiler pads its hazards and is free to
                // park query fragments in AGPRs); sched_barrier(0) after every piece pins the order below
                auto mfma1 = [&](auto T, const half8 &fa, auto QI) {
                    constexpr int t = decltype(T)::value, qi = decltype(QI)::value;
                    if constexpr (qi == 0) {
                        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, qf[qi], zero, 0, 0, 0);
                    } else {
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, qf[qi], acc[t], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                auto fstep = [&](auto F) {   // branch-free select step for register F of the previous tile
                    if constexpr (!NOSELECT) {
                        constexpr int f = decltype(F)::value;
                        constexpr int t = f >> 4, r = f & 15;
                        filter_asm<t * 32 + (r & 3) + 8 * (r >> 2)>(st.aw, pacc[t][r], st.thr, prev_rowbase, st.inc);
                    }
                };
                auto fcmp = [&](auto F) {
                    constexpr int f = decltype(F)::value;
                    if constexpr (!NOSELECT && f < 64) msk[f & 3] = fcmp_asm(pacc[f >> 4][f & 15], st.thr);
                };
                auto fapp = [&](auto F) {
                    constexpr int f = decltype(F)::value;
                    if constexpr (!NOSELECT && f >= 0 && f < 64) {
                        constexpr int t = f >> 4, r = f & 15;
                        fapp_asm<t * 32 + (r & 3) + 8 * (r >> 2)>(st.aw, pacc[t][r], msk[f & 3], prev_rowbase, st.inc);
                        if constexpr (r % CHK == CHK - 1)
                            Ops::check(st, lane, smem, wave_qbase, wave_scratch, false, LIM, STAMPS ? cprof : nullptr);
                    }
                };
                auto kstep = [&](const half8 (&f)[4], auto S) {
                    constexpr int sidx = decltype(S)::value;
                    constexpr int j = ks * 4 + sidx;
                    constexpr int f0 = (j * 64) / NKSTEP, f1 = ((j + 1) * 64) / NKSTEP;
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (SPLITSEL) {
                        // registers compared in the previous k-step are appended now; this k-step's are compared
                        constexpr int p0 = j > 0 ? ((j - 1) * 64) / NKSTEP : 0, p1 = j > 0 ? f0 : 0;
                        mfma1(std::integral_constant<int, 0>{}, f[0], std::integral_constant<int, j>{});
                        fcmp(std::integral_constant<int, f0>{});
                        mfma1(std::integral_constant<int, 1>{}, f[1], std::integral_constant<int, j>{});
                        if constexpr (p1 > p0) fapp(std::integral_constant<int, p0>{});
                        mfma1(std::integral_constant<int, 2>{}, f[2], std::integral_constant<int, j>{});
                        if constexpr (f1 > f0 + 1) fcmp(std::integral_constant<int, f0 + 1>{});
                        mfma1(std::integral_constant<int, 3>{}, f[3], std::integral_constant<int, j>{});
                        if constexpr (p1 > p0 + 1) fapp(std::integral_constant<int, p0 + 1>{});
                        if constexpr (j == NKSTEP - 1) {   // last k-step: its own registers cannot wait for a next one
                            fapp(std::integral_constant<int, f0>{});
                            if constexpr (f1 > f0 + 1) fapp(std::integral_constant<int, f0 + 1>{});
                        }
                    } else {
                        mfma1(std::integral_constant<int, 0>{}, f[0], std::integral_constant<int, j>{});
                        if constexpr (f1 > f0) fstep(std::integral_constant<int, f0>{});
                        mfma1(std::integral_constant<int, 1>{}, f[1], std::integral_constant<int, j>{});
                        mfma1(std::integral_constant<int, 2>{}, f[2], std::integral_constant<int, j>{});
                        if constexpr (f1 > f0 + 1) fstep(std::integral_constant<int, f0 + 1>{});
                        mfma1(std::integral_constant<int, 3>{}, f[3], std::integral_constant<int, j>{});
                        if constexpr (!NOSELECT && ((f1 - 1) % CHK) == CHK - 1 && f1 > f0)
                            Ops::check(st, lane, smem, wave_qbase, wave_scratch, false, LIM, STAMPS ? cprof : nullptr);
                    }
                };
                half8 f1[4], f2[4], f3[4];
                if (STAMPS) ICD_STAMP(t1);
                read_frags(f1, slot, 1);
                kstep(afn, std::integral_constant<int, 0>{});
                read_frags(f2, slot, 2);
                kstep(f1, std::integral_constant<int, 1>{});
                if (STAMPS) { ICD_STAMP(t2); t_body += t2 - t1; }
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if (STAMPS) { ICD_STAMP(t1); t_wait += t1 - t2; }
                {
                    constexpr int nks = ks + CO_S - 1;
                    issue_stage(tile + (nks >= KS ? 1 : 0), nks % KS, nks % CO_S);
                }
                read_frags(f3, slot, 3);
                kstep(f2, std::integral_constant<int, 2>{});
                read_frags(afn, (ks + 1) % CO_S, 0);
                kstep(f3, std::integral_constant<int, 3>{});
                if (STAMPS) { ICD_STAMP(t2); t_body += t2 - t1; }
            } else if constexpr (PIPE) {
                // stage g = (tile, ks) was published by the previous mid-stage barrier; afn holds its k-step 0
                half8 f1[4], f2[4], f3[4];
                if (STAMPS) ICD_STAMP(t1);
                read_frags(f1, slot, 1);
                mfma4(afn, ks * 4 + 0);
                overlap_filter(std::integral_constant<int, 0>{});
                read_frags(f2, slot, 2);
                mfma4(f1, ks * 4 + 1);
                overlap_filter(std::integral_constant<int, 1>{});
                // pin: reads of k-step s+1 go out before the MFMAs of k-step s
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                if (STAMPS) { ICD_STAMP(t2); t_body += t2 - t1; }
                // publish stage g+1: this wave's pieces of g+1 have landed when only g+2 is outstanding
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (NOBAR) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if (STAMPS) { ICD_STAMP(t1); t_wait += t1 - t2; }
                {   // every wave is past stage g-1: its slot takes stage g+3
                    constexpr int nks = ks + CO_S - 1;
                    issue_stage(tile + (nks >= KS ? 1 : 0), nks % KS, nks % CO_S);
                }
                read_frags(f3, slot, 3);
                mfma4(f2, ks * 4 + 2);
                overlap_filter(std::integral_constant<int, 2>{});
                read_frags(afn, (ks + 1) % CO_S, 0);
                mfma4(f3, ks * 4 + 3);
                overlap_filter(std::integral_constant<int, 3>{});
                __builtin_amdgcn_sched_group_barrier(0x020, 4, 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 1);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 1);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 1);
                __builtin_amdgcn_sched_barrier(0);
                if (STAMPS) { ICD_STAMP(t2); t_body += t2 - t1; }
            } else {
                if (STAMPS) ICD_STAMP(t0);
                // stage (tile,ks) landed for this wave when all but the youngest 2 stages are done
                asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (STAMPS) ICD_STAMP(t1);
                {
                    constexpr int nks = ks + CO_S - 1;
                    issue_stage(tile + (nks >= KS ? 1 : 0), nks % KS, nks % CO_S);
                }
                // fragment reads run one k-step ahead of the MFMAs that consume them
                half8 af[2][4];
                read_frags(af[0], slot, 0);
                static_for<0, 4>([&](auto SI) {
                    constexpr int s = decltype(SI)::value;
                    if constexpr (s + 1 < 4) read_frags(af[(s + 1) & 1], slot, s + 1);
                    mfma4(af[s & 1], ks * 4 + s);
                    overlap_filter(SI);
                });
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
                if (STAMPS) { ICD_STAMP(t2); t_wait += t1 - t0; t_body += t2 - t1; }
            }
        });
        if (STAMPS) ICD_STAMP(t0);

        if constexpr (NOSELECT) {
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[t][r];
            if (sum == 1.2345e30f) st.kept = 1;  // keeps the MFMAs live, never true
        } else if constexpr (OVERLAP) {
            // an asm MFMA's result needs its wait states before any non-MFMA reader (compiler cannot see it)
#pragma unroll
            for (int t = 0; t < 4; ++t) pacc[t] = acc[t];
            if constexpr (TILE_END_COMPACT) Ops::check(st, lane, smem, wave_qbase, wave_scratch, false, CO_KP + 8);
        } else {
            filter_tile(acc, tile_row0);
        }
        if (STAMPS) { ICD_STAMP(t1); t_epi += t1 - t0; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the run-ahead stages
    if constexpr (PIPE) asm volatile("" ::"v"(afn[0]), "v"(afn[1]), "v"(afn[2]), "v"(afn[3]));
    if constexpr (OVERLAP && !NOSELECT) filter_tile(pacc, row_begin + (ntiles - 1) * CO_BN);
    if (STAMPS && a.dbg && lane == 0) {
        unsigned long long *d = a.dbg + ((size_t)blockIdx.x * 4 + wave) * 4;
        d[0] = t_wait; d[1] = t_body; d[2] = t_epi; d[3] = (unsigned long long)ntiles;
        unsigned long long *e = a.dbg + (size_t)8192 * 16 / 2 + ((size_t)blockIdx.x * 4 + wave) * 2;
        e[0] = cprof[0]; e[1] = cprof[1];
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

// A/B builds only: the kernels of this file leave sorted lists; a full list's last entry bounds what it dropped
__global__ void bounds_from_sorted_lists_kernel(const float *scores, const int *rows, float *bounds, int nlists, int kp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nlists) return;
    bounds[i] = rows[(size_t)i * kp + kp - 1] >= 0 ? scores[(size_t)i * kp + kp - 1] : -INFINITY;
}

constexpr int CO_PRODUCT_VAR = 512;   // software-pipelined stage; select after the tile (best measured, profiles/r01_ablate*.log)

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
    unsigned int *zero_u32;  // nullable [rows_pad]: cleared (the coarse pass's shared per-query thresholds)
    long long perm_mul;      // with perm_mod > 0: dst row p holds src row (p * perm_mul) mod perm_mod (an affine permutation)
    int perm_mod;
};

__global__ __launch_bounds__(256) void convert_rows_kernel(ConvertArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= a.rows_pad) return;
    if (a.zero_u32 && lane == 0) a.zero_u32[row] = 0u;
    _Float16 *d = a.dst + (size_t)row * a.dim;
    if (row >= a.rows) {
        for (int i = lane * 4; i < a.dim; i += 256) {
            d[i] = (_Float16)0.0f; d[i + 1] = (_Float16)0.0f; d[i + 2] = (_Float16)0.0f; d[i + 3] = (_Float16)0.0f;
        }
        return;
    }
    const int srow = a.perm_mod > 0 ? (int)(((long long)row * a.perm_mul) % a.perm_mod) : row;
    const float *s = a.src + (size_t)srow * a.dim;
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
