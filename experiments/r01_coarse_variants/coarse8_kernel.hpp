// coarse8_kernel.hpp — fp16 MFMA scoring with a fused per-query top-KP, 8 waves per work-group,
// TWO waves per SIMD (K-split wave pairs).
//
// Why: with one wave per SIMD (coarse_kernel.hpp) every non-MFMA instruction — LDS-DMA issue, the
// threshold filter, the compactions — sits on the critical path of the only instruction stream of its
// SIMD (measured: select = 47 % of the kernel, profiles/r01_ablate*.log). Here each SIMD runs a PAIR:
//
//   wave A (kh = 0) and wave B (kh = 1) own the same 32 queries. A stage is 64 corpus rows x 128 k
//   (two 8-KiB sub-stages of 64 k); A multiplies sub-stage 0, B sub-stage 1, so each holds only HALF
//   of the query fragments in registers (D/32 fragments = 96 VGPRs at D = 768 -> fits 256 regs/wave).
//   After the last stage of a 64-row tile B stores its partial accumulators to an LDS exchange buffer
//   (8 KiB per pair) and starts the next tile; A picks them up after the next barrier, adds its own
//   partials and runs the select (filter + compactions) spread over the next tile's stages, while
//   B's MFMAs keep the matrix pipe of the shared SIMD busy. B also issues ALL LDS-DMA.
//
// Everything else (swizzled LDS image, ring + counted vmcnt + one barrier per stage, Sel2 candidate
// buffers, partial-list format) is as in coarse_kernel.hpp; results are consumed by finalize.hpp.
#pragma once
#include "coarse_kernel.hpp"

namespace icd {

constexpr int C8_BN = 64;                 // corpus rows per tile
constexpr int C8_BK = 128;                // k per stage (two sub-stages of 64)
constexpr int C8_S = 3;                   // ring slots
constexpr int C8_STAGE_BYTES = C8_BN * C8_BK * 2;      // 16384
constexpr int C8_SUB_BYTES = C8_STAGE_BYTES / 2;       // 8192
constexpr int C8_RING_BYTES = C8_S * C8_STAGE_BYTES;   // 49152
constexpr int C8_CAND_OFF = C8_RING_BYTES;
constexpr int C8_EXCH_OFF = C8_CAND_OFF + CO_BM * CO_CAP * 8;   // + 65536
constexpr int C8_EXCH_BYTES = 8192;                             // per wave pair: 32 regs x 64 lanes x 4 B
constexpr int C8_SCRATCH_OFF = C8_EXCH_OFF + 4 * C8_EXCH_BYTES;
constexpr int C8_LDS_BYTES = C8_SCRATCH_OFF + 4 * 256;          // 148480

template <int D, int VAR>
__global__ __launch_bounds__(512, 2) void coarse8_topk_kernel(CoarseArgs a) {
    constexpr int KS = D / C8_BK;   // stages per tile (6 at D = 768)
    constexpr int NF = D / 32;      // query fragments per wave (its half of K)
    constexpr bool NOSELECT = (VAR & 1) != 0;
    static_assert(KS % C8_S == 0, "ring slot must be a compile-time function of the stage");
    using Ops = Sel2Ops<CO_KP>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave & 3;        // query group: queries 32 g .. 32 g + 31 of the tile
    const int kh = wave >> 2;      // 0 = wave A (sub-stage 0 + select), 1 = wave B (sub-stage 1 + DMA)
    const int c = lane & 31, h = lane >> 5;
    const int mtile = blockIdx.x / a.P, chunk = blockIdx.x % a.P;
    const int slot0 = mtile * CO_BM;
    const int row_begin = chunk * a.rows_per_chunk;
    const int row_end = min(a.n_pad, row_begin + a.rows_per_chunk);
    const int ntiles = (row_end - row_begin) / C8_BN;
    if (ntiles <= 0) return;

    // ---- this wave's half of the query fragments: k16-steps 8 ks + 4 kh + i -----------------------
    half8 qf[NF];
    {
        const _Float16 *qrow = a.q16 + (size_t)(slot0 + g * 32 + c) * D + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) qf[ks * 4 + i] = *reinterpret_cast<const half8 *>(qrow + 16 * (ks * 8 + kh * 4 + i));
    }

    // ---- LDS-DMA (B waves): 16 one-KiB pieces per stage, 4 per B wave -------------------------------
    // piece p = 4 g + i -> row block rb = p >> 1 (rows 8 rb .. 8 rb + 7), k-half = p & 1 (one 128-B line)
    uint32_t src_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = g * 4 + i, rb = p >> 1, half = p & 1;
        const int row_local = rb * 8 + (lane >> 3);
        const int piece = (lane & 7) ^ ((row_local >> 1) & 7);
        src_off[i] = (uint32_t)row_local * (uint32_t)(D * 2) + (uint32_t)half * 128u + (uint32_t)piece * 16u;
    }
    const char *cbase = reinterpret_cast<const char *>(a.c16);
    const int last_tile_row0 = a.n_pad - C8_BN;
    const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(cbase) + (size_t)row_begin * (size_t)(D * 2), 0,
        (int)min((size_t)(a.n_pad - row_begin) * (size_t)(D * 2), (size_t)0x7FFFFFFF), 0x00020000);
    auto issue_stage = [&](int g_tile, int g_ks, int ring_slot) {
        int trow0 = row_begin + g_tile * C8_BN;
        trow0 = min(trow0, last_tile_row0);  // stages past the sweep re-read valid memory, never consumed
        const uint32_t soff = (uint32_t)(trow0 - row_begin) * (uint32_t)(D * 2) + (uint32_t)g_ks * (C8_BK * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = g * 4 + i;
            char *dst = smem + ring_slot * C8_STAGE_BYTES + (p & 1) * C8_SUB_BYTES + (p >> 1) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, (__attribute__((address_space(3))) void *)dst, 16,
                                                     src_off[i], soff, 0, 0);
        }
    };

    // ---- A-fragment reads: this wave's sub-stage, rows 32 t + c, k16-step s of the sub-stage ----------
    uint32_t rd_off[4];
    {
        const int sw = (c >> 1) & 7;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            rd_off[s] = (uint32_t)kh * C8_SUB_BYTES + (uint32_t)c * 128u + (uint32_t)(((2 * s + h) ^ sw) * 16);
    }
    auto read_frags = [&](half8 (&f)[2], int ring_slot, int s) {
        const char *sb = smem + ring_slot * C8_STAGE_BYTES + rd_off[s];
        f[0] = *reinterpret_cast<const half8 *>(sb);
        f[1] = *reinterpret_cast<const half8 *>(sb + 4096);
    };

    // ---- select state (A waves) ---------------------------------------------------------------------
    const uint32_t wave_qbase = (uint32_t)C8_CAND_OFF + (uint32_t)(g * 32) * Ops::QBYTES;
    const uint32_t wave_scratch = (uint32_t)C8_SCRATCH_OFF + (uint32_t)g * 256u;
    Sel2 st;
    Ops::init(st, wave_qbase + (uint32_t)c * Ops::QBYTES, h, (slot0 + g * 32 + c) < a.nq);
    char *exch = smem + C8_EXCH_OFF + g * C8_EXCH_BYTES;   // [8 x b128 per lane], lane-linear

    // filter register F (flat 16 t + r, 32 per tile) of the summed scores of the PREVIOUS tile
    auto filter_reg = [&](const f32x16 (&ps)[2], auto F, uint32_t rowbase, auto GUARD) {
        constexpr int f = decltype(F)::value;
        constexpr int t = f >> 4, r = f & 15;
        constexpr uint32_t roff = (uint32_t)(t * 32 + (r & 3) + 8 * (r >> 2));
        float v = ps[t][r];
        if constexpr (decltype(GUARD)::value) {
            if ((int)(rowbase + roff) >= a.n) v = -INFINITY;
        }
        if (v > st.thr) {
            *reinterpret_cast<float *>(smem + st.aw) = v;
            *reinterpret_cast<uint32_t *>(smem + st.aw + Ops::ROW_OFF) = rowbase + roff;
            st.aw += st.inc;
        }
        if constexpr (f % CO_CHECK_EVERY == CO_CHECK_EVERY - 1)
            Ops::check(st, lane, smem, wave_qbase, wave_scratch, false, CO_LIMIT);
    };

    // B -> A: partial accumulators of one tile through the exchange buffer
    auto send_partials = [&](const f32x16 (&acc)[2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 v = make_float4(acc[t][4 * q4], acc[t][4 * q4 + 1], acc[t][4 * q4 + 2], acc[t][4 * q4 + 3]);
                *reinterpret_cast<float4 *>(exch + ((t * 4 + q4) * 64 + lane) * 16) = v;
            }
    };
    auto add_partials = [&](f32x16 (&ps)[2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 v = *reinterpret_cast<const float4 *>(exch + ((t * 4 + q4) * 64 + lane) * 16);
                ps[t][4 * q4] += v.x; ps[t][4 * q4 + 1] += v.y; ps[t][4 * q4 + 2] += v.z; ps[t][4 * q4 + 3] += v.w;
            }
    };

    // prologue: stages 0 and 1 in flight (B waves)
    if (kh == 1) {
        issue_stage(0, 0, 0);
        issue_stage(1 / KS, 1 % KS, 1 % C8_S);
    }

    f32x16 ps[2];   // A: previous tile's scores (own partials, then + B's)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) ps[t][r] = -INFINITY;

    for (int tile = 0; tile < ntiles; ++tile) {
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        const int tile_row0 = row_begin + tile * C8_BN;
        const uint32_t prev_rowbase = (uint32_t)(tile_row0 - C8_BN + 4 * h);

        static_for<0, KS>([&](auto KSI) {
            constexpr int ks = decltype(KSI)::value;
            constexpr int slot = ks % C8_S;
            // stage (tile, ks) is published once every B wave has seen its own pieces land
            if (kh == 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (kh == 1) {
                constexpr int nks = ks + C8_S - 1;   // every wave is past stage g-1: its slot takes stage g+2
                issue_stage(tile + (nks >= KS ? 1 : 0), nks % KS, nks % C8_S);
            }
            half8 af[2][2];
            read_frags(af[0], slot, 0);
            if constexpr (!NOSELECT) {
                if (kh == 0) {
                    // the pair's scores of the previous tile: B stored its half before this barrier
                    if constexpr (ks == 0) add_partials(ps);
                    constexpr int f0 = (ks * 32) / KS, f1 = ((ks + 1) * 32) / KS;
                    // rows >= n exist only in the last 128 rows of the corpus (n_pad - n < 128): guard there
                    if (tile_row0 > a.n) static_for<f0, f1>([&](auto F) { filter_reg(ps, F, prev_rowbase, std::true_type{}); });
                    else static_for<f0, f1>([&](auto F) { filter_reg(ps, F, prev_rowbase, std::false_type{}); });
                }
            }
            static_for<0, 4>([&](auto SI) {
                constexpr int s = decltype(SI)::value;
                if constexpr (s + 1 < 4) read_frags(af[(s + 1) & 1], slot, s + 1);
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s & 1][t], qf[ks * 4 + s], acc[t], 0, 0, 0);
            });
        });
        if (kh == 1) {
            send_partials(acc);
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t) ps[t] = acc[t];
        }
    }
    if (kh == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // drain; last partials visible
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (kh == 1) return;

    // ---- A: the last tile's select (it may hold padded rows), then the sorted lists ----------------------
    if constexpr (NOSELECT) {
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += ps[t][r];
        if (sum == 1.2345e30f) st.kept = 1;
    } else {
        add_partials(ps);
        const uint32_t rowbase = (uint32_t)(row_begin + (ntiles - 1) * C8_BN + 4 * h);
        static_for<0, 32>([&](auto F) { filter_reg(ps, F, rowbase, std::true_type{}); });
    }
    Ops::check(st, lane, smem, wave_qbase, wave_scratch, true);
    for (int b = 0; b < 32; ++b) {
        const int slot = slot0 + g * 32 + b;
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

}  // namespace icd
