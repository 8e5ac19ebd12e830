// coarse_ksplit_kernel.hpp — round-4 experiment (A/B builds only, never in the product library): candidate (a) of VERDICT r3.
//
// The flat coarse kernel (rag_project_icd10_amd/csrc/coarse_flat_kernel.hpp) with the reduction dimension split over wave
// PAIRS: a wave holds 64 queries x HALF of K in registers (the same 192 registers as 32 queries x all of K), so every corpus
// fragment it reads from LDS feeds FOUR MFMAs instead of two: 8 ds_read_b128 per stage and wave instead of 16 (0.25 per MFMA).
// A stage of the ring still holds 128 rows x 64 halves = two 32-deep k-steps; wave kh = wave & 1 takes k-step kh of every
// stage (the coarse score's summation order is free: finalize's eps covers it). After a tile the two waves of a pair hold
// partial sums of the same 64 queries x 128 rows; they are exchanged through LDS BY QUERY HALVES - wave kh keeps query groups
// {2 kh, 2 kh + 1} and adds its partner's partials of those - after which wave w owns queries 32 w .. 32 w + 31 in exactly
// the accumulator layout of the product: lane swaps, select, compaction, flush and finalize are the product's, unchanged.
//
// LDS: ring 4 x 16 KB + candidate buffers 64 KB + exchange area 32 KB = 160 KB (no compaction scratch: the tie-ranking path
// ranks with v_readlane), so the 64 registers a wave hands over travel in TWO rounds of 32 (three more barriers per tile).
// Registers: 192 query fragments + 128 accumulators + 64 fragment registers + the select's state.
#pragma once
#include "../../rag_project_icd10_amd/csrc/coarse_flat_kernel.hpp"

namespace icd {

constexpr int KS_EXCH_BYTES = 32768;
__host__ __device__ constexpr int ks_lds_bytes() { return CO_S * CO_STAGE_BYTES + CO_BM * CO_CAP * 8 + KS_EXCH_BYTES; }

// TV: timing-only switches. 1: no select at all  2: thresholds at +inf (compares only)  4: no exchange (results are not the scores)
template <int D, int KP = CO_KP, int TV = 0>
__global__ __launch_bounds__(256, 1) void coarse_ksplit_kernel(CoarseFlatArgs a) {
    constexpr int CO_QUOTA = (CO_CAP - KP) / 2 - CO_CHECK_EVERY;
    static_assert(CO_QUOTA >= 8 && KP + 2 * (CO_QUOTA + CO_CHECK_EVERY) <= CO_CAP, "candidate buffer layout");
    constexpr bool NOSEL = (TV & 1) != 0, NOPASS = (TV & 2) != 0, NOEXCH = (TV & 4) != 0;
    constexpr int EPOCH = 24;
    constexpr int S = CO_S;
    constexpr int VM_MID = 4 * (S - 3);
    constexpr int PRO = S - 1;
    constexpr int VM_TILE_END = 4 * (S - 2);
    constexpr int KS = D / CO_BK;      // stages per tile = k-steps per wave
    static_assert(KS % S == 0 && D % 64 == 0, "ring slot must be a compile-time function of the stage");
    using Ops = Sel2Ops<KP>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave & 1, pair = wave >> 1;
    const int c = lane & 31, h = lane >> 5;
    const int q16 = lane & 15, g16 = lane >> 4;
    const int total_units = a.total_units;
    const int nwg_logical = (int)gridDim.x;
    const int nq_act = a.nq;

    uint32_t src_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row_local = (wave * 4 + i) * 8 + (lane >> 3);
        const int piece = (lane & 7) ^ ((row_local >> 1) & 7);
        src_off[i] = (uint32_t)row_local * (uint32_t)(D * 2) + (uint32_t)piece * 16u - (uint32_t)(i * 1024);
    }
    // A fragment of a 16-row group, THIS wave's k-step of the stage: row q16, 16-B piece 4 kh + g16 of its line
    const uint32_t rd_off = (uint32_t)q16 * 128u + (uint32_t)(((4 * kh + g16) ^ ((q16 >> 1) & 7)) * 16);
    auto read_frags = [&](half8 (&f)[4], int ring_slot, int hsel) {   // row groups 4 hsel .. 4 hsel + 3
        const char *sb = smem + ring_slot * CO_STAGE_BYTES + hsel * 8192 + rd_off;
#pragma unroll
        for (int t = 0; t < 4; ++t) f[t] = *reinterpret_cast<const half8 *>(sb + t * 2048);
    };
    constexpr uint32_t RING_BYTES = (uint32_t)S * CO_STAGE_BYTES;
    const uint32_t wave_qbase = RING_BYTES + (uint32_t)(wave * 32) * Ops::QBYTES;
    const uint32_t exch_base = RING_BYTES + (uint32_t)CO_BM * Ops::QBYTES;
    const uint32_t my_exch = exch_base + (uint32_t)wave * 8192u + (uint32_t)lane * 16u;
    const uint32_t partner_exch = exch_base + (uint32_t)(wave ^ 1) * 8192u + (uint32_t)lane * 16u;
    const int last_tile = a.ctiles - 1;

    // local query groups: lg 0, 1 = the groups this wave KEEPS (global group 2 kh + lg), lg 2, 3 = the ones it hands over
    half8 qf[4 * KS];
    int cur_mtile = -1;
    const int wg = flat_workgroup_of_block((int)blockIdx.x, nwg_logical, a.pos_period);
    const int u_begin = wg * a.units_per_wg;
    const int u_end = min(total_units, u_begin + a.units_per_wg);
    if (u_begin >= u_end) return;
    int u = u_begin;
    while (u < u_end) {
        const int mtile = u / a.ctiles;
        const int t0 = u - mtile * a.ctiles;
        const int run0 = max(u_begin - mtile * a.ctiles, 0);
        const int run1 = min(u_end - mtile * a.ctiles, a.ctiles);
        const int j = (t0 - run0) / a.list_tiles;
        const int t1 = min(run1, run0 + (j + 1) * a.list_tiles);
        const int ntiles = t1 - t0;
        const int ord = flat_first_ordinal(mtile, wg, a.ctiles, a.units_per_wg, a.list_tiles) + j;
        const int slot0 = mtile * CO_BM;

        if (mtile != cur_mtile) {
#pragma unroll
            for (int lg = 0; lg < 4; ++lg) {
                const int gr = lg < 2 ? 2 * kh + lg : 2 * (1 - kh) + (lg - 2);
                const int qr = slot0 + pair * 64 + 16 * gr + q16;
                const _Float16 *qrow = a.q16 + (size_t)qr * D + 8 * g16 + 32 * kh;
#pragma unroll
                for (int s = 0; s < KS; ++s) qf[lg * KS + s] = *reinterpret_cast<const half8 *>(qrow + 64 * s);
            }
#pragma unroll
            for (int s = 0; s < 4 * KS; ++s) asm volatile("" : "+a"(qf[s]));
            cur_mtile = mtile;
        }
        const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<_Float16 *>(a.c16) + (size_t)t0 * CO_BN * D, 0,
            (int)min((size_t)(a.ctiles - t0) * CO_BN * (size_t)(D * 2), (size_t)0x7FFFFFFF), 0x00020000);
        auto issue_stage = [&](int g_tile, int g_ks, int ring_slot) {
            const int trow = min(g_tile, last_tile - t0);
            char *dst = smem + ring_slot * CO_STAGE_BYTES + wave * 4096;
            const uint32_t soff = (uint32_t)trow * (uint32_t)(CO_BN * D * 2) + (uint32_t)g_ks * (CO_BK * 2);
            __attribute__((address_space(3))) void *ldst = (__attribute__((address_space(3))) void *)dst;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, ldst, 16, src_off[0], soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, ldst, 16, src_off[1], soff, 1024, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, ldst, 16, src_off[2], soff, 2048, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, ldst, 16, src_off[3], soff, 3072, 0);
        };

        Sel2 st;
        Ops::init(st, wave_qbase + (uint32_t)c * Ops::QBYTES, h, (slot0 + wave * 32 + c) < nq_act && !NOPASS);
        float boot1 = -INFINITY, boot2 = -INFINITY, boot3 = -INFINITY;
        const int boot_tiles = ntiles >= CO_BOOT_MIN_TILES ? min(a.boot_tiles, ntiles / 3) : 0;
        unsigned int *my_shared = a.shared_thr + (slot0 + wave * 32 + c);
        const bool publish = (slot0 + wave * 32 + c) < nq_act;
        uint32_t published = 0u;

        auto filter_quad = [&](const f32x4 (&px)[16], auto Q, uint32_t rowbase, auto GUARD) {
            constexpr int q = decltype(Q)::value;
            constexpr uint32_t qoff = (uint32_t)(16 * (q >> 1) + 4 * (q & 1));
            float v0 = px[q][0], v1 = px[q][1], v2 = px[q][2], v3 = px[q][3];
            const uint32_t rowq = rowbase + qoff;
            if constexpr (decltype(GUARD)::value) {
                if ((int)(rowq + 0u) >= a.n) v0 = -INFINITY;
                if ((int)(rowq + 1u) >= a.n) v1 = -INFINITY;
                if ((int)(rowq + 2u) >= a.n) v2 = -INFINITY;
                if ((int)(rowq + 3u) >= a.n) v3 = -INFINITY;
            }
            const unsigned long long m0 = __builtin_amdgcn_ballot_w64(v0 > st.thr), m1 = __builtin_amdgcn_ballot_w64(v1 > st.thr);
            const unsigned long long m2 = __builtin_amdgcn_ballot_w64(v2 > st.thr), m3 = __builtin_amdgcn_ballot_w64(v3 > st.thr);
            if (__builtin_expect(((m0 | m1) | (m2 | m3)) != 0ull, 0)) {
                uint32_t r1, r2, r3;
                asm volatile("v_or_b32_e32 %1, 1, %12\n\t"
                             "v_or_b32_e32 %2, 2, %12\n\t"
                             "v_or_b32_e32 %3, 3, %12\n\t"
                             "s_mov_b64 exec, %4\n\t"
                             "ds_write2st64_b32 %0, %8, %12 offset1:1\n\t"
                             "v_add_u32_e32 %0, %0, %13\n\t"
                             "s_mov_b64 exec, %5\n\t"
                             "ds_write2st64_b32 %0, %9, %1 offset1:1\n\t"
                             "v_add_u32_e32 %0, %0, %13\n\t"
                             "s_mov_b64 exec, %6\n\t"
                             "ds_write2st64_b32 %0, %10, %2 offset1:1\n\t"
                             "v_add_u32_e32 %0, %0, %13\n\t"
                             "s_mov_b64 exec, %7\n\t"
                             "ds_write2st64_b32 %0, %11, %3 offset1:1\n\t"
                             "v_add_u32_e32 %0, %0, %13\n\t"
                             "s_mov_b64 exec, -1"
                             : "+v"(st.aw), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                             : "s"(m0), "s"(m1), "s"(m2), "s"(m3), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(rowq), "v"(st.inc)
                             : "memory");
            }
            if constexpr (q % 2 == 1) {
                if (__builtin_amdgcn_ballot_w64(Ops::used(st, h) > CO_QUOTA) != 0ull)
                    Ops::template check<true>(st, lane, smem, wave_qbase, 0u, false, CO_LIMIT, nullptr, CO_QUOTA);
            }
        };
        static_assert(Ops::ROW_OFF == 256, "ds_write2st64_b32 offset1:1 = the row array of the query's buffer");

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int p = 0; p < PRO; ++p) issue_stage(p / KS, p % KS, p % S);
        // fragments of a stage: fa[parity of the stage][row-group half][t]
        half8 fa[2][2][4];
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"i"(4 * (PRO - 1)) : "memory");
        read_frags(fa[0][0], 0, 0);
        read_frags(fa[0][1], 0, 1);

        for (int tile = 0; tile < ntiles; ++tile) {
            uint32_t seen_early = 0u;
            f32x4 xs[32];   // accumulator of row group rg and LOCAL query group lg at xs[4 rg + lg]
#pragma unroll
            for (int t = 0; t < 32; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) xs[t][r] = 0.0f;
            static_for<0, KS>([&](auto KSI) {
                constexpr int ks = decltype(KSI)::value;
                constexpr int nslot = (ks + 1) % S;
                constexpr int par = ks & 1, npar = (ks + 1) & 1;
                auto mfma16 = [&](const half8 (&f)[4], int hsel, int t_lo, int t_hi) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (t < t_lo || t >= t_hi) continue;
                        const int rg = 4 * hsel + t;
#pragma unroll
                        for (int lg = 0; lg < 4; ++lg)
                            xs[4 * rg + lg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[t], qf[lg * KS + ks], xs[4 * rg + lg], 0, 0, 0);
                    }
                };
                // front half: row groups 0-3 of this stage (read behind the previous stage's barrier)
                mfma16(fa[par][0], 0, 0, 4);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"i"(VM_MID) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ks == KS - 2) {
                    asm volatile("global_load_dword %0, %1, off sc1" : "=v"(seen_early) : "v"(my_shared) : "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
                {
                    constexpr int nks = ks + S - 1;
                    issue_stage(tile + nks / KS, nks % KS, nks % S);
                }
                // back half: the next stage's eight fragments, row groups 4-7 of this one, the LDS-DMA pieces behind its first MFMAs
                read_frags(fa[npar][0], nslot, 0);
                mfma16(fa[par][1], 1, 0, 2);
                read_frags(fa[npar][1], nslot, 1);
                mfma16(fa[par][1], 1, 2, 4);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 1);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 1);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 1);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 1);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 1);
                __builtin_amdgcn_sched_barrier(0);
            });
            static_assert(KS % 2 == 0, "fragment set parity must return to 0 at a tile's end");

            // ---- exchange by query halves: two rounds of four row groups x the two handed-over query groups ------------------
            if constexpr (!NOEXCH) {
                // (inline asm: for compiler-visible LDS accesses next to the LDS-DMA ring hipcc waits vmcnt(0) - the whole
                //  prefetch pipeline - because it cannot tell the exchange area from the ring)
#pragma unroll
                for (int r = 0; r < 2; ++r) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj)
                            asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(my_exch), "v"(xs[4 * (4 * r + t) + 2 + jj]), "i"((2 * t + jj) * 1024) : "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    f32x4 p[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(p[e]) : "v"(partner_exch), "i"(e * 1024) : "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])::"memory");
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) xs[4 * (4 * r + t) + jj] += p[2 * t + jj];
                    if (r == 0) asm volatile("s_barrier" ::: "memory");   // (round 1 overwrites the area: every wave has read round 0)
                }
                // (the area is next written at the following tile's end, twelve stage barriers from here)
            }
            // the product's accumulator layout: ys[2 rg + gr], this wave's queries 32 wave .. + 31
            f32x4 ys[16];
#pragma unroll
            for (int rg = 0; rg < 8; ++rg) { ys[2 * rg] = xs[4 * rg]; ys[2 * rg + 1] = xs[4 * rg + 1]; }
            if constexpr (NOEXCH) {
                float keep = 0.0f;
#pragma unroll
                for (int rg = 0; rg < 8; ++rg) keep += xs[4 * rg + 2][0] + xs[4 * rg + 3][0];
                asm volatile("" ::"v"(keep));
            }
#pragma unroll
            for (int rg = 0; rg < 8; ++rg)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(ys[2 * rg][i]), __float_as_uint(ys[2 * rg + 1][i]), false, false);
                    ys[2 * rg][i] = __uint_as_float(sw[0]);
                    ys[2 * rg + 1][i] = __uint_as_float(sw[1]);
                }
            {
                uint32_t seen;
                asm volatile("s_waitcnt vmcnt(%1)" : "+v"(seen_early) : "i"(VM_TILE_END) : "memory");
                seen = seen_early;
                const uint32_t mine_key = order_f32(st.thr);
                if (seen > mine_key) st.thr = unorder_f32(seen);
                else if (h == 0 && publish && mine_key > seen && mine_key > published) {
                    __hip_atomic_fetch_max(my_shared, mine_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    published = mine_key;
                }
            }
            const int tile_row0 = (t0 + tile) * CO_BN;
            const uint32_t rowbase = (uint32_t)(tile_row0 + 8 * h);
            if (tile < boot_tiles && tile_row0 + CO_BN <= a.n) {
#pragma unroll
                for (int t = 0; t < 16; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = ys[t][r];
                        const float lo1 = raw_min_f32(boot1, v);
                        boot1 = raw_max_f32(boot1, v);
                        const float lo2 = raw_min_f32(boot2, lo1);
                        boot2 = raw_max_f32(boot2, lo1);
                        boot3 = raw_max_f32(boot3, lo2);
                    }
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(boot3), __float_as_uint(boot3), false, false);
                const float thr0 = fminf(boot3, __uint_as_float(h ? sw[0] : sw[1]));
                if (thr0 > st.thr) st.thr = thr0;
            }
            if constexpr (NOSEL) {
                float keep = 0.0f;
#pragma unroll
                for (int t = 0; t < 16; ++t) keep += ys[t][0] + ys[t][1] + ys[t][2] + ys[t][3];
                asm volatile("" ::"v"(keep));
            } else {
                if (tile_row0 + CO_BN > a.n) static_for<0, 16>([&](auto Q) { filter_quad(ys, Q, rowbase, std::true_type{}); });
                else static_for<0, 16>([&](auto Q) { filter_quad(ys, Q, rowbase, std::false_type{}); });
            }
            if ((tile + 1) % EPOCH == 0 && tile + 1 < ntiles && __builtin_amdgcn_ballot_w64(Ops::used(st, h) > 3) != 0ull)
                compact_all_parallel<KP, Ops>(smem, st, wave_qbase + (uint32_t)c * Ops::QBYTES, h, c);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" ::"v"(fa[0][0][0]), "v"(fa[0][0][1]), "v"(fa[0][0][2]), "v"(fa[0][0][3]), "v"(fa[0][1][0]), "v"(fa[0][1][1]), "v"(fa[0][1][2]), "v"(fa[0][1][3]));
        {
            const int mine = Ops::used(st, h);
            const auto swm = __builtin_amdgcn_permlane32_swap((unsigned)mine, (unsigned)mine, false, false);
            const int other = (int)(h ? swm[0] : swm[1]);
            const int nlo = st.kept + (h ? other : mine), nhi = h ? mine : other;
            const int slot = slot0 + wave * 32 + c;
            const bool store = slot < nq_act;
            const size_t o = ((size_t)min(slot, nq_act - 1) * a.P + ord) * KP;
            const float bound = flush_emit_parallel<KP>(smem, wave_qbase + (uint32_t)c * Ops::QBYTES, h, c, nlo, nhi, st.thr,
                                                           store, a.part_scores + o, a.part_rows + o);
            if (store && h == 0) {
                a.bounds[(size_t)slot * a.P + ord] = bound;
                if (t1 == a.ctiles) {
                    for (int e = ord + 1; e < a.P; ++e) {
                        const size_t oe = ((size_t)slot * a.P + e) * KP;
                        for (int d = 0; d < KP; ++d) { a.part_scores[oe + d] = -INFINITY; a.part_rows[oe + d] = -1; }
                        a.bounds[(size_t)slot * a.P + e] = -INFINITY;
                    }
                }
            }
        }
        __syncthreads();
        u += ntiles;
    }
}

}  // namespace icd
