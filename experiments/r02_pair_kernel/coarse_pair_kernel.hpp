// coarse_pair_kernel.hpp — the fp16-MFMA coarse pass with TWO waves per SIMD (8 waves per work-group, K-split pairs).
//
// Why. In coarse_flat_kernel.hpp one wave per SIMD owns 32 queries over the whole K: its B operand alone is 192 VGPRs, so a
// second wave cannot share the SIMD, and every non-MFMA instruction of that only wave - LDS-DMA issue (~55 cycles a piece),
// the fused select (~70 cycles per score register with its appends), barrier waits - leaves the matrix pipe idle: it is
// busy 40 % of the time (profiles/r01_pmc_counters.json, profiles/r02_ab_flat_variants.log). Here each SIMD runs a PAIR:
//
//   wave A (kh = 0) and wave B (kh = 1) own the same 32 queries; A multiplies K-half 0 (k < D/2), B K-half 1, so each holds
//   D/32 query fragments (96 VGPRs at D = 768) and both fit the 256-register budget of two waves per SIMD. Both sweep the
//   same 64-row corpus tile at the same time; a stage of the LDS ring is 64 rows x 256 B: 64 k of K-half 0 next to the
//   same 64 k of K-half 1 (one 128-B line of each per row). At the end of a tile B hands its partial sums to A through a
//   4-KB LDS buffer per pair, one 32-row half at a time, and A - which adds them to its own - runs the select of tile T
//   spread over the six stages of tile T+1, between its MFMAs. B issues ALL LDS-DMA. So on every SIMD the partner's MFMAs
//   cover the other wave's DMA issue / select, and the two instruction streams are of about equal length.
//
// Everything outside the stage loop is coarse_flat_kernel.hpp's: the flat partition of the (query tile x 128-row unit) grid
// (a unit = two 64-row tiles here), lists / ordinals / bounds, the two-sided candidate buffers (Sel2Ops), bootstrap and
// shared thresholds, the lane-parallel end-of-list flush, the XCD-aware block remap. Results are consumed by finalize.hpp.
//
// Replaces the scoring + k-selection inside MilvusClient.search on the FLAT/IP index (services/milvus_service.py:280-285).
#pragma once
#include "coarse_flat_kernel.hpp"

namespace icd {

constexpr int CP_BN = 64;                          // corpus rows per tile
constexpr int CP_STAGE_BYTES = CP_BN * 256;        // 64 rows x (128 B of K-half 0 | 128 B of K-half 1)
constexpr int CP_S = 4;                            // ring slots
constexpr int CP_RING_BYTES = CP_S * CP_STAGE_BYTES;
constexpr int CP_CAND_OFF = CP_RING_BYTES;
constexpr int CP_EXCH_OFF = CP_CAND_OFF + CO_BM * CO_CAP * 8;
constexpr int CP_EXCH_BYTES = 4096;                // per pair: 16 registers x 64 lanes x 4 B
constexpr int CP_SCRATCH_OFF = CP_EXCH_OFF + 4 * CP_EXCH_BYTES;
constexpr int CP_LDS_BYTES = CP_SCRATCH_OFF + 4 * 256;

// Overflow checks of the candidate buffers: once per 32-row half tile (16 registers), not every 8 registers as in
// coarse_flat_kernel.hpp: every inlined copy of the compaction raises the register pressure of the stage loop, and at
// four copies per tile the 256-register budget of two waves per SIMD spills lane constants that every filter reloads.
// A lane may append 16 entries between checks, so it asks for a compaction once it holds more than 8:
// kept 16 + 2 x (8 + 16) = 64 slots.
constexpr int CP_CHECK_EVERY = 16;
constexpr int CP_QUOTA = (CO_CAP - CO_KP) / 2 - CP_CHECK_EVERY;   // 8

// PV: timing-only ablation bits of A/B builds (results are garbage): 1 no select, 2 no LDS-DMA, 4 no s_barrier, 8 no
// fragment reads (MFMAs on whatever the registers hold), 16 no partial-sum exchange
template <int D, int PV = 0>
__global__ __launch_bounds__(512, 2) void coarse_pair_kernel(CoarseFlatArgs a) {
    constexpr bool NOSEL = (PV & 1) != 0, NODMA = (PV & 2) != 0, NOBAR = (PV & 4) != 0, NOREAD = (PV & 8) != 0, NOXCH = (PV & 16) != 0;
    constexpr int DH = D / 2;           // k per half
    constexpr int KS = DH / 64;         // stages per 64-row tile (6 at D = 768)
    constexpr int NF = DH / 16;         // query fragments per wave
    static_assert(DH % 64 == 0 && KS >= 4, "K-half must be a multiple of 64 with at least four stages");
    using Ops = Sel2Ops<CO_KP>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave & 3;                 // query group: queries 32 g .. 32 g + 31 of the tile
    const int kh = wave >> 2;               // 0 = wave A (K-half 0, select), 1 = wave B (K-half 1, LDS-DMA)
    const bool is_a = kh == 0;
    const int c = lane & 31, h = lane >> 5;
    const int wg = flat_workgroup_of_block((int)blockIdx.x, (int)gridDim.x, a.pos_period);
    const int u_begin = wg * a.units_per_wg;
    const int u_end = min(a.total_units, u_begin + a.units_per_wg);
    if (u_begin >= u_end) return;

    // LDS-DMA (B waves): a piece is 1 KiB = 4 rows x 256 B; piece j of B wave g covers rows 16 g + 4 j .. + 3. Lane i
    // writes row 4 (4 g + j) + (i >> 4), slot i & 15 of the stage image; the slot holds 16-B piece p = slot ^ (row & 15)
    // of the row's 256 B (p < 8: K-half 0, else K-half 1): the swizzle goes on the SOURCE address, the LDS write is lane-linear.
    uint32_t src_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 16 * g + 4 * j + (lane >> 4);
        const int p = (lane & 15) ^ (row & 15);
        src_off[j] = (uint32_t)row * (uint32_t)(D * 2) + (uint32_t)(p >> 3) * (uint32_t)(DH * 2) + (uint32_t)(p & 7) * 16u - (uint32_t)(j * 1024);
    }
    // A-fragment reads: row 32 t + c, k-step i of the stage (16 k = pieces 2 i + h of this wave's half)
    uint32_t rd_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rd_off[i] = (uint32_t)c * 256u + (uint32_t)(((kh * 8 + 2 * i + h) ^ (c & 15)) * 16);
    auto read_frags = [&](half8 (&f)[2], int ring_slot, int i) {
        if constexpr (NOREAD) { asm volatile("" : "+v"(f[0]), "+v"(f[1])); return; }
        const char *sb = smem + ring_slot * CP_STAGE_BYTES + rd_off[i];
        f[0] = *reinterpret_cast<const half8 *>(sb);
        f[1] = *reinterpret_cast<const half8 *>(sb + 32 * 256);
    };
    const uint32_t wave_qbase = (uint32_t)CP_CAND_OFF + (uint32_t)(g * 32) * Ops::QBYTES;
    const uint32_t wave_scratch = (uint32_t)CP_SCRATCH_OFF + (uint32_t)g * 256u;
    char *exch = smem + CP_EXCH_OFF + g * CP_EXCH_BYTES;   // [4 x b128 per lane], lane-linear
    const int last_unit = a.ctiles - 1;

    int u = u_begin;
    while (u < u_end) {
        // ---- the list [t0, t1) of 128-row units of query tile mtile, and its ordinal (as coarse_flat_kernel.hpp) ----
        const int mtile = u / a.ctiles;
        const int t0 = u - mtile * a.ctiles;
        const int run0 = max(u_begin - mtile * a.ctiles, 0);
        const int run1 = min(u_end - mtile * a.ctiles, a.ctiles);
        const int jl = (t0 - run0) / a.list_tiles;
        const int t1 = min(run1, run0 + (jl + 1) * a.list_tiles);
        const int nunits = t1 - t0;
        const int ntiles = 2 * nunits;   // 64-row tiles
        const int ord = flat_first_ordinal(mtile, wg, a.ctiles, a.units_per_wg, a.list_tiles) + jl;
        const int slot0 = mtile * CO_BM;

        half8 qf[NF];
        {   // this wave's half of the query fragments -> registers. Reloaded for EVERY list, also of the same query tile:
            // the end-of-list flush needs 64 registers of its own, and fragments kept alive across it would be spilled -
            // with their reloads landing in the stage loop (24 KB per wave and list from L2 instead)
            const _Float16 *qrow = a.q16 + (size_t)(slot0 + g * 32 + c) * D + kh * DH + 8 * h;
#pragma unroll
            for (int s = 0; s < NF; ++s) qf[s] = *reinterpret_cast<const half8 *>(qrow + 16 * s);
        }
        const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<_Float16 *>(a.c16) + (size_t)t0 * CO_BN * D, 0,
            (int)min((size_t)(a.ctiles - t0) * CO_BN * (size_t)(D * 2), (size_t)0x7FFFFFFF), 0x00020000);
        const int last_tile64 = 2 * (last_unit - t0) + 1;
        // piece J (0..3) of stage (tile64, ks) into ring slot `ring_slot` (B waves)
        auto issue_piece = [&](auto J, int tile64, int ks, int ring_slot) {
            if constexpr (NODMA) return;
            constexpr int j = decltype(J)::value;
            const int trow = min(tile64, last_tile64);   // stages past the sweep re-read valid memory, never consumed
            char *dst = smem + ring_slot * CP_STAGE_BYTES + g * 4096;
            const uint32_t soff = (uint32_t)trow * (uint32_t)(CP_BN * D * 2) + (uint32_t)ks * 128u;
            __attribute__((address_space(3))) void *ldst = (__attribute__((address_space(3))) void *)dst;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, ldst, 16, src_off[j], soff, j * 1024, 0);
        };
        auto issue_stage = [&](int tile64, int ks, int ring_slot) {
            static_for<0, 4>([&](auto J) { issue_piece(J, tile64, ks, ring_slot); });
        };

        // ---- select state (A waves) ----------------------------------------------------------------------------------
        Sel2 st;
        Ops::init(st, wave_qbase + (uint32_t)c * Ops::QBYTES, h, (slot0 + g * 32 + c) < a.nq);
        float boot1 = -INFINITY, boot2 = -INFINITY, boot3 = -INFINITY;
        const int boot_tiles = nunits >= CO_BOOT_MIN_TILES ? 2 * min(a.boot_tiles, nunits / 3) : 0;   // in 64-row tiles
        unsigned int *my_shared = a.shared_thr + (slot0 + g * 32 + c);
        const bool publish = (slot0 + g * 32 + c) < a.nq;
        uint32_t published = 0u;
        // one score register of a finished 32-row half tile (r = register 0..15; rowbase = first row of the half + 4 h)
        auto filter_one = [&](float v, uint32_t row) {
            if constexpr (NOSEL) { asm volatile("" ::"v"(v)); return; }
            const bool pass = v > st.thr;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(pass) != 0ull, 0)) {
                asm volatile("" ::: "memory");
                if (pass) {
                    *reinterpret_cast<float *>(smem + st.aw) = v;
                    *reinterpret_cast<uint32_t *>(smem + st.aw + Ops::ROW_OFF) = row;
                    // +4 for the low lane of the query, -4 for the high one, recomputed from the lane id on the spot: as a
                    // loop-invariant register (st.inc) it is spilled under the 256-register budget and every append
                    // would wait for a scratch reload
                    uint32_t z = 0u;
                    asm volatile("" : "+v"(z));   // (opaque zero: keeps the two instructions below from being hoisted and spilled)
                    const uint32_t l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
                    st.aw += 4u - ((l >> 5) << 3);
                }
            }
        };
        auto quota_check = [&]() {
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(Ops::used(st, h) > CP_QUOTA) != 0ull, 0))
                Ops::check(st, lane, smem, wave_qbase, wave_scratch, false, CO_CAP - 2 * CP_CHECK_EVERY, nullptr, CP_QUOTA);
        };
        // registers [R0, R1) of half tile `half` (f32x16) of the 64-row tile whose first row is tile_row0
        auto filter_range = [&](const f32x16 &ps, auto R0, auto R1, int tile_row0, int half) {
            constexpr int r0 = decltype(R0)::value, r1 = decltype(R1)::value;
            const uint32_t rowbase = (uint32_t)(tile_row0 + 32 * half + 4 * h);
            const bool guard = tile_row0 + CP_BN > a.n;   // rows >= n exist only at the very end of the corpus
            static_for<r0, r1>([&](auto R) {
                constexpr int r = decltype(R)::value;
                constexpr uint32_t roff = (uint32_t)((r & 3) + 8 * (r >> 2));
                float v = ps[r];
                if (guard && (int)(rowbase + roff) >= a.n) v = -INFINITY;
                filter_one(v, rowbase + roff);
                if constexpr (r % CP_CHECK_EVERY == CP_CHECK_EVERY - 1) quota_check();
            });
        };
        // score register IDX (0..31: 16 per 32-row half) of the tile whose first row is tile_row0; -1 = nothing
        auto filter_idx = [&](const f32x16 (&ps)[2], auto IDX, int tile_row0) {
            constexpr int idx = decltype(IDX)::value;
            if constexpr (idx >= 0) {
                constexpr int half = idx >> 4, r = idx & 15;
                constexpr uint32_t roff = (uint32_t)(32 * half + (r & 3) + 8 * (r >> 2));
                const uint32_t row = (uint32_t)(tile_row0 + 4 * h) + roff;
                float v = ps[half][r];
                if (tile_row0 + CP_BN > a.n && (int)row >= a.n) v = -INFINITY;   // rows >= n exist only at the very end of the corpus
                filter_one(v, row);
                if constexpr (r % CP_CHECK_EVERY == CP_CHECK_EVERY - 1) quota_check();
            }
        };
        // a half tile's scores have just become final: bootstrap the threshold / exchange it with the query's other lists
        auto on_scores_final = [&](const f32x16 &ps, int tile64, int tile_row0, int half) {
            if (half == 0) {
                const uint32_t seen = __hip_atomic_load(my_shared, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t mine_key = order_f32(st.thr);
                if (seen > mine_key) st.thr = unorder_f32(seen);
                else if (h == 0 && publish && mine_key > seen && mine_key > published) {
                    __hip_atomic_fetch_max(my_shared, mine_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    published = mine_key;
                }
            }
            if (tile64 < boot_tiles && tile_row0 + CP_BN <= a.n) {   // (coarse_flat_kernel.hpp: threshold bootstrap)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = ps[r];
                    const float lo1 = fminf(boot1, v);
                    boot1 = fmaxf(boot1, v);
                    const float lo2 = fminf(boot2, lo1);
                    boot2 = fmaxf(boot2, lo1);
                    boot3 = fmaxf(boot3, lo2);
                }
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(boot3), __float_as_uint(boot3), false, false);
                const float thr0 = fminf(boot3, __uint_as_float(h ? sw[0] : sw[1]));
                if (thr0 > st.thr) st.thr = thr0;
            }
        };
        // B -> A: 16 registers through the pair's exchange buffer
        auto send16 = [&](const f32x16 &v) {
            if constexpr (NOXCH) { asm volatile("" ::"v"(v)); return; }
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
                *reinterpret_cast<float4 *>(exch + (q4 * 64 + lane) * 16) = make_float4(v[4 * q4], v[4 * q4 + 1], v[4 * q4 + 2], v[4 * q4 + 3]);
        };
        auto recv_add16 = [&](f32x16 &v) {
            if constexpr (NOXCH) { asm volatile("" : "+v"(v)); return; }
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 x = *reinterpret_cast<const float4 *>(exch + (q4 * 64 + lane) * 16);
                v[4 * q4] += x.x; v[4 * q4 + 1] += x.y; v[4 * q4 + 2] += x.z; v[4 * q4 + 3] += x.w;
            }
        };

        // prologue: stages 0..S-2 in flight (B), stage 0 published, its first fragments read
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (query fragment loads: the vmcnt accounting starts from zero)
        if (!is_a) {
#pragma unroll
            for (int p = 0; p < CP_S - 1; ++p) issue_stage(p / KS, p % KS, p % CP_S);
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(4 * (CP_S - 2)) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        half8 fa[2];
        read_frags(fa, 0, 0);

        f32x16 ps[2];      // A: the previous tile's scores (own partial sums, then + B's); B: ps[1] = stash of the second half
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) ps[t][r] = 0.0f;

        for (int tile = 0; tile < ntiles; ++tile) {
            f32x16 acc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
            const bool have_prev = tile > 0;
            const int prev_row0 = (t0 * 2 + tile - 1) * CP_BN;   // first row of the previous 64-row tile
            static_for<0, KS>([&](auto KSI) {
                constexpr int ks = decltype(KSI)::value;
                const int gs = tile * KS + ks;                   // stage number within the list
                const int slot = gs % CP_S, nslot = (gs + 1) % CP_S;
                half8 fb[2];
                auto M = [&](const half8 &f, auto T, int qi) {
                    constexpr int t = decltype(T)::value;
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f, qf[qi], acc[t], 0, 0, 0);
                };
                using T0 = std::integral_constant<int, 0>;
                using T1 = std::integral_constant<int, 1>;
                // A wave: which score register of the previous tile is filtered behind each of the stage's eight MFMAs (-1 =
                // none). A register costs ~70 issue cycles with its appends, an MFMA hides ~24 of a wave's own cycles: spread
                // one per MFMA the select runs in the shadow of BOTH waves' MFMAs; as one block it adds to the stage time.
                // Registers 0-15 (first half) are final after the barrier of stage 0, 16-31 after the barrier of stage 2.
                constexpr int SCHED[6][8] = {{-1, -1, -1, -1, 0, 1, 2, 3},     {4, 5, 6, -1, 7, 8, 9, -1},       {10, 11, 12, -1, 13, 14, 15, -1},
                                             {16, 17, 18, -1, 19, 20, 21, -1}, {22, 23, 24, -1, 25, 26, -1, -1}, {27, 28, 29, -1, 30, 31, -1, -1}};
                constexpr int kq = ks < 6 ? ks : 5;   // (D = 768: KS = 6; larger K-halves filter in their first six stages)
                auto F = [&](auto SLOT) {
                    constexpr int sl = decltype(SLOT)::value;
                    if constexpr (ks < 6) {
                        if (have_prev) filter_idx(ps, std::integral_constant<int, SCHED[kq][sl]>{}, prev_row0);
                    }
                };
                if (is_a) {
                    read_frags(fb, slot, 1);
                    M(fa[0], T0{}, ks * 4 + 0); F(std::integral_constant<int, 0>{});
                    M(fa[1], T1{}, ks * 4 + 0); F(std::integral_constant<int, 1>{});
                    read_frags(fa, slot, 2);
                    M(fb[0], T0{}, ks * 4 + 1); F(std::integral_constant<int, 2>{});
                    M(fb[1], T1{}, ks * 4 + 1); F(std::integral_constant<int, 3>{});
                    if constexpr (NOBAR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    if (have_prev) {   // the previous tile's scores: B stored a half before the barrier A has just passed
                        if constexpr (ks == 0) { recv_add16(ps[0]); on_scores_final(ps[0], tile - 1, prev_row0, 0); }
                        if constexpr (ks == 2) { recv_add16(ps[1]); on_scores_final(ps[1], tile - 1, prev_row0, 1); }
                    }
                    read_frags(fb, slot, 3);
                    M(fa[0], T0{}, ks * 4 + 2); F(std::integral_constant<int, 4>{});
                    M(fa[1], T1{}, ks * 4 + 2); F(std::integral_constant<int, 5>{});
                    read_frags(fa, nslot, 0);
                    M(fb[0], T0{}, ks * 4 + 3); F(std::integral_constant<int, 6>{});
                    M(fb[1], T1{}, ks * 4 + 3); F(std::integral_constant<int, 7>{});
                } else {
                    read_frags(fb, slot, 1);
                    M(fa[0], T0{}, ks * 4 + 0);
                    M(fa[1], T1{}, ks * 4 + 0);
                    read_frags(fa, slot, 2);
                    M(fb[0], T0{}, ks * 4 + 1);
                    M(fb[1], T1{}, ks * 4 + 1);
                    // publish stage gs+1: B's pieces of it have landed when only the stages behind it are outstanding
                    if constexpr (!NODMA) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(4 * (CP_S - 3)) : "memory");
                    if constexpr (NOBAR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    if constexpr (ks == 1) {
                        if (have_prev) send16(ps[1]);            // second half of the previous tile (A has consumed the first)
                    }
                    // every wave is past stage gs-1: its slot takes stage gs+S-1, one piece behind each of the next four MFMAs
                    const int ngs = gs + CP_S - 1;
                    const int nt = ngs / KS, nk = ngs % KS, nsl = ngs % CP_S;
                    read_frags(fb, slot, 3);
                    M(fa[0], T0{}, ks * 4 + 2); issue_piece(std::integral_constant<int, 0>{}, nt, nk, nsl);
                    M(fa[1], T1{}, ks * 4 + 2); issue_piece(std::integral_constant<int, 1>{}, nt, nk, nsl);
                    read_frags(fa, nslot, 0);
                    M(fb[0], T0{}, ks * 4 + 3); issue_piece(std::integral_constant<int, 2>{}, nt, nk, nsl);
                    M(fb[1], T1{}, ks * 4 + 3); issue_piece(std::integral_constant<int, 3>{}, nt, nk, nsl);
                }
            });
            // end of the tile: A keeps its partial sums, B hands over the first half and stashes the second
            if (is_a) {
                ps[0] = acc[0];
                ps[1] = acc[1];
            } else {
                send16(acc[0]);
                ps[1] = acc[1];
            }
        }
        if (!is_a) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the run-ahead stages
        asm volatile("" ::"v"(fa[0]), "v"(fa[1]));

        // ---- the last tile's scores: two more hand-overs, then the whole tile is filtered at once --------------------
        {
            const int last_row0 = (t0 * 2 + ntiles - 1) * CP_BN;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // B's first half is in the buffer
            if (is_a) { recv_add16(ps[0]); on_scores_final(ps[0], ntiles - 1, last_row0, 0); }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // A has read it
            if (!is_a) send16(ps[1]);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // B's second half is in the buffer
            if (is_a) {
                recv_add16(ps[1]);
                on_scores_final(ps[1], ntiles - 1, last_row0, 1);
                filter_range(ps[0], std::integral_constant<int, 0>{}, std::integral_constant<int, 16>{}, last_row0, 0);
                filter_range(ps[1], std::integral_constant<int, 0>{}, std::integral_constant<int, 16>{}, last_row0, 1);
            }
        }

        // ---- end of the list: every query's top-KP entries (unsorted) and its bound go to global memory (A waves) -----
        if (is_a) {
            const int mine = Ops::used(st, h);
            const auto swm = __builtin_amdgcn_permlane32_swap((unsigned)mine, (unsigned)mine, false, false);
            const int other = (int)(h ? swm[0] : swm[1]);
            const int nlo = st.kept + (h ? other : mine), nhi = h ? mine : other;
            const int slot = slot0 + g * 32 + c;
            const bool store = slot < a.nq;
            const size_t o = ((size_t)min(slot, a.nq - 1) * a.P + ord) * CO_KP;
            // (rot / qb are made opaque here: otherwise the flush's 64 per-slot LDS addresses are hoisted out of the list loop
            //  as loop invariants, live through the stage loop, and the 256-register budget spills query fragments for them)
            int rot = c;
            uint32_t qb = wave_qbase + (uint32_t)c * Ops::QBYTES;
            asm volatile("" : "+v"(rot), "+v"(qb));
            const float bound = flush_emit_parallel<CO_KP>(smem, qb, h, rot, nlo, nhi, st.thr, store, a.part_scores + o, a.part_rows + o);
            if (store && h == 0) {
                a.bounds[(size_t)slot * a.P + ord] = bound;
                if (t1 == a.ctiles) {   // last list of the query tile: the unused ordinals are empty
                    for (int e = ord + 1; e < a.P; ++e) {
                        const size_t oe = ((size_t)slot * a.P + e) * CO_KP;
                        for (int d = 0; d < CO_KP; ++d) { a.part_scores[oe + d] = -INFINITY; a.part_rows[oe + d] = -1; }
                        a.bounds[(size_t)slot * a.P + e] = -INFINITY;
                    }
                }
            }
        }
        __syncthreads();   // every wave is done with the ring, the exchange buffers and the candidate buffers
        u += nunits;
    }
}

}  // namespace icd
