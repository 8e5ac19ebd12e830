// coarse_rg_kernel.hpp — EXPERIMENT (round 2, measured and not shipped; README.md next to this file has the numbers):
// the fp16-MFMA coarse pass with the corpus tile swept ROW GROUP BY ROW GROUP, so that the select of a finished row
// group can run inside the MFMA stream of the next one. Built only into ablation libraries (`make ABLATE=1`, env
// ICD_RG_VAR); bit-exact through the harness's 24 parity cases in every variant that computes the scores.
//
// Replaces the scoring + k-selection inside MilvusClient.search on the FLAT/IP index
// (services/milvus_service.py:280-285) for batches; exactness is restored by finalize.hpp.
//
// Same partition, lists, bounds, candidate buffers and outputs as coarse_flat_kernel.hpp (CoarseFlatArgs; the two
// kernels are interchangeable launch for launch). What differs is the order in which a 128-row corpus tile is
// multiplied:
//   coarse_flat: stage = 128 rows x 64 halves; the four 32-row accumulators of a wave advance together over the
//                12 stages of a tile and all 64 score registers are final at the tile's end, where the MFMA pipe
//                stands still while they are compared, appended and compacted.
//   coarse_rg:   stage = 32 rows x 256 halves (the same 16 KiB, the same 16 one-KiB LDS-DMA pieces, the same 16
//                ds_read_b128 and 16 MFMAs per wave and stage); ONE 32-row accumulator takes D/16 MFMAs in a row
//                (v_mfma_f32_32x32x16 issues back to back on a single accumulator), and while the next row group
//                accumulates into the other of two accumulators, the 16 finished registers are tested in the MFMA
//                gaps. 32 accumulator registers instead of 64; no copy of the scores is kept.
// Result: the loop itself runs as fast as coarse_flat's (no select: 0.495 vs 0.503 ms), but the select does not hide:
// at one wave per SIMD this kernel is bound by the wave's instruction issue, not by the MFMA pipe (about 55-75 cycles
// per MFMA against 32), so an instruction placed in an "MFMA gap" costs what it costs anywhere else, and whatever one
// wave does out of step with the others is paid by all four at the next stage barrier.
#pragma once
#include "../../rag_project_icd10_amd/csrc/coarse_flat_kernel.hpp"

namespace icd {

constexpr int RG_ROWS = 32;                         // rows of a row group = M of the MFMA
constexpr int RG_BK = 256;                          // halves of a row per stage: 4 lines of 128 B
constexpr int RG_STAGE_BYTES = RG_ROWS * RG_BK * 2; // 16384 = CO_STAGE_BYTES
static_assert(RG_STAGE_BYTES == CO_STAGE_BYTES, "the ring slots of both coarse kernels are 16 KiB");
constexpr int RG_QBYTES = CO_CAP * 8 + 4;           // query buffers one LDS bank apart: every lane of a wave writes in one instruction
__host__ __device__ constexpr int rg_lds_bytes() { return CO_S * RG_STAGE_BYTES + CO_BM * RG_QBYTES + 4 * 256; }

// MFMA gaps of a stage: gap i follows MFMA i (0..15). Gap 7 carries the stage's wait + barrier and gaps 8..11 one LDS-DMA
// piece each: the select uses the others, from gap 2 of the row group on (the last MFMA of the finished accumulator is
// two MFMAs old by then). A register's test is three pieces (compare, store, advance) that go into consecutive usable
// gaps: 48 pieces over the usable gaps of the NEXT row group's stages, at most two per gap.
__host__ __device__ constexpr bool rg_gap_usable(int g) { return g >= 2 && ((g % 16) <= 6 || (g % 16) >= 12); }
__host__ __device__ constexpr int rg_usable_gaps(int ksl_count) {
    int n = 0;
    for (int g = 0; g < ksl_count * 16; ++g) n += rg_gap_usable(g) ? 1 : 0;
    return n;
}
// ordinal of gap g among the usable gaps (g itself must be usable)
__host__ __device__ constexpr int rg_gap_ordinal(int g) {
    int n = 0;
    for (int x = 0; x < g; ++x) n += rg_gap_usable(x) ? 1 : 0;
    return n;
}
// pieces [first, last) of the 48 that go into the usable gap of ordinal o out of U
__host__ __device__ constexpr int rg_piece_first(int o, int U) { return (o * 48 + U - 1) / U; }

__device__ __forceinline__ float rg_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float rg_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// VAR (A/B and timing builds, `make ABLATE=1`): 1 = no select at all (TIMING ONLY), 4 = no compaction at tile ends,
// 8 = a row group's 16 registers tested at its end instead of inside the next one's MFMA stream, 16 = tile-end
// compaction from half the quota instead of three quarters, 32 = no compaction at all (TIMING ONLY), 64 = exec-masked
// append (v_cmpx) in one gap instead of the three spread pieces
template <int D, int KP = CO_KP, int VAR = 0>
__global__ __launch_bounds__(256, 1) void coarse_rg_kernel(CoarseFlatArgs a) {
    constexpr int CO_QUOTA = (CO_CAP - KP) / 2 - CO_CHECK_EVERY;   // appends per lane between compactions (coarse_flat_kernel.hpp)
    static_assert(CO_QUOTA >= 8 && KP + 2 * (CO_QUOTA + CO_CHECK_EVERY) <= CO_CAP, "candidate buffer layout");
    constexpr bool NOSEL = (VAR & 1) != 0, NO_TILE_COMPACT = (VAR & 4) != 0, AT_END = (VAR & 8) != 0;
    constexpr bool MASKED = (VAR & 64) != 0;
    constexpr bool NO_COMPACT = (VAR & 32) != 0;   // TIMING ONLY: the buffers overrun into each other (all inside LDS), results are garbage
    constexpr int TILE_QUOTA = (VAR & 16) ? CO_QUOTA / 2 : (CO_QUOTA * 3) / 4;   // appends of a lane that ask for a compaction at a tile end
    constexpr int S = CO_S;                 // ring slots
    constexpr int KSL = D / RG_BK;          // stages per row group
    constexpr int KS = 4 * KSL;             // stages per tile
    constexpr int NF = D / 16;              // query fragments per lane
    constexpr int VM_MID = 4 * (S - 3);     // LDS-DMA pieces that may stay in flight at the mid-stage wait
    static_assert(D % RG_BK == 0 && KS % S == 0, "ring slot must be a compile-time function of the stage");
    using Ops = Sel2Ops<KP, RG_QBYTES>;
    constexpr int RG_FLUSH_ROT = 0;         // (the buffers are a bank apart already: no slot rotation in the end-of-list flush)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int wg = flat_workgroup_of_block((int)blockIdx.x, (int)gridDim.x, a.pos_period);
    const int u_begin = wg * a.units_per_wg;
    const int u_end = min(a.total_units, u_begin + a.units_per_wg);
    if (u_begin >= u_end) return;

    // LDS-DMA: wave w fills line w of the stage (halves [64 w, 64 w + 64) of the stage's 256, 32 rows x 128 B = 4 KiB) in
    // four pieces of 8 rows; 16-B pieces XOR-swizzled on the source side exactly as in coarse_flat, so the A-fragment
    // reads of line j are the reads of row tile j there: conflict-free.
    uint32_t src_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row_local = i * 8 + (lane >> 3);
        const int piece = (lane & 7) ^ ((row_local >> 1) & 7);
        src_off[i] = (uint32_t)row_local * (uint32_t)(D * 2) + (uint32_t)wave * 128u + (uint32_t)piece * 16u - (uint32_t)(i * 1024);
    }
    uint32_t rd_off[4];
    {
        const int sw = (c >> 1) & 7;
#pragma unroll
        for (int s = 0; s < 4; ++s) rd_off[s] = (uint32_t)c * 128u + (uint32_t)(((2 * s + h) ^ sw) * 16);
    }
    auto read_frag = [&](int ring_slot, int line, int s) __attribute__((always_inline)) -> half8 {
        return *reinterpret_cast<const half8 *>(smem + ring_slot * RG_STAGE_BYTES + line * 4096 + rd_off[s]);
    };
    constexpr uint32_t RING_BYTES = (uint32_t)S * RG_STAGE_BYTES;
    const uint32_t wave_qbase = RING_BYTES + (uint32_t)(wave * 32) * Ops::QBYTES;
    const uint32_t wave_scratch = RING_BYTES + (uint32_t)CO_BM * Ops::QBYTES + (uint32_t)wave * 256u;
    const int last_tile = a.ctiles - 1;

    half8 qf[NF];
    int cur_mtile = -1;
    int u = u_begin;
    while (u < u_end) {
        // ---- the list [t0, t1) of query tile mtile, and its ordinal (as coarse_flat) ------------------------------
        const int mtile = u / a.ctiles;
        const int t0 = u - mtile * a.ctiles;
        const int run0 = max(u_begin - mtile * a.ctiles, 0);
        const int run1 = min(u_end - mtile * a.ctiles, a.ctiles);
        const int j = (t0 - run0) / a.list_tiles;
        const int t1 = min(run1, run0 + (j + 1) * a.list_tiles);
        const int ntiles = t1 - t0;
        const int ord = flat_first_ordinal(mtile, wg, a.ctiles, a.units_per_wg, a.list_tiles) + j;
        const int slot0 = mtile * CO_BM;

        if (mtile != cur_mtile) {   // query fragments -> accumulator registers (B operand: lane holds Q[query c][16 s + 8 h + j])
            const _Float16 *qrow = a.q16 + (size_t)(slot0 + wave * 32 + c) * D + 8 * h;
#pragma unroll
            for (int s = 0; s < NF; ++s) qf[s] = *reinterpret_cast<const half8 *>(qrow + 16 * s);
#pragma unroll
            for (int s = 0; s < NF; ++s) asm volatile("" : "+a"(qf[s]));
            cur_mtile = mtile;
        }
        const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<_Float16 *>(a.c16) + (size_t)t0 * CO_BN * D, 0,
            (int)min((size_t)(a.ctiles - t0) * CO_BN * (size_t)(D * 2), (size_t)0x7FFFFFFF), 0x00020000);
        // stage (g_tile, g_ks): row group g_ks / KSL of the tile, halves [256 (g_ks % KSL), +256)
        auto stage_soff = [&](int g_tile, int g_ks) __attribute__((always_inline)) -> uint32_t {
            const int trow = min(g_tile, last_tile - t0);   // stages past the sweep re-read valid memory, never consumed
            return (uint32_t)trow * (uint32_t)(CO_BN * D * 2) + (uint32_t)(g_ks / KSL) * (uint32_t)(RG_ROWS * D * 2) +
                   (uint32_t)(g_ks % KSL) * (uint32_t)(RG_BK * 2);
        };
        auto issue_piece = [&](uint32_t soff, int ring_slot, auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            // one M0 for the four pieces of a stage: the instruction offset applies to the LDS and to the buffer address
            __attribute__((address_space(3))) void *ldst =
                (__attribute__((address_space(3))) void *)(smem + ring_slot * RG_STAGE_BYTES + wave * 4096);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, ldst, 16, src_off[i], soff, i * 1024, 0);
        };
        auto issue_stage = [&](int g_tile, int g_ks, int ring_slot) __attribute__((always_inline)) {
            const uint32_t soff = stage_soff(g_tile, g_ks);
            static_for<0, 4>([&](auto I) __attribute__((always_inline)) { issue_piece(soff, ring_slot, I); });
        };

        Sel2 st;
        Ops::init(st, wave_qbase + (uint32_t)c * Ops::QBYTES, h, (slot0 + wave * 32 + c) < a.nq);
        float boot1 = -INFINITY, boot2 = -INFINITY, boot3 = -INFINITY;   // bootstrap: the lane's three best scores so far
        const int boot_tiles = ntiles >= CO_BOOT_MIN_TILES ? min(a.boot_tiles, ntiles / 3) : 0;
        unsigned int *my_shared = a.shared_thr + (slot0 + wave * 32 + c);
        const bool publish = (slot0 + wave * 32 + c) < a.nq;   // (padding queries sit at +inf and never publish)
        uint32_t published = 0u;
        // One score register of a finished row group, WITHOUT a branch, in three pieces: (0) row id and compare, (1) EVERY
        // lane stores (score, row) at its next slot - a lane that did not pass writes a slot it will overwrite -, (2) the
        // lanes that passed advance. Five instructions whatever the outcome, spread over the MFMA gaps of the next row
        // group: the waves of a work-group stay in step (they meet at a barrier every 16 MFMAs, so a data-dependent detour
        // of one wave inside the stream is paid by all four: a branch per register there measured 0.70 ms against the
        // 0.65 of coarse_flat, which takes its branches at the tile end).
        uint32_t t_row[16];
        bool t_pass[16];
        auto test_piece = [&](const f32x16 &pa, auto R, auto K, uint32_t rowbase) __attribute__((always_inline)) {
            constexpr int r = decltype(R)::value, k = decltype(K)::value;
            constexpr uint32_t roff = (uint32_t)((r & 3) + 8 * (r >> 2));
            if constexpr (MASKED) {
                // VAR & 64: the whole test in one gap, only the passing lanes store: v_cmpx leaves them active, EXEC is
                // restored behind the append (EXEC is all ones here: 256-thread blocks, wave-uniform control flow)
                if constexpr (k == 0) {
                    const uint32_t row = rowbase | roff;
                    asm volatile("v_cmpx_gt_f32_e32 vcc, %1, %2\n\t"
                                 "ds_write2st64_b32 %0, %1, %3 offset1:1\n\t"
                                 "v_add_u32_e32 %0, %0, %4\n\t"
                                 "s_mov_b64 exec, -1"
                                 : "+v"(st.aw)
                                 : "v"(pa[r]), "v"(st.thr), "v"(row), "v"(st.inc)
                                 : "vcc", "memory");
                }
                if constexpr (k == 2 && r % CO_CHECK_EVERY == CO_CHECK_EVERY - 1 && !NO_COMPACT) {
                    if (__builtin_amdgcn_ballot_w64(Ops::used(st, h) >= CO_QUOTA) != 0ull)
                        Ops::check(st, lane, smem, wave_qbase, wave_scratch, false, CO_LIMIT - 2, nullptr, CO_QUOTA - 1);
                }
            } else if constexpr (k == 0) {
                t_row[r] = rowbase | roff;   // (rowbase = a multiple of 32 plus 4 h: the bits are disjoint)
                t_pass[r] = pa[r] > st.thr;
            } else if constexpr (k == 1) {
                *reinterpret_cast<float *>(smem + st.aw) = pa[r];
                *reinterpret_cast<uint32_t *>(smem + st.aw + Ops::ROW_OFF) = t_row[r];
            } else {
                st.aw += t_pass[r] ? st.inc : 0u;
                if constexpr (r % CO_CHECK_EVERY == CO_CHECK_EVERY - 1 && !NO_COMPACT) {
                    // overflow guard (coarse_flat_kernel.hpp): a lane may append CO_QUOTA entries between compactions, and
                    // the slot behind its last entry must stay its own (every lane writes there): one less than that
                    if (__builtin_amdgcn_ballot_w64(Ops::used(st, h) >= CO_QUOTA) != 0ull)
                        Ops::check(st, lane, smem, wave_qbase, wave_scratch, false, CO_LIMIT - 2, nullptr, CO_QUOTA - 1);
                }
            }
        };
        auto test_reg = [&](const f32x16 &pa, auto R, uint32_t rowbase) __attribute__((always_inline)) {
            static_for<0, 3>([&](auto K) __attribute__((always_inline)) { test_piece(pa, R, K, rowbase); });
        };
        static_assert(Ops::ROW_OFF == 256, "ds_write2st64_b32 offset1:1 = the row array of the query's buffer");

        // prologue: stages 0..S-2 in flight, stage 0 published, its first two k-quads of fragments read
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (query fragment loads: the vmcnt accounting starts from zero)
#pragma unroll
        for (int p = 0; p < S - 1; ++p) issue_stage(p / KS, p % KS, p % S);
        half8 fa[4], fb[4], fc[4], fd[4];   // k-quads 0, 1, 2, 3 of the stage in flight (fa / fb are read one stage ahead)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"i"(4 * (S - 2)) : "memory");
#pragma unroll
        for (int s = 0; s < 4; ++s) { fa[s] = read_frag(0, 0, s); fb[s] = read_frag(0, 1, s); }

        f32x16 accA, accB;
#pragma unroll
        for (int r = 0; r < 16; ++r) { accA[r] = -INFINITY; accB[r] = -INFINITY; }
        uint32_t rb_prev = 0u;   // row of register 0 of the row group under test

        for (int tile = 0; tile < ntiles; ++tile) {
            uint32_t seen_early = 0u;
            const int tile_row0 = (t0 + tile) * CO_BN;
            static_for<0, KS>([&](auto KSI) __attribute__((always_inline)) {
                constexpr int ks = decltype(KSI)::value;
                constexpr int t = ks / KSL, ksl = ks % KSL;
                constexpr int slot = ks % S, nslot = (ks + 1) % S;
                f32x16 &cur = (t & 1) ? accB : accA;
                const f32x16 &prev = (t & 1) ? accA : accB;
                uint32_t soff_next = 0u;
                static_for<0, 16>([&](auto II) __attribute__((always_inline)) {
                    constexpr int i = decltype(II)::value;
                    constexpr int qi = ksl * 16 + i;
                    const half8 &af = i < 4 ? fa[i & 3] : (i < 8 ? fb[i & 3] : (i < 12 ? fc[i & 3] : fd[i & 3]));
                    if constexpr (ksl == 0 && i == 0) {
                        f32x16 zero;
#pragma unroll
                        for (int r = 0; r < 16; ++r) zero[r] = 0.0f;
                        cur = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, qf[qi], zero, 0, 0, 0);
                    } else {
                        cur = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, qf[qi], cur, 0, 0, 0);
                    }
                    // fragments: k-quads 2 and 3 of this stage behind MFMAs 0..7, k-quads 0 and 1 of the next behind 8..15
                    if constexpr (i < 4) fc[i] = read_frag(slot, 2, i);
                    else if constexpr (i < 8) fd[i - 4] = read_frag(slot, 3, i - 4);
                    else if constexpr (i < 12) {
                        // every wave is past stage g-1 (the barrier below): its slot takes stage g+S-1, one piece per gap
                        constexpr int nks = ks + S - 1;
                        if constexpr (i == 8) soff_next = stage_soff(tile + nks / KS, nks % KS);
                        issue_piece(soff_next, nks % S, std::integral_constant<int, i - 8>{});
                        fa[i - 8] = read_frag(nslot, 0, i - 8);
                    } else fb[i - 12] = read_frag(nslot, 1, i - 12);
                    if constexpr (i == 7) {
                        // publish stage g+1: this wave's pieces of it have landed when only the stages behind it are outstanding
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(VM_MID) : "memory");
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (ks == KS - 2) {
                            // the query's shared threshold for the end of this tile: older than this stage's and the next
                            // stage's LDS-DMA pieces, so the next stage's counted wait covers it
                            asm volatile("global_load_dword %0, %1, off sc1" : "=v"(seen_early) : "v"(my_shared) : "memory");
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    // the select of the row group that finished before this one, one register per usable gap
                    if constexpr (!NOSEL && !AT_END && rg_gap_usable(ksl * 16 + i)) {
                        constexpr int U = rg_usable_gaps(KSL), o = rg_gap_ordinal(ksl * 16 + i);
                        static_for<rg_piece_first(o, U), rg_piece_first(o + 1, U)>([&](auto P) __attribute__((always_inline)) {
                            constexpr int p = decltype(P)::value;
                            test_piece(prev, std::integral_constant<int, p / 3>{}, std::integral_constant<int, p % 3>{}, rb_prev);
                        });
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                if constexpr (ksl == KSL - 1) {
                    // ---- row group t of the tile is complete ------------------------------------------------------
                    const uint32_t rb_cur = (uint32_t)(tile_row0 + t * RG_ROWS + 4 * h);
                    if constexpr (t == 3) {
                        // Threshold sharing between the lists of a query (coarse_flat_kernel.hpp): adopt the largest
                        // threshold any of them has published, publish this list's when it is larger.
                        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(seen_early) : "i"(4 * (S - 2)) : "memory");
                        const uint32_t seen = seen_early;
                        const uint32_t mine_key = order_f32(st.thr);
                        if (seen > mine_key) st.thr = unorder_f32(seen);
                        else if (h == 0 && publish && mine_key > seen && mine_key > published) {
                            __hip_atomic_fetch_max(my_shared, mine_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            published = mine_key;
                        }
                    }
                    if constexpr (NOSEL) {
                        asm volatile("" ::"v"(cur));
                    } else {
                        const int rg_row0 = tile_row0 + t * RG_ROWS;
                        if (tile < boot_tiles && rg_row0 + RG_ROWS <= a.n) {
                            // Threshold bootstrap (coarse_flat_kernel.hpp): over a list's first tiles every lane tracks the
                            // three best scores it has seen; the threshold follows the smaller of the two lanes' third
                            // best. (The rows of this row group are tested after it, against the threshold they helped set.)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const float v = cur[r];
                                const float lo1 = rg_min(boot1, v);
                                boot1 = rg_max(boot1, v);
                                const float lo2 = rg_min(boot2, lo1);
                                boot2 = rg_max(boot2, lo1);
                                boot3 = rg_max(boot3, lo2);
                            }
                            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(boot3), __float_as_uint(boot3), false, false);
                            const float thr0 = fminf(boot3, __uint_as_float(h ? sw[0] : sw[1]));
                            if (thr0 > st.thr) st.thr = thr0;   // (padding queries keep +inf)
                        }
                        if (rg_row0 + RG_ROWS > a.n) {   // rows past the corpus (its last tile, the planner's zero tiles) never pass
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                if ((int)(rb_cur + (uint32_t)((r & 3) + 8 * (r >> 2))) >= a.n) cur[r] = -INFINITY;
                        }
                        if constexpr (AT_END) {
                            static_for<0, 16>([&](auto R) __attribute__((always_inline)) { test_reg(cur, R, rb_cur); });
                        }
                        if constexpr (t == 3 && !NO_TILE_COMPACT) {
                            // all four waves are at a tile end together: compact here, where the others do the same, the
                            // queries that would otherwise ask for it inside the next tile's MFMA stream
                            if (__builtin_amdgcn_ballot_w64(Ops::used(st, h) > TILE_QUOTA) != 0ull)
                                Ops::check(st, lane, smem, wave_qbase, wave_scratch, false, KP + 2 * TILE_QUOTA, nullptr, TILE_QUOTA);
                        }
                        rb_prev = rb_cur;
                    }
                }
            });
        }
        if constexpr (!NOSEL && !AT_END) {   // the list's last row group (row group 3 of a tile: accumulator B)
            static_for<0, 16>([&](auto R) __attribute__((always_inline)) { test_reg(accB, R, rb_prev); });
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the run-ahead stages
        asm volatile("" ::"v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]));
        asm volatile("" ::"v"(fb[0]), "v"(fb[1]), "v"(fb[2]), "v"(fb[3]));

        // ---- end of the list: every query's top-KP entries (unsorted) and its bound go to global memory ----------
        {
            const int mine = Ops::used(st, h);
            const auto swm = __builtin_amdgcn_permlane32_swap((unsigned)mine, (unsigned)mine, false, false);
            const int other = (int)(h ? swm[0] : swm[1]);
            const int nlo = st.kept + (h ? other : mine), nhi = h ? mine : other;
            const int slot = slot0 + wave * 32 + c;
            const bool store = slot < a.nq;
            const size_t o = ((size_t)min(slot, a.nq - 1) * a.P + ord) * KP;
            const float bound = flush_emit_parallel<KP>(smem, wave_qbase + (uint32_t)c * Ops::QBYTES, h, RG_FLUSH_ROT, nlo, nhi, st.thr,
                                                           store, a.part_scores + o, a.part_rows + o);
            if (store && h == 0) {
                a.bounds[(size_t)slot * a.P + ord] = bound;
                if (t1 == a.ctiles) {   // last list of the query tile: the unused ordinals are empty
                    for (int e = ord + 1; e < a.P; ++e) {
                        const size_t oe = ((size_t)slot * a.P + e) * KP;
                        for (int d = 0; d < KP; ++d) { a.part_scores[oe + d] = -INFINITY; a.part_rows[oe + d] = -1; }
                        a.bounds[(size_t)slot * a.P + e] = -INFINITY;
                    }
                }
            }
        }
        __syncthreads();   // every wave is done with the ring and its buffers before the next list's prologue
        u += ntiles;
    }
}

}  // namespace icd
