// coarse_w8_kernel.hpp — EXPERIMENT (round 2): the fp16-MFMA coarse pass with EIGHT waves per work-group, two per
// SIMD, 16 queries per wave on v_mfma_f32_16x16x32_f16.
//
// Replaces the scoring + k-selection inside MilvusClient.search on the FLAT/IP index
// (services/milvus_service.py:280-285) for batches; exactness is restored by finalize.hpp.
//
// Why: coarse_flat_kernel (four waves, one per SIMD, 32 queries per wave in 192 registers) is bound by the wave's own
// instruction issue - about 75 cycles per 32-cycle MFMA (experiments/r02_rg_kernel/README.md): with one wave on a SIMD
// nothing runs beside an LDS-DMA issue, a compare-and-branch or an append. Sixteen queries per wave need 96 registers
// for the query fragments and 32 accumulators: a wave fits in 256 registers and TWO share a SIMD, one issuing while the
// other's MFMA runs. The price: every wave still reads every corpus fragment from LDS, for half the arithmetic - one
// ds_read_b128 per 16-cycle MFMA, which with four SIMDs is exactly the LDS array's 256 B/clk: the loop is LDS-bound at
// the MFMA's own rate instead of issue-bound at 40 % of it.
//
// Same partition, lists, bounds and outputs as coarse_flat_kernel.hpp (CoarseFlatArgs), same LDS ring (stage = 128
// rows x 64 halves, 16 one-KiB LDS-DMA pieces - two per wave -, XOR swizzle: the A fragments of the 16x16x32 shape
// read it conflict-free as well), same candidate buffers (one per query, 64 entries).
//   lane l: query l & 15 of the wave (B operand, N), k-octet l >> 4; as A operand row l & 15 of a 16-row group.
//   accumulators: 8 row groups x 4 registers: register i of group rg = row 16 rg + 4 (l >> 4) + i of the tile.
//   a query's scores sit in FOUR lanes (l & 15 equal); they append to ONE buffer, slots assigned from the ballot.
#pragma once
#include "../../rag_project_icd10_amd/csrc/coarse_flat_kernel.hpp"

namespace icd {


constexpr int W8_WAVES = 8;
constexpr int W8_QPW = 16;   // queries per wave
__host__ __device__ constexpr int w8_lds_bytes() { return CO_S * CO_STAGE_BYTES + CO_BM * CO_CAP * 8 + W8_WAVES * 256; }


// Compact the candidate buffer of ONE query of the wave (wave-cooperative, one entry per lane; entries [0, nb) are
// valid): keep the entries above the KP-th best score - found by bisection on wave ballots, or by ranking unique keys
// when scores tie (both as Sel2Ops::check, topk_select.hpp) - at the front, unsorted. Returns (wave-uniform) the new
// threshold = an upper bound on everything dropped, and the number of entries kept; nb <= KP keeps everything and
// returns thr unchanged. force: rank even when bisection would do (the list's end wants nothing but <= KP entries).
template <int KP>
__device__ __forceinline__ void w8_compact_one(char *smem, uint32_t qb, int nb, float thr, int lane, uint32_t scratch,
                                               float &new_thr, int &kept) {
    constexpr uint32_t ROW_OFF = 256;
    new_thr = thr;
    kept = nb;
    if (nb <= KP) return;
    const bool valid = lane < nb;
    float v = 0.f;
    uint32_t row = 0;
    if (valid) {
        v = *reinterpret_cast<const float *>(smem + qb + lane * 4);
        row = *reinterpret_cast<const uint32_t *>(smem + qb + ROW_OFF + lane * 4);
    }
    const uint32_t okey = valid ? order_f32(v) : 0u;
    uint32_t mx = okey;
    mx = max(mx, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mx, 0x111, 0xf, 0xf, false));   // row_shr:1
    mx = max(mx, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mx, 0x112, 0xf, 0xf, false));   // row_shr:2
    mx = max(mx, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mx, 0x114, 0xf, 0xf, false));   // row_shr:4
    mx = max(mx, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mx, 0x118, 0xf, 0xf, false));   // row_shr:8
    mx = max(mx, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mx, 0x142, 0xa, 0xf, false));   // row_bcast:15
    mx = max(mx, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mx, 0x143, 0xc, 0xf, false));   // row_bcast:31
    uint32_t hi_s = readlane<uint32_t>(mx, 63);
    uint32_t lo_s = order_f32(thr);
    int c = __popcll(__ballot(okey > lo_s));
    if (c <= KP) hi_s = lo_s;   // nothing to bisect: keep what is above the threshold
    while (c > KP && hi_s - lo_s > 1u) {
        const uint32_t mid = lo_s + ((hi_s - lo_s) >> 1);
        const int cm = __popcll(__ballot(okey > mid));
        if (cm >= KP) { lo_s = mid; c = cm; } else hi_s = mid;
    }
    if (c <= KP) {
        const bool keep = okey > lo_s;
        const u64 km = __ballot(keep);
        const int dest = __popcll(km & ((1ull << lane) - 1ull));
        if (keep) {
            *reinterpret_cast<float *>(smem + qb + dest * 4) = v;
            *reinterpret_cast<uint32_t *>(smem + qb + ROW_OFF + dest * 4) = row;
        }
        new_thr = unorder_f32(lo_s);   // every kept entry is above it, every dropped one at or below
        kept = c;
        return;
    }
    // score ties at the KP-th place: rank unique keys (ordered score with its 6 low bits replaced by 63 - slot)
    const uint32_t key = valid ? ((order_f32(v) & ~63u) | (uint32_t)(63 - lane)) : 0u;
    *reinterpret_cast<uint32_t *>(smem + scratch + lane * 4) = key;
    int rank = 0;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        if (ch * 16 < nb) {   // wave-uniform
            uint4 k0, k1, k2, k3;
            const uint32_t addr = scratch + (uint32_t)ch * 64u;
            asm volatile("ds_read_b128 %0, %4\n\t"
                         "ds_read_b128 %1, %4 offset:16\n\t"
                         "ds_read_b128 %2, %4 offset:32\n\t"
                         "ds_read_b128 %3, %4 offset:48\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(k0), "=&v"(k1), "=&v"(k2), "=&v"(k3)
                         : "v"(addr)
                         : "memory");
            rank += (k0.x > key) + (k0.y > key) + (k0.z > key) + (k0.w > key);
            rank += (k1.x > key) + (k1.y > key) + (k1.z > key) + (k1.w > key);
            rank += (k2.x > key) + (k2.y > key) + (k2.z > key) + (k2.w > key);
            rank += (k3.x > key) + (k3.y > key) + (k3.z > key) + (k3.w > key);
        }
    }
    if (valid && rank < KP) {
        *reinterpret_cast<float *>(smem + qb + rank * 4) = v;
        *reinterpret_cast<uint32_t *>(smem + qb + ROW_OFF + rank * 4) = row;
    }
    const u64 mk = __ballot(valid && rank == KP - 1);
    float nthr = 0.f;
    if (mk) nthr = __builtin_bit_cast(float, readlane<uint32_t>(__float_as_uint(v), __ffsll((long long)mk) - 1));
    new_thr = fmaxf(thr, nthr);   // (never below a threshold adopted from another list)
    kept = KP;
}

// VAR: 1 = no select (TIMING ONLY)
template <int D, int KP = CO_KP, int VAR = 0>
__global__ __launch_bounds__(512, 2) void coarse_w8_kernel(CoarseFlatArgs a) {
    constexpr bool NOSEL = (VAR & 1) != 0;
    constexpr int S = CO_S;
    constexpr int KS = D / CO_BK;       // stages per tile
    constexpr int NF = D / 32;          // query fragments per lane (one per 32-deep k-step)
    constexpr int VM_MID = 2 * (S - 3); // LDS-DMA pieces (two per wave and stage) that may stay in flight at the mid-stage wait
    static_assert(KS % S == 0, "ring slot must be a compile-time function of the stage");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = lane & 15, g = lane >> 4;
    const int wg = flat_workgroup_of_block((int)blockIdx.x, (int)gridDim.x, a.pos_period);
    const int u_begin = wg * a.units_per_wg;
    const int u_end = min(a.total_units, u_begin + a.units_per_wg);
    if (u_begin >= u_end) return;

    // LDS-DMA: piece p = rows 8 p .. 8 p + 7 of the stage, one full 128-B line each; wave w issues pieces 2 w and 2 w + 1
    uint32_t src_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row_local = (wave * 2 + i) * 8 + (lane >> 3);
        const int piece = (lane & 7) ^ ((row_local >> 1) & 7);
        src_off[i] = (uint32_t)row_local * (uint32_t)(D * 2) + (uint32_t)piece * 16u - (uint32_t)(i * 1024);
    }
    // A fragment of row group rg, k-step ks2 (0 / 1) of a stage: row 16 rg + qi, 16-B piece 4 ks2 + g of its line
    uint32_t rd_off[2];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) rd_off[k2] = (uint32_t)qi * 128u + (uint32_t)(((4 * k2 + g) ^ ((qi >> 1) & 7)) * 16);
    // quad j of a stage = k-step j >> 1, row groups 4 (j & 1) .. + 3
    auto read_quad = [&](half8 (&f)[4], int ring_slot, int j) __attribute__((always_inline)) {
        const char *sb = smem + ring_slot * CO_STAGE_BYTES + (j & 1) * 8192 + rd_off[j >> 1];
#pragma unroll
        for (int t = 0; t < 4; ++t) f[t] = *reinterpret_cast<const half8 *>(sb + t * 2048);
    };
    constexpr uint32_t RING_BYTES = (uint32_t)S * CO_STAGE_BYTES;
    const uint32_t wave_qbase = RING_BYTES + (uint32_t)(wave * W8_QPW) * 512u;
    const uint32_t wave_scratch = RING_BYTES + (uint32_t)CO_BM * 512u + (uint32_t)wave * 256u;
    static_assert(KP % 4 == 0 && KP <= CO_CAP - 16, "a query's list is written by its four lanes");
    const int last_tile = a.ctiles - 1;

    half8 qf[NF];
    int cur_mtile = -1;
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
        const int myq = slot0 + wave * W8_QPW + qi;

        if (mtile != cur_mtile) {   // query fragments (B operand: lane holds Q[query qi][32 s + 8 g + 0..7])
            const _Float16 *qrow = a.q16 + (size_t)myq * D + 8 * g;
#pragma unroll
            for (int s = 0; s < NF; ++s) qf[s] = *reinterpret_cast<const half8 *>(qrow + 32 * s);
            cur_mtile = mtile;
        }
        const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<_Float16 *>(a.c16) + (size_t)t0 * CO_BN * D, 0,
            (int)min((size_t)(a.ctiles - t0) * CO_BN * (size_t)(D * 2), (size_t)0x7FFFFFFF), 0x00020000);
        auto issue_stage = [&](int g_tile, int g_ks, int ring_slot) __attribute__((always_inline)) {
            const int trow = min(g_tile, last_tile - t0);   // stages past the sweep re-read valid memory, never consumed
            const uint32_t soff = (uint32_t)trow * (uint32_t)(CO_BN * D * 2) + (uint32_t)g_ks * (CO_BK * 2);
            __attribute__((address_space(3))) void *ldst =
                (__attribute__((address_space(3))) void *)(smem + ring_slot * CO_STAGE_BYTES + wave * 2048);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, ldst, 16, src_off[0], soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, ldst, 16, src_off[1], soff, 1024, 0);
        };

        // ---- select state: one query per lane, replicated in the four lanes that hold its scores ---------------------
        const bool valid_q = myq < a.nq;
        float thr = valid_q ? -INFINITY : INFINITY;    // the query's threshold (upper bound on every score dropped)
        int cnt = 0;                                    // entries in its buffer (all at the front)
        const uint32_t qb = wave_qbase + (uint32_t)qi * 512u;
        const uint32_t gmask = g == 0 ? 0u : (g == 1 ? 0x1u : (g == 2 ? 0x10001u : 0x10003u));   // bits of the lanes before this one (below)
        float boot1 = -INFINITY, boot2 = -INFINITY;    // bootstrap: the lane's two best scores so far
        const int boot_tiles = ntiles >= CO_BOOT_MIN_TILES ? min(a.boot_tiles, ntiles / 3) : 0;
        unsigned int *my_shared = a.shared_thr + myq;
        uint32_t published = 0u;
        auto compact_need = [&](uint32_t need) __attribute__((always_inline)) {   // need: 16-bit mask of the wave's queries
            while (need) {
                const int b = __ffs((int)need) - 1;
                need &= need - 1;
                const int nb = readlane<int>(cnt, b);
                const float tb = __builtin_bit_cast(float, readlane<uint32_t>(__float_as_uint(thr), b));
                float nt;
                int kept;
                w8_compact_one<KP>(smem, wave_qbase + (uint32_t)b * 512u, nb, tb, lane, wave_scratch, nt, kept);
                if (qi == b) { thr = nt; cnt = kept; }
            }
        };
        // one score register: wave-uniform skip first; the passing lanes of a query take consecutive slots in lane order
        auto test_reg = [&](float v, uint32_t row, auto GUARD) __attribute__((always_inline)) {
            if constexpr (decltype(GUARD)::value) {
                if ((int)row >= a.n) v = -INFINITY;
            }
            const bool pass = v > thr;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
            if (__builtin_expect(m != 0ull, 0)) {
                asm volatile("" ::: "memory");
                // the query's four lanes are qi, qi + 16, qi + 32, qi + 48: their pass bits, packed as g0 -> bit 0, g2 -> bit 1,
                // g1 -> bit 16, g3 -> bit 17
                const unsigned long long mq = m >> qi;
                const uint32_t z = ((uint32_t)mq & 0x10001u) | ((((uint32_t)(mq >> 32)) & 0x10001u) << 1);
                const int slot = cnt + __popc(z & gmask);
                if (pass) {
                    *reinterpret_cast<float *>(smem + qb + slot * 4) = v;
                    *reinterpret_cast<uint32_t *>(smem + qb + 256 + slot * 4) = row;
                }
                cnt += __popc(z);
            }
        };

        // prologue: stages 0..S-2 in flight, stage 0 published, its first two quads of fragments read
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int p = 0; p < S - 1; ++p) issue_stage(p / KS, p % KS, p % S);
        half8 afn[4], bfn[4];
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"i"(2 * (S - 2)) : "memory");
        read_quad(afn, 0, 0);
        read_quad(bfn, 0, 1);

        for (int tile = 0; tile < ntiles; ++tile) {
            uint32_t seen_early = 0u;
            f32x4 acc[8];
#pragma unroll
            for (int rg = 0; rg < 8; ++rg)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[rg][i] = 0.0f;
            static_for<0, KS>([&](auto KSI) __attribute__((always_inline)) {
                constexpr int ks = decltype(KSI)::value;
                constexpr int slot = ks % S, nslot = (ks + 1) % S;
                auto mfma_quad = [&](const half8 (&f)[4], int jq) __attribute__((always_inline)) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int rg = 4 * (jq & 1) + t;
                        acc[rg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[t], qf[ks * 2 + (jq >> 1)], acc[rg], 0, 0, 0);
                    }
                };
                half8 f2[4], f3[4];
                read_quad(f2, slot, 2);
                mfma_quad(afn, 0);
                read_quad(f3, slot, 3);
                mfma_quad(bfn, 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_barrier(0);
                // publish stage g+1: this wave's pieces of it have landed when only the stage behind it is outstanding
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"i"(VM_MID) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!NOSEL && ks == KS - 2) {
                    // the query's shared threshold for the end of this tile: older than this stage's and the next stage's
                    // LDS-DMA pieces, so a counted wait at the tile end covers it without draining them
                    asm volatile("global_load_dword %0, %1, off sc1" : "=v"(seen_early) : "v"(my_shared) : "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
                {   // every wave is past stage g-1: its slot takes stage g+S-1
                    constexpr int nks = ks + S - 1;
                    issue_stage(tile + nks / KS, nks % KS, nks % S);
                }
                read_quad(afn, nslot, 0);
                mfma_quad(f2, 2);
                read_quad(bfn, nslot, 1);
                mfma_quad(f3, 3);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 1);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 1);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 1);
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (NOSEL) {
#pragma unroll
                for (int rg = 0; rg < 8; ++rg) asm volatile("" ::"v"(acc[rg]));
            } else {
                // threshold sharing between the lists of a query (coarse_flat_kernel.hpp)
                {
                    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(seen_early) : "i"(2 * (S - 2)) : "memory");
                    const uint32_t seen = seen_early;
                    const uint32_t mine_key = order_f32(thr);
                    if (seen > mine_key) thr = unorder_f32(seen);
                    else if (g == 0 && valid_q && mine_key > seen && mine_key > published) {
                        __hip_atomic_fetch_max(my_shared, mine_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        published = mine_key;
                    }
                }
                const int tile_row0 = (t0 + tile) * CO_BN;
                const uint32_t rowbase = (uint32_t)(tile_row0 + 4 * g);
                if (tile < boot_tiles && tile_row0 + CO_BN <= a.n) {
                    // Threshold bootstrap (coarse_flat_kernel.hpp): here a query's rows sit in four lanes of 32 registers a
                    // tile; every lane tracks its two best scores, the threshold follows the smallest of the four second
                    // best: 8 rows seen so far score at or above it.
#pragma unroll
                    for (int rg = 0; rg < 8; ++rg)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float v = acc[rg][i];
                            const float lo1 = fminf(boot1, v);
                            boot1 = fmaxf(boot1, v);
                            boot2 = fmaxf(boot2, lo1);
                        }
                    const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(boot2), __float_as_uint(boot2), false, false);
                    const float m1 = fminf(boot2, __uint_as_float((g & 1) ? s16[0] : s16[1]));
                    const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m1), __float_as_uint(m1), false, false);
                    const float thr0 = fminf(m1, __uint_as_float((g & 2) ? s32[0] : s32[1]));
                    if (thr0 > thr) thr = thr0;   // (padding queries keep +inf)
                }
                const bool ragged = tile_row0 + CO_BN > a.n;
                static_for<0, 8>([&](auto RG) __attribute__((always_inline)) {
                    constexpr int rg = decltype(RG)::value;
                    if (ragged) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) test_reg(acc[rg][i], rowbase + (uint32_t)(16 * rg + i), std::true_type{});
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) test_reg(acc[rg][i], rowbase + (uint32_t)(16 * rg + i), std::false_type{});
                    }
                    // overflow guard: the four lanes of a query append at most 16 entries per row group
                    const uint32_t need = (uint32_t)__builtin_amdgcn_ballot_w64(cnt > CO_CAP - 16) & 0xffffu;
                    if (need) compact_need(need);
                });
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the run-ahead stages
        asm volatile("" ::"v"(afn[0]), "v"(afn[1]), "v"(afn[2]), "v"(afn[3]));
        asm volatile("" ::"v"(bfn[0]), "v"(bfn[1]), "v"(bfn[2]), "v"(bfn[3]));
        if constexpr (!NOSEL) {
            // ---- end of the list: every query's top-KP entries (unsorted) and its bound go to global memory ----------
            // (one query at a time for now: a compaction leaves at most KP entries at the front and the threshold it ends
            //  on bounds everything the list dropped)
            compact_need((uint32_t)__builtin_amdgcn_ballot_w64(cnt > KP) & 0xffffu);
            if (valid_q) {
                constexpr int PER = KP / 4;
                const size_t o = ((size_t)myq * a.P + ord) * KP;
#pragma unroll
                for (int e = 0; e < PER; ++e) {
                    const int d = g * PER + e;
                    float sv = -INFINITY;
                    int rw = -1;
                    if (d < cnt) {
                        sv = *reinterpret_cast<const float *>(smem + qb + d * 4);
                        rw = (int)*reinterpret_cast<const uint32_t *>(smem + qb + 256 + d * 4);
                    }
                    a.part_scores[o + d] = sv;
                    a.part_rows[o + d] = rw;
                }
                if (g == 0) {
                    a.bounds[(size_t)myq * a.P + ord] = thr;
                    if (t1 == a.ctiles) {   // last list of the query tile: the unused ordinals are empty
                        for (int e = ord + 1; e < a.P; ++e) {
                            const size_t oe = ((size_t)myq * a.P + e) * KP;
                            for (int d = 0; d < KP; ++d) { a.part_scores[oe + d] = -INFINITY; a.part_rows[oe + d] = -1; }
                            a.bounds[(size_t)myq * a.P + e] = -INFINITY;
                        }
                    }
                }
            }
        }
        __syncthreads();
        u += ntiles;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// coarse_w8rg_kernel: the eight-wave kernel with the tile swept ROW SET BY ROW SET (experiments/r02_rg_kernel): a stage
// of the ring holds 32 rows x 256 halves; a wave's two 16-row accumulators (2 x 4 registers) take D/32 MFMAs each and are
// final after D/256 stages, and while the next row set accumulates into a second pair, the 8 finished registers are
// tested between its MFMAs. With two waves on a SIMD the partner's MFMAs run while a wave compares, appends or compacts;
// the tile-end select of coarse_w8_kernel cannot have that (all eight waves reach the tile end together).
// ---------------------------------------------------------------------------------------------------------------------
__host__ __device__ constexpr bool w8rg_gap_usable(int gap) { return gap >= 2 && (gap % 16) != 7 && (gap % 16) != 8 && (gap % 16) != 9; }
// the 8 registers of a row set are tested in every fifth usable gap of the next row set's stages
__host__ __device__ constexpr int w8rg_test_reg(int gap, int ngaps) {
    if (!w8rg_gap_usable(gap)) return -1;
    int ord = 0, total = 0;
    for (int x = 0; x < ngaps; ++x) total += w8rg_gap_usable(x) ? 1 : 0;
    for (int x = 0; x < gap; ++x) ord += w8rg_gap_usable(x) ? 1 : 0;
    const int step = total / 8;
    return (ord % step == 0 && ord / step < 8) ? ord / step : -1;
}

// VAR: 1 = no select (TIMING ONLY), 2 / 4 = synchronised compaction of every query every 30 / 20 tiles
template <int D, int KP = CO_KP, int VAR = 0>
__global__ __launch_bounds__(512, 2) void coarse_w8rg_kernel(CoarseFlatArgs a) {
    constexpr bool NOSEL = (VAR & 1) != 0;
    constexpr int EPOCH = (VAR & 2) ? 30 : ((VAR & 4) ? 20 : 0);   // every EPOCH tiles ALL waves compact ALL their queries, at the same time
    constexpr int S = CO_S;
    constexpr int KSL = D / 256;        // stages per row set
    constexpr int KS = 4 * KSL;         // stages per tile
    constexpr int NF = D / 32;          // query fragments per lane
    constexpr int VM_MID = 2 * (S - 3);
    static_assert(D % 256 == 0 && KS % S == 0, "ring slot must be a compile-time function of the stage");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = lane & 15, g = lane >> 4;
    const int wg = flat_workgroup_of_block((int)blockIdx.x, (int)gridDim.x, a.pos_period);
    const int u_begin = wg * a.units_per_wg;
    const int u_end = min(a.total_units, u_begin + a.units_per_wg);
    if (u_begin >= u_end) return;

    // LDS-DMA: a stage = 4 k-lines (64 halves each) x 32 rows x 128 B; wave w fills 2 KiB of it: k-line w >> 1, rows
    // 16 (w & 1) .. + 15 in two pieces of 8 rows, 16-B pieces XOR-swizzled on the source side (coarse_common.hpp)
    uint32_t src_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row_local = ((wave & 1) * 2 + i) * 8 + (lane >> 3);
        const int piece = (lane & 7) ^ ((row_local >> 1) & 7);
        src_off[i] = (uint32_t)row_local * (uint32_t)(D * 2) + (uint32_t)(wave >> 1) * 128u + (uint32_t)piece * 16u - (uint32_t)(i * 1024);
    }
    // A fragment (row group rg2 of the row set, k-line j, k-step k2 of the line): row 16 rg2 + qi, 16-B piece 4 k2 + g
    uint32_t rd_off[2];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) rd_off[k2] = (uint32_t)qi * 128u + (uint32_t)(((4 * k2 + g) ^ ((qi >> 1) & 7)) * 16);
    // fragment t of quad j (= k-line j): k-step t >> 1, row group t & 1
    auto read_frag = [&](int ring_slot, int j, int t) __attribute__((always_inline)) -> half8 {
        return *reinterpret_cast<const half8 *>(smem + ring_slot * CO_STAGE_BYTES + j * 4096 + (t & 1) * 2048 + rd_off[t >> 1]);
    };
    constexpr uint32_t RING_BYTES = (uint32_t)S * CO_STAGE_BYTES;
    const uint32_t wave_qbase = RING_BYTES + (uint32_t)(wave * W8_QPW) * 512u;
    const uint32_t wave_scratch = RING_BYTES + (uint32_t)CO_BM * 512u + (uint32_t)wave * 256u;
    static_assert(KP % 4 == 0 && KP <= CO_CAP - 16, "a query's list is written by its four lanes");
    const int last_tile = a.ctiles - 1;

    half8 qf[NF];
    int cur_mtile = -1;
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
        const int myq = slot0 + wave * W8_QPW + qi;

        if (mtile != cur_mtile) {
            const _Float16 *qrow = a.q16 + (size_t)myq * D + 8 * g;
#pragma unroll
            for (int s = 0; s < NF; ++s) qf[s] = *reinterpret_cast<const half8 *>(qrow + 32 * s);
            cur_mtile = mtile;
        }
        const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<_Float16 *>(a.c16) + (size_t)t0 * CO_BN * D, 0,
            (int)min((size_t)(a.ctiles - t0) * CO_BN * (size_t)(D * 2), (size_t)0x7FFFFFFF), 0x00020000);
        auto stage_soff = [&](int g_tile, int g_ks) __attribute__((always_inline)) -> uint32_t {
            const int trow = min(g_tile, last_tile - t0);
            return (uint32_t)trow * (uint32_t)(CO_BN * D * 2) + (uint32_t)(g_ks / KSL) * (uint32_t)(32 * D * 2) +
                   (uint32_t)(g_ks % KSL) * 512u;
        };
        auto issue_piece = [&](uint32_t soff, int ring_slot, auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            __attribute__((address_space(3))) void *ldst =
                (__attribute__((address_space(3))) void *)(smem + ring_slot * CO_STAGE_BYTES + wave * 2048);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, ldst, 16, src_off[i], soff, i * 1024, 0);
        };

        const bool valid_q = myq < a.nq;
        float thr = valid_q ? -INFINITY : INFINITY;
        int cnt = 0;
        const uint32_t qb = wave_qbase + (uint32_t)qi * 512u;
        const uint32_t gmask = g == 0 ? 0u : (g == 1 ? 0x1u : (g == 2 ? 0x10001u : 0x10003u));
        float boot1 = -INFINITY, boot2 = -INFINITY;
        const int boot_tiles = ntiles >= CO_BOOT_MIN_TILES ? min(a.boot_tiles, ntiles / 3) : 0;
        unsigned int *my_shared = a.shared_thr + myq;
        uint32_t published = 0u;
        auto compact_need = [&](uint32_t need) __attribute__((always_inline)) {
            while (need) {
                const int b = __ffs((int)need) - 1;
                need &= need - 1;
                const int nb = readlane<int>(cnt, b);
                const float tb = __builtin_bit_cast(float, readlane<uint32_t>(__float_as_uint(thr), b));
                float nt;
                int kept;
                w8_compact_one<KP>(smem, wave_qbase + (uint32_t)b * 512u, nb, tb, lane, wave_scratch, nt, kept);
                if (qi == b) { thr = nt; cnt = kept; }
            }
        };
        auto test_reg = [&](float v, uint32_t row) __attribute__((always_inline)) {
            const bool pass = v > thr;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
            if (__builtin_expect(m != 0ull, 0)) {
                asm volatile("" ::: "memory");
                const unsigned long long mq = m >> qi;
                const uint32_t z = ((uint32_t)mq & 0x10001u) | ((((uint32_t)(mq >> 32)) & 0x10001u) << 1);
                const int slot = cnt + __popc(z & gmask);
                if (pass) {
                    *reinterpret_cast<float *>(smem + qb + slot * 4) = v;
                    *reinterpret_cast<uint32_t *>(smem + qb + 256 + slot * 4) = row;
                }
                cnt += __popc(z);
            }
        };
        // register r (0..7) of a finished row set: row group r >> 2, register r & 3; every 4 registers the overflow guard
        // (the four lanes of a query append at most 16 entries in between)
        auto test_r = [&](const f32x4 (&pa)[2], auto R, uint32_t rowbase) __attribute__((always_inline)) {
            constexpr int r = decltype(R)::value;
            test_reg(pa[r >> 2][r & 3], rowbase + (uint32_t)(16 * (r >> 2) + (r & 3)));
            if constexpr ((r & 3) == 3) {
                const uint32_t need = (uint32_t)__builtin_amdgcn_ballot_w64(cnt > CO_CAP - 16) & 0xffffu;
                if (need) compact_need(need);
            }
        };

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int p = 0; p < S - 1; ++p) {
            const uint32_t soff = stage_soff(p / KS, p % KS);
            issue_piece(soff, p % S, std::integral_constant<int, 0>{});
            issue_piece(soff, p % S, std::integral_constant<int, 1>{});
        }
        half8 fa[4], fb[4], fc[4], fd[4];
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"i"(2 * (S - 2)) : "memory");
#pragma unroll
        for (int t = 0; t < 4; ++t) { fa[t] = read_frag(0, 0, t); fb[t] = read_frag(0, 1, t); }

        f32x4 accA[2], accB[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) { accA[0][i] = -INFINITY; accA[1][i] = -INFINITY; accB[0][i] = -INFINITY; accB[1][i] = -INFINITY; }
        uint32_t rb_prev = 0u;

        for (int tile = 0; tile < ntiles; ++tile) {
            uint32_t seen_early = 0u;
            const int tile_row0 = (t0 + tile) * CO_BN;
            static_for<0, KS>([&](auto KSI) __attribute__((always_inline)) {
                constexpr int ks = decltype(KSI)::value;
                constexpr int t = ks / KSL, ksl = ks % KSL;
                constexpr int slot = ks % S, nslot = (ks + 1) % S;
                f32x4 (&cur)[2] = (t & 1) ? accB : accA;
                const f32x4 (&prev)[2] = (t & 1) ? accA : accB;
                uint32_t soff_next = 0u;
                static_for<0, 16>([&](auto II) __attribute__((always_inline)) {
                    constexpr int i = decltype(II)::value;
                    constexpr int jq = i >> 2, tt = i & 3;            // quad (k-line) and fragment of the quad
                    constexpr int qidx = ksl * 8 + 2 * jq + (tt >> 1);
                    const half8 &af = jq == 0 ? fa[tt] : (jq == 1 ? fb[tt] : (jq == 2 ? fc[tt] : fd[tt]));
                    if constexpr (ksl == 0 && jq == 0 && tt < 2) {
                        f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
                        cur[tt & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, qf[qidx], zero, 0, 0, 0);
                    } else {
                        cur[tt & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, qf[qidx], cur[tt & 1], 0, 0, 0);
                    }
                    if constexpr (i < 4) fc[i] = read_frag(slot, 2, i);
                    else if constexpr (i < 8) fd[i - 4] = read_frag(slot, 3, i - 4);
                    else if constexpr (i < 12) {
                        constexpr int nks = ks + S - 1;
                        if constexpr (i == 8) soff_next = stage_soff(tile + nks / KS, nks % KS);
                        if constexpr (i < 10) issue_piece(soff_next, nks % S, std::integral_constant<int, i - 8>{});
                        fa[i - 8] = read_frag(nslot, 0, i - 8);
                    } else fb[i - 12] = read_frag(nslot, 1, i - 12);
                    if constexpr (i == 7) {
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(VM_MID) : "memory");
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (!NOSEL && ks == KS - 2) {
                            asm volatile("global_load_dword %0, %1, off sc1" : "=v"(seen_early) : "v"(my_shared) : "memory");
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if constexpr (!NOSEL) {
                        constexpr int r = w8rg_test_reg(ksl * 16 + i, KSL * 16);
                        if constexpr (r >= 0) test_r(prev, std::integral_constant<int, r>{}, rb_prev);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                if constexpr (ksl == KSL - 1) {
                    // ---- row set t of the tile is complete ---------------------------------------------------------------
                    const uint32_t rb_cur = (uint32_t)(tile_row0 + t * 32 + 4 * g);
                    if constexpr (NOSEL) {
                        asm volatile("" ::"v"(cur[0]), "v"(cur[1]));
                    } else {
                        if constexpr (t == 3) {   // threshold sharing between the lists of a query (coarse_flat_kernel.hpp)
                            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(seen_early) : "i"(2 * (S - 2)) : "memory");
                            const uint32_t seen = seen_early;
                            const uint32_t mine_key = order_f32(thr);
                            if (seen > mine_key) thr = unorder_f32(seen);
                            else if (g == 0 && valid_q && mine_key > seen && mine_key > published) {
                                __hip_atomic_fetch_max(my_shared, mine_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                published = mine_key;
                            }
                        }
                        const int rs_row0 = tile_row0 + t * 32;
                        if (tile < boot_tiles && rs_row0 + 32 <= a.n) {   // threshold bootstrap (coarse_w8_kernel above)
#pragma unroll
                            for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const float v = cur[rg][i];
                                    const float lo1 = fminf(boot1, v);
                                    boot1 = fmaxf(boot1, v);
                                    boot2 = fmaxf(boot2, lo1);
                                }
                            const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(boot2), __float_as_uint(boot2), false, false);
                            const float m1 = fminf(boot2, __uint_as_float((g & 1) ? s16[0] : s16[1]));
                            const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m1), __float_as_uint(m1), false, false);
                            const float thr0 = fminf(m1, __uint_as_float((g & 2) ? s32[0] : s32[1]));
                            if (thr0 > thr) thr = thr0;
                        }
                        if (rs_row0 + 32 > a.n) {   // rows past the corpus never pass
#pragma unroll
                            for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                                for (int i = 0; i < 4; ++i)
                                    if ((int)(rb_cur + (uint32_t)(16 * rg + i)) >= a.n) cur[rg][i] = -INFINITY;
                        }
                        if constexpr (t == 3 && EPOCH > 0) {
                            // A compaction anywhere else stalls the other seven waves at the next stage barrier for its
                            // whole length, one wave after the other; here all eight compact together.
                            if ((tile + 1) % EPOCH == 0 && tile + 1 < ntiles)
                                compact_need((uint32_t)__builtin_amdgcn_ballot_w64(cnt > KP) & 0xffffu);
                        }
                        rb_prev = rb_cur;
                    }
                }
            });
        }
        if constexpr (!NOSEL) {   // the list's last row set (row set 3 of a tile: accumulator pair B)
            static_for<0, 8>([&](auto R) __attribute__((always_inline)) { test_r(accB, R, rb_prev); });
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" ::"v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]));
        asm volatile("" ::"v"(fb[0]), "v"(fb[1]), "v"(fb[2]), "v"(fb[3]));
        if constexpr (!NOSEL) {
            compact_need((uint32_t)__builtin_amdgcn_ballot_w64(cnt > KP) & 0xffffu);
            if (valid_q) {
                constexpr int PER = KP / 4;
                const size_t o = ((size_t)myq * a.P + ord) * KP;
#pragma unroll
                for (int e = 0; e < PER; ++e) {
                    const int d = g * PER + e;
                    float sv = -INFINITY;
                    int rw = -1;
                    if (d < cnt) {
                        sv = *reinterpret_cast<const float *>(smem + qb + d * 4);
                        rw = (int)*reinterpret_cast<const uint32_t *>(smem + qb + 256 + d * 4);
                    }
                    a.part_scores[o + d] = sv;
                    a.part_rows[o + d] = rw;
                }
                if (g == 0) {
                    a.bounds[(size_t)myq * a.P + ord] = thr;
                    if (t1 == a.ctiles) {
                        for (int e = ord + 1; e < a.P; ++e) {
                            const size_t oe = ((size_t)myq * a.P + e) * KP;
                            for (int d = 0; d < KP; ++d) { a.part_scores[oe + d] = -INFINITY; a.part_rows[oe + d] = -1; }
                            a.bounds[(size_t)myq * a.P + e] = -INFINITY;
                        }
                    }
                }
            }
        }
        __syncthreads();
        u += ntiles;
    }
}

}  // namespace icd
