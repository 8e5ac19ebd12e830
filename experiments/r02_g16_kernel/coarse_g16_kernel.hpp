// coarse_g16_kernel.hpp — EXPERIMENT (round 2): coarse_flat_kernel's geometry (four waves, one per SIMD, 32 queries per
// wave in 192 accumulator registers, 128-row x 64-half stages) on v_mfma_f32_16x16x32_f16: a wave's queries are two
// groups of 16, every corpus fragment feeds two MFMAs. Same LDS traffic and MFMA cycles per flop as the 32x32x16 form;
// the chip holds a higher clock on this shape (MI355X_MICROARCH.md, DVFS give-back (7); measured -7 % on the loop in
// experiments/r02_flat_variants). The select is the four-lanes-per-query one of coarse_w8_kernel.hpp, once per group.
//
// Replaces the scoring + k-selection inside MilvusClient.search on the FLAT/IP index
// (services/milvus_service.py:280-285) for batches; exactness is restored by finalize.hpp.
#pragma once
#include "../r02_w8_kernel/coarse_w8_kernel.hpp"

namespace icd {

// VAR: 1 = no select (TIMING ONLY)
template <int D, int KP = CO_KP, int VAR = 0>
__global__ __launch_bounds__(256, 1) void coarse_g16_kernel(CoarseFlatArgs a) {
    constexpr int NG = 2;               // query groups of 16 per wave (B operands of v_mfma_f32_16x16x32_f16)
    constexpr bool NOSEL = (VAR & 1) != 0;
    constexpr int S = CO_S;
    constexpr int KS = D / CO_BK;       // stages per tile
    constexpr int NF = D / 32;          // query fragments per lane (one per 32-deep k-step)
    constexpr int VM_MID = 4 * (S - 3); // LDS-DMA pieces (four per wave and stage) that may stay in flight at the mid-stage wait
    static_assert(KS % S == 0, "ring slot must be a compile-time function of the stage");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = lane & 15, g = lane >> 4;
    const int wg = flat_workgroup_of_block((int)blockIdx.x, (int)gridDim.x, a.pos_period);
    const int u_begin = wg * a.units_per_wg;
    const int u_end = min(a.total_units, u_begin + a.units_per_wg);
    if (u_begin >= u_end) return;

    // LDS-DMA: piece p = rows 8 p .. 8 p + 7 of the stage, one full 128-B line each; wave w issues pieces 4 w .. 4 w + 3
    uint32_t src_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row_local = (wave * 4 + i) * 8 + (lane >> 3);
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
    const uint32_t wave_qbase = RING_BYTES + (uint32_t)(wave * 32) * 512u;   // the wave's 32 buffers: group 0 first
    const uint32_t wave_scratch = RING_BYTES + (uint32_t)CO_BM * 512u + (uint32_t)wave * 256u;
    static_assert(KP % 4 == 0 && KP <= CO_CAP - 16, "a query's list is written by its four lanes");
    const int last_tile = a.ctiles - 1;

    half8 qf[NG][NF];
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
        const int myq0 = slot0 + wave * 32 + qi;   // group 0's query of this lane; group 1's is myq0 + 16

        if (mtile != cur_mtile) {   // query fragments (B operand: lane holds Q[query][32 s + 8 g + 0..7]) -> accumulator registers
#pragma unroll
            for (int gr = 0; gr < NG; ++gr) {
                const _Float16 *qrow = a.q16 + (size_t)(myq0 + 16 * gr) * D + 8 * g;
#pragma unroll
                for (int s = 0; s < NF; ++s) qf[gr][s] = *reinterpret_cast<const half8 *>(qrow + 32 * s);
            }
#pragma unroll
            for (int gr = 0; gr < NG; ++gr)
#pragma unroll
                for (int s = 0; s < NF; ++s) asm volatile("" : "+a"(qf[gr][s]));
            cur_mtile = mtile;
        }
        const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<_Float16 *>(a.c16) + (size_t)t0 * CO_BN * D, 0,
            (int)min((size_t)(a.ctiles - t0) * CO_BN * (size_t)(D * 2), (size_t)0x7FFFFFFF), 0x00020000);
        auto issue_stage = [&](int g_tile, int g_ks, int ring_slot) __attribute__((always_inline)) {
            const int trow = min(g_tile, last_tile - t0);   // stages past the sweep re-read valid memory, never consumed
            const uint32_t soff = (uint32_t)trow * (uint32_t)(CO_BN * D * 2) + (uint32_t)g_ks * (CO_BK * 2);
            __attribute__((address_space(3))) void *ldst =
                (__attribute__((address_space(3))) void *)(smem + ring_slot * CO_STAGE_BYTES + wave * 4096);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, ldst, 16, src_off[0], soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, ldst, 16, src_off[1], soff, 1024, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, ldst, 16, src_off[2], soff, 2048, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, ldst, 16, src_off[3], soff, 3072, 0);
        };

        // ---- select state: TWO queries per lane (one per group), each replicated in the four lanes that hold its scores ----
        bool valid_q[NG];
        float thr[NG];       // the query's threshold (upper bound on every score dropped)
        int cnt[NG];         // entries in its buffer (all at the front)
        uint32_t qb[NG], published[NG];
        float boot1[NG], boot2[NG];   // bootstrap: the lane's two best scores so far
#pragma unroll
        for (int gr = 0; gr < NG; ++gr) {
            valid_q[gr] = (myq0 + 16 * gr) < a.nq;
            thr[gr] = valid_q[gr] ? -INFINITY : INFINITY;
            cnt[gr] = 0;
            qb[gr] = wave_qbase + (uint32_t)(16 * gr + qi) * 512u;
            published[gr] = 0u;
            boot1[gr] = -INFINITY; boot2[gr] = -INFINITY;
        }
        const uint32_t gmask = g == 0 ? 0u : (g == 1 ? 0x1u : (g == 2 ? 0x10001u : 0x10003u));   // bits of the lanes before this one (below)
        const int boot_tiles = ntiles >= CO_BOOT_MIN_TILES ? min(a.boot_tiles, ntiles / 3) : 0;
        unsigned int *my_shared = a.shared_thr + myq0;   // (group 1: + 16)
        auto compact_need = [&](uint32_t need, auto GR) __attribute__((always_inline)) {   // need: 16-bit mask of the group's queries
            constexpr int gr = decltype(GR)::value;
            while (need) {
                const int b = __ffs((int)need) - 1;
                need &= need - 1;
                const int nb = readlane<int>(cnt[gr], b);
                const float tb = __builtin_bit_cast(float, readlane<uint32_t>(__float_as_uint(thr[gr]), b));
                float nt;
                int kept;
                w8_compact_one<KP>(smem, wave_qbase + (uint32_t)(16 * gr + b) * 512u, nb, tb, lane, wave_scratch, nt, kept);
                if (qi == b) { thr[gr] = nt; cnt[gr] = kept; }
            }
        };
        // the four registers of one row group and query group (four consecutive rows): ONE branch when no lane passes any;
        // the passing lanes of a query take consecutive slots in lane order, register by register
        auto test_quad = [&](const f32x4 &pa, uint32_t row0, auto GR, bool ragged) __attribute__((always_inline)) {
            constexpr int gr = decltype(GR)::value;
            float v[4] = {pa[0], pa[1], pa[2], pa[3]};
            if (ragged) {   // (wave-uniform: the corpus's last tile only)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if ((int)(row0 + i) >= a.n) v[i] = -INFINITY;
            }
            unsigned long long m[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) m[i] = __builtin_amdgcn_ballot_w64(v[i] > thr[gr]);
            if (__builtin_expect(((m[0] | m[1]) | (m[2] | m[3])) != 0ull, 0)) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (m[i] != 0ull) {   // (wave-uniform)
                        // the query's four lanes are qi, qi + 16, qi + 32, qi + 48: their pass bits, packed as g0 -> bit 0,
                        // g2 -> bit 1, g1 -> bit 16, g3 -> bit 17
                        const unsigned long long mq = m[i] >> qi;
                        const uint32_t z = ((uint32_t)mq & 0x10001u) | ((((uint32_t)(mq >> 32)) & 0x10001u) << 1);
                        const int slot = cnt[gr] + __popc(z & gmask);
                        if (v[i] > thr[gr]) {
                            *reinterpret_cast<float *>(smem + qb[gr] + slot * 4) = v[i];
                            *reinterpret_cast<uint32_t *>(smem + qb[gr] + 256 + slot * 4) = row0 + i;
                        }
                        cnt[gr] += __popc(z);
                    }
                }
            }
        };

        // prologue: stages 0..S-2 in flight, stage 0 published, its first two quads of fragments read
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int p = 0; p < S - 1; ++p) issue_stage(p / KS, p % KS, p % S);
        half8 afn[4], bfn[4];
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"i"(4 * (S - 2)) : "memory");
        read_quad(afn, 0, 0);
        read_quad(bfn, 0, 1);

        for (int tile = 0; tile < ntiles; ++tile) {
            uint32_t seen_early[NG] = {0u, 0u};
            f32x4 acc[8][NG];
#pragma unroll
            for (int rg = 0; rg < 8; ++rg)
#pragma unroll
                for (int gr = 0; gr < NG; ++gr)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[rg][gr][i] = 0.0f;
            static_for<0, KS>([&](auto KSI) __attribute__((always_inline)) {
                constexpr int ks = decltype(KSI)::value;
                constexpr int slot = ks % S, nslot = (ks + 1) % S;
                auto mfma_quad = [&](const half8 (&f)[4], int jq) __attribute__((always_inline)) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int rg = 4 * (jq & 1) + t;
#pragma unroll
                        for (int gr = 0; gr < NG; ++gr)
                            acc[rg][gr] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[t], qf[gr][ks * 2 + (jq >> 1)], acc[rg][gr], 0, 0, 0);
                    }
                };
                half8 f2[4], f3[4];
                read_quad(f2, slot, 2);
                mfma_quad(afn, 0);
                read_quad(f3, slot, 3);
                mfma_quad(bfn, 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                __builtin_amdgcn_sched_barrier(0);
                // publish stage g+1: this wave's pieces of it have landed when only the stage behind it is outstanding
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"i"(VM_MID) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!NOSEL && ks == KS - 2) {
                    // the query's shared threshold for the end of this tile: older than this stage's and the next stage's
                    // LDS-DMA pieces, so a counted wait at the tile end covers it without draining them
                    asm volatile("global_load_dword %0, %2, off sc1\n\tglobal_load_dword %1, %2, off offset:64 sc1"
                                 : "=&v"(seen_early[0]), "=&v"(seen_early[1]) : "v"(my_shared) : "memory");
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
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 1);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 1);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 1);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 1);
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (NOSEL) {
#pragma unroll
                for (int rg = 0; rg < 8; ++rg) asm volatile("" ::"v"(acc[rg][0]), "v"(acc[rg][1]));
            } else {
                // threshold sharing between the lists of a query (coarse_flat_kernel.hpp), once per group
                asm volatile("s_waitcnt vmcnt(%2)" : "+v"(seen_early[0]), "+v"(seen_early[1]) : "i"(4 * (S - 2)) : "memory");
                static_for<0, NG>([&](auto GR) __attribute__((always_inline)) {
                    constexpr int gr = decltype(GR)::value;
                    const uint32_t seen = seen_early[gr];
                    const uint32_t mine_key = order_f32(thr[gr]);
                    if (seen > mine_key) thr[gr] = unorder_f32(seen);
                    else if (g == 0 && valid_q[gr] && mine_key > seen && mine_key > published[gr]) {
                        __hip_atomic_fetch_max(my_shared + 16 * gr, mine_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        published[gr] = mine_key;
                    }
                });
                const int tile_row0 = (t0 + tile) * CO_BN;
                const uint32_t rowbase = (uint32_t)(tile_row0 + 4 * g);
                if (tile < boot_tiles && tile_row0 + CO_BN <= a.n) {
                    // Threshold bootstrap (coarse_w8_kernel): every lane tracks its two best scores per group, the
                    // threshold follows the smallest of the four lanes' second best: 8 rows seen so far score at or above it.
                    static_for<0, NG>([&](auto GR) __attribute__((always_inline)) {
                        constexpr int gr = decltype(GR)::value;
#pragma unroll
                        for (int rg = 0; rg < 8; ++rg)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float v = acc[rg][gr][i];
                                const float lo1 = raw_min_f32(boot1[gr], v);
                                boot1[gr] = raw_max_f32(boot1[gr], v);
                                boot2[gr] = raw_max_f32(boot2[gr], lo1);
                            }
                        const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(boot2[gr]), __float_as_uint(boot2[gr]), false, false);
                        const float m1 = fminf(boot2[gr], __uint_as_float((g & 1) ? s16[0] : s16[1]));
                        const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m1), __float_as_uint(m1), false, false);
                        const float thr0 = fminf(m1, __uint_as_float((g & 2) ? s32[0] : s32[1]));
                        if (thr0 > thr[gr]) thr[gr] = thr0;   // (padding queries keep +inf)
                    });
                }
                const bool ragged = tile_row0 + CO_BN > a.n;
                static_for<0, 8>([&](auto RG) __attribute__((always_inline)) {
                    constexpr int rg = decltype(RG)::value;
                    static_for<0, NG>([&](auto GR) __attribute__((always_inline)) {
                        constexpr int gr = decltype(GR)::value;
                        test_quad(acc[rg][gr], rowbase + (uint32_t)(16 * rg), GR, ragged);
                        // overflow guard: the four lanes of a query append at most 16 entries per row group
                        const uint32_t need = (uint32_t)__builtin_amdgcn_ballot_w64(cnt[gr] > CO_CAP - 16) & 0xffffu;
                        if (__builtin_expect(need != 0u, 0)) compact_need(need, GR);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the run-ahead stages
        asm volatile("" ::"v"(afn[0]), "v"(afn[1]), "v"(afn[2]), "v"(afn[3]));
        asm volatile("" ::"v"(bfn[0]), "v"(bfn[1]), "v"(bfn[2]), "v"(bfn[3]));
        if constexpr (!NOSEL) {
            // ---- end of the list: every query's top-KP entries (unsorted) and its bound go to global memory ----------
            static_for<0, NG>([&](auto GR) __attribute__((always_inline)) {
                constexpr int gr = decltype(GR)::value;
                compact_need((uint32_t)__builtin_amdgcn_ballot_w64(cnt[gr] > KP) & 0xffffu, GR);
                if (valid_q[gr]) {
                    constexpr int PER = KP / 4;
                    const int myq = myq0 + 16 * gr;
                    const size_t o = ((size_t)myq * a.P + ord) * KP;
#pragma unroll
                    for (int e = 0; e < PER; ++e) {
                        const int d = g * PER + e;
                        float sv = -INFINITY;
                        int rw = -1;
                        if (d < cnt[gr]) {
                            sv = *reinterpret_cast<const float *>(smem + qb[gr] + d * 4);
                            rw = (int)*reinterpret_cast<const uint32_t *>(smem + qb[gr] + 256 + d * 4);
                        }
                        a.part_scores[o + d] = sv;
                        a.part_rows[o + d] = rw;
                    }
                    if (g == 0) {
                        a.bounds[(size_t)myq * a.P + ord] = thr[gr];
                        if (t1 == a.ctiles) {   // last list of the query tile: the unused ordinals are empty
                            for (int e = ord + 1; e < a.P; ++e) {
                                const size_t oe = ((size_t)myq * a.P + e) * KP;
                                for (int d = 0; d < KP; ++d) { a.part_scores[oe + d] = -INFINITY; a.part_rows[oe + d] = -1; }
                                a.bounds[(size_t)myq * a.P + e] = -INFINITY;
                            }
                        }
                    }
                }
            });
        }
        __syncthreads();
        u += ntiles;
    }
}


// ---------------------------------------------------------------------------------------------------------------------


// ---- select with a private region per lane -----------------------------------------------------------------------------
// A query's 64-entry buffer is four regions of 16, one per lane that holds its scores: an append is the exec-masked
// store-and-advance of coarse_flat_kernel (no slot arithmetic across lanes). Compaction handles ALL 16 queries of a group
// at once, lane = quarter of a query: its 16 slots in registers, the KP-th best of the four lanes' entries by bisection
// with two lane swaps per probe, survivors stay in their own region (compacted in place). No lane keeps more than
// G16_LANE_KEEP entries, so the four registers of the next row group always fit; a lane that would (most of a query's best
// rows in ONE residue class of rows mod 16) raises the query's threshold instead - the list then ends on a bound inside
// its top KP and the query may fail the certificate: slower, never wrong.
constexpr int G16_REGION = 16;
constexpr int G16_LANE_KEEP = G16_REGION - 4;
constexpr int G16_QBYTES = 516;   // query buffers one LDS bank apart: the 64 lanes of a wave load slot j of their regions in one instruction
__host__ __device__ constexpr int g16_lds_bytes() { return CO_S * CO_STAGE_BYTES + CO_BM * G16_QBYTES + 256; }

struct G16Res { float thr; uint32_t aw; };

// the lane's (G16_LANE_KEEP + 1)-th best value: at most G16_LANE_KEEP of its entries are above it (rare path, out of line)
__device__ __attribute__((noinline)) float g16_lane_cut(const float (&v)[G16_REGION]) {
    float lane_cut = -INFINITY;
#pragma unroll
    for (int j = 0; j < G16_REGION; ++j) {
        int rank = 0, eq = 0;
#pragma unroll
        for (int k = 0; k < G16_REGION; ++k) { rank += (v[k] > v[j]) ? 1 : 0; eq += (v[k] == v[j]) ? 1 : 0; }
        if (rank <= G16_LANE_KEEP && rank + eq > G16_LANE_KEEP) lane_cut = v[j];
    }
    return lane_cut;
}



// EMIT: the survivors (<= KP per query, the lanes' shares in lane order) go to out_scores / out_rows[0..KP) of the query
// instead of back into the regions; the returned thr is the list's bound either way.
// (a real call spills the caller's live accumulators through scratch, and the scratch reloads wait for vmcnt(0): every call
//  drained the LDS-DMA pipeline - 0.82 ms; inlined at its 16 sites per tile the kernel has no scratch traffic)
template <int KP, bool EMIT>
__device__ __forceinline__ G16Res g16_compact(uint32_t aw0, uint32_t aw, float thr, int g, bool store, float *out_scores, int *out_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int R = G16_REGION;
    const int mine = (int)(aw - aw0) >> 2;
    float v[R];
    uint32_t rw[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const float sv = *reinterpret_cast<const float *>(smem + aw0 + 4 * j);
        rw[j] = *reinterpret_cast<const uint32_t *>(smem + aw0 + 256 + 4 * j);
        v[j] = j < mine ? sv : -INFINITY;
    }
    // values of the lane's partners: g ^ 1 (16 lanes away), then the other pair (32 lanes away)
    auto p16 = [&](uint32_t x) -> uint32_t { const auto s = __builtin_amdgcn_permlane16_swap(x, x, false, false); return (g & 1) ? s[0] : s[1]; };
    auto p32 = [&](uint32_t x) -> uint32_t { const auto s = __builtin_amdgcn_permlane32_swap(x, x, false, false); return (g & 2) ? s[0] : s[1]; };
    auto quad_sum = [&](int x) -> int { x += (int)p16((uint32_t)x); x += (int)p32((uint32_t)x); return x; };
    auto quad_maxf = [&](float x) -> float { x = fmaxf(x, __uint_as_float(p16(__float_as_uint(x)))); return fmaxf(x, __uint_as_float(p32(__float_as_uint(x)))); };
    const int total = quad_sum(mine);
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < R; ++j) mx = fmaxf(mx, v[j]);
    mx = quad_maxf(mx);
    // bisection on the order-preserving key; invariant count(v > lo) >= KP > count(v > hi)
    uint32_t lo = order_f32(thr), hi = order_f32(mx);
    int cnt_lo = 0;
#pragma unroll
    for (int j = 0; j < R; ++j) cnt_lo += (v[j] > thr) ? 1 : 0;
    cnt_lo = quad_sum(cnt_lo);
    bool active = total > KP && cnt_lo > KP && hi > lo + 1u;
    while (__builtin_amdgcn_ballot_w64(active) != 0ull) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        const float midf = unorder_f32(mid);
        int c = 0;
#pragma unroll
        for (int j = 0; j < R; ++j) c += (v[j] > midf) ? 1 : 0;
        c = quad_sum(c);
        if (active) {
            if (c >= KP) { lo = mid; cnt_lo = c; } else hi = mid;
            active = cnt_lo > KP && hi > lo + 1u;
        }
    }
    float cut = total > KP ? unorder_f32(lo) : -INFINITY;   // keep v > cut
    float bound = total > KP ? cut : thr;
    int quota_ties = 0;
    float tie = 0.0f;
    const bool tied_case = cnt_lo > KP;   // (possible only after the loop ended on hi == lo + 1: scores tied at key hi)
    bool rewrite = total > KP;
    if (tied_case) {
        tie = unorder_f32(hi);
        int above = 0, tied = 0;
#pragma unroll
        for (int j = 0; j < R; ++j) { above += (v[j] > tie) ? 1 : 0; tied += (v[j] == tie) ? 1 : 0; }
        const int room = KP - quad_sum(above);
        const int t16 = (int)p16((uint32_t)tied);
        const int tpair32 = (int)p32((uint32_t)(tied + t16));
        const int before = ((g & 1) ? t16 : 0) + ((g & 2) ? tpair32 : 0);   // ties of the lanes in front of this one
        quota_ties = max(0, min(tied, room - min(before, room)));
        cut = tie;
        bound = tie;
    }
    int mine_n = quota_ties;
#pragma unroll
    for (int j = 0; j < R; ++j) mine_n += (v[j] > cut) ? 1 : 0;
    if constexpr (!EMIT) {
        // no lane may keep more than G16_LANE_KEEP entries (rare: wave-uniform branch)
        if (__builtin_amdgcn_ballot_w64(mine_n > G16_LANE_KEEP) != 0ull) {
            const float lane_cut = mine_n > G16_LANE_KEEP ? g16_lane_cut(v) : -INFINITY;
            // the query's cut rises to the largest of its lanes' (tied entries at the cut are dropped with it)
            cut = quad_maxf(fmaxf(lane_cut, cut));
            bound = cut;
            quota_ties = 0;
            rewrite = true;
            mine_n = 0;
#pragma unroll
            for (int j = 0; j < R; ++j) mine_n += (v[j] > cut) ? 1 : 0;
        }
    }
    G16Res res;
    res.thr = bound;
    if constexpr (EMIT) {
        const int n16 = (int)p16((uint32_t)mine_n);
        const int npair32 = (int)p32((uint32_t)(mine_n + n16));
        int dest = ((g & 1) ? n16 : 0) + ((g & 2) ? npair32 : 0);
        const int kept = mine_n + n16 + npair32;   // <= KP
        if (store) {
            int ties_left = quota_ties;
#pragma unroll
            for (int j = 0; j < R; ++j) {
                bool take = v[j] > cut;
                if (!take && ties_left > 0 && tied_case && v[j] == tie) { take = true; --ties_left; }
                if (take) { out_scores[dest] = v[j]; out_rows[dest] = (int)rw[j]; ++dest; }
            }
            if (g == 3) for (int d = kept; d < KP; ++d) { out_scores[d] = -INFINITY; out_rows[d] = -1; }
        }
        res.aw = aw;
    } else {
        uint32_t w = aw0;
        int ties_left = quota_ties;
        if (rewrite) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                bool take = v[j] > cut;
                if (!take && ties_left > 0 && tied_case && v[j] == tie) { take = true; --ties_left; }
                if (take) {
                    *reinterpret_cast<float *>(smem + w) = v[j];
                    *reinterpret_cast<uint32_t *>(smem + w + 256) = rw[j];
                    w += 4;
                }
            }
            res.aw = w;
        } else {
            res.aw = aw;
        }
    }
    return res;
}

// coarse_g16r_kernel: the same sweep with the per-lane-region select above. VAR: 1 = no select (TIMING ONLY)
template <int D, int KP = CO_KP, int VAR = 0>
__global__ __launch_bounds__(256, 1) void coarse_g16r_kernel(CoarseFlatArgs a) {
    constexpr int NG = 2;               // query groups of 16 per wave (B operands of v_mfma_f32_16x16x32_f16)
    constexpr bool NOSEL = (VAR & 1) != 0;
    constexpr int S = CO_S;
    constexpr int KS = D / CO_BK;       // stages per tile
    constexpr int NF = D / 32;          // query fragments per lane (one per 32-deep k-step)
    constexpr int VM_MID = 4 * (S - 3); // LDS-DMA pieces (four per wave and stage) that may stay in flight at the mid-stage wait
    static_assert(KS % S == 0, "ring slot must be a compile-time function of the stage");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = lane & 15, g = lane >> 4;
    const int wg = flat_workgroup_of_block((int)blockIdx.x, (int)gridDim.x, a.pos_period);
    const int u_begin = wg * a.units_per_wg;
    const int u_end = min(a.total_units, u_begin + a.units_per_wg);
    if (u_begin >= u_end) return;

    // LDS-DMA: piece p = rows 8 p .. 8 p + 7 of the stage, one full 128-B line each; wave w issues pieces 4 w .. 4 w + 3
    uint32_t src_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row_local = (wave * 4 + i) * 8 + (lane >> 3);
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
    const uint32_t wave_qbase = RING_BYTES + (uint32_t)(wave * 32) * (uint32_t)G16_QBYTES;   // the wave's 32 buffers: group 0 first
    static_assert(KP % 4 == 0 && KP <= CO_CAP - 16, "a query's list is written by its four lanes");
    const int last_tile = a.ctiles - 1;

    half8 qf[NG][NF];
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
        const int myq0 = slot0 + wave * 32 + qi;   // group 0's query of this lane; group 1's is myq0 + 16

        if (mtile != cur_mtile) {   // query fragments (B operand: lane holds Q[query][32 s + 8 g + 0..7]) -> accumulator registers
#pragma unroll
            for (int gr = 0; gr < NG; ++gr) {
                const _Float16 *qrow = a.q16 + (size_t)(myq0 + 16 * gr) * D + 8 * g;
#pragma unroll
                for (int s = 0; s < NF; ++s) qf[gr][s] = *reinterpret_cast<const half8 *>(qrow + 32 * s);
            }
#pragma unroll
            for (int gr = 0; gr < NG; ++gr)
#pragma unroll
                for (int s = 0; s < NF; ++s) asm volatile("" : "+a"(qf[gr][s]));
            cur_mtile = mtile;
        }
        const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<_Float16 *>(a.c16) + (size_t)t0 * CO_BN * D, 0,
            (int)min((size_t)(a.ctiles - t0) * CO_BN * (size_t)(D * 2), (size_t)0x7FFFFFFF), 0x00020000);
        auto issue_stage = [&](int g_tile, int g_ks, int ring_slot) __attribute__((always_inline)) {
            const int trow = min(g_tile, last_tile - t0);   // stages past the sweep re-read valid memory, never consumed
            const uint32_t soff = (uint32_t)trow * (uint32_t)(CO_BN * D * 2) + (uint32_t)g_ks * (CO_BK * 2);
            __attribute__((address_space(3))) void *ldst =
                (__attribute__((address_space(3))) void *)(smem + ring_slot * CO_STAGE_BYTES + wave * 4096);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, ldst, 16, src_off[0], soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, ldst, 16, src_off[1], soff, 1024, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, ldst, 16, src_off[2], soff, 2048, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, ldst, 16, src_off[3], soff, 3072, 0);
        };

        // ---- select state: TWO queries per lane (one per group); a lane appends to its own region of each ---------------
        bool valid_q[NG];
        float thr[NG];         // the query's threshold (upper bound on every score dropped)
        uint32_t aw[NG], aw0[NG], published[NG];   // next slot of the lane's region (LDS byte address), the region's first slot
        float boot1[NG], boot2[NG];
#pragma unroll
        for (int gr = 0; gr < NG; ++gr) {
            valid_q[gr] = (myq0 + 16 * gr) < a.nq;
            thr[gr] = valid_q[gr] ? -INFINITY : INFINITY;
            aw0[gr] = wave_qbase + (uint32_t)(16 * gr + qi) * (uint32_t)G16_QBYTES + (uint32_t)(g * G16_REGION * 4);
            aw[gr] = aw0[gr];
            published[gr] = 0u;
            boot1[gr] = -INFINITY; boot2[gr] = -INFINITY;
        }
        const int boot_tiles = ntiles >= CO_BOOT_MIN_TILES ? min(a.boot_tiles, ntiles / 3) : 0;
        unsigned int *my_shared = a.shared_thr + myq0;   // (group 1: + 16)
        // the four registers of one row group and query group (four consecutive rows): ONE branch when no lane passes any;
        // the appends run exec-masked from the four masks (coarse_flat_kernel's quad select); then the overflow guard
        auto test_quad = [&](const f32x4 &pa, uint32_t row0, auto GR, bool ragged) __attribute__((always_inline)) {
            constexpr int gr = decltype(GR)::value;
            float v0 = pa[0], v1 = pa[1], v2 = pa[2], v3 = pa[3];
            if (ragged) {   // (wave-uniform: the corpus's last tile only)
                if ((int)(row0 + 0u) >= a.n) v0 = -INFINITY;
                if ((int)(row0 + 1u) >= a.n) v1 = -INFINITY;
                if ((int)(row0 + 2u) >= a.n) v2 = -INFINITY;
                if ((int)(row0 + 3u) >= a.n) v3 = -INFINITY;
            }
            const unsigned long long m0 = __builtin_amdgcn_ballot_w64(v0 > thr[gr]), m1 = __builtin_amdgcn_ballot_w64(v1 > thr[gr]);
            const unsigned long long m2 = __builtin_amdgcn_ballot_w64(v2 > thr[gr]), m3 = __builtin_amdgcn_ballot_w64(v3 > thr[gr]);
            if (__builtin_expect(((m0 | m1) | (m2 | m3)) != 0ull, 0)) {
                // EXEC is all ones here (256-thread blocks, wave-uniform control flow down to this point)
                uint32_t r1, r2, r3;
                asm volatile("v_or_b32_e32 %1, 1, %12\n\t"
                             "v_or_b32_e32 %2, 2, %12\n\t"
                             "v_or_b32_e32 %3, 3, %12\n\t"
                             "s_mov_b64 exec, %4\n\t"
                             "ds_write2st64_b32 %0, %8, %12 offset1:1\n\t"
                             "v_add_u32_e32 %0, 4, %0\n\t"
                             "s_mov_b64 exec, %5\n\t"
                             "ds_write2st64_b32 %0, %9, %1 offset1:1\n\t"
                             "v_add_u32_e32 %0, 4, %0\n\t"
                             "s_mov_b64 exec, %6\n\t"
                             "ds_write2st64_b32 %0, %10, %2 offset1:1\n\t"
                             "v_add_u32_e32 %0, 4, %0\n\t"
                             "s_mov_b64 exec, %7\n\t"
                             "ds_write2st64_b32 %0, %11, %3 offset1:1\n\t"
                             "v_add_u32_e32 %0, 4, %0\n\t"
                             "s_mov_b64 exec, -1"
                             : "+v"(aw[gr]), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                             : "s"(m0), "s"(m1), "s"(m2), "s"(m3), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(row0)
                             : "memory");
                // a lane's region takes the four registers of the next quad as long as it holds at most G16_LANE_KEEP
                if (__builtin_amdgcn_ballot_w64((aw[gr] - aw0[gr]) > (uint32_t)(G16_LANE_KEEP * 4)) != 0ull) {
                    const G16Res r = g16_compact<KP, false>(aw0[gr], aw[gr], thr[gr], g, false, nullptr, nullptr);
                    thr[gr] = r.thr;
                    aw[gr] = r.aw;
                }
            }
        };

        // prologue: stages 0..S-2 in flight, stage 0 published, its first two quads of fragments read
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int p = 0; p < S - 1; ++p) issue_stage(p / KS, p % KS, p % S);
        half8 afn[4], bfn[4];
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"i"(4 * (S - 2)) : "memory");
        read_quad(afn, 0, 0);
        read_quad(bfn, 0, 1);

        for (int tile = 0; tile < ntiles; ++tile) {
            uint32_t seen_early[NG] = {0u, 0u};
            f32x4 acc[8][NG];
#pragma unroll
            for (int rg = 0; rg < 8; ++rg)
#pragma unroll
                for (int gr = 0; gr < NG; ++gr)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[rg][gr][i] = 0.0f;
            static_for<0, KS>([&](auto KSI) __attribute__((always_inline)) {
                constexpr int ks = decltype(KSI)::value;
                constexpr int slot = ks % S, nslot = (ks + 1) % S;
                auto mfma_quad = [&](const half8 (&f)[4], int jq) __attribute__((always_inline)) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int rg = 4 * (jq & 1) + t;
#pragma unroll
                        for (int gr = 0; gr < NG; ++gr)
                            acc[rg][gr] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[t], qf[gr][ks * 2 + (jq >> 1)], acc[rg][gr], 0, 0, 0);
                    }
                };
                half8 f2[4], f3[4];
                read_quad(f2, slot, 2);
                mfma_quad(afn, 0);
                read_quad(f3, slot, 3);
                mfma_quad(bfn, 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                __builtin_amdgcn_sched_barrier(0);
                // publish stage g+1: this wave's pieces of it have landed when only the stage behind it is outstanding
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"i"(VM_MID) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!NOSEL && ks == KS - 2) {
                    // the query's shared threshold for the end of this tile: older than this stage's and the next stage's
                    // LDS-DMA pieces, so a counted wait at the tile end covers it without draining them
                    asm volatile("global_load_dword %0, %2, off sc1\n\tglobal_load_dword %1, %2, off offset:64 sc1"
                                 : "=&v"(seen_early[0]), "=&v"(seen_early[1]) : "v"(my_shared) : "memory");
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
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 1);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 1);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 1);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 1);
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (NOSEL) {
#pragma unroll
                for (int rg = 0; rg < 8; ++rg) asm volatile("" ::"v"(acc[rg][0]), "v"(acc[rg][1]));
            } else {
                // threshold sharing between the lists of a query (coarse_flat_kernel.hpp), once per group
                asm volatile("s_waitcnt vmcnt(%2)" : "+v"(seen_early[0]), "+v"(seen_early[1]) : "i"(4 * (S - 2)) : "memory");
                static_for<0, NG>([&](auto GR) __attribute__((always_inline)) {
                    constexpr int gr = decltype(GR)::value;
                    const uint32_t seen = seen_early[gr];
                    const uint32_t mine_key = order_f32(thr[gr]);
                    if (seen > mine_key) thr[gr] = unorder_f32(seen);
                    else if (g == 0 && valid_q[gr] && mine_key > seen && mine_key > published[gr]) {
                        __hip_atomic_fetch_max(my_shared + 16 * gr, mine_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        published[gr] = mine_key;
                    }
                });
                const int tile_row0 = (t0 + tile) * CO_BN;
                const uint32_t rowbase = (uint32_t)(tile_row0 + 4 * g);
                if (tile < boot_tiles && tile_row0 + CO_BN <= a.n) {
                    // Threshold bootstrap (coarse_w8_kernel): every lane tracks its two best scores per group, the
                    // threshold follows the smallest of the four lanes' second best: 8 rows seen so far score at or above it.
                    static_for<0, NG>([&](auto GR) __attribute__((always_inline)) {
                        constexpr int gr = decltype(GR)::value;
#pragma unroll
                        for (int rg = 0; rg < 8; ++rg)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float v = acc[rg][gr][i];
                                const float lo1 = raw_min_f32(boot1[gr], v);
                                boot1[gr] = raw_max_f32(boot1[gr], v);
                                boot2[gr] = raw_max_f32(boot2[gr], lo1);
                            }
                        const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(boot2[gr]), __float_as_uint(boot2[gr]), false, false);
                        const float m1 = fminf(boot2[gr], __uint_as_float((g & 1) ? s16[0] : s16[1]));
                        const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(m1), __float_as_uint(m1), false, false);
                        const float thr0 = fminf(m1, __uint_as_float((g & 2) ? s32[0] : s32[1]));
                        if (thr0 > thr[gr]) thr[gr] = thr0;   // (padding queries keep +inf)
                    });
                }
                const bool ragged = tile_row0 + CO_BN > a.n;
                static_for<0, 8>([&](auto RG) __attribute__((always_inline)) {
                    constexpr int rg = decltype(RG)::value;
                    static_for<0, NG>([&](auto GR) __attribute__((always_inline)) {
                        constexpr int gr = decltype(GR)::value;
                        test_quad(acc[rg][gr], rowbase + (uint32_t)(16 * rg), GR, ragged);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the run-ahead stages
        asm volatile("" ::"v"(afn[0]), "v"(afn[1]), "v"(afn[2]), "v"(afn[3]));
        asm volatile("" ::"v"(bfn[0]), "v"(bfn[1]), "v"(bfn[2]), "v"(bfn[3]));
        if constexpr (!NOSEL) {
            // ---- end of the list: every query's top-KP entries (unsorted) and its bound go to global memory ----------
            static_for<0, NG>([&](auto GR) __attribute__((always_inline)) {
                constexpr int gr = decltype(GR)::value;
                const int myq = myq0 + 16 * gr;
                const size_t o = ((size_t)min(myq, a.nq - 1) * a.P + ord) * KP;
                const G16Res r = g16_compact<KP, true>(aw0[gr], aw[gr], thr[gr], g, valid_q[gr], a.part_scores + o, a.part_rows + o);
                if (valid_q[gr] && g == 0) {
                    a.bounds[(size_t)myq * a.P + ord] = r.thr;
                    if (t1 == a.ctiles) {   // last list of the query tile: the unused ordinals are empty
                        for (int e = ord + 1; e < a.P; ++e) {
                            const size_t oe = ((size_t)myq * a.P + e) * KP;
                            for (int d = 0; d < KP; ++d) { a.part_scores[oe + d] = -INFINITY; a.part_rows[oe + d] = -1; }
                            a.bounds[(size_t)myq * a.P + e] = -INFINITY;
                        }
                    }
                }
            });
        }
        __syncthreads();
        u += ntiles;
    }
}


// ---------------------------------------------------------------------------------------------------------------------


}  // namespace icd
