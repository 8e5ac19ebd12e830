// coarse_flat_kernel.hpp — the product form of the fp16-MFMA coarse pass: a software-pipelined
// stage over a FLAT partition of the (query tile x corpus tile) grid.
//
// Replaces the scoring + k-selection inside MilvusClient.search on the FLAT/IP index
// (services/milvus_service.py:280-285) for batches; exactness is restored by finalize.hpp.
//
// Work unit = one 128-query x 128-row tile. Units are numbered u = mtile * ctiles + tile and work-group w
// takes the contiguous range [w U, (w+1) U): with U = ceil(units / CUs) every CU gets the same number of
// tiles whatever the ratio of query tiles to CUs is (the (mtile, chunk) grid of coarse_kernel.hpp left 19 of
// 256 CUs idle at 79 query tiles x 3 chunks). A work-group's run inside one query tile is cut into lists of at
// most `list_tiles` tiles; every list is one top-KP candidate list of that query tile:
//     part[query][ordinal][KP] (unsorted, -1 rows = empty) and bounds[query][ordinal]
// where bounds = the list's final threshold = an upper bound on every score the list dropped (-inf if it
// dropped nothing). Ordinals count the lists of a query tile in row order; the work-group that reaches the end
// of a query tile also writes the unused ordinals as empty.
#pragma once
#include "coarse_common.hpp"
#include "flat_partition.hpp"

namespace icd {

struct CoarseFlatArgs {
    const _Float16 *q16;     // [nq_pad][D], rows >= nq are zero
    const _Float16 *c16;     // [n_pad][D], rows >= n are zero
    int nq;
    int n;                   // valid rows
    int n_pad;               // multiple of 128
    int ctiles;              // n_pad / 128
    int total_units;         // query tiles * ctiles
    int units_per_wg;        // U
    int pos_period;          // ctiles / gcd(U, ctiles): work-groups l and l + pos_period start on the same corpus tile
    int list_tiles;          // a list covers at most this many tiles
    int boot_tiles;          // threshold bootstrap over at most this many first tiles of a list (less for larger k)
    int sparse_from;         // (VAR & 16777216) from this tile of a list on, 8-register groups are pre-filtered by their maximum
    int P;                   // list slots per query (>= the largest number of lists of any query tile)
    float *part_scores;      // [nq][P][KP] (KP = the instantiation's candidates per list)
    int *part_rows;
    float *bounds;           // [nq][P]
    unsigned int *shared_thr; // [nq_pad] order_f32 keys, cleared before the launch: max over a query's lists of their thresholds
    // second pass over the queries the first pass could not certify (icd_search.hip, "pass 2"): the query slots are the
    // entries of a device-side list whose length is only known on the device
    const int *nq_ptr;       // nullable: number of active query slots = min(*nq_ptr, nq); total_units follows from it
    const int *qlist;        // nullable: query slot s reads row qlist[s] of q16 (lists, bounds, shared_thr are indexed by slot)
    int nwg_virtual;         // logical work-groups of a full batch (PERSIST launches size their own count from *nq_ptr)
    int skip_below;          // PERSIST: nothing to do when at most this many slots are active (the streaming kernel is cheaper there)
    unsigned long long *dbg;  // diagnostic builds only (VAR & 1024): [block][wave][8] cycle sums
    // (VAR & 67108864) pacing of the work-groups that sweep the same corpus tiles (see the VAR list); pace == nullptr: off
    unsigned int *pace;      // [pace_period][pace_epochs] arrivals per (class, epoch), zeroed before the launch
    int pace_period;         // work-groups l and l + pace_period start on the same corpus tile (flat_class_period: BEFORE the placement split)
    int pace_epochs;         // epochs per class (row stride of `pace`)
    int pace_shift;          // an epoch = 2^pace_shift tiles
    int pace_lead;           // a work-group starts epoch e only when every member of its class has finished epoch e - pace_lead
};

constexpr int CO_BOOT_MIN_TILES = 6;   // lists at least this long bootstrap their threshold ...
constexpr int CO_KP_WIDE = 24;         // candidates per list of the instantiation for larger k (see icd_search.hip)
constexpr int CO_BOOT_TILES = 8;       // ... over their first tiles
constexpr int PACE_SPIN_LIMIT = 20000;   // polls (s_sleep 16 + one sc1 load each, ~1 us) before a work-group gives pacing up: ~20 ms
constexpr int CO_SPARSE_FROM = 40;     // (VAR & 16777216) tiles of a list before its 8-register groups are pre-filtered by their maximum

// End of a list, all 32 queries of the wave at once (lane = half a query; the one-query-at-a-time compaction of
// Sel2Ops costs ~1 400 cycles per query: 45 000 per list and wave, 7 % of a 90-tile sweep). Every lane loads the 32
// scores and rows of its half of the query's buffer (slots rotated by the query index: conflict-free), the two lanes
// of a query bisect the KP-th best score together (counts exchanged with v_permlane32_swap), and the survivors go
// straight to the list in global memory - the LDS buffer is not rewritten, the list ends here.
//   slots [0, nlo) and [64 - nhi, 64) of the buffer are valid; thr = the query's threshold (entries at or below an
//   adopted threshold may still be present: they are candidates like any other).
// Returns the list's bound (upper bound on everything dropped). Ties at the KP-th score: the tied entries that fit
// are kept, the bound is the tied score.
template <int KP>
__device__ __forceinline__ float flush_emit_parallel(const char *smem, uint32_t qb, int h, int rot, int nlo, int nhi, float thr,
                                                      bool store, float *out_scores, int *out_rows) {
    constexpr int CAP = 64, HALF = 32;
    float v[HALF];
    uint32_t rw[HALF];
    const int lo_end = nlo, hi_begin = CAP - nhi;
#pragma unroll
    for (int j = 0; j < HALF; ++j) {
        const int slot = HALF * h + ((j + rot) & (HALF - 1));
        const bool valid = slot < lo_end || slot >= hi_begin;
        const float sv = *reinterpret_cast<const float *>(smem + qb + slot * 4);
        rw[j] = *reinterpret_cast<const uint32_t *>(smem + qb + CAP * 4 + slot * 4);
        v[j] = valid ? sv : -INFINITY;
    }
    auto pair_sum = [&](int x) {
        const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
        return x + (int)(h ? sw[0] : sw[1]);
    };
    const int total = nlo + nhi;
    // bisection on the order-preserving key; invariant count(v > lo) >= KP > count(v > hi)
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < HALF; ++j) mx = fmaxf(mx, v[j]);
    {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(mx, __uint_as_float(h ? sw[0] : sw[1]));
    }
    uint32_t lo = order_f32(thr), hi = order_f32(mx);
    int cnt_lo = 0;
#pragma unroll
    for (int j = 0; j < HALF; ++j) cnt_lo += (v[j] > thr) ? 1 : 0;
    cnt_lo = pair_sum(cnt_lo);
    bool active = total > KP && cnt_lo > KP && hi > lo + 1u;
    while (__builtin_amdgcn_ballot_w64(active) != 0ull) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        const float midf = unorder_f32(mid);
        int c = 0;
#pragma unroll
        for (int j = 0; j < HALF; ++j) c += (v[j] > midf) ? 1 : 0;
        c = pair_sum(c);
        if (active) {
            if (c >= KP) { lo = mid; cnt_lo = c; } else hi = mid;
            active = cnt_lo > KP && hi > lo + 1u;
        }
    }
    // survivors: above lo when that leaves at most KP, else (ties at key hi = lo + 1) above hi plus as many tied as fit
    // (a buffer of more than KP entries keeps those above lo >= thr - at most KP of them unless scores tie; what it
    //  drops is at or below lo. A buffer of at most KP entries keeps everything and has dropped nothing new.)
    float cut = total > KP ? unorder_f32(lo) : -INFINITY;   // keep v > cut
    float bound = total > KP ? cut : thr;
    int quota_ties = 0;
    float tie = 0.0f;
    if (cnt_lo > KP) {   // (possible only after the loop ended on hi == lo + 1)
        tie = unorder_f32(hi);
        int above = 0, tied = 0;
#pragma unroll
        for (int j = 0; j < HALF; ++j) { above += (v[j] > tie) ? 1 : 0; tied += (v[j] == tie) ? 1 : 0; }
        const int above_all = pair_sum(above);
        const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)tied, (unsigned)tied, false, false);
        const int tied_low_lane = h ? (int)sw[0] : tied;           // the low lane's ties are served first
        const int room = KP - above_all;
        quota_ties = h ? max(0, min(tied, room - min(tied_low_lane, room))) : min(tied, room);
        cut = tie;
        bound = tie;
    }
    // destination slots: the low lane's survivors first
    int mine_n = 0;
#pragma unroll
    for (int j = 0; j < HALF; ++j) mine_n += (v[j] > cut) ? 1 : 0;
    mine_n += quota_ties;
    const auto swn = __builtin_amdgcn_permlane32_swap((unsigned)mine_n, (unsigned)mine_n, false, false);
    const int other_n = (int)(h ? swn[0] : swn[1]);
    int dest = h ? other_n : 0;
    const int kept = mine_n + other_n;   // <= KP
    if (store) {
        int ties_left = quota_ties;
#pragma unroll
        for (int j = 0; j < HALF; ++j) {
            bool take = v[j] > cut;
            if (!take && ties_left > 0 && v[j] == tie && cnt_lo > KP) { take = true; --ties_left; }
            if (take) {
                out_scores[dest] = v[j];
                out_rows[dest] = (int)rw[j];
                ++dest;
            }
        }
        if (h == 0) {
            for (int d = kept; d < KP; ++d) { out_scores[d] = -INFINITY; out_rows[d] = -1; }
        }
    }
    return bound;
}

// Compaction of ALL 32 queries of a wave at once (lane = half a query, as flush_emit_parallel): every buffer that holds
// more than KP entries keeps its KP best at the front (unsorted; ties at the KP-th score: as many of the tied as fit) and
// the query's threshold becomes the bound on what was dropped. The one-query-at-a-time compaction of Sel2Ops::check costs
// ~1 200 cycles and happens at a different register in every wave, so each tile end lasts as long as the slowest wave's
// compactions (about one per tile and work-group: 8 % of the sweep); this one costs ~250 cycles per query and all four
// waves run it at the same tile ends (CoarseFlatArgs-independent schedule, see the kernel).
template <int KP, typename Ops>
__device__ __forceinline__ void compact_all_parallel(char *smem, Sel2 &st, uint32_t qb, int h, int rot) {
    constexpr int CAP = 64, HALF = 32;
    const int mine = Ops::used(st, h);
    const auto swm = __builtin_amdgcn_permlane32_swap((unsigned)mine, (unsigned)mine, false, false);
    const int other = (int)(h ? swm[0] : swm[1]);
    const int nlo = st.kept + (h ? other : mine), nhi = h ? mine : other;
    const int total = nlo + nhi;
    float v[HALF];
    uint32_t rw[HALF];
    const int lo_end = nlo, hi_begin = CAP - nhi;
#pragma unroll
    for (int j = 0; j < HALF; ++j) {
        const int slot = HALF * h + ((j + rot) & (HALF - 1));
        const bool valid = slot < lo_end || slot >= hi_begin;
        const float sv = *reinterpret_cast<const float *>(smem + qb + slot * 4);
        rw[j] = *reinterpret_cast<const uint32_t *>(smem + qb + CAP * 4 + slot * 4);
        v[j] = valid ? sv : -INFINITY;
    }
    auto pair_sum = [&](int x) {
        const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
        return x + (int)(h ? sw[0] : sw[1]);
    };
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < HALF; ++j) mx = fmaxf(mx, v[j]);
    {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(mx, __uint_as_float(h ? sw[0] : sw[1]));
    }
    // bisection on the order-preserving key; invariant count(v > lo) >= KP > count(v > hi)
    uint32_t lo = order_f32(st.thr), hi = order_f32(mx);
    int cnt_lo = 0;
#pragma unroll
    for (int j = 0; j < HALF; ++j) cnt_lo += (v[j] > st.thr) ? 1 : 0;
    cnt_lo = pair_sum(cnt_lo);
    bool active = total > KP && cnt_lo > KP && hi > lo + 1u;
    while (__builtin_amdgcn_ballot_w64(active) != 0ull) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        const float midf = unorder_f32(mid);
        int c = 0;
#pragma unroll
        for (int j = 0; j < HALF; ++j) c += (v[j] > midf) ? 1 : 0;
        c = pair_sum(c);
        if (active) {
            if (c >= KP) { lo = mid; cnt_lo = c; } else hi = mid;
            active = cnt_lo > KP && hi > lo + 1u;
        }
    }
    float cut = unorder_f32(lo);   // keep v > cut
    int quota_ties = 0;
    float tie = 0.0f;
    if (cnt_lo > KP) {   // (possible only after the loop ended on hi == lo + 1: scores tied at key hi)
        tie = unorder_f32(hi);
        int above = 0, tied = 0;
#pragma unroll
        for (int j = 0; j < HALF; ++j) { above += (v[j] > tie) ? 1 : 0; tied += (v[j] == tie) ? 1 : 0; }
        const int above_all = pair_sum(above);
        const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)tied, (unsigned)tied, false, false);
        const int tied_low_lane = h ? (int)sw[0] : tied;           // the low lane's ties are served first
        const int room = KP - above_all;
        quota_ties = h ? max(0, min(tied, room - min(tied_low_lane, room))) : min(tied, room);
        cut = tie;
    }
    int mine_n = 0;
#pragma unroll
    for (int j = 0; j < HALF; ++j) mine_n += (v[j] > cut) ? 1 : 0;
    mine_n += quota_ties;
    const auto swn = __builtin_amdgcn_permlane32_swap((unsigned)mine_n, (unsigned)mine_n, false, false);
    const int other_n = (int)(h ? swn[0] : swn[1]);
    if (total > KP) {   // (both lanes of a query agree; a buffer of at most KP entries stays as it is)
        int dest = h ? other_n : 0;
        int ties_left = quota_ties;
#pragma unroll
        for (int j = 0; j < HALF; ++j) {
            bool take = v[j] > cut;
            if (!take && ties_left > 0 && v[j] == tie && cnt_lo > KP) { take = true; --ties_left; }
            if (take) {
                *reinterpret_cast<float *>(smem + qb + dest * 4) = v[j];
                *reinterpret_cast<uint32_t *>(smem + qb + CAP * 4 + dest * 4) = rw[j];
                ++dest;
            }
        }
        const int kept = mine_n + other_n;   // <= KP
        st.thr = cut;   // every kept entry is above it (or tied with it), every dropped one at or below; >= the old threshold
        st.kept = kept;
        st.aw0 = h ? qb + (CAP - 1) * 4 : qb + (uint32_t)kept * 4;
        st.aw = st.aw0;
    }
}

// VAR: bit flags of the stage variants (A/B builds instantiate several, `make ABLATE=1`; the product one is CF_PRODUCT_VAR)
//   1    shared-threshold load issued two stages before the tile end (asm, counted wait) instead of a drained load after it
//   2    LDS-DMA pieces one behind each of the four MFMAs that follow the barrier instead of a burst in front of them
//   4    the MFMAs of a k-step issued in the reverse order of their fragment reads (one s_waitcnt per k-step instead of four)
//   8    query fragments pinned to accumulator registers
//   16   select by QUADS of score registers (four consecutive rows): four v_cmp into four SGPR pairs, OR-ed, ONE scalar
//        branch per quad (one VALU -> SALU round trip instead of four); the appends of a quad with a passing lane run
//        exec-masked from the four masks, no compare is repeated
//   128  A fragments read two k-steps ahead of their MFMAs instead of one
//   1024 diagnostic: s_memtime stamps around the mid-stage wait, the barrier and the select (CoarseFlatArgs::dbg)
//   32768  the sweep on v_mfma_f32_16x16x32_f16 (the chip holds a higher clock on this shape: loop -12 %): the wave's 32 queries
//        are two B-operand groups of 16, every corpus fragment (16 rows x 32 halves) feeds two MFMAs; a query's scores then
//        sit in FOUR lanes (l & 15 equal) - 32 v_permlane16_swap per tile exchange group 1 of the even 16-lane rows with
//        group 0 of the odd ones, after which lane l holds 64 scores of query l & 31 again (rows 16 rg + 8 (l >> 5) + 0..7)
//        and the two-lane select below runs unchanged
//   32 / 2048  every 16 / 24 tiles all four waves compact ALL their queries at once (compact_all_parallel) at the tile end
//   131072 __builtin_amdgcn_s_setprio(1) behind every stage barrier, (0) in front of the next (cdna_hip_programming.md T5: on the
//        8-phase GEMM template the pair keeps hipcc from moving MFMAs across the raw barriers; here the sched_barrier(0)
//        pins already do that and one wave per SIMD has nobody to take priority from: A/B in profiles/r03_ab_coarse_variants.log)
//   262144 / 524288  the shared-threshold exchange of a tile end (load of the query's published threshold, adopt or publish) only on
//        every 2nd / 4th tile (and always on a list's last tile but one) instead of every tile
//   1048576  ONE barrier per TWO stages: ring of six 16-KB slots (no compaction scratch: the rare tie-ranking path ranks with
//        v_readlane), the mid-stage wait + barrier of the even stages publishes the next two stages and is followed by
//        the LDS-DMA pieces of two stages (eight, two behind each MFMA group); odd stages run without wait or barrier
//   2097152  the four LDS-DMA pieces of a stage spread over a WHOLE stage interval, one per eight MFMAs (pieces 0, 1 in the half
//        stage behind the barrier, pieces 2, 3 in the next stage's half in front of its barrier) instead of one behind each of
//        the first four MFMA pairs: the CU's one address unit takes 16 cycles per piece and the four waves issue in step, so
//        four pieces per 32 cycles queue (a piece stalls its wave ~45 cycles); at one per 128 cycles they do not
//   4194304  threshold bootstrap with three INDEPENDENT instructions per score (b1' = max(b1, v), b2' = med3(b1, b2, v),
//        b3' = med3(b2, b3, v): with b1 >= b2 >= b3 these are the three largest of {b1, b2, b3, v}) on two register chains
//        (even / odd registers, merged once per tile) instead of the five-instruction min / max ladder on one chain
//   8388608  the overflow guard of the select (every 8 registers) only when some lane has appended since the last one: a
//        wave-uniform flag set by the append path; a warm list passes most 8-register groups without an append
//   16777216  from tile `sparse_from` of a list on, the select first reduces every 8 registers (two quads) to their lane
//        maximum (three v_max3 + one v_max) and tests THAT against the threshold: one compare and one scalar branch per 8
//        registers while nothing passes, the two quads' own tests only behind it. A warm list passes most groups untouched
//        (a 1.25 M-row shard's lists are 4 900 tiles long and append a handful of rows per hundred tiles)
//   67108864  PACING of a class (the work-groups that start on the same corpus tile and sweep the same tiles in the same order:
//        l, l + T, l + 2 T, ...). On a 1.25 M-row shard a sweep is 4 900 tiles long and the members drift apart by more tiles
//        than an XCD's L2 holds (20 tiles of 196 KB): every work-group then fetches the image for itself - 20 to 70 GB of
//        fabric traffic per launch against 1.96 GB algorithmic, different from run to run (profiles/r04_pmc_traffic_rowshard.json).
//        Every 2^pace_shift tiles a work-group reports the epoch it has finished (one no-return agent-scope atomic add per
//        class and epoch) and may start the next one only when ALL members have finished the epoch pace_lead back: the class
//        stays within pace_lead epochs. The counter is polled with the early threshold load (same place, same counted
//        wait: no extra latency when the class is in step); a work-group that is ahead spins with s_sleep, BOUNDED - after
//        PACE_SPIN_LIMIT polls it gives pacing up for the rest of the launch (speed only: results never depend on it)
//   33554432  diagnostic: every wave stamps s_memtime / s_memrealtime once at its start and once at its end and leaves the two
//        differences in CoarseFlatArgs::dbg: the in-kernel clock the chip holds under this kernel (MI355X_MICROARCH.md,
//        DVFS give-back item 6); nothing inside the loops changes
//   TIMING ONLY (the results are not the scores; they size the parts of the kernel, profiles/r02_coarse_loop_decomposition.log):
//   256 no s_barrier   512 no wait for the LDS-DMA pieces   4096 thresholds at +inf (nothing passes)
//   8192 no select at all   16384 no LDS-DMA inside the tile loop   65536 (with 32768) no lane swaps
// Variants that were measured and dropped (per-wave DMA slots, branch-free select, 3/6-stage rings, select deferred into
// the next tile's MFMA gaps, the first two selects built for the 16x16x32 shape before the lane swap let it keep this
// one - bit 32768 IS that shape and is part of the product) live in experiments/r02_flat_variants/ with their logs.
constexpr int CF_PRODUCT_VAR = 1 + 2 + 8 + 128 + 16 + 2048 + 32768;   // measured: profiles/r02_ab_flat_variants.log, profiles/r02_coarse_variants_rg_w8_all.log (+ 16 + 2048: -2 %; + 32768: -2 % at 37 000 rows, -4 % on 1.25 M-row shards)
// The first pass over a corpus whose fp16 image stays in the Infinity Cache (the 37k - 40k-row ICD corpus: 57 - 62 MB): the
// LDS-DMA pieces spread over a whole stage interval and the three-instruction bootstrap. Interleaved A/Bs on four boxes
// (profiles/r04_ab_coarse_variants.log): 10 000 x 37 000 -0.9 ... -2.6 %, the family corpus in wide mode (a third of a
// list's tiles are bootstrap tiles) -3 %; on a 1.25 M-row shard streamed from HBM the later issue of half the pieces costs
// +6 % (less run-ahead for a longer latency), so shards keep CF_PRODUCT_VAR (icd_search.hip picks by the image's size).
constexpr int CF_CACHED_VAR = CF_PRODUCT_VAR + 2097152 + 4194304;
// images streamed from HBM (row shards): the classes are paced (VAR list, 67108864)
constexpr int CF_PACED_VAR = CF_PRODUCT_VAR + 67108864;
__host__ __device__ constexpr int cf_ring_stages(int var) { return (var & 1048576) ? 6 : CO_S; }
__host__ __device__ constexpr int cf_lds_bytes(int var) { return cf_ring_stages(var) * CO_STAGE_BYTES + CO_BM * CO_CAP * 8 + ((var & 1048576) ? 0 : 4 * 256); }
#define ICD_CF_STAMP(t) do { __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); \
    __builtin_amdgcn_sched_barrier(0); } while (0)

// PERSIST: the grid is smaller than the logical work-group count (one block per CU) and every block loops over its share -
// the form of the second pass, whose work is sized on the device (most launches find nothing to do and must cost nothing:
// 256 blocks that read one counter and leave). The first pass launches one block per logical work-group.
template <int D, int VAR = CF_PRODUCT_VAR, int KP = CO_KP, bool PERSIST = false>
__global__ __launch_bounds__(256, 1) void coarse_flat_kernel(CoarseFlatArgs a) {
    // appends per lane between compactions: the buffer holds the KP kept entries + 2 x (quota + one check interval)
    constexpr int CO_QUOTA = (CO_CAP - KP) / 2 - CO_CHECK_EVERY;   // 16 at KP = 16, 12 at KP = 24
    static_assert(CO_QUOTA >= 8 && KP + 2 * (CO_QUOTA + CO_CHECK_EVERY) <= CO_CAP, "candidate buffer layout");
    constexpr bool EARLY_THR = (VAR & 1) != 0;
    constexpr bool DMA_SPREAD = (VAR & 2) != 0;
    constexpr bool Q_AGPR = (VAR & 8) != 0;
    constexpr bool PF2 = (VAR & 128) != 0;
    constexpr bool REV_WAIT = (VAR & 4) != 0;
    constexpr bool QUAD = (VAR & 16) != 0;
    constexpr bool SETPRIO = (VAR & 131072) != 0;
    constexpr int EXCH = (VAR & 524288) ? 4 : ((VAR & 262144) ? 2 : 1);   // tiles between two shared-threshold exchanges
    constexpr bool X16 = (VAR & 32768) != 0;   // v_mfma_f32_16x16x32_f16: a wave's 32 queries as two groups of 16 (see the VAR list)
    static_assert(!X16 || (QUAD && (VAR & 8) != 0 && D % 32 == 0), "the 16x16x32 form is built on the quad select and pinned queries");
    constexpr int EPOCH = ((VAR & 32) && (VAR & 2048)) ? 32 : ((VAR & 32) ? 16 : ((VAR & 2048) ? 24 : 0));   // tiles between the synchronised compactions of all queries
    constexpr bool NOBAR = (VAR & 256) != 0, NOVM = (VAR & 512) != 0, STAMPS = (VAR & 1024) != 0;
    constexpr bool NOPASS = (VAR & 4096) != 0, NOSEL = (VAR & 8192) != 0, NODMA = (VAR & 16384) != 0;
    constexpr bool PAIRBAR = (VAR & 1048576) != 0;
    constexpr bool DMA_WIDE = (VAR & 2097152) != 0;
    constexpr bool BOOT_MED3 = (VAR & 4194304) != 0;
    constexpr bool DIRTY_GUARD = (VAR & 8388608) != 0;
    constexpr bool SPARSE_PRE = (VAR & 16777216) != 0;
    static_assert(!SPARSE_PRE || (QUAD && X16), "the group pre-filter is built on the quad select of the 16x16x32 form");
    // (only the 16x16x32 branch of mfma4 honours the half-calls the wide spread splits a k-step into)
    static_assert(!DMA_WIDE || (X16 && DMA_SPREAD && PF2 && !NODMA), "the wide spread is built on the product's 16x16x32 stage");
    constexpr bool CLOCKS = (VAR & 33554432) != 0;
    constexpr bool PACE = (VAR & 67108864) != 0;
    static_assert(!PACE || (EARLY_THR && EXCH == 1 && !PERSIST), "the pace poll rides on the early threshold load of every tile");
    static_assert(!(CLOCKS && STAMPS), "one use of the debug buffer at a time");
    constexpr int S = cf_ring_stages(VAR);            // ring slots
    constexpr int VM_MID = NOVM ? 63 : (PAIRBAR ? 4 : 4 * (S - 3));   // LDS-DMA pieces that may stay in flight at the mid-stage wait
    constexpr int PRO = PAIRBAR ? 4 : S - 1;          // stages issued by a list's prologue
    constexpr int VM_TILE_END = NOVM ? 0 : (PAIRBAR ? (DMA_WIDE ? 6 : 8) : (DMA_WIDE ? 4 * (S - 2) - 2 : 4 * (S - 2)));   // pieces younger than the early threshold load at the tile end
    constexpr int KS = D / CO_BK;      // stages per tile
    constexpr int NF = D / 16;         // query fragments per lane
    static_assert(KS % S == 0, "ring slot must be a compile-time function of the stage");
    using Ops = Sel2Ops<KP>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int q16 = lane & 15, g16 = lane >> 4;   // (X16: query of the group / row of the 16-row group, and k-octet / row quartet)
    // Which range does this work-group take? Work-groups l, l + T, l + 2T, ... (T = pos_period) sweep the same corpus
    // tiles at the same time (for other query tiles). The dispatcher deals consecutive blockIdx round-robin over the
    // 8 XCDs, each with its own L2: remap so that an XCD gets a contiguous stretch of the CLASS-MAJOR order
    // (class 0's members, class 1's, ...) - then the ~G/T work-groups that stream the same tiles share one L2 and
    // the corpus is fetched from the Infinity Cache once per class instead of once per work-group (measured:
    // FETCH_SIZE 4.3 GB -> see profiles/). Pure placement: any bijection is correct.
    const int nq_act = PERSIST && a.nq_ptr ? min(__builtin_amdgcn_readfirstlane(*a.nq_ptr), a.nq) : a.nq;
    const int total_units = PERSIST ? ((nq_act + CO_BM - 1) / CO_BM) * a.ctiles : a.total_units;
    // PERSIST: block b takes the logical work-groups b, b + grid, ... of the nwg_logical that the active slots need - the
    // placement bijection runs over THAT count (over the full batch's count the few work-groups of a small flagged set
    // all map to two blocks: 0.44 ms for 15 queries, measured)
    const int nwg_logical = PERSIST ? (total_units + a.units_per_wg - 1) / a.units_per_wg : (int)gridDim.x;
    if (PERSIST && nq_act <= a.skip_below) return;
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if constexpr (CLOCKS) {
        clk_c0 = __builtin_amdgcn_s_memtime();
        clk_r0 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);   // (lgkmcnt(0) alone: see cdna_hip_programming.md section 7, In-kernel stamps)
    }

    // LDS-DMA: per-lane source offsets (bytes from the tile's first row, k = 0); piece i of this wave =
    // rows 8 (4 wave + i) .. +7, one full 128-B line each, 16-B pieces XOR-swizzled on the source side
    uint32_t src_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row_local = (wave * 4 + i) * 8 + (lane >> 3);
        const int piece = (lane & 7) ^ ((row_local >> 1) & 7);
        src_off[i] = (uint32_t)row_local * (uint32_t)(D * 2) + (uint32_t)piece * 16u - (uint32_t)(i * 1024);   // (piece i is issued with instruction offset 1024 i, see issue_stage)
    }
    uint32_t rd_off[4];
    if constexpr (X16) {   // A fragment of a 16-row group, k-step k2 of the stage: row q16, 16-B piece 4 k2 + g16 of its line
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) rd_off[k2] = (uint32_t)q16 * 128u + (uint32_t)(((4 * k2 + g16) ^ ((q16 >> 1) & 7)) * 16);
        rd_off[2] = rd_off[3] = 0u;
    } else {
        const int sw = (c >> 1) & 7;
#pragma unroll
        for (int s = 0; s < 4; ++s) rd_off[s] = (uint32_t)c * 128u + (uint32_t)(((2 * s + h) ^ sw) * 16);
    }
    auto read_frags = [&](half8 (&f)[4], int ring_slot, int s) {
        if constexpr (X16) {   // quad s of the stage: k-step s >> 1, row groups 4 (s & 1) .. + 3
            const char *sb = smem + ring_slot * CO_STAGE_BYTES + (s & 1) * 8192 + rd_off[s >> 1];
#pragma unroll
            for (int t = 0; t < 4; ++t) f[t] = *reinterpret_cast<const half8 *>(sb + t * 2048);
        } else {
            const char *sb = smem + ring_slot * CO_STAGE_BYTES + rd_off[s];
#pragma unroll
            for (int t = 0; t < 4; ++t) f[t] = *reinterpret_cast<const half8 *>(sb + t * 4096);
        }
    };
    constexpr uint32_t RING_BYTES = (uint32_t)S * CO_STAGE_BYTES;
    const uint32_t wave_qbase = RING_BYTES + (uint32_t)(wave * 32) * Ops::QBYTES;
    const uint32_t wave_scratch = RING_BYTES + (uint32_t)CO_BM * Ops::QBYTES + (uint32_t)wave * 256u;
    unsigned long long st_vm = 0, st_bar = 0, st_body = 0, st_sel = 0, st_tiles = 0, st_prev = 0;   // (STAMPS)
    unsigned long long st_comp[2] = {0, 0}, st_thr = 0, st_boot = 0;                              // compactions, their cycles
    const int last_tile = a.ctiles - 1;

    half8 qf[NF];
    int cur_mtile = -1;
    for (int vblock = (int)blockIdx.x; vblock < (PERSIST ? nwg_logical : (int)blockIdx.x + 1); vblock += PERSIST ? (int)gridDim.x : 1) {
    const int wg = flat_workgroup_of_block(vblock, nwg_logical, a.pos_period);
    const int u_begin = wg * a.units_per_wg;
    const int u_end = min(total_units, u_begin + a.units_per_wg);
    if (u_begin >= u_end) continue;
    int u = u_begin;
    // pacing state (wave-uniform): class, members that take part in an epoch, whether this work-group still paces
    [[maybe_unused]] bool pacing = false;
    [[maybe_unused]] int pace_cls = 0, pace_members = 0, pace_last_epochs = 0;
    [[maybe_unused]] bool pace_last_mine = false;
    if constexpr (PACE) {
        if (a.pace && a.pace_period > 0) {
            pacing = true;
            pace_cls = wg % a.pace_period;
            pace_members = (nwg_logical - pace_cls + a.pace_period - 1) / a.pace_period;
            // the last work-group holds fewer units: it takes part in fewer epochs
            const int last_units = total_units - (nwg_logical - 1) * a.units_per_wg;
            pace_last_mine = ((nwg_logical - 1) % a.pace_period) == pace_cls;
            pace_last_epochs = last_units >> a.pace_shift;
        }
    }
    while (u < u_end) {
        // ---- the list [t0, t1) of query tile mtile, and its ordinal ---------------------------------------
        const int mtile = u / a.ctiles;
        const int t0 = u - mtile * a.ctiles;
        const int run0 = max(u_begin - mtile * a.ctiles, 0);                 // this work-group's run in the query tile
        const int run1 = min(u_end - mtile * a.ctiles, a.ctiles);
        const int j = (t0 - run0) / a.list_tiles;
        const int t1 = min(run1, run0 + (j + 1) * a.list_tiles);
        const int ntiles = t1 - t0;
        const int ord = flat_first_ordinal(mtile, wg, a.ctiles, a.units_per_wg, a.list_tiles) + j;
        const int slot0 = mtile * CO_BM;

        if (mtile != cur_mtile) {   // query fragments -> registers (B operand: lane holds Q[query c][16 s + 8 h + j])
            if constexpr (X16) {   // group gr's fragments at qf[gr NF/2 ..): lane holds Q[query 16 gr + q16][32 s + 8 g16 + 0..7]
#pragma unroll
                for (int gr = 0; gr < 2; ++gr) {
                    int qr = slot0 + wave * 32 + 16 * gr + q16;
                    if (PERSIST && a.qlist) qr = a.qlist[min(qr, nq_act - 1)];   // (slots past the list sit at +inf: any row will do)
                    const _Float16 *qrow = a.q16 + (size_t)qr * D + 8 * g16;
#pragma unroll
                    for (int s = 0; s < NF / 2; ++s) qf[gr * (NF / 2) + s] = *reinterpret_cast<const half8 *>(qrow + 32 * s);
                }
            } else {
                int qr = slot0 + wave * 32 + c;
                if (PERSIST && a.qlist) qr = a.qlist[min(qr, nq_act - 1)];
                const _Float16 *qrow = a.q16 + (size_t)qr * D + 8 * h;
#pragma unroll
                for (int s = 0; s < NF; ++s) qf[s] = *reinterpret_cast<const half8 *>(qrow + 16 * s);
            }
            if constexpr (Q_AGPR) {
#pragma unroll
                for (int s = 0; s < NF; ++s) asm volatile("" : "+a"(qf[s]));
            }
            cur_mtile = mtile;
        }
        // buffer_load ... lds: per-lane part in voffset, tile/stage part in a scalar soffset, base = the list's first row
        const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<_Float16 *>(a.c16) + (size_t)t0 * CO_BN * D, 0,
            (int)min((size_t)(a.ctiles - t0) * CO_BN * (size_t)(D * 2), (size_t)0x7FFFFFFF), 0x00020000);
        auto issue_pieces = [&](int g_tile, int g_ks, int ring_slot, auto LO, auto HI) {   // pieces [LO, HI) of a stage
            const int trow = min(g_tile, last_tile - t0);   // stages past the sweep re-read valid memory, never consumed
            char *dst = smem + ring_slot * CO_STAGE_BYTES + wave * 4096;
            const uint32_t soff = (uint32_t)trow * (uint32_t)(CO_BN * D * 2) + (uint32_t)g_ks * (CO_BK * 2);
            // The instruction offset is added to the LDS address AND the buffer address: piece i lands at dst + 1024 i
            // with ONE M0 for the four pieces; src_off[i] was reduced by 1024 i to compensate on the buffer side.
            __attribute__((address_space(3))) void *ldst = (__attribute__((address_space(3))) void *)dst;
            constexpr int lo = decltype(LO)::value, hi = decltype(HI)::value;
            if constexpr (lo <= 0 && 0 < hi) __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, ldst, 16, src_off[0], soff, 0, 0);
            if constexpr (lo <= 1 && 1 < hi) __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, ldst, 16, src_off[1], soff, 1024, 0);
            if constexpr (lo <= 2 && 2 < hi) __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, ldst, 16, src_off[2], soff, 2048, 0);
            if constexpr (lo <= 3 && 3 < hi) __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, ldst, 16, src_off[3], soff, 3072, 0);
        };
        auto issue_stage = [&](int g_tile, int g_ks, int ring_slot) {
            issue_pieces(g_tile, g_ks, ring_slot, std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
        };

        Sel2 st;
        Ops::init(st, wave_qbase + (uint32_t)c * Ops::QBYTES, h, (slot0 + wave * 32 + c) < nq_act && !NOPASS);
        float boot1 = -INFINITY, boot2 = -INFINITY, boot3 = -INFINITY;   // bootstrap: the lane's three best scores so far
        // (a third of the list at most: 6 rows above the level per boot_tiles tiles -> >= 18 in the whole list)
        const int boot_tiles = ntiles >= CO_BOOT_MIN_TILES ? min(a.boot_tiles, ntiles / 3) : 0;
        unsigned int *my_shared = a.shared_thr + (slot0 + wave * 32 + c);
        const bool publish = (slot0 + wave * 32 + c) < nq_act;   // (padding queries sit at +inf and never publish)
        uint32_t published = 0u;
        int dirty = 0;   // (DIRTY_GUARD; wave-uniform) some lane has appended since the last overflow check
        auto filter_reg = [&](const f32x16 (&pa)[4], auto F, uint32_t rowbase, auto GUARD) {
            constexpr int f = decltype(F)::value;
            constexpr int t = f >> 4, r = f & 15;
            constexpr uint32_t roff = (uint32_t)(t * 32 + (r & 3) + 8 * (r >> 2));
            float v = pa[t][r];
            if constexpr (decltype(GUARD)::value) {
                if ((int)(rowbase + roff) >= a.n) v = -INFINITY;
            }
            // wave-uniform skip first (v_cmp + one scalar branch when no lane passes), the per-lane append behind it
            const bool pass = v > st.thr;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(pass) != 0ull, 0)) {   // unlikely: laid out of line, the common path falls through
                asm volatile("" ::: "memory");   // keeps the scalar branch: without it the two conditions merge into a predicate
                if (pass) {
                    *reinterpret_cast<float *>(smem + st.aw) = v;
                    *reinterpret_cast<uint32_t *>(smem + st.aw + Ops::ROW_OFF) = rowbase + roff;
                    st.aw += st.inc;
                }
            }
            if constexpr (r % CO_CHECK_EVERY == CO_CHECK_EVERY - 1) {
                // overflow guard, cheap form: each lane of a query may append CO_QUOTA entries on its side of the buffer
                // between compactions (kept 16 + 2 x (16 + the 8 of one check interval) = 64 slots); only when some lane
                // is past its quota does the wave run the full check (partner counts, compaction)
                if (__builtin_amdgcn_ballot_w64(Ops::used(st, h) > CO_QUOTA) != 0ull)
                    Ops::template check<PAIRBAR>(st, lane, smem, wave_qbase, wave_scratch, false, CO_LIMIT, STAMPS ? st_comp : nullptr, CO_QUOTA);
            }
        };

        auto filter_quad = [&](const f32x16 (&pa)[X16 ? 1 : 4], const f32x4 (&px)[X16 ? 16 : 1], auto Q, uint32_t rowbase, auto GUARD) {
            constexpr int q = decltype(Q)::value;
            constexpr int t = q >> 2, g = q & 3;
            // registers 4 g .. 4 g + 3 of row tile t: rows qoff + 0..3 (+ 4 h); X16: quad q = row group q >> 1, half q & 1 (+ 8 h)
            constexpr uint32_t qoff = X16 ? (uint32_t)(16 * (q >> 1) + 4 * (q & 1)) : (uint32_t)(t * 32 + 8 * g);
            float v0 = X16 ? px[X16 ? q : 0][0] : pa[X16 ? 0 : t][4 * g + 0], v1 = X16 ? px[X16 ? q : 0][1] : pa[X16 ? 0 : t][4 * g + 1];
            float v2 = X16 ? px[X16 ? q : 0][2] : pa[X16 ? 0 : t][4 * g + 2], v3 = X16 ? px[X16 ? q : 0][3] : pa[X16 ? 0 : t][4 * g + 3];
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
                // EXEC is all ones here (256-thread blocks, wave-uniform control flow down to this point); the four appends
                // run under the four masks and EXEC is restored
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
                if constexpr (DIRTY_GUARD) dirty = 1;
            }
            if constexpr (q % 2 == 1) {   // every 8 registers: the overflow guard of filter_reg
                if constexpr (DIRTY_GUARD) {
                    if (dirty) {
                        dirty = 0;
                        if (__builtin_amdgcn_ballot_w64(Ops::used(st, h) > CO_QUOTA) != 0ull)
                            Ops::template check<PAIRBAR>(st, lane, smem, wave_qbase, wave_scratch, false, CO_LIMIT, STAMPS ? st_comp : nullptr, CO_QUOTA);
                    }
                } else
                if (__builtin_amdgcn_ballot_w64(Ops::used(st, h) > CO_QUOTA) != 0ull)
                    Ops::template check<PAIRBAR>(st, lane, smem, wave_qbase, wave_scratch, false, CO_LIMIT, STAMPS ? st_comp : nullptr, CO_QUOTA);
            }
        };
        static_assert(!QUAD || Ops::ROW_OFF == 256, "ds_write2st64_b32 offset1:1 = the row array of the query's buffer");
        // SPARSE_PRE: quads 2 G and 2 G + 1 (registers xs[2 G], xs[2 G + 1]) behind one test of their lane maximum. Rows past
        // the corpus are only masked inside filter_quad: a zero pad row above the threshold costs the detour, nothing else.
        auto filter_group = [&](const f32x16 (&pa)[X16 ? 1 : 4], const f32x4 (&px)[X16 ? 16 : 1], auto G, uint32_t rowbase, auto GUARD) {
            constexpr int g = decltype(G)::value;
            const f32x4 &x0 = px[X16 ? 2 * g : 0], &x1 = px[X16 ? 2 * g + 1 : 0];
            float m = raw_max3_f32(x0[0], x0[1], x0[2]);
            m = raw_max3_f32(m, x0[3], x1[0]);
            m = raw_max3_f32(m, x1[1], x1[2]);
            m = raw_max_f32(m, x1[3]);
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(m > st.thr) != 0ull, 0)) {
                filter_quad(pa, px, std::integral_constant<int, 2 * g>{}, rowbase, GUARD);
                filter_quad(pa, px, std::integral_constant<int, 2 * g + 1>{}, rowbase, GUARD);
            }
        };

        // prologue: stages 0..S-2 in flight, stage 0 published, its first fragments read
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (query fragment loads: the vmcnt accounting starts from zero)
#pragma unroll
        for (int p = 0; p < PRO - (DMA_WIDE ? 1 : 0); ++p) issue_stage(p / KS, p % KS, p % S);
        // (DMA_WIDE: the last prologue stage's pieces 2, 3 go out in the first stage's front half, like every later stage's)
        if constexpr (DMA_WIDE) issue_pieces((PRO - 1) / KS, (PRO - 1) % KS, (PRO - 1) % S, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
        half8 afn[4], bfn[4];
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"i"(4 * (PRO - 1) - (DMA_WIDE ? 2 : 0)) : "memory");
        read_frags(afn, 0, 0);
        if constexpr (PF2) read_frags(bfn, 0, 1);
        if constexpr (STAMPS) ICD_CF_STAMP(st_prev);

        for (int tile = 0; tile < ntiles; ++tile) {
            uint32_t seen_early = 0u;
            [[maybe_unused]] uint32_t pace_seen = 0u;
            [[maybe_unused]] bool pace_due = false;
            [[maybe_unused]] int pace_need = -1, pace_done_epoch = 0;
            if constexpr (PACE) {
                if (pacing) {
                    const int done_after = (u - u_begin) + tile + 1;   // units this work-group will have finished at the end of this tile
                    pace_due = (done_after & ((1 << a.pace_shift) - 1)) == 0;
                    pace_done_epoch = (done_after >> a.pace_shift) - 1;   // the epoch that ends with this tile
                    pace_need = pace_done_epoch + 1 - a.pace_lead;        // ... and the one the next epoch waits for
                }
            }
            f32x16 acc[X16 ? 1 : 4];
            f32x4 xs[X16 ? 16 : 1];   // X16: accumulator of row group rg and query group gr at xs[2 rg + gr]
            if constexpr (!X16) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[X16 ? 0 : t][r] = 0.0f;
            }
#pragma unroll
            for (int t = 0; t < (X16 ? 16 : 1); ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) xs[t][r] = 0.0f;
            static_for<0, KS>([&](auto KSI) {
                constexpr int ks = decltype(KSI)::value;
                constexpr int slot = ks % S, nslot = (ks + 1) % S;
                auto mfma4 = [&](const half8 (&f)[4], int qi, int t_lo = 0, int t_hi = 4) {
                    // (REV_WAIT: the four MFMAs of a k-step in the reverse of the order their fragments were read - the first one
                    //  waits for the youngest read, the other three need no s_waitcnt at all)
                    if constexpr (X16) {   // qi = 4 ks + quad: k-step 2 ks + (quad >> 1), row groups 4 (quad & 1) + t, both query groups
                        const int quad = qi & 3, kq = (qi >> 2) * 2 + (quad >> 1);
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            if (t < t_lo || t >= t_hi) continue;
                            const int rg = 4 * (quad & 1) + t;
#pragma unroll
                            for (int gr = 0; gr < 2; ++gr)
                                xs[2 * rg + gr] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[t], qf[gr * (NF / 2) + kq], xs[2 * rg + gr], 0, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) {
                            const int t = REV_WAIT ? 3 - tt : tt;
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[t], qf[qi], acc[t], 0, 0, 0);
                        }
                    }
                };
                constexpr int M4 = X16 ? 8 : 4;   // MFMAs per k-step group
                // stage g = (tile, ks) was published by the previous mid-stage barrier; afn holds its k-step 0 (PF2: bfn its k-step 1)
                half8 f1[4], f2[4], f3[4];
                if constexpr (DMA_WIDE) {   // pieces 2, 3 of the stage whose pieces 0, 1 went out behind the previous barrier
                    constexpr int pks = PAIRBAR ? ks + 3 : ks + S - 2;   // (PAIRBAR: the eight pieces of two stages over two stage intervals)
                    read_frags(f2, slot, 2);
                    mfma4(afn, ks * 4 + 0, 0, 2);
                    __builtin_amdgcn_sched_barrier(0);
                    issue_pieces(tile + pks / KS, pks % KS, pks % S, std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{});
                    __builtin_amdgcn_sched_barrier(0);
                    mfma4(afn, ks * 4 + 0, 2, 4);
                    read_frags(f3, slot, 3);
                    mfma4(bfn, ks * 4 + 1, 0, 2);
                    __builtin_amdgcn_sched_barrier(0);
                    issue_pieces(tile + pks / KS, pks % KS, pks % S, std::integral_constant<int, 3>{}, std::integral_constant<int, 4>{});
                    __builtin_amdgcn_sched_barrier(0);
                    mfma4(bfn, ks * 4 + 1, 2, 4);
                } else if constexpr (PF2) {
                    read_frags(f2, slot, 2);
                    mfma4(afn, ks * 4 + 0);
                    read_frags(f3, slot, 3);
                    mfma4(bfn, ks * 4 + 1);
                } else {
                    read_frags(f1, slot, 1);
                    mfma4(afn, ks * 4 + 0);
                    read_frags(f2, slot, 2);
                    mfma4(f1, ks * 4 + 1);
                }
                // pin: the reads go out before the MFMAs of the k-step in front of them
                if constexpr (DMA_WIDE) {
                } else {
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, M4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, M4, 0);
                }
                // publish stage g+1: this wave's pieces of g+1 have landed when only the stages behind it are outstanding
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (STAMPS) {
                    unsigned long long ta, tb, tc;
                    ICD_CF_STAMP(ta);
                    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(VM_MID) : "memory");
                    ICD_CF_STAMP(tb);
                    if constexpr (!NOBAR) asm volatile("s_barrier" ::: "memory");
                    ICD_CF_STAMP(tc);
                    st_body += ta - st_prev; st_vm += tb - ta; st_bar += tc - tb; st_prev = tc;
                } else if constexpr (NOBAR) {
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(VM_MID) : "memory");
                } else if constexpr (PAIRBAR && (ks & 1) == 1) {
                    // (odd stage: the next stage was published together with this one, nothing is issued, nobody waits)
                } else {
                    if constexpr (SETPRIO) __builtin_amdgcn_s_setprio(0);
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"i"(VM_MID) : "memory");
                    if constexpr (SETPRIO) __builtin_amdgcn_s_setprio(1);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (EARLY_THR && ks == KS - 2) {
                    // the query's shared threshold for the end of this tile: issued here, older than this stage's and the
                    // next stage's LDS-DMA pieces, so a counted wait at the tile end covers it without draining them
                    if (EXCH == 1 || (tile & (EXCH - 1)) == EXCH - 1) {
                        asm volatile("global_load_dword %0, %1, off sc1" : "=v"(seen_early) : "v"(my_shared) : "memory");
                    }
                    if constexpr (PACE) {
                        // the epoch whose completion this work-group needs before it starts its next epoch: polled HERE, next to the
                        // threshold load (the same counted wait at the tile end covers both)
                        if (pacing && pace_due && pace_need >= 0 && wave == 0) {
                            const unsigned int *pp = a.pace + (size_t)pace_cls * a.pace_epochs + pace_need;
                            asm volatile("global_load_dword %0, %1, off sc1" : "=v"(pace_seen) : "v"(pp) : "memory");
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (!NODMA && !PAIRBAR) {   // every wave is past stage g-1: its slot takes stage g+S-1
                    constexpr int nks = ks + S - 1;
                    if constexpr (!DMA_WIDE) issue_stage(tile + nks / KS, nks % KS, nks % S);
                }
                if constexpr (!NODMA && PAIRBAR && !DMA_WIDE && (ks & 1) == 0) {   // every wave is past stages g-2, g-1: their slots take g+4, g+5
                    constexpr int n4 = ks + 4, n5 = ks + 5;
                    issue_stage(tile + n4 / KS, n4 % KS, n4 % S);
                    issue_stage(tile + n5 / KS, n5 % KS, n5 % S);
                }
                if constexpr (DMA_WIDE) {
                    constexpr int nks = PAIRBAR ? ks + 4 : ks + S - 1;
                    read_frags(afn, nslot, 0);
                    mfma4(f2, ks * 4 + 2, 0, 2);
                    __builtin_amdgcn_sched_barrier(0);
                    issue_pieces(tile + nks / KS, nks % KS, nks % S, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
                    __builtin_amdgcn_sched_barrier(0);
                    mfma4(f2, ks * 4 + 2, 2, 4);
                    read_frags(bfn, nslot, 1);
                    mfma4(f3, ks * 4 + 3, 0, 2);
                    __builtin_amdgcn_sched_barrier(0);
                    issue_pieces(tile + nks / KS, nks % KS, nks % S, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
                    __builtin_amdgcn_sched_barrier(0);
                    mfma4(f3, ks * 4 + 3, 2, 4);
                } else if constexpr (PF2) {
                    read_frags(afn, nslot, 0);
                    mfma4(f2, ks * 4 + 2);
                    read_frags(bfn, nslot, 1);
                    mfma4(f3, ks * 4 + 3);
                } else {
                    read_frags(f3, slot, 3);
                    mfma4(f2, ks * 4 + 2);
                    read_frags(afn, nslot, 0);
                    mfma4(f3, ks * 4 + 3);
                }
                if constexpr (!DMA_SPREAD) {          // the four pieces in a burst behind the barrier
                    __builtin_amdgcn_sched_group_barrier(0x020, 4, 1);
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 1);
                    __builtin_amdgcn_sched_group_barrier(0x008, M4, 1);
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 1);
                    __builtin_amdgcn_sched_group_barrier(0x008, M4, 1);
                } else if constexpr (DMA_WIDE) {      // (one piece in the middle of each k-step group: pinned by source order above)
                } else {                               // one piece behind each of the next four MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 1);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, M4 / 4, 1);
                        if constexpr (!PAIRBAR) __builtin_amdgcn_sched_group_barrier(0x020, 1, 1);
                        else if constexpr ((ks & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x020, 2, 1);   // (two stages' pieces)
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 1);
                    __builtin_amdgcn_sched_group_barrier(0x008, M4, 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (X16 && (VAR & 65536) == 0) {   // (VAR & 65536: TIMING ONLY, without the lane swaps)
                // Lane (q16, g16) holds, per row group rg, rows 4 g16 + 0..3 of query q16 (xs[2 rg]) and of query 16 + q16
                // (xs[2 rg + 1]). Swapping xs[2 rg]'s odd 16-lane rows with xs[2 rg + 1]'s even ones gives every lane 8
                // consecutive rows of ONE query: lane l -> query l & 31, xs[2 rg] = rows 16 rg + 8 (l >> 5) + 0..3,
                // xs[2 rg + 1] = rows + 4..7: the lane pair (l, l + 32) of the two-lane select.
#pragma unroll
                for (int rg = 0; rg < 8; ++rg)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(xs[2 * rg][i]), __float_as_uint(xs[2 * rg + 1][i]), false, false);
                        xs[2 * rg][i] = __uint_as_float(sw[0]);
                        xs[2 * rg + 1][i] = __uint_as_float(sw[1]);
                    }
            }
            unsigned long long ts0 = 0;
            if constexpr (STAMPS) ICD_CF_STAMP(ts0);
            // Threshold sharing between the lists of a query (they are swept by different work-groups at the same
            // time): adopt the largest threshold any of them has published, publish this list's when it is larger.
            // A list still reports the threshold it ends on as its bound, and the largest bound over the lists - what
            // finalize certifies against - is the largest of the lists' OWN k'-th best scores with or without sharing;
            // the weaker lists just stop collecting rows that could never matter. Stale reads are harmless.
            if (EXCH == 1 || (tile & (EXCH - 1)) == EXCH - 1) {
                uint32_t seen;
                if constexpr (EARLY_THR) {
                    // (every destination of the asm loads this wait covers is named "+v": cdna_hip_programming.md 5.7 item 1, form (ii))
                    if constexpr (PACE) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(seen_early), "+v"(pace_seen) : "i"(VM_TILE_END) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(seen_early) : "i"(VM_TILE_END) : "memory");
                    seen = seen_early;
                } else {
                    seen = __hip_atomic_load(my_shared, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if constexpr (PACE) {
                    if (pacing && pace_due && wave == 0) {   // (the wait above has covered the pace poll too)
                        if (pace_done_epoch < a.pace_epochs && lane == 0)
                            __hip_atomic_fetch_add(a.pace + (size_t)pace_cls * a.pace_epochs + pace_done_epoch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (pace_need >= 0 && pace_need < a.pace_epochs) {
                            const uint32_t want = (uint32_t)(pace_members - ((pace_last_mine && pace_need >= pace_last_epochs) ? 1 : 0));
                            uint32_t got = (uint32_t)__builtin_amdgcn_readfirstlane((int)pace_seen);
                            int spins = 0;
                            while (got < want) {   // this work-group is ahead of its class: wait for the slowest member, BOUNDED
                                if (++spins > PACE_SPIN_LIMIT) { pacing = false; break; }
                                __builtin_amdgcn_s_sleep(16);
                                got = __hip_atomic_load(a.pace + (size_t)pace_cls * a.pace_epochs + pace_need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
                    }
                }
                const uint32_t mine_key = order_f32(st.thr);
                if (seen > mine_key) st.thr = unorder_f32(seen);
                else if (h == 0 && publish && mine_key > seen && mine_key > published) {
                    __hip_atomic_fetch_max(my_shared, mine_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    published = mine_key;
                }
            }
            unsigned long long ts_a = 0;
            if constexpr (STAMPS) { ICD_CF_STAMP(ts_a); st_thr += ts_a - ts0; }
            // fused select of the finished tile (rows >= n exist only in the corpus's last tile)
            const int tile_row0 = (t0 + tile) * CO_BN;
            const uint32_t rowbase = (uint32_t)(tile_row0 + (X16 ? 8 : 4) * h);
            if (tile < boot_tiles && tile_row0 + CO_BN <= a.n) {
                // Threshold bootstrap. A list that starts at -inf appends all 128 rows of its first tile and the
                // next few hundred, and compacts 5-8 times per query before its threshold means anything (~60 us
                // per list and wave). Any threshold is VALID - the list reports the threshold it ends on as the
                // bound on what it dropped, and finalize certifies against the bounds - so over the list's first
                // CO_BOOT_TILES tiles every lane tracks the three best scores it has seen (5 VALU per score) and the
                // threshold follows the smaller of the two lanes' third best: about the 6th best of the rows
                // seen so far. In a list of >= CO_BOOT_MIN_TILES tiles more than KP rows beat the level this
                // reaches (0.6 % of the rows), so the list still fills and ends on its own KP-th best. Third best,
                // not second: with the second best a list whose first rows happen to hold four of the query's
                // global top hits ends on a bound inside the top-k and fails the certificate (measured 4 of 10 000).
                if constexpr (BOOT_MED3) {
                    // chain A = (boot1, boot2, boot3) takes the even registers, chain B the odd ones; every step is three
                    // independent instructions on the chain's previous state
                    float c1 = -INFINITY, c2 = -INFINITY, c3 = -INFINITY;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int r = 0; r < 16; r += 2) {
                            const float va = X16 ? xs[X16 ? 4 * t + (r >> 2) : 0][r & 3] : acc[X16 ? 0 : t][r];
                            const float vb = X16 ? xs[X16 ? 4 * t + ((r + 1) >> 2) : 0][(r + 1) & 3] : acc[X16 ? 0 : t][r + 1];
                            const float a3 = __builtin_amdgcn_fmed3f(boot2, boot3, va), a2 = __builtin_amdgcn_fmed3f(boot1, boot2, va);
                            boot1 = raw_max_f32(boot1, va); boot2 = a2; boot3 = a3;
                            const float b3 = __builtin_amdgcn_fmed3f(c2, c3, vb), b2 = __builtin_amdgcn_fmed3f(c1, c2, vb);
                            c1 = raw_max_f32(c1, vb); c2 = b2; c3 = b3;
                        }
                    // merge: the three largest of the six (insert chain B's three into chain A)
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const float v = i == 0 ? c1 : (i == 1 ? c2 : c3);
                        const float a3 = __builtin_amdgcn_fmed3f(boot2, boot3, v), a2 = __builtin_amdgcn_fmed3f(boot1, boot2, v);
                        boot1 = raw_max_f32(boot1, v); boot2 = a2; boot3 = a3;
                    }
                } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = X16 ? xs[X16 ? 4 * t + (r >> 2) : 0][r & 3] : acc[X16 ? 0 : t][r];
                        const float lo1 = raw_min_f32(boot1, v);
                        boot1 = raw_max_f32(boot1, v);
                        const float lo2 = raw_min_f32(boot2, lo1);
                        boot2 = raw_max_f32(boot2, lo1);
                        boot3 = raw_max_f32(boot3, lo2);
                    }
                }
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(boot3), __float_as_uint(boot3), false, false);
                const float thr0 = fminf(boot3, __uint_as_float(h ? sw[0] : sw[1]));
                if (thr0 > st.thr) st.thr = thr0;   // (padding queries keep +inf)
            }
            if constexpr (STAMPS) { unsigned long long ts_b; ICD_CF_STAMP(ts_b); st_boot += ts_b - ts_a; }
            if constexpr (NOSEL) {   // the scores stay live (cdna_hip_programming.md rule 17), nothing else happens to them
                if constexpr (X16) {
#pragma unroll
                    for (int t = 0; t < 16; ++t) asm volatile("" ::"v"(xs[X16 ? t : 0]));
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) asm volatile("" ::"v"(acc[X16 ? 0 : t]));
                }
            } else if constexpr (QUAD) {
                if (tile_row0 + CO_BN > a.n) static_for<0, 16>([&](auto Q) { filter_quad(acc, xs, Q, rowbase, std::true_type{}); });
                else if (SPARSE_PRE && tile >= a.sparse_from) static_for<0, 8>([&](auto G) { filter_group(acc, xs, G, rowbase, std::false_type{}); });
                else static_for<0, 16>([&](auto Q) { filter_quad(acc, xs, Q, rowbase, std::false_type{}); });
            } else if (tile_row0 + CO_BN > a.n) static_for<0, 64>([&](auto F) { filter_reg(acc, F, rowbase, std::true_type{}); });
            else static_for<0, 64>([&](auto F) { filter_reg(acc, F, rowbase, std::false_type{}); });
            if constexpr (EPOCH > 0) {
                // (nothing to do where no lane has appended more than a few entries since the last compaction: the long lists
                //  of a row-sharded corpus, 4 900 tiles, append a handful of rows per hundred tiles once they are warm)
                if ((tile + 1) % EPOCH == 0 && tile + 1 < ntiles && __builtin_amdgcn_ballot_w64(Ops::used(st, h) > 3) != 0ull)
                    compact_all_parallel<KP, Ops>(smem, st, wave_qbase + (uint32_t)c * Ops::QBYTES, h, c);
            }
            if constexpr (STAMPS) {
                unsigned long long ts1;
                ICD_CF_STAMP(ts1);
                st_sel += ts1 - ts0; st_prev += ts1 - ts0; st_tiles += 1;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the run-ahead stages
        asm volatile("" ::"v"(afn[0]), "v"(afn[1]), "v"(afn[2]), "v"(afn[3]));
        if constexpr (PF2) asm volatile("" ::"v"(bfn[0]), "v"(bfn[1]), "v"(bfn[2]), "v"(bfn[3]));

        // ---- end of the list: every query's top-KP entries (unsorted) and its bound go to global memory ----------
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
    }   // (logical work-groups of this block)
    if constexpr (CLOCKS) {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && a.dbg && blockIdx.x < 2048) {   // (a buffer of its own: nothing in the kernel reads it)
            unsigned long long *d = a.dbg + ((size_t)blockIdx.x * 4 + wave) * 8;
            d[0] = c1 - clk_c0; d[1] = r1 - clk_r0; d[4] = 1;
        }
    }
    if constexpr (STAMPS) {
        if (lane == 0 && a.dbg) {
            unsigned long long *d = a.dbg + ((size_t)blockIdx.x * 4 + wave) * 8;
            d[0] = st_vm; d[1] = st_bar; d[2] = st_body; d[3] = st_sel; d[4] = st_tiles; d[5] = st_comp[0]; d[6] = st_comp[1];
            d[7] = (st_thr << 32) | (st_boot & 0xffffffffull);
        }
    }
}

}  // namespace icd
