// topk_select.hpp — lane-local threshold filter + LDS candidate buffers + wave-cooperative compaction.
//
// Shared by the exact (fp32 MFMA) and coarse (fp16 MFMA) kernels. Both compute 32x32 score tiles with
// "corpus rows = M, queries = N", so after an MFMA tile lane l holds 16 scores of ONE query
// (query column l&31) for rows (r&3) + 8*(r>>2) + 4*(l>>5), r = register index (C/D layout of every
// 32x32 MFMA on gfx950). Lanes l and l+32 own the same query and share its candidate buffer.
//
// Replaces the k-selection inside the Milvus FLAT search called at
// services/milvus_service.py:280-285 (reference has no code of its own for it).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace icd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

// ---- 64-bit sortable key: larger key = better hit (score desc, row asc) -----------------------------
__device__ __forceinline__ uint32_t order_f32(float v) {
    uint32_t u = __float_as_uint(v);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float unorder_f32(uint32_t o) {
    uint32_t u = o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu);
    return __uint_as_float(u);
}
__device__ __forceinline__ u64 make_key(float score, uint32_t row) {
    return ((u64)order_f32(score) << 32) | (u64)(~row);
}
__device__ __forceinline__ float key_score(u64 k) { return unorder_f32((uint32_t)(k >> 32)); }
__device__ __forceinline__ uint32_t key_row(u64 k) { return ~(uint32_t)k; }
// key 0 is below every real key (a real key has row <= 0x7FFFFFFE so its low word is >= 0x80000001)

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

template <typename T>
__device__ __forceinline__ T readlane(T v, int l);
template <>
__device__ __forceinline__ int readlane<int>(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
template <>
__device__ __forceinline__ uint32_t readlane<uint32_t>(uint32_t v, int l) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}
__device__ __forceinline__ u64 readlane_u64(u64 v, int l) {
    uint32_t lo = readlane<uint32_t>((uint32_t)v, l), hi = readlane<uint32_t>((uint32_t)(v >> 32), l);
    return ((u64)hi << 32) | lo;
}

// Rank the wave's candidates for a top-T selection. key[e] = candidate lane + 64 e (0 = empty), also staged
// at keys_lds[0..ncand). On return rank[e] is EXACT for every key whose true rank is below T, and >= T for
// every other key (a key of true rank T keeps rank == T or ties with others at the survivor count).
// Small sets: plain rank-by-counting over all candidates. Large sets with T <= 64: first bound the T-th
// best from below by the T-th largest of the 64 lane maxima (T distinct keys are >= it), keep only the
// keys >= that bound (at most T lanes hold any: <= T * NE survivors, typically ~T), and count over the
// survivors only. keys_lds is overwritten by the survivor list in that case.
template <int NE>
__device__ __forceinline__ void rank_top(const u64 (&key)[NE], int (&rank)[NE], int ncand, int T, u64 *keys_lds, int lane) {
    const int ef = (ncand + 63) >> 6;
#pragma unroll
    for (int e = 0; e < NE; ++e) rank[e] = 0;
    int nlist = ncand;
    if (T <= 64 && ncand > 128) {
        u64 lmax = 0ull;
#pragma unroll
        for (int e = 0; e < NE; ++e) lmax = key[e] > lmax ? key[e] : lmax;
        int above = 0;   // lanes whose maximum beats this lane's (keys are unique; empty lanes hold 0)
        for (int l = 0; l < 64; ++l) above += (readlane_u64(lmax, l) > lmax) ? 1 : 0;
        const u64 hit = __ballot(lmax != 0ull && above == T - 1);
        const u64 bound = hit ? readlane_u64(lmax, __ffsll((long long)hit) - 1) : 0ull;   // < T non-empty lanes: keep all
        const u64 lt = (1ull << lane) - 1ull;
        int ns = 0;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            if (e < ef) {
                const bool keep = key[e] != 0ull && key[e] >= bound;
                const u64 m = __ballot(keep);
                if (keep) keys_lds[ns + __popcll(m & lt)] = key[e];
                ns += __popcll(m);
            }
        }
        nlist = ns;
    }
    for (int j = 0; j < nlist; ++j) {
        const u64 kj = keys_lds[j];
#pragma unroll
        for (int e = 0; e < NE; ++e)
            if (e < ef) rank[e] += (kj > key[e]) ? 1 : 0;
    }
}

// Per-lane select state (query-level values are duplicated in lanes l and l+32).
struct SelState {
    float thr;         // score of the query's current KP-th best (−inf until KP candidates exist)
    uint32_t thr_row;  // its row (exact tie rule)
    int cnt;           // entries in the query's LDS buffer
};

// Capacity of one query's buffer = 64*E entries (E keys per lane during compaction).
// A 16-register group appends at most 32 entries per query, so compaction is triggered when
// cnt > CAP-32 after a group.

// Compact the buffer of ONE query (wave-cooperative, nb entries, wave-uniform): keep the best
// min(nb,KP), rewritten sorted best-first at qbuf[0..]. Returns the key of rank KP-1 via kth
// (valid iff nb >= KP).
template <int KP, int E>
__device__ __forceinline__ void compact_one(u64 *qbuf, int nb, int lane, u64 &kth) {
    u64 key[E];
    int rank[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = lane + 64 * e;
        key[e] = (i < nb) ? qbuf[i] : 0ull;
        rank[e] = 0;
    }
    // rank by counting; the reads below are wave-uniform addresses (LDS broadcast, conflict-free)
    for (int j = 0; j < nb; ++j) {
        const u64 kj = qbuf[j];
#pragma unroll
        for (int e = 0; e < E; ++e) rank[e] += (kj > key[e]) ? 1 : 0;
    }
    // all reads of this wave precede the writes in program order; LDS serves one wave's ops in order
    u64 kth_local = 0ull;
    bool have = false;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = lane + 64 * e;
        if (i < nb && rank[e] < KP) qbuf[rank[e]] = key[e];
        if (i < nb && rank[e] == KP - 1) { kth_local = key[e]; have = true; }
    }
    const u64 m = __ballot(have);
    kth = 0ull;
    if (m) {
        const int src = __ffsll((long long)m) - 1;
        kth = readlane_u64(kth_local, src);
    }
}

// Compact every query of this wave whose buffer could overflow in the next group (or all when
// force is set). wbuf = this wave's 32 buffers, CAP apart.
// GROUP: the most entries one filter call can append to a query's buffer (32: a whole 32x32 tile, 16: half of one)
template <int KP, int E, int CAP = 64 * E, int GROUP = 32>
__device__ __forceinline__ void compact_wave(u64 *wbuf, SelState &st, int lane, bool force) {
    constexpr int LIMIT = CAP - GROUP;
    uint32_t need = (uint32_t)__ballot(force ? (st.cnt > 0) : (st.cnt > LIMIT));
    while (need) {
        const int b = __ffs((int)need) - 1;
        need &= need - 1;
        const int nb = readlane<int>(st.cnt, b);
        u64 kth;
        compact_one<KP, E>(wbuf + (size_t)b * CAP, nb, lane, kth);
        if ((lane & 31) == b) {
            if (nb >= KP) {
                st.thr = key_score(kth);
                st.thr_row = key_row(kth);
                st.cnt = KP;
            }
        }
    }
}

// Threshold-filter one 32x32 MFMA tile's 16 registers. row0 = absolute row of the tile's row 0.
// EXACT_TIES: also accept score == thr with a lower row than the current KP-th (canonical order).
// R0, R1: the registers of the tile this call tests (all 16, or one half: at most 16 appends per query)
template <bool EXACT_TIES, int R0 = 0, int R1 = 16>
__device__ __forceinline__ void filter16(const f32x16 &acc, uint32_t row0, SelState &st, u64 *qbuf, int lane) {
    const uint32_t rbase = row0 + 4u * (uint32_t)(lane >> 5);
    const uint32_t c = (uint32_t)lane & 31u;
    const uint32_t hi_half = (uint32_t)lane >> 5;
#pragma unroll
    for (int r = R0; r < R1; ++r) {
        const float v = acc[r];
        const uint32_t row = rbase + (uint32_t)((r & 3) + 8 * (r >> 2));
        bool pass = v > st.thr;
        if (EXACT_TIES) pass = pass || (v == st.thr && row < st.thr_row);
        const u64 m = __ballot(pass);
        if (m) {
            const uint32_t plo = ((uint32_t)m >> c) & 1u;
            const uint32_t phi = ((uint32_t)(m >> 32) >> c) & 1u;
            if (pass) {
                const int slot = st.cnt + (int)(hi_half ? plo : 0u);
                qbuf[slot] = make_key(v, row);
            }
            st.cnt += (int)(plo + phi);
        }
    }
}

}  // namespace icd

// =====================================================================================================
// Select v2 (coarse kernel): two-sided SoA candidate buffers and a fixed-trip rank compaction.
//
// Per query: score[64] f32 | row[64] u32 (512 B). Lane h=0 of the query's lane pair appends upward
// from slot `kept`, lane h=1 downward from slot 63, so an append needs no cross-lane slot arithmetic:
//     if (v > thr) { score[w] = v; row[w] = r; w += step; }
// Compaction during the sweep: one entry per lane, the KP-th best score is bisected with wave ballots
// (v_cmp + s_bcnt1 per probe, no LDS traffic), the KP entries above it move to the front (unsorted)
// and the probe becomes the new threshold. The final compaction (and score ties) rank the <= 64
// entries by a unique 32-bit key = (ordered score with its 6 low bits replaced by 63 - slot):
// 16 broadcast ds_read_b128 + 64 compare/add pairs; it leaves the list sorted best-first. The 6 dropped
// bits are a relative 2^-17 perturbation of the coarse score; finalize.hpp widens tau by that much
// (COARSE_KEY_SLACK).
// =====================================================================================================
namespace icd {

constexpr float COARSE_KEY_SLACK = 1.6e-5f;  // > 2^-17 * 2: relative slack of the truncated ranking key

struct Sel2 {
    float thr;      // current KP-th best score of the query (valid lower bound)
    int kept;       // entries at the front after the last compaction (<= KP); query-level
    uint32_t aw;    // LDS byte address of this lane's next score slot
    uint32_t aw0;   // its value right after the last compaction
    uint32_t inc;   // +4 (low lane, grows up) or -4 (high lane, grows down)
};

// QB: bytes from one query's buffer to the next (512 = packed; 516 puts the 32 buffers of a wave one bank apart, for
// a select in which all lanes of a wave write at once: coarse_rg_kernel.hpp)
template <int KP, int QB = 512>
struct Sel2Ops {
    static constexpr int CAP = 64;
    // byte offsets inside a query buffer
    static constexpr uint32_t ROW_OFF = CAP * 4;
    static constexpr uint32_t QBYTES = QB;
    static_assert(QB >= CAP * 8 && QB % 4 == 0, "a query buffer holds CAP scores and CAP rows");

    __device__ static __forceinline__ void init(Sel2 &s, uint32_t qbase, int h, bool valid) {
        s.thr = valid ? -INFINITY : INFINITY;
        s.kept = 0;
        s.aw0 = h ? qbase + (CAP - 1) * 4 : qbase;
        s.aw = s.aw0;
        s.inc = h ? (uint32_t)-4 : 4u;
    }
    __device__ static __forceinline__ int used(const Sel2 &s, int h) {
        return h ? (int)(s.aw0 - s.aw) >> 2 : (int)(s.aw - s.aw0) >> 2;
    }
    // one score of the finished tile
    __device__ static __forceinline__ void step(Sel2 &s, float v, uint32_t row, char *smem) {
        if (v > s.thr) {
            *reinterpret_cast<float *>(smem + s.aw) = v;
            *reinterpret_cast<uint32_t *>(smem + s.aw + ROW_OFF) = row;
            s.aw += s.inc;
        }
    }
    // after a 16-register group: compact the queries that could overflow in the next group
    // limit: compact the queries whose entry count exceeds it (CAP - 32 = overflow guard for the next
    // 16-register group; a lower value at tile ends compacts early, while all waves are in step)
    // NOSCRATCH: the tie-ranking path ranks with v_readlane instead of the 256-B LDS scratch (a kernel whose LDS is full)
    template <bool NOSCRATCH = false>
    __device__ static __forceinline__ void check(Sel2 &s, int lane, char *smem, uint32_t wave_qbase,
                                                 uint32_t wave_scratch, bool force, int limit = CAP - 32,
                                                 unsigned long long *prof = nullptr, int quota = CAP) {
        const int h = lane >> 5;
        const int mine = used(s, h);
        // partner lane's count without touching LDS: v_permlane32_swap exchanges the two wave halves
        const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)mine, (unsigned)mine, false, false);
        const int other = (int)(h ? sw[0] : sw[1]);
        const int total = s.kept + mine + other;
        // (quota: a lane that has appended more than this since the last compaction asks for one - the caller's cheap
        //  per-lane pre-check uses the same rule, so the query that triggered the call is always compacted)
        uint32_t need = (uint32_t)__ballot(force ? (total > 0) : (total > limit || mine > quota || other > quota));
        unsigned long long pt0 = 0;
        if (prof && need) {
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pt0)::"memory");
            prof[0] += (unsigned long long)__popc(need);
        }
        while (need) {
            const int b = __ffs((int)need) - 1;
            need &= need - 1;
            // counts of query b (lane b = low lane, lane b+32 = high lane)
            const int kept_b = readlane<int>(s.kept, b);
            const int nlo = kept_b + readlane<int>(mine, b);
            const int nhi = readlane<int>(mine, b + 32);
            const uint32_t qb = wave_qbase + (uint32_t)b * QBYTES;
            const bool valid = (lane < nlo) || (lane >= CAP - nhi);
            float v = 0.f;
            uint32_t row = 0;
            if (valid) {
                v = *reinterpret_cast<const float *>(smem + qb + lane * 4);
                row = *reinterpret_cast<const uint32_t *>(smem + qb + ROW_OFF + lane * 4);
            }
            const int nvalid = nlo + nhi;
            if (!force && nvalid > KP) {
                // ---- fast path: bisect the KP-th best score with wave ballots (no LDS traffic) -----------
                // Invariant: count(okey > lo) >= KP > count(okey > hi). Every entry beats the query's
                // current threshold, so lo starts there; hi starts at the largest entry. Ends when a probe
                // leaves exactly KP entries above it; score ties that make that impossible fall through
                // to the ranking path below (unique keys).
                const uint32_t okey = valid ? order_f32(v) : 0u;
                uint32_t mx = okey;
                mx = max(mx, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mx, 0x111, 0xf, 0xf, false));   // row_shr:1
                mx = max(mx, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mx, 0x112, 0xf, 0xf, false));   // row_shr:2
                mx = max(mx, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mx, 0x114, 0xf, 0xf, false));   // row_shr:4
                mx = max(mx, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mx, 0x118, 0xf, 0xf, false));   // row_shr:8
                mx = max(mx, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mx, 0x142, 0xa, 0xf, false));   // row_bcast:15
                mx = max(mx, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mx, 0x143, 0xc, 0xf, false));   // row_bcast:31
                uint32_t hi_s = readlane<uint32_t>(mx, 63);
                uint32_t lo_s = order_f32(__builtin_bit_cast(float, readlane<uint32_t>(__float_as_uint(s.thr), b)));
                // (a threshold adopted from another list of the query may sit above some of the entries)
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
                    if ((lane & 31) == b) {
                        s.thr = unorder_f32(lo_s);   // every kept entry is above it, every dropped one at or below
                        s.kept = c;
                        s.aw0 = h ? qb + (CAP - 1) * 4 : qb + (uint32_t)c * 4;
                        s.aw = s.aw0;
                    }
                    continue;
                }
            }
            const uint32_t key = valid ? ((order_f32(v) & ~63u) | (uint32_t)(63 - lane)) : 0u;
            int rank = 0;
            if constexpr (NOSCRATCH) {
                for (int j = 0; j < 64; ++j) rank += (readlane<uint32_t>(key, j) > key) ? 1 : 0;   // (empty slots carry key 0 and never count)
            } else {
            *reinterpret_cast<uint32_t *>(smem + wave_scratch + lane * 4) = key;
            // Rank against the keys in 16-key chunks, only the chunks that hold entries ([0, nlo) at the
            // front, [64 - nhi, 64) at the back; empty slots carry key 0 and never count). The reads are
            // an asm block on purpose: for a compiler-visible ds_read in this loop hipcc emits
            // s_waitcnt vmcnt(0) (it cannot tell the scratch from the LDS-DMA ring), which drains the
            // whole prefetch pipeline of the caller on every compaction.
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                if (ch * 16 < nlo || (ch + 1) * 16 > CAP - nhi) {   // wave-uniform
                    uint4 k0, k1, k2, k3;
                    const uint32_t addr = wave_scratch + (uint32_t)ch * 64u;
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
            }
            if (valid && rank < KP) {
                *reinterpret_cast<float *>(smem + qb + rank * 4) = v;
                *reinterpret_cast<uint32_t *>(smem + qb + ROW_OFF + rank * 4) = row;
            }
            const u64 mk = __ballot(valid && rank == KP - 1);
            float nthr = 0.f;
            if (mk) nthr = __builtin_bit_cast(float, readlane<uint32_t>(__float_as_uint(v), __ffsll((long long)mk) - 1));
            if ((lane & 31) == b) {
                if (nvalid >= KP) s.thr = fmaxf(s.thr, nthr);   // (never below a threshold adopted from another list)
                s.kept = min(nvalid, KP);
                s.aw0 = h ? qb + (CAP - 1) * 4 : qb + (uint32_t)s.kept * 4;
                s.aw = s.aw0;
            }
        }
        if (prof && pt0) {
            unsigned long long pt1;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pt1)::"memory");
            prof[1] += pt1 - pt0;
        }
    }
};

}  // namespace icd
