// stream_kernel.hpp — exact FLAT / IP search for SPARSE query sets: the corpus is streamed once per
// pass of ST_QB queries and every lane walks the canonical fp32 fmaf chain of its own row.
//
// Used where the MFMA kernels are the wrong tool (bound: HBM / Infinity-Cache bandwidth, not MFMA):
//   * the AUTO fallback when only a few queries failed certification (the fp32-MFMA exact kernel
//     would sweep the whole corpus with 128-query tiles for them: 70 ms for 5 queries on a 1.25 M-row
//     shard, profiles/r01_sizes_before_pmin2.log);
//   * the reference's own call shape, one query per call (services/milvus_service.py:280-285),
//     and any small batch / large k.
//
// Work-group = 4 waves over a contiguous row range; a wave takes 64 rows at a time (lane = row).
// 32-float slices (full 128-B lines) go HBM -> LDS by buffer_load ... lds into a wave-private ring of
// 2-4 stages (no VGPR staging, no barriers, counted vmcnt), 16-B pieces XOR-swizzled on the source side;
// each lane reads its row back with ds_read_b128 and consumes it d-ascending, so each (row, query)
// score is bit-identical to oracle/icd_oracle.c chain_score(). Bound: HBM (one corpus sweep per pass). Selection is the exact
// rule (score desc, row asc) with a per-query threshold and the u64-key compaction of topk_select.hpp.
// The candidate buffer of a (wave, query) holds 64 E entries with E one larger than the exact MFMA
// kernel uses for the same KP: a 64-row step can append 64 entries at once.
// Output: one best-first list per (query slot, wave) -> reduce_lists_kernel -> finalize_kernel<false>.
//
// ONE = true: the reference's own call shape (ONE query per MilvusService.search call, services/milvus_service.py:280-285;
// also up to four: the searches of one /query request's diagnoses) as ONE launch. Three differences from the form above:
//   * the rows are spread over EVERY CU: a wave takes rows_per_step <= 64 rows per step (a multiple of 8: a stage is
//     rows_per_step / 8 LDS-DMA pieces), so that 40 474 rows are 256 work-groups x 4 waves x 40 rows instead of 159 x 4 x 64;
//     the ring gets the stages the smaller slices leave room for (up to 8);
//   * the list reduction is folded in: a work-group merges its four waves' lists in LDS, publishes ONE list of KP keys
//     (plain stores -> every wave's vmcnt(0) -> barrier -> agent-scope release -> ticket), and the work-group that draws
//     the last ticket (agent-scope acquire) merges the <= 256 lists per query and
//   * writes the FINAL outputs itself (emit_outputs of finalize.hpp: raw order, level reweight in double, stable re-sort,
//     services/milvus_service.py:290-295,314): no reduce_lists launch, no finalize launch, no memset in front.
// The ticket is a 64-bit counter that only ever counts up (work-groups per launch is a constant of the index): the last
// arriver is the one whose ticket is = nwg - 1 modulo nwg, nothing is reset, a replayed graph needs no memset node.
#pragma once
#include "finalize.hpp"
#include "topk_select.hpp"

namespace icd {

constexpr int ST_QB = 8;          // queries per pass (template QB <= ST_QB: fewer for tiny batches)
constexpr int ST_PF = 1;          // dim must be a multiple of 32 * ST_PF
constexpr int ST_MAX_ACTIVE = 64; // direct calls: batches up to this size take the streaming kernel (EXACT mode)
constexpr int ST_FALLBACK_MAX_ACTIVE = 40;    // AUTO fallback: flagged lists up to this long (less for large k: workspace). One sweep per 8 queries costs ~37 us at 37 000 rows; the MFMA kernel, its short flagged list cut into up to 2048 / KP row chunks, ~0.22 ms for any count up to a few hundred even when the handed-over threshold is useless (all-zero queries): they meet near 45 (profiles/r04_sparse_fallback_policy.log; round 3, 16 chunks at most: 0.68 ms, 146)
constexpr int ST_STAGE_BYTES = 8192;   // one wave stage: 64 rows x 32 floats
constexpr int ST_PAD_ROWS = 512;       // zero rows the index keeps behind the corpus (stages may run past n)

struct StreamArgs {
    const float *corpus;  // [n + ST_PAD_ROWS][dim], the tail zero-filled
    const float *queries;
    const int *qlist;     // nullable: slot -> query index
    const int *nq_ptr;    // nullable: device-side number of slots
    int nq;               // slots (upper bound when nq_ptr is given)
    int max_active;       // run only if the actual slot count is <= this (else the MFMA kernel runs)
    int n, dim;           // dim multiple of 32
    int rows_per_wg;      // multiple of 256
    int nwg;              // work-groups = row ranges
    int ring_stages;      // 2..4 wave-private LDS stages
    float *list_scores;   // [slot][4 * nwg][KP]
    int *list_rows;
    const float *thr0;    // nullable: [query] a score that k distinct rows are known to reach (finalize.hpp: an uncertified query's k best
                          // coarse candidates, rescored) - the query's lists start there instead of at -inf (ties pass). Without it a
                          // re-search at k > 32 kept EVERY row (a wave sees 64 rows per step, its list holds KP >= 64) and the list
                          // reduction ranked the whole corpus per query: 0.15-0.25 ms for one to five queries (profiles/r06_k100_lists.log)
    // ONE = true only
    int rows_per_step;    // rows a wave takes per step (multiple of 8, <= 64); rows_per_wg is a multiple of 4 * rows_per_step
    u64 *wg_keys;         // [slot][nwg][KP] one merged best-first list per work-group (keys, 0 = empty)
    u64 *ticket;          // monotonic arrival counter (zeroed once, at index create)
    FinArgs fin;          // the outputs, levels, id_base and k of the search (emit_outputs), counters / host_counters
    // a host caller's ONE query (icd_search.hip, search_common): the last work-group stores done_value to *done - a word of the
    // index's mapped host block - behind its outputs (system-scope release); nullptr = nobody polls
    u64 *done;
    u64 done_value;
};

// ONE query handed over IN the kernel arguments (a host caller's call: no H2D copy command in front of the launch). The
// dispatch packet's argument segment is written by the host at launch and read here like any global memory.
constexpr int ST_INLINE_FLOATS = 768;
struct alignas(16) StreamInlineQuery { float v[ST_INLINE_FLOATS]; };
static_assert(sizeof(StreamArgs) + sizeof(StreamInlineQuery) + 16 <= 4096, "kernel arguments: 4 KB at most");

template <int KP, int E, int QB>
__host__ __device__ constexpr size_t stream_lds_bytes(int dim, int ring_stages) {
    return (size_t)QB * dim * 4 + (size_t)4 * ring_stages * ST_STAGE_BYTES + (size_t)4 * QB * 64 * E * 8;
}
// ONE = true: stages of rows_per_step x 128 B; the tail (merge of <= 256 lists per query: 64 KP keys per wave, + 4 KP merged
// keys, + sorted / adjusted buffers of emit_outputs) reuses the same memory once the ring has drained
__host__ __device__ constexpr size_t stream_one_tail_bytes(int kp) { return (size_t)4 * ((size_t)4 * kp * kp * 8 + 128 * 8 + 128 * 8); }   // (the per-wave form of 3-8 queries; the four-wave form of 1-2 needs a quarter)
template <int KP, int E, int QB>
__host__ __device__ constexpr size_t stream_one_lds_bytes(int dim, int ring_stages, int rows_per_step) {
    const size_t body = (size_t)QB * dim * 4 + (size_t)4 * ring_stages * rows_per_step * 128 + (size_t)4 * QB * 64 * E * 8;
    return body > stream_one_tail_bytes(KP) ? body : stream_one_tail_bytes(KP);
}

typedef float st_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u64 gu64;   // a GLOBAL (never flat) word of an inter-work-group hand-off

// acc[qi] = fma(q[qi][d], c[d], acc[qi]) for the four d of one 16-B piece, d ascending, as QB independent
// chains. The query values live one per lane (lane & 15 = d & 15, replicated in the four DPP rows), and
// row_newbcast:n hands lane n's value to every lane of the row: no LDS broadcast reads, no SGPR traffic.
// v_fmac_f32 is the fused multiply-add of the canonical chain. The leading s_nop covers the
// VALU-write -> DPP-read hazard should the compiler have placed a copy right before the block.
#define ICD_F(a, q, c, n) "v_fmac_f32_dpp %" #a ", %" #q ", %" #c " row_newbcast:%" #n " row_mask:0xf bank_mask:0xf\n\t"
template <int QB, int N0>
__device__ __forceinline__ void stream_fma4(float (&acc)[QB], const float (&q)[QB], const st_f32x4 &c) {
    if constexpr (QB == 1) {
        asm volatile("s_nop 1\n\t" ICD_F(0, 1, 2, 6) ICD_F(0, 1, 3, 7) ICD_F(0, 1, 4, 8) ICD_F(0, 1, 5, 9)
                     : "+v"(acc[0]) : "v"(q[0]), "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w),
                       "n"(N0), "n"(N0 + 1), "n"(N0 + 2), "n"(N0 + 3));
    } else if constexpr (QB == 2) {
#define ICD_R2(c, n) ICD_F(0, 2, c, n) ICD_F(1, 3, c, n)
        asm volatile("s_nop 1\n\t" ICD_R2(4, 8) ICD_R2(5, 9) ICD_R2(6, 10) ICD_R2(7, 11)
                     : "+v"(acc[0]), "+v"(acc[1]) : "v"(q[0]), "v"(q[1]), "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w),
                       "n"(N0), "n"(N0 + 1), "n"(N0 + 2), "n"(N0 + 3));
#undef ICD_R2
    } else if constexpr (QB == 4) {
#define ICD_R4(c, n) ICD_F(0, 4, c, n) ICD_F(1, 5, c, n) ICD_F(2, 6, c, n) ICD_F(3, 7, c, n)
        asm volatile("s_nop 1\n\t" ICD_R4(8, 12) ICD_R4(9, 13) ICD_R4(10, 14) ICD_R4(11, 15)
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
                     : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w),
                       "n"(N0), "n"(N0 + 1), "n"(N0 + 2), "n"(N0 + 3));
#undef ICD_R4
    } else {
        static_assert(QB == 8, "queries per pass: 1, 2, 4 or 8");
#define ICD_R8(c, n) ICD_F(0, 8, c, n) ICD_F(1, 9, c, n) ICD_F(2, 10, c, n) ICD_F(3, 11, c, n) \
                     ICD_F(4, 12, c, n) ICD_F(5, 13, c, n) ICD_F(6, 14, c, n) ICD_F(7, 15, c, n)
        asm volatile("s_nop 1\n\t" ICD_R8(16, 20) ICD_R8(17, 21) ICD_R8(18, 22) ICD_R8(19, 23)
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                     : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7]),
                       "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w), "n"(N0), "n"(N0 + 1), "n"(N0 + 2), "n"(N0 + 3));
#undef ICD_R8
    }
}
#undef ICD_F

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate): ONE = true, whose stages hold 1-8 pieces
__device__ __forceinline__ void wait_vmcnt_uniform(int n) {
#define ICD_VM(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        ICD_VM(0) ICD_VM(1) ICD_VM(2) ICD_VM(3) ICD_VM(4) ICD_VM(5) ICD_VM(6) ICD_VM(7) ICD_VM(8) ICD_VM(9) ICD_VM(10) ICD_VM(11) ICD_VM(12)
        ICD_VM(13) ICD_VM(14) ICD_VM(15) ICD_VM(16) ICD_VM(17) ICD_VM(18) ICD_VM(19) ICD_VM(20) ICD_VM(21) ICD_VM(22) ICD_VM(23) ICD_VM(24)
        ICD_VM(25) ICD_VM(26) ICD_VM(27) ICD_VM(28) ICD_VM(29) ICD_VM(30) ICD_VM(31) ICD_VM(32) ICD_VM(33) ICD_VM(34) ICD_VM(35) ICD_VM(36)
        ICD_VM(37) ICD_VM(38) ICD_VM(39) ICD_VM(40) ICD_VM(41) ICD_VM(42) ICD_VM(43) ICD_VM(44) ICD_VM(45) ICD_VM(46) ICD_VM(47) ICD_VM(48)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef ICD_VM
}

template <int KP, int E, int QB, bool ONE>
__device__ __forceinline__ void stream_topk_body(const StreamArgs &a, const float *qsrc) {
    constexpr int CAP = 64 * E;
    static_assert(!ONE || (QB <= 4 && KP <= 16), "the single-launch form serves up to four queries at k <= 16");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // (the gate looks at the device's count itself: a.nq is already clamped to max_active by the host, and a list longer
    //  than that belongs to the MFMA kernel alone - round 3: the clamped count used to pass the gate and cost 24 sweeps)
    const int cnt = a.nq_ptr ? *a.nq_ptr : a.nq;
    if (cnt <= 0 || cnt > a.max_active) return;   // work-group-uniform
    const int nq = min(cnt, a.nq);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dim = a.dim, nsl = dim >> 5, D = a.ring_stages;
    const int RPS = ONE ? a.rows_per_step : 64;            // rows of a wave per step
    const int NP = RPS >> 3;                               // LDS-DMA pieces (8 rows x 128 B) per stage
    const uint32_t stage_bytes = ONE ? (uint32_t)RPS * 128u : (uint32_t)ST_STAGE_BYTES;
    float *qs = reinterpret_cast<float *>(smem);                                   // [QB][dim]
    const uint32_t ring_off = (uint32_t)QB * dim * 4 + (uint32_t)(wave * D) * stage_bytes;   // wave-private ring
    u64 *bufs = reinterpret_cast<u64 *>(smem + (size_t)QB * dim * 4 + (size_t)4 * D * stage_bytes) +
                (size_t)wave * QB * CAP;                                           // [QB][CAP]
    const int row_begin = blockIdx.x * a.rows_per_wg;
    const int row_end = min(a.n, row_begin + a.rows_per_wg);
    const int nsteps = ONE ? (row_end - row_begin + 4 * RPS - 1) / (4 * RPS) : (row_end - row_begin + 255) >> 8;   // steps of the work-group: 4 waves x RPS rows
    const int nlists = 4 * a.nwg;

    // LDS-DMA source: lane L of piece i brings 16 B of row 8 i + (L >> 3); the 16-B pieces of a row are
    // XOR-swizzled with (row >> 1) & 7 so that the row-per-lane ds_read_b128 below is conflict-free.
    // The swizzle of piece i depends on i only through its parity: two per-lane offsets.
    const uint32_t row_bytes = (uint32_t)dim * 4u;
    const uint32_t voff_even = (uint32_t)(lane >> 3) * row_bytes + (uint32_t)(((lane & 7) ^ (lane >> 4)) * 16);
    const uint32_t voff_odd = (uint32_t)(lane >> 3) * row_bytes + (uint32_t)(((lane & 7) ^ (4 + (lane >> 4))) * 16);
    const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.corpus) + (size_t)row_begin * dim, 0, (int)((size_t)nsteps * (4 * RPS) * row_bytes), 0x00020000);
    const uint32_t rd_base = (uint32_t)lane * 128u + (uint32_t)(((lane >> 1) & 7) * 16);
    const uint32_t q_lane = (uint32_t)(lane & 15) * 4u;

    for (int q0 = 0; q0 < nq; q0 += QB) {
        const int nqp = min(QB, nq - q0);
        __syncthreads();   // previous pass is done with qs
        for (int i = tid * 4; i < QB * dim; i += 1024) {
            const int qi = i / dim, d = i - qi * dim;
            const int slot = q0 + min(qi, nqp - 1);   // unused slots repeat the last query (never emitted)
            const int gq = a.qlist ? a.qlist[slot] : slot;
            *reinterpret_cast<float4 *>(qs + i) = *reinterpret_cast<const float4 *>(qsrc + (size_t)gq * dim + d);
        }
        __syncthreads();
        float thr[QB];
        uint32_t thr_row[QB];
        int cnt[QB];
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
            thr[qi] = -INFINITY; thr_row[qi] = 0u; cnt[qi] = 0;
            if (a.thr0) {
                const int slot = q0 + min(qi, nqp - 1);
                const float t0 = a.thr0[a.qlist ? a.qlist[slot] : slot];
                if (t0 > -INFINITY) { thr[qi] = t0; thr_row[qi] = 0xFFFFFFFFu; }   // (rows that tie with it pass)
            }
        }

        // stage cursor of the DMA stream: (step, slice) -> ring slot; runs D - 1 stages ahead of the reader.
        // Past the last stage it re-reads the last step (valid memory, never consumed).
        int is_t = 0, is_s = 0, is_slot = 0;
        auto issue = [&]() {
            const uint32_t soff = (uint32_t)(wave * RPS + 4 * RPS * min(is_t, nsteps - 1)) * row_bytes + (uint32_t)is_s * 128u;
            const uint32_t dst = ring_off + (uint32_t)is_slot * stage_bytes;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (!ONE || i < NP)   // (wave-uniform)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(crsrc, (__attribute__((address_space(3))) void *)(smem + dst + i * 1024),
                                                             16, (i & 1) ? voff_odd : voff_even, soff + (uint32_t)i * 8u * row_bytes, 0, 0);
            if (++is_s == nsl) { is_s = 0; ++is_t; }
            if (++is_slot == D) is_slot = 0;
        };
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the vmcnt accounting below starts from zero
        for (int p = 0; p < D - 1; ++p) issue();
        int rslot = 0;

        for (int t = 0; t < nsteps; ++t) {
            const int r0 = row_begin + wave * RPS + 4 * RPS * t;
            float acc[QB];
#pragma unroll
            for (int qi = 0; qi < QB; ++qi) acc[qi] = 0.0f;
            for (int s = 0; s < nsl; ++s) {
                // stage (t, s) has landed when at most D - 2 younger stages are outstanding
                if constexpr (ONE) wait_vmcnt_uniform((D - 2) * NP);
                else if (D == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (D == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                issue();   // into the slot read one slice ago (those reads completed before its fmas)
                const uint32_t ad = ring_off + (uint32_t)rslot * stage_bytes + (ONE && lane >= RPS ? 0u : rd_base);   // (idle lanes of a short stage read row 0's bytes: in bounds, never used)
                if (++rslot == D) rslot = 0;
                st_f32x4 c0, c1, c2, c3, c4, c5, c6, c7;
                asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %11\n\t"
                             "ds_read_b128 %4, %12\n\tds_read_b128 %5, %13\n\tds_read_b128 %6, %14\n\tds_read_b128 %7, %15"
                             : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3), "=&v"(c4), "=&v"(c5), "=&v"(c6), "=&v"(c7)
                             : "v"(ad), "v"(ad ^ 16u), "v"(ad ^ 32u), "v"(ad ^ 48u), "v"(ad ^ 64u), "v"(ad ^ 80u), "v"(ad ^ 96u), "v"(ad ^ 112u)
                             : "memory");
                float qlo[QB], qhi[QB];   // q[qi][32 s + (lane & 15)], q[qi][32 s + 16 + (lane & 15)]
#pragma unroll
                for (int qi = 0; qi < QB; ++qi) {
                    const uint32_t qa = q_lane + (uint32_t)s * 128u + (uint32_t)qi * row_bytes;
                    asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:64" : "=&v"(qlo[qi]), "=&v"(qhi[qi]) : "v"(qa) : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) :: "memory");
#pragma unroll
                for (int qi = 0; qi < QB; ++qi) asm volatile("" : "+v"(qlo[qi]), "+v"(qhi[qi]));
                stream_fma4<QB, 0>(acc, qlo, c0);
                stream_fma4<QB, 4>(acc, qlo, c1);
                stream_fma4<QB, 8>(acc, qlo, c2);
                stream_fma4<QB, 12>(acc, qlo, c3);
                stream_fma4<QB, 0>(acc, qhi, c4);
                stream_fma4<QB, 4>(acc, qhi, c5);
                stream_fma4<QB, 8>(acc, qhi, c6);
                stream_fma4<QB, 12>(acc, qhi, c7);
            }
            // exact select: (score desc, row asc); NaN and -inf never enter
            const uint32_t row = (uint32_t)(r0 + lane);
            const bool rvalid = (int)row < row_end && (!ONE || lane < RPS);
#pragma unroll
            for (int qi = 0; qi < QB; ++qi) {
                const float v = acc[qi];
                const bool pass = rvalid && (v > thr[qi] || (v == thr[qi] && row < thr_row[qi])) && v != -INFINITY;
                const u64 m = __ballot(pass);
                if (m) {
                    u64 *qb = bufs + (size_t)qi * CAP;
                    if (pass) qb[cnt[qi] + __popcll(m & ((1ull << lane) - 1ull))] = make_key(v, row);
                    cnt[qi] += __popcll(m);
                    if (cnt[qi] > CAP - 64) {
                        u64 kth;
                        compact_one<KP, E>(qb, cnt[qi], lane, kth);
                        if (cnt[qi] >= KP) {
                            thr[qi] = key_score(kth);
                            thr_row[qi] = key_row(kth);
                            cnt[qi] = KP;
                        }
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the run-ahead stages before the ring is reused
        if constexpr (ONE) {
            // ---- single-launch form: work-group merge -> publish -> the last arriver merges and writes the outputs ------------
            constexpr int NE4 = (4 * KP + 63) / 64;   // keys per lane of a 4 x KP merge
            const uint32_t ring_base = (uint32_t)QB * dim * 4;
            u64 *bufs_all = reinterpret_cast<u64 *>(smem + (size_t)QB * dim * 4 + (size_t)4 * D * stage_bytes);
            int nb_mine[QB];
#pragma unroll
            for (int qi = 0; qi < QB; ++qi) {   // every wave: its buffers sorted best-first
                nb_mine[qi] = 0;
                if (qi < nqp) {
                    u64 kth;
                    if (cnt[qi] > 0) compact_one<KP, E>(bufs + (size_t)qi * CAP, cnt[qi], lane, kth);
                    nb_mine[qi] = min(cnt[qi], KP);
                }
            }
            __syncthreads();   // every wave is past its loop (and drained its own LDS-DMA): the ring's memory is free
            int *nbw = reinterpret_cast<int *>(smem + ring_base + 6144);   // [4][QB] (the ring is >= 8 KB: 4 waves x >= 2 stages x >= 1 KB; keysA below it: QB x 512 B)
            int *flag = nbw + 32;
            if (lane == 0) {
#pragma unroll
                for (int qi = 0; qi < QB; ++qi) nbw[wave * QB + qi] = nb_mine[qi];
            }
            __syncthreads();
            for (int qi = wave; qi < nqp; qi += 4) {   // wave qi (mod 4) merges query qi's four lists into the work-group's one
                u64 *keysA = reinterpret_cast<u64 *>(smem + ring_base) + (size_t)qi * (64 * NE4);
                u64 key[NE4];
                int rank[NE4];
#pragma unroll
                for (int e = 0; e < NE4; ++e) {
                    const int i = lane + 64 * e, sw = i / KP, j = i - sw * KP;
                    key[e] = (i < 4 * KP && j < nbw[sw * QB + qi]) ? bufs_all[((size_t)sw * QB + qi) * CAP + j] : 0ull;
                    keysA[i] = key[e];
                }
                rank_top<NE4>(key, rank, 4 * KP, KP, keysA, lane);   // (<= 128 candidates: plain counting, every rank exact)
                int nvalid = 0;
#pragma unroll
                for (int e = 0; e < NE4; ++e) nvalid += __popcll(__ballot(key[e] != 0ull));
                // write-through (sc1) stores: the list leaves this XCD's L2 without a release fence (MI355X_MICROARCH.md,
                // visibility: the form "sc1 stores + every storing wave's vmcnt(0) + barrier + ONE lane's agent-scope atomic add;
                // the work-group whose add came last reads with sc1 loads", one work-group per CU, 8-byte accesses)
                gu64 *dst = (gu64 *)(a.wg_keys) + ((size_t)(q0 + qi) * gridDim.x + blockIdx.x) * KP;
                for (int j = lane; j < KP; j += 64)
                    if (j >= nvalid) __hip_atomic_store(dst + j, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int e = 0; e < NE4; ++e)
                    if (key[e] != 0ull && rank[e] < KP) __hip_atomic_store(dst + rank[e], key[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // publish: every storing wave's vmcnt(0) -> barrier -> ONE lane's ticket (agent-scope atomic add)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const u64 t = __hip_atomic_fetch_add((gu64 *)(a.ticket), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *flag = (t % (u64)gridDim.x) == (u64)gridDim.x - 1ull ? 1 : 0;   // (written after the add has returned)
            }
            __syncthreads();
            const int last_arriver = *flag;
            __syncthreads();   // (every thread has read the flag: the last arriver's scratch below may overlap it - at small dims also in the 1-2 query branch)
            if (last_arriver == 0) return;   // (work-group-uniform; ONE = true runs a single pass: nq <= QB)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // (no instruction: keeps the compiler from moving the sc1 loads up)
            if constexpr (QB > 2) {
                // ---- the last arriver, 3-4 queries: every wave takes a whole query (wave w: query w), four rounds of 64
                // lists each; the survivors of the rounds are ranked by the same wave. No barrier between the waves.
                const int nl = (int)gridDim.x;
                char *wbase = smem + (size_t)wave * (4 * KP * KP * 8 + 128 * 8 + 128 * 8);
                u64 *surv = reinterpret_cast<u64 *>(wbase);                       // [4 rounds][KP * KP]
                u64 *sorted = surv + 4 * KP * KP;                                  // [128]
                double *adjbuf = reinterpret_cast<double *>(sorted + 128);         // [128]
                for (int qi = wave; qi < nqp; qi += 4) {
                    const __amdgpu_buffer_rsrc_t krsrc = __builtin_amdgcn_make_buffer_rsrc(a.wg_keys + (size_t)(q0 + qi) * nl * KP, 0, nl * KP * 8, 0x00020000);
                    int tot = 0;
                    for (int rd = 0; rd * 64 < nl; ++rd) {
                        const int li = rd * 64 + lane;
                        u64 key[KP];
#pragma unroll
                        for (int e = 0; e < KP; e += 2) {
                            const auto v = __builtin_amdgcn_raw_buffer_load_b128(krsrc, (uint32_t)(min(li, nl - 1) * KP + e) * 8u, 0, 16);
                            key[e] = li < nl ? (((u64)v[1] << 32) | (u64)v[0]) : 0ull;
                            key[e + 1] = li < nl ? (((u64)v[3] << 32) | (u64)v[2]) : 0ull;
                        }
                        const u64 head = key[0];
                        int above = 0;
                        for (int l = 0; l < 64; ++l) above += (readlane_u64(head, l) > head) ? 1 : 0;
                        const u64 hit = __ballot(head != 0ull && above == KP - 1);
                        const u64 bound = hit ? readlane_u64(head, __ffsll((long long)hit) - 1) : 0ull;
                        const u64 lt = (1ull << lane) - 1ull;
#pragma unroll
                        for (int e = 0; e < KP; ++e) {
                            const bool keep = key[e] != 0ull && key[e] >= bound;
                            const u64 m = __ballot(keep);
                            if (m == 0ull) break;
                            if (keep) surv[tot + __popcll(m & lt)] = key[e];
                            tot += __popcll(m);
                        }
                    }
                    for (int base = 0; base < tot; base += 64) {
                        const int i = base + lane;
                        const u64 ku = i < tot ? surv[i] : 0ull;
                        int ru = 0;
                        for (int j = 0; j < tot; ++j) ru += (surv[j] > ku) ? 1 : 0;
                        if (ku != 0ull && ru < 128) sorted[ru] = ku;
                    }
                    const int slot = q0 + qi;
                    const int qidx = a.qlist ? a.qlist[slot] : slot;
                    emit_outputs(a.fin, qidx, sorted, min(min(tot, KP), a.fin.k), adjbuf, lane);
                }
                __syncthreads();
            } else {
            // ---- the last arriver: <= 256 lists per query, lane = list (best first) ------------------------------------------
            // A list's head is its best key. The KP-th largest of a wave's 64 heads bounds the wave's KP-th best from below (KP
            // distinct keys reach it); only keys at or above that bound can rank: the four waves compact theirs into one LDS list
            // (tens of keys for Gaussian data, at most 4 KP KP) and wave 0 ranks that list alone.
            const int nl = (int)gridDim.x;
            u64 *surv = reinterpret_cast<u64 *>(smem);              // [4][KP * KP] survivors of the four waves
            int *nsurv = reinterpret_cast<int *>(surv + 4 * KP * KP);   // [4]
            u64 *sorted = reinterpret_cast<u64 *>(nsurv + 4);       // [128]
            double *adjbuf = reinterpret_cast<double *>(sorted + 128);   // [128]
            for (int qi = 0; qi < nqp; ++qi) {
                const int li = wave * 64 + lane;
                u64 key[KP];
                // (sc1 on EVERY load of the handed-off bytes: 16-byte buffer loads, aux 16 = sc1)
                const __amdgpu_buffer_rsrc_t krsrc = __builtin_amdgcn_make_buffer_rsrc(a.wg_keys + (size_t)(q0 + qi) * nl * KP, 0, nl * KP * 8, 0x00020000);
#pragma unroll
                for (int e = 0; e < KP; e += 2) {
                    const auto v = __builtin_amdgcn_raw_buffer_load_b128(krsrc, (uint32_t)(min(li, nl - 1) * KP + e) * 8u, 0, 16);
                    key[e] = li < nl ? (((u64)v[1] << 32) | (u64)v[0]) : 0ull;
                    key[e + 1] = li < nl ? (((u64)v[3] << 32) | (u64)v[2]) : 0ull;
                }
                const u64 head = key[0];
                int above = 0;   // lanes whose head beats this lane's (keys are unique; empty lists hold 0)
                for (int l = 0; l < 64; ++l) above += (readlane_u64(head, l) > head) ? 1 : 0;
                const u64 hit = __ballot(head != 0ull && above == KP - 1);
                const u64 bound = hit ? readlane_u64(head, __ffsll((long long)hit) - 1) : 0ull;   // fewer than KP lists: keep all
                const u64 lt = (1ull << lane) - 1ull;
                int ns = 0;
                u64 *mine = surv + (size_t)wave * (KP * KP);
#pragma unroll
                for (int e = 0; e < KP; ++e) {   // (a list is sorted: once a key falls below the bound the rest do; at most KP lanes hold any)
                    const bool keep = key[e] != 0ull && key[e] >= bound;
                    const u64 m = __ballot(keep);
                    if (m == 0ull) break;   // wave-uniform
                    if (keep) mine[ns + __popcll(m & lt)] = key[e];
                    ns += __popcll(m);
                }
                if (lane == 0) nsurv[wave] = ns;
                __syncthreads();
                // the union of the four waves' survivors, ranked by counting among themselves: the overall top KP is in it (each
                // wave's top KP is, and a key outside its wave's top KP is outside the overall one). Tens of keys: every wave
                // ranks a 64-key slice of the union against all of it (wave-uniform LDS addresses: broadcasts).
                const int n0 = nsurv[0], n1 = nsurv[1], n2 = nsurv[2], n3 = nsurv[3];
                const int tot = n0 + n1 + n2 + n3;
                for (int base = wave * 64; base < tot; base += 256) {
                    const int i = base + lane;
                    u64 ku = 0ull;
                    if (i < tot) {
                        const int w = i < n0 ? 0 : (i < n0 + n1 ? 1 : (i < n0 + n1 + n2 ? 2 : 3));
                        const int j = i - (w == 0 ? 0 : (w == 1 ? n0 : (w == 2 ? n0 + n1 : n0 + n1 + n2)));
                        ku = surv[(size_t)w * (KP * KP) + j];
                    }
                    int ru = 0;
                    for (int w = 0; w < 4; ++w) {
                        const int nw = nsurv[w];
                        const u64 *sw = surv + (size_t)w * (KP * KP);
                        for (int j = 0; j < nw; ++j) ru += (sw[j] > ku) ? 1 : 0;
                    }
                    if (ku != 0ull && ru < 128) sorted[ru] = ku;
                }
                __syncthreads();
                if (wave == 0) {
                    const int slot = q0 + qi;
                    const int qidx = a.qlist ? a.qlist[slot] : slot;
                    emit_outputs(a.fin, qidx, sorted, min(min(tot, KP), a.fin.k), adjbuf, lane);
                }
                __syncthreads();
            }
            }
            if (tid == 0 && a.fin.host_counters) {   // what the memset in front and finalize<false>'s block 0 did for this path
                a.fin.counters[0] = 0;
                a.fin.host_counters[0] = 0;
                for (int i = 1; i < 5; ++i) a.fin.host_counters[i] = a.fin.counters[i];
            }
            if (a.done) {   // (work-group-uniform) the host polls this word instead of waiting for the stream
                __threadfence_system();   // every wave: its output stores have reached the host block ...
                __syncthreads();
                if (tid == 0) __hip_atomic_store(a.done, a.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // ... before the word does
            }
            return;
        } else {
        // this wave's best-first list of every query of the pass
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
            if (qi < nqp) {
                u64 *qb = bufs + (size_t)qi * CAP;
                u64 kth;
                if (cnt[qi] > 0) compact_one<KP, E>(qb, cnt[qi], lane, kth);
                const int nb = min(cnt[qi], KP);
                const size_t o = ((size_t)(q0 + qi) * nlists + (size_t)blockIdx.x * 4 + wave) * KP;
                for (int j = lane; j < KP; j += 64) {
                    float s = -INFINITY;
                    int r = -1;
                    if (j < nb) { const u64 k = qb[j]; s = key_score(k); r = (int)key_row(k); }
                    a.list_scores[o + j] = s;
                    a.list_rows[o + j] = r;
                }
            }
        }
        }
    }
}

template <int KP, int E, int QB, bool ONE = false>
__global__ __launch_bounds__(256) void stream_topk_kernel(StreamArgs a) {
    stream_topk_body<KP, E, QB, ONE>(a, a.queries);
}
// the single-launch form of ONE query whose vector arrives in the kernel arguments (StreamInlineQuery)
template <int KP, int E>
__global__ __launch_bounds__(256) void stream_one_inline_kernel(StreamArgs a, StreamInlineQuery q) {
    stream_topk_body<KP, E, 1, true>(a, q.v);
}

// One wave per (slot, output list g): merge lists g, g + P_out, g + 2 P_out, ... of the slot's nlists
// best-first lists (up to 512 candidates per wave) into the best KP. Output layout [slot][P_out][KP]
// = what finalize_kernel<false> reads.
struct ReduceArgs {
    const float *list_scores;
    const int *list_rows;
    int nlists, KP, P_out;
    const int *nq_ptr;
    int nq, max_active;
    float *part_scores;
    int *part_rows;
};

__global__ __launch_bounds__(256) void reduce_lists_kernel(ReduceArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cnt = a.nq_ptr ? *a.nq_ptr : a.nq;
    if (cnt <= 0 || cnt > a.max_active) return;
    const int nq = min(cnt, a.nq);
    const int w = blockIdx.x * 4 + wave;
    const int slot = w / a.P_out, g = w - slot * a.P_out;
    if (slot >= nq) return;
    u64 *keys = reinterpret_cast<u64 *>(smem) + (size_t)wave * 512;
    const int per = (a.nlists + a.P_out - 1) / a.P_out;   // lists merged by this wave
    const int ncand = per * a.KP;                          // <= 512 (checked on the host)
    u64 key[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int i = lane + 64 * e;
        key[e] = 0ull;
        if (i < ncand) {
            const int li = g + (i / a.KP) * a.P_out, j = i % a.KP;
            if (li < a.nlists) {
                const size_t src = ((size_t)slot * a.nlists + li) * a.KP + j;
                const int row = a.list_rows[src];
                if (row >= 0) key[e] = make_key(a.list_scores[src], (uint32_t)row);
            }
            keys[i] = key[e];
        }
    }
    int rank[8];
    rank_top<8>(key, rank, ncand, a.KP, keys, lane);
    int nvalid = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) nvalid += __popcll(__ballot(key[e] != 0ull));
    const size_t o = ((size_t)slot * a.P_out + g) * a.KP;
    for (int j = lane; j < a.KP; j += 64)
        if (j >= nvalid) { a.part_scores[o + j] = -INFINITY; a.part_rows[o + j] = -1; }
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if (key[e] != 0ull && rank[e] < a.KP) {
            a.part_scores[o + rank[e]] = key_score(key[e]);
            a.part_rows[o + rank[e]] = (int)key_row(key[e]);
        }
}

}  // namespace icd
