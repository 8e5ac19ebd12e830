// gemm_x.hip — probe for a different decomposition of the coarse pass ("design X"): is a plain
// Q x C^T sweep with TWO waves per SIMD and BOTH operands streamed through LDS faster on this chip than the shipped loop
// (queries in registers, one wave per SIMD: 1 141 TFLOP/s without any select)?
//
//   work-group   8 waves (2 per SIMD), tile = 256 queries x 128 corpus rows, BK = 64 halves per stage
//   wave (qg, rh) 64 queries x 64 rows: 4 x 4 accumulators of v_mfma_f32_16x16x32_f16 (64 VGPRs); per k-step of 32 it
//                reads 4 row fragments + 4 query fragments (ds_read_b128) for 16 MFMAs: 0.5 reads per MFMA, as the product
//   stage        48 KB = 16 KB of rows + 32 KB of queries by LDS-DMA (6 one-KiB pieces per wave), 16-B chunks XOR-swizzled
//                on the source side; ring of S stages; one raw s_barrier per stage, counted vmcnt
//   sweep        work-group w takes query tile w / LISTS and the row tiles [t0, t1) of its list, like the product's lists
// No select, no output but a checksum: this measures the ceiling of the decomposition, nothing else.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -o gemm_x gemm_x.hip && ./gemm_x
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int D = 768, BK = 64, KS = D / BK;
constexpr int TQ = 256;

// RG = 16-row groups per wave: 4 -> tile of 128 rows (16 KB of rows + 32 KB of queries per stage), 8 -> 256 rows (32 + 32 KB:
// half the LDS-DMA bytes per FLOP, 128 accumulator registers per wave - the geometry of the guide's 256 x 256 template)
// SEL (round 4): the select VERDICT r3 proposed for this form - no LDS candidate area: per quad of accumulator registers four
// v_cmp against the query's threshold, OR-ed, one wave-uniform branch; a passing lane takes a slot of the (query, list)'s
// global spill list with an atomic and stores (score, row) there. The thresholds are FIXED at the level that lets 0.5 % of
// the scores pass (the product's append rate): no bootstrap, no threshold upkeep, no compaction, no flush - a LOWER bound
// of what a real select costs in this form.
template <int S, int RG, int SEL = 0>
__device__ __forceinline__ void gemm_x_body(const _Float16 *q16, const _Float16 *c16, int ctiles, int tiles_per_wg, int lists, float *out,
                                            float thr = 0.f, int *spill_cnt = nullptr, unsigned long long *spill = nullptr, int spill_cap = 0) {
    constexpr int TR = 32 * RG, NA = TR / 64;   // rows per tile, A pieces per wave and stage
    constexpr int A_BYTES = TR * BK * 2, B_BYTES = TQ * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qg = wave & 3, rh = wave >> 2;
    const int r16 = lane & 15, g16 = lane >> 4;
    const int mtile = blockIdx.x / lists, li = blockIdx.x % lists;
    const int t0 = li * tiles_per_wg, t1 = min(ctiles, t0 + tiles_per_wg);
    if (t0 >= t1) return;
    // LDS-DMA source offsets of this wave's six pieces of a stage: A pieces 2w, 2w+1 (8 rows each), B pieces 4w .. 4w+3
    uint32_t a_off[NA], b_off[4];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = (wave * NA + i) * 8 + (lane >> 3);
        a_off[i] = (uint32_t)row * (D * 2) + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        b_off[i] = (uint32_t)row * (D * 2) + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
    }
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(c16), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(q16) + (size_t)mtile * TQ * D, 0, 0x7FFFFFFF, 0x00020000);
    auto issue = [&](int tile, int ks, int slot) {
        char *sb = smem + slot * STAGE_BYTES;
        const int trow = min(tile, ctiles - 1);
        const uint32_t asoff = (uint32_t)trow * (uint32_t)(TR * D * 2) + (uint32_t)ks * (BK * 2);
        const uint32_t bsoff = (uint32_t)ks * (BK * 2);
#pragma unroll
        for (int i = 0; i < NA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(crs, (__attribute__((address_space(3))) void *)(sb + (wave * NA + i) * 1024), 16, a_off[i], asoff, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(qrs, (__attribute__((address_space(3))) void *)(sb + A_BYTES + (wave * 4 + i) * 1024), 16, b_off[i], bsoff, 0, 0);
    };
    // fragment read offsets inside a stage: row / query r16 of a 16-group, chunk 4 k2 + g16
    uint32_t rd[2];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) rd[k2] = (uint32_t)r16 * 128u + (uint32_t)(((4 * k2 + g16) ^ ((r16 >> 1) & 7)) * 16);
    const uint32_t a_base = (uint32_t)(rh * 16 * RG) * 128u, b_base = (uint32_t)A_BYTES + (uint32_t)(qg * 64) * 128u;

    f32x4 acc[RG][4];
    float keep = 0.f;
    int mypos[4] = {0, 0, 0, 0};   // (SEL == 2)
    const int ntiles = t1 - t0, nstages = ntiles * KS;
    // prologue
#pragma unroll
    for (int p = 0; p < S - 1; ++p) issue(t0 + p / KS, p % KS, p % S);
    constexpr int NP = NA + 4;   // LDS-DMA pieces per wave and stage
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(NP * (S - 2)) : "memory");
    for (int g = 0; g < nstages; ++g) {
        const int ks = g % KS, slot = g % S;
        if (ks == 0) {
#pragma unroll
            for (int a = 0; a < RG; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
        }
        {   // every wave is past stage g-1: its slot takes stage g+S-1
            const int n = g + S - 1;
            issue(t0 + n / KS, n % KS, n % S);
        }
        const char *sb = smem + slot * STAGE_BYTES;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            half8 af[RG], bf[4];
#pragma unroll
            for (int t = 0; t < RG; ++t) af[t] = *reinterpret_cast<const half8 *>(sb + a_base + rd[k2] + t * 2048);
#pragma unroll
            for (int t = 0; t < 4; ++t) bf[t] = *reinterpret_cast<const half8 *>(sb + b_base + rd[k2] + t * 2048);
#pragma unroll
            for (int a = 0; a < RG; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        if (ks == KS - 1) {
            if constexpr (SEL != 0) {
                const int tile = t0 + g / KS;
#pragma unroll
                for (int a = 0; a < RG; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const f32x4 v = acc[a][b];
                        const bool any = v[0] > thr || v[1] > thr || v[2] > thr || v[3] > thr;
                        if (__builtin_amdgcn_ballot_w64(any)) {   // (wave-uniform, rarely taken)
                            const int q = mtile * TQ + qg * 64 + b * 16 + r16;
                            const int row0 = tile * TR + rh * 16 * RG + a * 16 + 4 * g16;
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (v[i] > thr) {
                                    if constexpr (SEL == 1) {
                                        const int pos = atomicAdd(&spill_cnt[q * lists + li], 1);
                                        if (pos < spill_cap)
                                            spill[((size_t)q * lists + li) * spill_cap + pos] = ((unsigned long long)__float_as_uint(v[i]) << 32) | (unsigned)(row0 + i);
                                    } else {   // SEL == 2: a private sub-list per (query, list, row half, lane group): the slot counter is a register
                                        const int pos = mypos[b]++;
                                        if (pos < spill_cap / 8)
                                            spill[(((size_t)q * lists + li) * 8 + rh * 4 + g16) * (spill_cap / 8) + pos] = ((unsigned long long)__float_as_uint(v[i]) << 32) | (unsigned)(row0 + i);
                                    }
                                }
                        }
                    }
            } else {
#pragma unroll
                for (int a = 0; a < RG; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) keep += acc[a][b][0] + acc[a][b][3];
            }
        }
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"i"(NP * (S - 2)) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (SEL == 2) {
#pragma unroll
        for (int b = 0; b < 4; ++b) atomicAdd(&spill_cnt[(mtile * TQ + qg * 64 + b * 16 + r16) * lists + li], mypos[b]);   // (the sub-lists' lengths: once per sweep)
    }
    out[blockIdx.x * 512 + tid] = keep;
}

// ---- ping-pong form (round 4): the same 256 x 256 tile, 8 waves, but the two waves of a SIMD (rh = 0 / 1, same qg) run ONE PHASE
// APART: a stage (32 halves of K: 32 KB, ring of four) is a LOAD phase (4 LDS-DMA pieces of stage g + 3, the stage's 12 fragment
// reads) and an MFMA phase (32 MFMAs under s_setprio), every phase ends in a raw s_barrier, and group rh = 1 enters the loop one
// barrier late - while one wave of a SIMD issues its MFMAs the other one loads (the guide's 8-phase template reduced to two phases).
// RAW: a stage is read two phases after the vmcnt that retires its pieces (vmcnt(4): one stage in flight) and a barrier;
// WAR: its slot is restaged one phase after its last read.
__device__ __forceinline__ void gemm_x_pp_body(const _Float16 *q16, const _Float16 *c16, int ctiles, int tiles_per_wg, int lists, float *out) {
    constexpr int BKP = 32, KSP = D / BKP, S = 4, RG = 8, TR = 256;
    constexpr int A_BYTES = TR * BKP * 2, B_BYTES = TQ * BKP * 2, STAGE_BYTES = A_BYTES + B_BYTES;   // 16 + 16 KB
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qg = wave & 3, rh = wave >> 2;
    const int r16 = lane & 15, g16 = lane >> 4;
    const int mtile = blockIdx.x / lists, li = blockIdx.x % lists;
    const int t0 = li * tiles_per_wg, t1 = min(ctiles, t0 + tiles_per_wg);
    if (t0 >= t1) return;
    // pieces of 1 KiB = 16 rows x 64 B: lane l -> row l >> 2, 16-byte slot l & 3 (LDS side linear); the source chunk is the slot
    // XOR (row >> 2) & 3, so that the fragment read of row r16, chunk g16 finds it at slot g16 ^ ((r16 >> 2) & 3): conflict-free
    uint32_t a_off[2], b_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 16 + (lane >> 2);
        const uint32_t o = (uint32_t)row * (D * 2) + (uint32_t)(((lane & 3) ^ ((row >> 2) & 3)) * 16);
        a_off[i] = o; b_off[i] = o;
    }
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(c16), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(q16) + (size_t)mtile * TQ * D, 0, 0x7FFFFFFF, 0x00020000);
    auto issue = [&](int tile, int ks, int slot) {
        char *sb = smem + slot * STAGE_BYTES;
        const int trow = min(tile, ctiles - 1);
        const uint32_t asoff = (uint32_t)trow * (uint32_t)(TR * D * 2) + (uint32_t)ks * (BKP * 2);
        const uint32_t bsoff = (uint32_t)ks * (BKP * 2);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(crs, (__attribute__((address_space(3))) void *)(sb + (wave * 2 + i) * 1024), 16, a_off[i], asoff, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(qrs, (__attribute__((address_space(3))) void *)(sb + A_BYTES + (wave * 2 + i) * 1024), 16, b_off[i], bsoff, 0, 0);
    };
    const uint32_t rd = (uint32_t)r16 * 64u + (uint32_t)((g16 ^ ((r16 >> 2) & 3)) * 16);
    const uint32_t a_base = (uint32_t)(rh * 16 * RG) * 64u, b_base = (uint32_t)A_BYTES + (uint32_t)(qg * 64) * 64u;
    f32x4 acc[RG][4];
    float keep = 0.f;
    const int ntiles = t1 - t0, nstages = ntiles * KSP;
#pragma unroll
    for (int p = 0; p < S - 1; ++p) issue(t0 + p / KSP, p % KSP, p % S);
    asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");   // stages 0 and 1 have landed everywhere
    if (rh == 1) asm volatile("s_barrier" ::: "memory");             // the stagger
    for (int g = 0; g < nstages; ++g) {
        const int ks = g % KSP, slot = g % S;
        if (ks == 0) {
#pragma unroll
            for (int a = 0; a < RG; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
        }
        // LOAD phase
        {
            const int n = g + S - 1;
            issue(t0 + n / KSP, n % KSP, n % S);
        }
        const char *sb = smem + slot * STAGE_BYTES;
        half8 af[RG], bf[4];
#pragma unroll
        for (int t = 0; t < RG; ++t) af[t] = *reinterpret_cast<const half8 *>(sb + a_base + rd + t * 1024);
#pragma unroll
        for (int t = 0; t < 4; ++t) bf[t] = *reinterpret_cast<const half8 *>(sb + b_base + rd + t * 1024);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // MFMA phase
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int a = 0; a < RG; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (ks == KSP - 1) {
#pragma unroll
            for (int a = 0; a < RG; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) keep += acc[a][b][0] + acc[a][b][3];
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    }
    if (rh == 0) asm volatile("s_barrier" ::: "memory");   // (the late group's last barrier)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * 512 + tid] = keep;
}
__global__ __launch_bounds__(512, 1) void gemm_x_pp(const _Float16 *q, const _Float16 *c, int ct, int tpw, int l, float *o) { gemm_x_pp_body(q, c, ct, tpw, l, o); }

// ---- two INDEPENDENT work-groups per CU (round 4): four waves (one per SIMD), tile = 128 rows x 256 queries, wave tile 128 x 64
// (128 accumulator registers), 32-deep stages of 24 KB in a ring of three (72 KB: two work-groups fit a CU), 256 registers per wave.
// The two work-groups of a CU share nothing and drift apart: the select of one can run under the MFMAs of the other. SEL as above
// (2 = private sub-lists, slot counters in registers).
template <int SEL>
__device__ __forceinline__ void gemm_x4_body(const _Float16 *q16, const _Float16 *c16, int ctiles, int tiles_per_wg, int lists, float *out,
                                             float thr, int *spill_cnt, unsigned long long *spill, int spill_cap) {
    constexpr int BKP = 32, KSP = D / BKP, S = 3, RG = 8, TR = 128;
    constexpr int A_BYTES = TR * BKP * 2, B_BYTES = TQ * BKP * 2, STAGE_BYTES = A_BYTES + B_BYTES;   // 8 + 16 KB
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int qg = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g16 = lane >> 4;
    const int mtile = blockIdx.x / lists, li = blockIdx.x % lists;
    const int t0 = li * tiles_per_wg, t1 = min(ctiles, t0 + tiles_per_wg);
    if (t0 >= t1) return;
    uint32_t a_off[2], b_off[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (qg * 2 + i) * 16 + (lane >> 2);
        a_off[i] = (uint32_t)row * (D * 2) + (uint32_t)(((lane & 3) ^ ((row >> 2) & 3)) * 16);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (qg * 4 + i) * 16 + (lane >> 2);
        b_off[i] = (uint32_t)row * (D * 2) + (uint32_t)(((lane & 3) ^ ((row >> 2) & 3)) * 16);
    }
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(c16), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(q16) + (size_t)mtile * TQ * D, 0, 0x7FFFFFFF, 0x00020000);
    auto issue = [&](int tile, int ks, int slot) {
        char *sb = smem + slot * STAGE_BYTES;
        const int trow = min(tile, ctiles - 1);
        const uint32_t asoff = (uint32_t)trow * (uint32_t)(TR * D * 2) + (uint32_t)ks * (BKP * 2);
        const uint32_t bsoff = (uint32_t)ks * (BKP * 2);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(crs, (__attribute__((address_space(3))) void *)(sb + (qg * 2 + i) * 1024), 16, a_off[i], asoff, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(qrs, (__attribute__((address_space(3))) void *)(sb + A_BYTES + (qg * 4 + i) * 1024), 16, b_off[i], bsoff, 0, 0);
    };
    const uint32_t rd = (uint32_t)r16 * 64u + (uint32_t)((g16 ^ ((r16 >> 2) & 3)) * 16);
    const uint32_t b_base = (uint32_t)A_BYTES + (uint32_t)(qg * 64) * 64u;
    f32x4 acc[RG][4];
    float keep = 0.f;
    int mypos[4] = {0, 0, 0, 0};
    const int ntiles = t1 - t0, nstages = ntiles * KSP;
#pragma unroll
    for (int p = 0; p < S - 1; ++p) issue(t0 + p / KSP, p % KSP, p % S);
    asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
    for (int g = 0; g < nstages; ++g) {
        const int ks = g % KSP, slot = g % S;
        if (ks == 0) {
#pragma unroll
            for (int a = 0; a < RG; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
        }
        {   // every wave is past stage g-1: its slot takes stage g+S-1
            const int n = g + S - 1;
            issue(t0 + n / KSP, n % KSP, n % S);
        }
        const char *sb = smem + slot * STAGE_BYTES;
        half8 af[RG], bf[4];
#pragma unroll
        for (int t = 0; t < RG; ++t) af[t] = *reinterpret_cast<const half8 *>(sb + rd + t * 1024);
#pragma unroll
        for (int t = 0; t < 4; ++t) bf[t] = *reinterpret_cast<const half8 *>(sb + b_base + rd + t * 1024);
#pragma unroll
        for (int a = 0; a < RG; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
        if (ks == KSP - 1) {
            if constexpr (SEL != 0) {
                const int tile = t0 + g / KSP;
#pragma unroll
                for (int a = 0; a < RG; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const f32x4 v = acc[a][b];
                        const bool any = v[0] > thr || v[1] > thr || v[2] > thr || v[3] > thr;
                        if (__builtin_amdgcn_ballot_w64(any)) {
                            const int q = mtile * TQ + qg * 64 + b * 16 + r16;
                            const int row0 = tile * TR + a * 16 + 4 * g16;
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (v[i] > thr) {
                                    const int pos = mypos[b]++;
                                    if (pos < spill_cap / 4)
                                        spill[(((size_t)q * lists + li) * 4 + g16) * (spill_cap / 4) + pos] = ((unsigned long long)__float_as_uint(v[i]) << 32) | (unsigned)(row0 + i);
                                }
                        }
                    }
            } else {
#pragma unroll
                for (int a = 0; a < RG; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) keep += acc[a][b][0] + acc[a][b][3];
            }
        }
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (SEL != 0) {
#pragma unroll
        for (int b = 0; b < 4; ++b) atomicAdd(&spill_cnt[(mtile * TQ + qg * 64 + b * 16 + r16) * lists + li], mypos[b]);
    }
    out[blockIdx.x * 256 + tid] = keep;
}
__global__ __launch_bounds__(256, 2) void gemm_x4_plain(const _Float16 *q, const _Float16 *c, int ct, int tpw, int l, float *o, float thr, int *cnt, unsigned long long *sp, int cap) {
    gemm_x4_body<0>(q, c, ct, tpw, l, o, thr, cnt, sp, cap);
}
__global__ __launch_bounds__(256, 2) void gemm_x4_sel(const _Float16 *q, const _Float16 *c, int ct, int tpw, int l, float *o, float thr, int *cnt, unsigned long long *sp, int cap) {
    gemm_x4_body<2>(q, c, ct, tpw, l, o, thr, cnt, sp, cap);
}

#define ICD_GX(SV, RGV)                                                                                              \
    __global__ __launch_bounds__(512, 1) void gemm_x_##SV##_##RGV(const _Float16 *q, const _Float16 *c, int ct, int tpw, int l, float *o) { \
        gemm_x_body<SV, RGV>(q, c, ct, tpw, l, o);                                                                        \
    }
ICD_GX(2, 4)
ICD_GX(3, 4)
ICD_GX(2, 8)
__global__ __launch_bounds__(512, 1) void gemm_x_sel_2_8(const _Float16 *q, const _Float16 *c, int ct, int tpw, int l, float *o, float thr, int *cnt,
                                                         unsigned long long *spill, int cap) {
    gemm_x_body<2, 8, 1>(q, c, ct, tpw, l, o, thr, cnt, spill, cap);
}
__global__ __launch_bounds__(512, 1) void gemm_x_sel2_2_8(const _Float16 *q, const _Float16 *c, int ct, int tpw, int l, float *o, float thr, int *cnt,
                                                          unsigned long long *spill, int cap) {
    gemm_x_body<2, 8, 2>(q, c, ct, tpw, l, o, thr, cnt, spill, cap);
}

#define RUN_CASE(SV, RGV)                                                                                                   \
    do {                                                                                                                    \
        constexpr int TR = 32 * RGV;                                                                                        \
        const int ctiles = n / TR, mtiles = nq / TQ;                                                                        \
        const int tiles_per_wg = (ctiles + lists - 1) / lists, grid = mtiles * lists;                                       \
        const size_t lds = (size_t)SV * (TR * BK * 2 + TQ * BK * 2);                                                        \
        float *o;                                                                                                           \
        hipMalloc(&o, (size_t)grid * 512 * 4);                                                                              \
        hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_x_##SV##_##RGV), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
        for (int w = 0; w < 100; ++w) hipLaunchKernelGGL(gemm_x_##SV##_##RGV, dim3(grid), dim3(512), lds, 0, dq, dc, ctiles, tiles_per_wg, lists, o); \
        hipDeviceSynchronize();                                                                                             \
        if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }                                       \
        hipEventRecord(e0);                                                                                                 \
        for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(gemm_x_##SV##_##RGV, dim3(grid), dim3(512), lds, 0, dq, dc, ctiles, tiles_per_wg, lists, o); \
        hipEventRecord(e1);                                                                                                 \
        hipEventSynchronize(e1);                                                                                            \
        float ms;                                                                                                           \
        hipEventElapsedTime(&ms, e0, e1);                                                                                   \
        hipFree(o);                                                                                                         \
        const double flop = 2.0 * nq * (double)n * D * iters;                                                               \
        printf("tile 256 x %3d, lists %2d (grid %3d, %2d tiles per work-group) S=%d (%3zu KB LDS): %.4f ms per launch, %.0f TFLOP/s (%.3f of 2500)\n", TR, \
               lists, grid, tiles_per_wg, SV, lds / 1024, ms / iters, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 2500.0); \
    } while (0)

int main(int argc, char **argv) {
    const int nq = 10240, n = 37120, lists = 6, iters = 20;   // 40 query tiles x 290 / 145 row tiles; 240 work-groups
    std::vector<_Float16> hq((size_t)nq * D), hc((size_t)(n + 256) * D);
    srand(1);
    for (auto &x : hq) x = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
    for (auto &x : hc) x = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
    _Float16 *dq, *dc;
    hipMalloc(&dq, hq.size() * 2); hipMalloc(&dc, hc.size() * 2);
    hipMemcpy(dq, hq.data(), hq.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dc, hc.data(), hc.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        RUN_CASE(2, 4);
        RUN_CASE(3, 4);
        RUN_CASE(2, 8);
    }
    {   // ping-pong form against the plain 256 x 256 kernel: same sums expected bit for bit
        constexpr int TR = 256;
        const int ctiles = n / TR, mtiles = nq / TQ;
        const int tiles_per_wg = (ctiles + lists - 1) / lists, grid = mtiles * lists;
        const size_t lds_ref = (size_t)2 * (TR * BK * 2 + TQ * BK * 2), lds_pp = (size_t)4 * (TR * 32 * 2 + TQ * 32 * 2);
        float *o1, *o2;
        hipMalloc(&o1, (size_t)grid * 512 * 4); hipMalloc(&o2, (size_t)grid * 512 * 4);
        hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_x_pp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pp);
        hipLaunchKernelGGL(gemm_x_2_8, dim3(grid), dim3(512), lds_ref, 0, dq, dc, ctiles, tiles_per_wg, lists, o1);
        for (int w = 0; w < 50; ++w) hipLaunchKernelGGL(gemm_x_pp, dim3(grid), dim3(512), lds_pp, 0, dq, dc, ctiles, tiles_per_wg, lists, o2);
        hipDeviceSynchronize();
        if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
        std::vector<float> h1((size_t)grid * 512), h2((size_t)grid * 512);
        int bad_total = 0;
        for (int rep = 0; rep < 5; ++rep) {   // (a race would show as sums that come and go)
            hipLaunchKernelGGL(gemm_x_pp, dim3(grid), dim3(512), lds_pp, 0, dq, dc, ctiles, tiles_per_wg, lists, o2);
            hipMemcpy(h1.data(), o1, h1.size() * 4, hipMemcpyDeviceToHost);
            hipMemcpy(h2.data(), o2, h2.size() * 4, hipMemcpyDeviceToHost);
            int bad = 0;
            for (size_t i = 0; i < h1.size(); ++i) bad += (h1[i] != h2[i]);
            bad_total += bad;
        }
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(gemm_x_pp, dim3(grid), dim3(512), lds_pp, 0, dq, dc, ctiles, tiles_per_wg, lists, o2);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = 2.0 * nq * (double)n * D * iters;
            printf("tile 256 x 256 PING-PONG (two waves of a SIMD one phase apart, 32-deep stages, ring of 4 = 128 KB): %.4f ms per launch, %.0f TFLOP/s (%.3f of 2500); checksums differing from the plain kernel's in 5 runs: %d\n",
                   ms / iters, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 2500.0, bad_total);
        }
        hipFree(o1); hipFree(o2);
    }
    {   // two independent four-wave work-groups per CU: 128 x 256 tiles, plain and with the private-sub-list select
        constexpr int TR = 128;
        const int ctiles = n / TR, mtiles = nq / TQ, cap = 256;
        const size_t lds = (size_t)3 * (TR * 32 * 2 + TQ * 32 * 2);
        for (int lists4 : {12, 13}) {
            const int tiles_per_wg = (ctiles + lists4 - 1) / lists4, grid = mtiles * lists4;
            float *o; int *cnt; unsigned long long *spill;
            hipMalloc(&o, (size_t)grid * 256 * 4);
            hipMalloc(&cnt, (size_t)nq * lists4 * 4);
            hipMalloc(&spill, (size_t)nq * lists4 * cap * 8);
            hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_x4_plain), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_x4_sel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            const float thrs[2] = {2.576f * 9.2376f, 1e30f};
            for (int var = 0; var < 3; ++var) {   // plain, select without appends, select at the product's append rate
                auto kern = var == 0 ? gemm_x4_plain : gemm_x4_sel;
                const float thr = var == 2 ? thrs[0] : thrs[1];
                for (int w = 0; w < 20; ++w) {
                    hipMemsetAsync(cnt, 0, (size_t)nq * lists4 * 4, 0);
                    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dq, dc, ctiles, tiles_per_wg, lists4, o, thr, cnt, spill, cap);
                }
                hipDeviceSynchronize();
                if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
                float tot = 0.f;
                for (int it = 0; it < iters; ++it) {
                    hipMemsetAsync(cnt, 0, (size_t)nq * lists4 * 4, 0);
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dq, dc, ctiles, tiles_per_wg, lists4, o, thr, cnt, spill, cap);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1); tot += ms;
                }
                std::vector<int> hc2((size_t)nq * lists4);
                hipMemcpy(hc2.data(), cnt, hc2.size() * 4, hipMemcpyDeviceToHost);
                double appended = 0; for (int v : hc2) appended += v;
                const double flop = 2.0 * nq * (double)n * D * iters;
                printf("TWO work-groups per CU, tile 128 x 256, 4 waves, lists %d (grid %d), %s: %.4f ms per launch, %.0f TFLOP/s (%.3f of 2500), %.2f M appends\n", lists4, grid,
                       var == 0 ? "plain" : (var == 1 ? "select, nothing passes" : "select at the product's append rate"), tot / iters, flop / (tot * 1e-3) / 1e12,
                       flop / (tot * 1e-3) / 1e12 / 2500.0, appended / 1e6);
            }
            hipFree(o); hipFree(cnt); hipFree(spill);
        }
    }
    // 256 x 256 tiles with the register / ballot select and global spill lists (fixed thresholds)
    {
        constexpr int TR = 256;
        const int ctiles = n / TR, mtiles = nq / TQ, cap = 256;
        const int tiles_per_wg = (ctiles + lists - 1) / lists, grid = mtiles * lists;
        const size_t lds = (size_t)2 * (TR * BK * 2 + TQ * BK * 2);
        float *o; int *cnt; unsigned long long *spill;
        hipMalloc(&o, (size_t)grid * 512 * 4);
        hipMalloc(&cnt, (size_t)nq * lists * 4);
        hipMalloc(&spill, (size_t)nq * lists * cap * 8);
        hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_x_sel_2_8), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_x_sel2_2_8), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        // scores of two uniform(-1, 1) vectors of 768 components: sigma = sqrt(768) / 3 = 9.24; pass rates 0.5 % / 0.1 % / none
        const float thrs[3] = {2.576f * 9.2376f, 3.09f * 9.2376f, 1e30f};
        for (int var = 1; var <= 2; ++var)
        for (int rep = 0; rep < 2; ++rep)
            for (int ti = 0; ti < 3; ++ti) {
                auto kern = var == 1 ? gemm_x_sel_2_8 : gemm_x_sel2_2_8;
                for (int w = 0; w < 20; ++w) {
                    hipMemsetAsync(cnt, 0, (size_t)nq * lists * 4, 0);
                    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, dq, dc, ctiles, tiles_per_wg, lists, o, thrs[ti], cnt, spill, cap);
                }
                hipDeviceSynchronize();
                if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
                float tot = 0.f;
                for (int it = 0; it < iters; ++it) {   // (the counter reset is outside the timed interval: the product clears its own in the prep launch)
                    hipMemsetAsync(cnt, 0, (size_t)nq * lists * 4, 0);
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, dq, dc, ctiles, tiles_per_wg, lists, o, thrs[ti], cnt, spill, cap);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1); tot += ms;
                }
                std::vector<int> hc2((size_t)nq * lists);
                hipMemcpy(hc2.data(), cnt, hc2.size() * 4, hipMemcpyDeviceToHost);
                double appended = 0; for (int v : hc2) appended += v;
                const double flop = 2.0 * nq * (double)n * D * iters;
                printf("tile 256 x 256 + register/ballot select, %s, threshold %.3g: %.4f ms per launch, %.0f TFLOP/s (%.3f of 2500), %.2f M appends (%.2f %% of the scores)\n",
                       var == 1 ? "one global spill list per (query, list), slots by atomics" : "eight private sub-lists per (query, list), slot counters in registers", thrs[ti], tot / iters, flop / (tot * 1e-3) / 1e12, flop / (tot * 1e-3) / 1e12 / 2500.0, appended / 1e6, 100.0 * appended / ((double)nq * n));
            }
        hipFree(o); hipFree(cnt); hipFree(spill);
    }
    return 0;
}
