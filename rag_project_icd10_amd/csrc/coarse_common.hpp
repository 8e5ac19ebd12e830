// coarse_common.hpp — geometry, helpers and the fp32 -> fp16 conversion kernel of the fp16 MFMA coarse pass
// (v_mfma_f32_32x32x16_f16 with a fused per-query top-KP; the kernel itself is coarse_flat_kernel.hpp).
//
// The dominant kernel of the hot path: replaces the FLAT/IP scan behind MilvusClient.search
// (services/milvus_service.py:280-285) for query batches. Its output is a candidate list per
// (query, corpus chunk); finalize.hpp certifies and rescoring restores exact fp32 results.
//
// Geometry
//   work-group  = 4 waves, one per SIMD, 128 queries (32 per wave, one query column per lane pair)
//   queries     = B operand, held in registers for the whole sweep (D/16 fragments of 8 halves)
//   corpus rows = A operand, 128-row tiles streamed through an LDS ring by LDS-DMA:
//                 stage = 128 rows x 64 halves (16 KiB), 4 ring slots, 16 one-KiB pieces per stage
//                 (4 per wave), each piece = 8 rows x one full 128-B line
//   swizzle     = LDS slot (row, p) holds 16-B piece p ^ ((row>>1)&7) of the row's 128-B segment:
//                 applied on the DMA SOURCE address and on the ds_read_b128 address (the LDS
//                 destination of an LDS-DMA is lane-linear); A-fragment reads are conflict-free.
#pragma once
#include <type_traits>

#include "topk_select.hpp"

namespace icd {

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E) — keeps every accumulator
// index a constant so the arrays stay in registers
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// v_min_f32 / v_max_f32 as they are: fminf / fmaxf canonicalise both operands first (a v_max x, x in front of every
// operand: 8 instead of 5 instructions per score in the threshold bootstrap); MFMA results need no quieting
__device__ __forceinline__ float raw_min_f32(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float raw_max_f32(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float raw_max3_f32(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

#ifndef ICD_CO_KP
#define ICD_CO_KP 16
#endif
constexpr int CO_KP = ICD_CO_KP;   // candidates kept per (query, list)
constexpr int CO_BM = 128;
constexpr int CO_BN = 128;
constexpr int CO_BK = 64;
constexpr int CO_S = 4;
constexpr int CO_STAGE_BYTES = CO_BN * CO_BK * 2;  // 16384
constexpr int CO_RING_BYTES = CO_S * CO_STAGE_BYTES;
constexpr int CO_CAP = 64;
constexpr int CO_CHECK_EVERY = 8;                       // registers between overflow checks (2 lanes append per register)
constexpr int CO_LIMIT = CO_CAP - 2 * CO_CHECK_EVERY;    // compact a query once it holds more entries than this
constexpr int CO_LDS_BYTES = CO_RING_BYTES + CO_BM * CO_CAP * 8 + 4 * 256;

// ---- fp32 -> fp16 images ---------------------------------------------------------------------------
// One wave per row. The fp16 image of a row is the row times a POWER OF TWO chosen so that its largest component
// lands in [1, 2): scaling by 2^e is exact, so the image keeps fp16's 11 significant bits whatever the magnitude of
// the input (unit-norm embeddings, rows scaled by 1e-30, rows scaled by 1e30 alike) and can neither overflow nor sink
// into fp16's subnormal range as a whole. Queries get a scale per row (mode 0); the corpus gets ONE scale from its
// largest component (mode 2 finds it, mode 1 applies it) because one coarse list holds scores of many rows.
// Coarse scores are therefore in SCALED units 2^(e_query + e_corpus); they are only ever compared with each other and
// with bounds of the same query, and finalize.hpp rescans the candidates from the unscaled fp32 data. The norms
// reported here are norms of the scaled rows (they size the certificate's error bound, also in scaled units).
struct ConvertArgs {
    const float *src;        // [rows][dim]
    _Float16 *dst;           // [rows_pad][dim]
    int rows, rows_pad, dim;
    int mode;                // 0: per-row scale; 1: fixed scale 2^fixed_exp; 2: scan only (amax_bits, any_bad)
    int fixed_exp;
    float *norm;             // nullable [rows]: L2 norm of the scaled row, rounded up
    int *scale_exp;          // nullable [rows]: e of the row (mode 0)
    unsigned char *bad;      // nullable [rows]: 1 = non-finite input
    unsigned int *rmax_bits; // nullable: atomicMax of the scaled norm's bits (norm >= 0)
    unsigned int *amax_bits; // nullable: atomicMax of max |x| bits over all rows (mode 2)
    unsigned int *any_bad;   // nullable: set to 1 if any row is bad
    unsigned int *zero_u32;  // nullable [rows_pad]: cleared (the coarse pass's shared per-query thresholds)
    unsigned int *zero_u32b; // nullable [rows_pad]: cleared (the same of the second coarse pass)
    int *zero_i32;           // nullable: four words cleared by the launch (the fallback counters of the search it starts)
    long long perm_mul;      // with perm_mod > 0: dst row p holds src row (p * perm_mul) mod perm_mod (an affine permutation)
    int perm_mod;
    const float *mu;         // nullable [dim]: subtracted from every row before anything else (the corpus's column mean, see below)
    float *norm_raw_max;     // nullable: atomicMax (as bits) of the L2 norm of the UNcentred, unscaled row (rounded up)
};

// ---- the corpus's fp16 image is CENTRED when its rows share a large common component ------------------------------------------
// Sentence embeddings are anisotropic: every row has a large common component (mean pairwise cosine 0.3 - 0.9; the synthetic
// encoder of this repo 0.98). q.c_r = q.(c_r - mu) + q.mu and q.mu is the same for every row of one query, so the ORDER of a
// query's hits - all the coarse pass decides - is that of q.(c_r - mu). The fp16 rounding error of the coarse score is
// relative to the norm of what is rounded: on centred rows it shrinks by |c - mu| / |c| (8 x at cosine 0.98), and with it
// the certificate's window. icd_index_create computes mu (two deterministic reduction launches), centres when
// |mu|^2 >= CENTER_MIN_SHARE of the mean squared row norm (Gaussian or family-shaped test corpora have mu ~ 0 and stay as
// they are, bit for bit), and finalize adds the one error term that does NOT shrink: the fp32 chain of the canonical score
// runs on the uncentred rows (finalize.hpp, eps).
constexpr float CENTER_MIN_SHARE = 0.25f;

// column sums of a row-major [rows][dim] matrix: block b sums rows b, b + grid, ... into part[b][dim] (fixed order), then
// ONE block adds the partials in order: the same bits on every run
__global__ __launch_bounds__(256) void column_sum_partial_kernel(const float *src, int rows, int dim, float *part, float *sq_part) {
    float sq = 0.0f;
    for (int d0 = threadIdx.x; d0 < dim; d0 += 256) {
        float acc = 0.0f;
        for (int r = blockIdx.x; r < rows; r += gridDim.x) {
            const float v = src[(size_t)r * dim + d0];
            acc += v;
            sq = __builtin_fmaf(v, v, sq);
        }
        part[(size_t)blockIdx.x * dim + d0] = acc;
    }
    __shared__ float red[256];
    red[threadIdx.x] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
        for (int i = 0; i < 256; ++i) t += red[i];
        sq_part[blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(256) void column_sum_final_kernel(const float *part, const float *sq_part, int nblocks, int rows, int dim, float *mu, float *out2) {
    for (int d0 = threadIdx.x; d0 < dim; d0 += 256) {
        float acc = 0.0f;
        for (int b = 0; b < nblocks; ++b) acc += part[(size_t)b * dim + d0];
        mu[d0] = acc / (float)rows;
    }
    __syncthreads();
    if (threadIdx.x == 0) {   // out2[0] = |mu|^2, out2[1] = mean squared row norm
        float m2 = 0.0f, sq = 0.0f;
        for (int d = 0; d < dim; ++d) m2 = __builtin_fmaf(mu[d], mu[d], m2);
        for (int b = 0; b < nblocks; ++b) sq += sq_part[b];
        out2[0] = m2;
        out2[1] = sq / (float)rows;
    }
}

// e with m * 2^e in [1, 2) for finite m > 0 (subnormals included); 0 otherwise
__host__ __device__ inline int scale_exp_for(float m) {
    if (!(m > 0.0f) || !(m <= 3.402823466e38f)) return 0;
    int ex;
    (void)frexpf(m, &ex);   // m = f * 2^ex, f in [0.5, 1)
    return 1 - ex;
}

__global__ __launch_bounds__(256) void convert_rows_kernel(ConvertArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (a.zero_i32 && blockIdx.x == 0 && threadIdx.x < 4) a.zero_i32[threadIdx.x] = 0;
    if (row >= a.rows_pad) return;
    if (a.zero_u32 && lane == 0) a.zero_u32[row] = 0u;
    if (a.zero_u32b && lane == 0) a.zero_u32b[row] = 0u;
    _Float16 *d = a.dst + (size_t)row * a.dim;
    if (row >= a.rows) {
        if (a.mode != 2)
            for (int i = lane * 4; i < a.dim; i += 256) {
                d[i] = (_Float16)0.0f; d[i + 1] = (_Float16)0.0f; d[i + 2] = (_Float16)0.0f; d[i + 3] = (_Float16)0.0f;
            }
        return;
    }
    const int srow = a.perm_mod > 0 ? (int)(((long long)row * a.perm_mul) % a.perm_mod) : row;
    const float *s = a.src + (size_t)srow * a.dim;
    // the row is read ONCE and stays in registers (the fp16 path exists for dim 768 and 1024 only: at most four float4 per
    // lane); the second pass over it used to be a second round trip
    constexpr int NV = 4;
    float4 v[NV];
    float m = 0.0f, raw_ss = 0.0f;
    bool bad = false;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const int i = lane * 4 + 256 * t;
        v[t] = i < a.dim ? *reinterpret_cast<const float4 *>(s + i) : float4{0.f, 0.f, 0.f, 0.f};
        if (a.norm_raw_max) raw_ss = __builtin_fmaf(v[t].x, v[t].x, __builtin_fmaf(v[t].y, v[t].y, __builtin_fmaf(v[t].z, v[t].z, __builtin_fmaf(v[t].w, v[t].w, raw_ss))));
        if (a.mu && i < a.dim) {
            const float4 m4 = *reinterpret_cast<const float4 *>(a.mu + i);
            v[t].x -= m4.x; v[t].y -= m4.y; v[t].z -= m4.z; v[t].w -= m4.w;
        }
        const float f[4] = {v[t].x, v[t].y, v[t].z, v[t].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bad |= !(fabsf(f[j]) <= 3.402823466e38f);   // NaN and infinities
            m = fmaxf(m, fabsf(f[j]));
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    const bool anybad = __any(bad);
    if (a.norm_raw_max) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) raw_ss += __shfl_xor(raw_ss, off);
        if (lane == 0 && !anybad) atomicMax(reinterpret_cast<unsigned int *>(a.norm_raw_max), __float_as_uint(sqrtf(raw_ss) * 1.000001f));
    }
    if (a.mode == 2) {
        if (lane == 0) {
            if (!anybad && a.amax_bits) atomicMax(a.amax_bits, __float_as_uint(m));
            if (a.any_bad && anybad) atomicOr(a.any_bad, 1u);
        }
        return;
    }
    const int e = anybad ? 0 : (a.mode == 1 ? a.fixed_exp : scale_exp_for(m));
    float ss = 0.0f;
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const int i = lane * 4 + 256 * t;
        if (i < a.dim) {
            const float f[4] = {v[t].x, v[t].y, v[t].z, v[t].w};
            half4 h;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = ldexpf(f[j], e);   // exact (a component 2^126 below the row's largest flushes: far below fp16's grid)
                ss = __builtin_fmaf(x, x, ss);
                h[j] = (_Float16)x;
            }
            *reinterpret_cast<half4 *>(d + i) = h;   // one 8-byte store per lane
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off);
    if (lane == 0) {
        float nrm = sqrtf(ss) * 1.000001f;  // round up: it multiplies an error bound
        if (!(nrm == nrm)) nrm = INFINITY;
        if (a.norm) a.norm[row] = nrm;
        if (a.scale_exp) a.scale_exp[row] = e;
        if (a.bad) a.bad[row] = anybad ? 1 : 0;
        if (a.rmax_bits) atomicMax(a.rmax_bits, __float_as_uint(nrm));
        if (a.any_bad && anybad) atomicOr(a.any_bad, 1u);
    }
}

}  // namespace icd
