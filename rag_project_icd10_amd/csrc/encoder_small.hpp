// encoder_small.hpp — the BERT-style sentence encoder for SMALL inputs: the reference's own call shape of the embed step,
// EmbeddingService.encode_query / encode_single (services/embedding_service.py:97-120: ONE string per
// SentenceTransformer.encode call) and the handful of diagnoses of one /query request.
//
// At 10-100 tokens the forward is ~85 MFLOP per token against 340 MB of fp32 weights: what sets its latency is not
// arithmetic but the NUMBER of kernels and the dependent memory round trips inside each (the framework's forward is ~220
// kernels of ~5 us: 1.1 ms replayed from a graph; a kernel that loads its arguments, one operand and stores takes 3-5 us on
// this part however little it computes, and a boundary 1.2-1.5: MI355X_MICROARCH.md, price list - a grid barrier inside
// one launch costs MORE than a boundary, 4-7 us, so this is not a persistent kernel). This file is that forward as FIVE
// launches per layer, activations and accumulation in fp32, the LayerNorms folded into their consumers; the GEMMs' products in one
// of two arithmetics chosen at create (icd_encoder_desc.arithmetic): fp32-input MFMAs (round 5), or - the default since round 6 -
// the split-bf16 form further down (three bf16 MFMAs per 32 k-values, ~1e-6 off the fp32 forward, a fifth of the matrix time):
//
//   enc_linear_kernel   Y = act(A W^T + b) (+ R): one work-group per NT (16 or 8) output columns and 16 tokens (grid.y: the token
//                       tiles of a longer input side by side), its four waves split K (192 columns each; 256 at hidden 1 024). A
//                       wave issues ALL its loads up front - its NT x 192 slice of W (12 x 16 B per lane: the whole slice in
//                       flight at once) and its tokens' operand rows - so that the kernel is one memory round trip deep, then
//                       runs v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate: exact products; the sum's order differs from the
//                       vendor GEMM's like any two GEMMs differ) or v_mfma_f32_16x16x32_bf16 on the split operands (BF); the waves'
//                       partial sums meet in LDS, where bias, erf-GELU and
//                       the residual are applied. Operands are STORED in the order these loads want them (enc_pa / enc_pw);
//                       the FFN-down GEMM splits K over four work-groups whose partial sums its readers add up (slabs).
//       LNPRO           A = LayerNorm(X) without touching X: LN(x) W^T + b = rstd (x (W diag g)^T - mean c1) + c2 with c1, c2
//                       and W diag(g) made once (enc_fold_ln_kernel). The MFMAs run on the PRE-norm rows of the previous
//                       sublayer the moment they land; mean and 1/std of a row come from the same registers (the work-group's
//                       waves hold whole rows between them), VALU work beside the MFMAs, and meet the sums behind the ONE
//                       barrier of the kernel. Work-group 0 leaves the two numbers per token for ...
//       EPI == 2        ... the residual of the next GEMM (BertSelfOutput / BertOutput: dense(x) + LayerNorm-ed input),
//                       rebuilt from the pre-norm row and those two numbers. No normalised activation is ever stored.
//   enc_attention_kernel  softmax(q K^T / 8) V, one wave per (token, head), lane = head dimension: every key / value row one
//                       coalesced 256-B load, a score one wave-wide sum (DPP), the softmax on wave-uniform numbers.
//   enc_embed_kernel    (word + type) + position, pre-norm.        enc_pool_kernel   the last LayerNorm, masked-mean (or
//                       [CLS]) pooling and L2 normalisation, one work-group per sequence.
//
// Token ids, positions and sequence bounds come from a small descriptor in device memory (layout below) that the host
// fills per call: the launches themselves depend only on the token BUCKET (16 / 32 / 64 / 128 / 256 / 512), so a forward is one replay
// of a captured graph (icd_encoder.hpp). Arithmetic restated from transformers' BertModel - the published architecture
// the reference reaches through sentence-transformers; the checkpoint's numerics are unpinned (no weights offline,
// DESIGN.md section 7): the tests compare with the framework's fp32 forward of the same weights (1e-5).
#pragma once
#include <hip/hip_runtime.h>
#include "attention_kernel.hpp"

namespace icd {

constexpr int ENC_TMAX = 512;   // packed tokens per call (any ONE sequence a BERT-style encoder takes fits: max_position_embeddings 512)
constexpr int ENC_BMAX = 64;    // sequences per call
// the batch form of the same forward (encoder_big.hpp: the GEMMs in large tiles, everything else these kernels): tokens / sequences per pass
constexpr int ENC_BIG_TMAX = 8192;
constexpr int ENC_BIG_BMAX = 2048;
// descriptor (int32 words): [0] T, [1] B, then per token (TMAX each): id, position, first row of its sequence, length of its
// sequence (0 past the call's tokens), then BMAX + 1 sequence starts (starts[b] = T for b >= B)
template <int TMAX, int BMAX>
struct EncMeta {
    static constexpr int IDS = 2, POS = 2 + TMAX, TOK_R0 = 2 + 2 * TMAX, TOK_LEN = 2 + 3 * TMAX, STARTS = 2 + 4 * TMAX;
    static constexpr int WORDS = STARTS + BMAX + 1;
    static constexpr int T_MAX = TMAX, B_MAX = BMAX;
};
using EncMetaSmall = EncMeta<ENC_TMAX, ENC_BMAX>;
using EncMetaBig = EncMeta<ENC_BIG_TMAX, ENC_BIG_BMAX>;
constexpr int ENC_META_WORDS = EncMetaSmall::WORDS;

typedef float enc_f32x4 __attribute__((ext_vector_type(4)));

// ---- operand layouts -----------------------------------------------------------------------------------------------------
// v_mfma_f32_16x16x4_f32 wants element [row r][k] of its A (and column r of its B) operand in lane (r, kq = k & 3 of the
// step). A wave that loads those straight from a row-major matrix touches 16 rows x 64 B per instruction: 16 half cache
// lines, ~40 cycles of the CU's address unit each - 24 such loads per wave were 2.2 of a GEMM kernel's 7.7 us (clock stamps,
// profiles/r05_encoder_small.log). So every matrix a GEMM reads is STORED in the order its loads want it:
//   activations [T][K] (K = 192 x waves): element (t, k) at ((((t / 16 * K / 192 + w) * 12 + i) * 4 + kq) * 16 + t % 16) * 4 + c
//       with w = k / 192, i = k % 192 / 16, kq = k % 16 / 4, c = k % 4      (192 = the columns a wave takes; 256 at hidden 1 024)
//       - lane (r, kq) of wave w reads step i of token tile t / 16 at  base + (i * 64 + lane) * 16 bytes: ONE KB, contiguous;
//   weights [N][K] in tiles of NT rows: the same with NT in place of 16 and the row tile n / NT in place of t / 16.
// Producers (the embedding sum, the GEMM epilogues, the attention) scatter their few values per thread into that order.
// KW = the columns of K a wave takes: 192 for hidden 768 / inter 3 072 (12 k-steps of 16), 256 for hidden 1 024 / inter 4 096 (16)
__host__ __device__ __forceinline__ size_t enc_pa(int t, int k, int K, int KW) {
    const int w = k / KW, kk = k - w * KW;
    return ((((size_t)(t >> 4) * (K / KW) + w) * (KW >> 4) + (kk >> 4)) * 4 + ((kk >> 2) & 3)) * 64 + (size_t)(t & 15) * 4 + (kk & 3);
}
__host__ __device__ __forceinline__ size_t enc_pw(int n, int k, int K, int NT, int KW) {
    const int w = k / KW, kk = k - w * KW;
    return ((((size_t)(n / NT) * (K / KW) + w) * (KW >> 4) + (kk >> 4)) * 4 + ((kk >> 2) & 3)) * (size_t)(NT * 4) + (size_t)(n % NT) * 4 + (kk & 3);
}

// one-time: a torch.nn.Linear weight [N][K] row-major -> the NT-row-tile order above (16 bytes per thread), its columns
// optionally scaled by a LayerNorm weight (LNPRO below: W' = W diag(g))
__global__ __launch_bounds__(256) void enc_permute_w_kernel(const float *src, const float *colscale, float *dst, int N, int K, int NT, int KW) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;   // float4 index in the source
    if (g >= (size_t)N * K / 4) return;
    const int n = (int)(g / (K / 4)), k = (int)(g % (K / 4)) * 4;
    float4 v = *reinterpret_cast<const float4 *>(src + (size_t)n * K + k);
    if (colscale) {
        const float4 sc = *reinterpret_cast<const float4 *>(colscale + k);
        v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w;
    }
    *reinterpret_cast<float4 *>(dst + enc_pw(n, k, K, NT, KW)) = v;
}
// ---- the split-bf16 arithmetic (ENC_ARITH_BF16X3) ------------------------------------------------------------------------------
// The fp32-input MFMA runs at 1 / 16 of the bf16 rate. With x = x_hi + x_lo and w = w_hi + w_lo in bf16 (x_lo = bf16(x - x_hi): 16
// mantissa bits each way) a block of 32 k-values is THREE v_mfma_f32_16x16x32_bf16 into one fp32 accumulator, in this order:
//   x_hi w_hi, x_hi w_lo, x_lo w_hi          (the dropped x_lo w_lo is 2^-18 relative; products exact in fp32, fp32 accumulation)
// - 48 cycles of the matrix pipe where eight fp32 MFMAs take 256. Weights are split ONCE (enc_permute_w_bf16_kernel), activations
// in registers on their way into the MFMA (enc_split8: the same function in both forms of the GEMM - same bits). The operand
// ORDER is the fp32 form's: the lane's two 16-byte slots of a 32-block hold k = 32 b + 4 kq + c and 32 b + 16 + 4 kq + c (c = 0 .. 3);
// the MFMA's eight slots per lane take them in that order for A and B alike (which k sits in which slot does not matter to a dot
// product as long as both operands agree). For W the two slots hold the block's eight w_hi and its eight w_lo.
typedef __bf16 enc_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void enc_split8(const float4 &p, const float4 &q, enc_bf16x8 &hi, enc_bf16x8 &lo) {
#pragma clang fp contract(off)
    const float v[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h = (__bf16)v[e];            // round to nearest even
        hi[e] = h;
        lo[e] = (__bf16)(v[e] - (float)h);
    }
}
__device__ __forceinline__ enc_bf16x8 enc_as_bf16x8(const float4 &v) { return __builtin_bit_cast(enc_bf16x8, v); }
// one-time: a Linear weight [N][K] row-major (columns optionally scaled: W' = W diag(g)) -> per (row, 32-block, kq) the eight w_hi
// in the slot of k-step 2 b and the eight w_lo in the slot of k-step 2 b + 1 of the NT-row-tile order; one thread per (n, b, kq)
__global__ __launch_bounds__(256) void enc_permute_w_bf16_kernel(const float *src, const float *colscale, float *dst, int N, int K, int NT, int KW) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= (size_t)N * K / 8) return;
    const int n = (int)(g / (K / 8)), r = (int)(g % (K / 8)), b = r >> 2, kq = r & 3;
    const int k0 = 32 * b + 4 * kq;
    float4 p = *reinterpret_cast<const float4 *>(src + (size_t)n * K + k0), q = *reinterpret_cast<const float4 *>(src + (size_t)n * K + k0 + 16);
    if (colscale) {
        const float4 s0 = *reinterpret_cast<const float4 *>(colscale + k0), s1 = *reinterpret_cast<const float4 *>(colscale + k0 + 16);
        p.x *= s0.x; p.y *= s0.y; p.z *= s0.z; p.w *= s0.w; q.x *= s1.x; q.y *= s1.y; q.z *= s1.z; q.w *= s1.w;
    }
    enc_bf16x8 hi, lo;
    enc_split8(p, q, hi, lo);
    *reinterpret_cast<enc_bf16x8 *>(dst + enc_pw(n, k0, K, NT, KW)) = hi;
    *reinterpret_cast<enc_bf16x8 *>(dst + enc_pw(n, k0 + 16, K, NT, KW)) = lo;
}

// one-time, for a Linear that reads a LayerNorm's output, LN(x) W^T + bias = rstd (x (W diag g)^T - mean c1) + c2 with
//   c1[n] = sum_k g[k] W[n][k]        c2[n] = sum_k b[k] W[n][k] + bias[n]          (sums in double); one wave per n
__global__ __launch_bounds__(256) void enc_fold_ln_kernel(const float *w, const float *g, const float *b, const float *bias, float *c1, float *c2, int N, int K) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    double s1 = 0.0, s2 = 0.0;
    for (int k = lane; k < K; k += 64) {
        const double wv = (double)w[(size_t)n * K + k];
        s1 += (double)g[k] * wv;
        s2 += (double)b[k] * wv;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
    if (lane == 0) { c1[n] = (float)s1; c2[n] = (float)(s2 + (double)bias[n]); }
}

__device__ __forceinline__ float enc_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// ---- the arithmetic BOTH forms of a GEMM share (enc_linear_kernel below; enc_linear_big_kernel, encoder_big.hpp) ----------------
// One embedding arithmetic whatever the call shape: the batch form must give every token the BITS the
// small-input form gives it. The products and sums of the MFMAs are fixed by the instruction order both kernels spell out;
// everything around them lives in these functions, with contraction OFF (every rounding is the one written here, whatever
// code surrounds the inlined body) and the two explicit fmaf where a fused step is meant.
// statistics of ONE wave's slice of a row (KW = 16 ITER columns): lane (r16, kq) holds ITER float4 of row r16; returns the
// slice's mean and centred sum of squares in every lane of the row
template <int ITER>
__device__ __forceinline__ void enc_piece_stats(const float4 (&areg)[ITER], float &mw, float &qw) {
#pragma clang fp contract(off)
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < ITER; ++i) sm += (areg[i].x + areg[i].y) + (areg[i].z + areg[i].w);
    sm += __shfl_xor(sm, 16);
    sm += __shfl_xor(sm, 32);
    mw = sm * (1.0f / (16 * ITER));
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
        const float dx = areg[i].x - mw, dy = areg[i].y - mw, dz = areg[i].z - mw, dw = areg[i].w - mw;
        q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    qw = q;
}
// the slices of a row combined (equal counts: mean = the average of the slices' means, M2 = sum of their M2 + KW sum (mean_w - mean)^2);
// mw / qw: the nw slices' numbers, `stride` floats apart
__device__ __forceinline__ void enc_combine_stats(const float *mw, const float *qw, int stride, int nw, int kw_cols, int K, float eps, float &mean, float &rstd) {
#pragma clang fp contract(off)
    float m = 0.f;
    for (int w = 0; w < nw; ++w) m += mw[(size_t)w * stride];
    m /= (float)nw;
    float m2 = 0.f;
    for (int w = 0; w < nw; ++w) { const float dm = mw[(size_t)w * stride] - m; m2 += qw[(size_t)w * stride] + (float)kw_cols * dm * dm; }
    mean = m;
    rstd = 1.0f / sqrtf(m2 / (float)K + eps);
}
// what becomes of a K slice's sum `s`: LNPRO rstd (s - mean c1); in the FIRST slice (slab 0) + bias (+ the LayerNorm-ed residual,
// EPI == 2); erf-GELU (EPI == 1)
template <int EPI, bool LNPRO>
__device__ __forceinline__ float enc_epilogue(float s, bool first_slice, float mean, float rstd, float c1, float bias, float rsrc, float rmean, float rrstd, float rg, float rb) {
#pragma clang fp contract(off)
    if constexpr (LNPRO) s = rstd * (s - mean * c1);
    if (first_slice) {
        s += bias;
        if constexpr (EPI == 2) s += (rsrc - rmean) * rrstd * rg + rb;
    }
    if constexpr (EPI == 1) s = 0.5f * s * (1.0f + erff(s * 0.70710678118654752440f));
    return s;
}

struct EncEmbedArgs {
    const int *meta;
    const float *word, *pos, *type0;   // embeddings [vocab][H], [max_pos][H], token type 0 [H]
    int H, KW;
    float *y;   // [TMAX][H] pre-norm, operand order
};
// y[t] = (word[id] + type0) + pos[p] (BertEmbeddings in front of its LayerNorm): one wave per token
template <int NV, typename M = EncMetaSmall>
__global__ __launch_bounds__(256) void enc_embed_kernel(EncEmbedArgs a) {
    const int lane = threadIdx.x & 63, t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= a.meta[0]) return;   // wave-uniform
    const size_t id = (size_t)a.meta[M::IDS + t], p = (size_t)a.meta[M::POS + t];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = 4 * (lane + 64 * j);
        const float4 w = *reinterpret_cast<const float4 *>(a.word + id * a.H + c);
        const float4 ty = *reinterpret_cast<const float4 *>(a.type0 + c);
        const float4 pe = *reinterpret_cast<const float4 *>(a.pos + p * a.H + c);
        float4 v;
        v.x = (w.x + ty.x) + pe.x; v.y = (w.y + ty.y) + pe.y; v.z = (w.z + ty.z) + pe.z; v.w = (w.w + ty.w) + pe.w;
        *reinterpret_cast<float4 *>(a.y + enc_pa(t, c, a.H, a.KW)) = v;
    }
}

struct EncLinearArgs {
    const int *meta;
    const float *x;        // [TMAX][K] operand order: the A operand, or (LNPRO) the pre-norm rows it is the LayerNorm of
    float ln_eps;          // LNPRO: the LayerNorm's epsilon; its weight and bias are folded into w / c1 / c2 (enc_fold_ln_kernel)
    const float *c1;       // LNPRO: [N]
    float *stats_out;      // LNPRO: [TMAX][2] mean, 1 / sqrt(var + eps) of every token (written by work-group 0)
    const float *w;        // [N][K] in NT-row tiles (enc_pw); LNPRO: its columns scaled by the LayerNorm weight
    const float *bias;     // [N]; LNPRO: c2
    // EPI == 2: + LayerNorm(res_src)[t][n], rebuilt from the pre-norm row and the statistics a LNPRO kernel left
    const float *res_src;  // [TMAX][N] operand order
    const float *res_stats, *res_g, *res_b;
    float *y;              // [TMAX][N]: operand order (OUT_PA) or row-major
    int K, N;
    // K split over work-groups (the FFN-down GEMM: K = 3072 in four work-groups of four waves): work-group b takes output tile
    // b / (nwk / waves) and K slice b % (nwk / waves) and writes its partial sums to slab `slice` of y; bias and residual go
    // into slab 0. Whoever reads such a matrix adds the slabs up (NSLAB below, res_nslab, EncPoolArgs::nslab): a split
    // costs its readers three more loads per value and no launch, no atomics, no second pass.
    int nwk;               // waves over the whole K (K / 192)
    long long slab;        // floats between the slabs of x / res_src / y
    int res_nslab;         // slabs of res_src
    unsigned long long *stamps;   // diagnostic builds (ICD_ABLATE): 8 x (s_memtime, s_memrealtime) of wave 0 of work-group 1; nullptr = none
};
#ifdef ICD_ABLATE
__device__ unsigned long long g_enc_first[8];   // diagnostic: clocks at the first instruction of a wave (no kernel argument has been read yet)
#define ENC_STAMP(i) do { if (a.stamps && blockIdx.x == 1 && tid == 0) { __builtin_amdgcn_sched_barrier(0); \
    a.stamps[2 * (i)] = __builtin_amdgcn_s_memtime(); a.stamps[2 * (i) + 1] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define ENC_STAMP_DRAIN() do { if (a.stamps) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); } while (0)
#else
#define ENC_STAMP(i) do { } while (0)
#define ENC_STAMP_DRAIN() do { } while (0)
#endif
// NT: output columns per work-group (grid.x = N / NT x K slices, grid.y = token tiles of 16: the tiles of a longer input run
// side by side on other CUs, each re-reading its slice of W from the caches); a wave takes 16 ITER columns of K (ITER = 12: hidden
// 768, 16: hidden 1 024), the block has at most MAXW waves;
// EPI: 0 bias, 1 bias + erf-GELU (BertIntermediate), 2 bias + LayerNorm-ed residual; OUT_PA: y in operand order
// NSLAB: the A operand is the sum of this many slabs
// BF: the split-bf16 arithmetic (three bf16 MFMAs per 32-block; w holds w_hi / w_lo) instead of eight fp32 MFMAs
template <int ITER, int NT, int EPI, bool LNPRO, bool OUT_PA, int MAXW, int NSLAB, bool BF = false>
__global__ __launch_bounds__(MAXW * 64) void enc_linear_kernel(EncLinearArgs a) {
    static_assert(NT == 16 || NT == 8 || NT == 4, "columns per work-group");
    constexpr int KW = 16 * ITER;   // this wave's columns of K
    __shared__ float red[16][256];
    __shared__ float lnred[2][16][16];   // LNPRO: per wave and row, the slice's mean and centred sum of squares
    const int tid = threadIdx.x, lane = tid & 63;
#ifdef ICD_ABLATE
    if (blockIdx.x == 1 && tid == 0) {
        const int slot = (LNPRO ? (EPI == 0 ? 0 : 2) : (NT == 8 ? 1 : 3)) * 2;   // QKV, attention output, FFN up, FFN down
        g_enc_first[slot] = __builtin_amdgcn_s_memtime(); g_enc_first[slot + 1] = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_sched_barrier(0);
    }
#endif
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = (int)blockDim.x >> 6;
    ENC_STAMP(0);
    const int ksn = a.nwk / nw;                                   // K slices (1: the work-group's waves cover K)
    const int ntile = (int)blockIdx.x / ksn, ks = (int)blockIdx.x - ntile * ksn;
    const int kw = ks * nw + wave;                                // this wave's 192 columns of K
    const int n0 = ntile * NT;
    const int r16 = lane & 15, kq = lane >> 4;
    // ---- everything this wave will read, issued before anything waits ------------------------------------------------------
    float4 wreg[ITER];   // W rows n0 .. n0 + NT - 1 (lanes r16 >= NT: zero columns of the MFMA's B operand)
    {
        const float *wp = a.w + (((size_t)ntile * a.nwk + kw) * ITER * 4 + kq) * (NT * 4) + (size_t)(r16 < NT ? r16 : 0) * 4;
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            const float4 v = *reinterpret_cast<const float4 *>(wp + (size_t)i * (4 * NT * 4));
            wreg[i] = r16 < NT ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float4 areg[ITER];   // rows t0 .. t0 + 15 of the operand (rows past T inside the last tile: finite stale values, never stored)
    auto load_tile = [&](float4 (&dst)[ITER], int tile) {
        const float *xp = a.x + (((size_t)tile * a.nwk + kw) * ITER * 64 + lane) * 4;
#pragma unroll
        for (int i = 0; i < ITER; ++i) dst[i] = *reinterpret_cast<const float4 *>(xp + (size_t)i * 256);
        if constexpr (NSLAB > 1) {
#pragma unroll
            for (int sl = 1; sl < NSLAB; ++sl) {
#pragma unroll
                for (int i = 0; i < ITER; ++i) {
                    const float4 v = *reinterpret_cast<const float4 *>(xp + (size_t)sl * a.slab + (size_t)i * 256);
                    dst[i].x += v.x; dst[i].y += v.y; dst[i].z += v.z; dst[i].w += v.w;
                }
            }
        }
    };
    auto load_res = [&](int t, float &src, float &mean, float &rstd) {   // residual operands of (token t, this thread's column)
        const size_t idx = enc_pa(t, n0 + (tid & 15), a.N, KW);
        float v = a.res_src[idx];
        for (int sl = 1; sl < a.res_nslab; ++sl) v += a.res_src[idx + (size_t)sl * a.slab];
        src = v; mean = a.res_stats[2 * t]; rstd = a.res_stats[2 * t + 1];
    };
    const int t0 = (int)blockIdx.y * 16;   // this work-group's 16 tokens (grid.y = token tiles of the bucket)
    load_tile(areg, (int)blockIdx.y);
    // the epilogue's operands of thread tid < 256: output (token 4 (l >> 4) + j, column l & 15), l = tid & 63, j = tid >> 6
    const int ej = tid >> 6, el = tid & 63;
    const int en = el & 15, et = 4 * (el >> 4) + ej;
    const bool ecol = tid < 256 && en < NT;
    float bias_v = 0.f, c1_v = 0.f, rg = 0.f, rb = 0.f, rsrc = 0.f, rmean = 0.f, rrstd = 0.f;
    if (ecol) {
        bias_v = a.bias[n0 + en];
        if constexpr (LNPRO) c1_v = a.c1[n0 + en];
        if constexpr (EPI == 2) {
            rg = a.res_g[n0 + en]; rb = a.res_b[n0 + en];
            load_res(t0 + et, rsrc, rmean, rrstd);
        }
    }
    const int T = a.meta[0];
    __builtin_amdgcn_sched_barrier(0);   // (left alone, hipcc sinks the loads to their uses: one round trip per four MFMAs instead of one in all)
    ENC_STAMP(1);        // loads issued
    ENC_STAMP_DRAIN();
    ENC_STAMP(2);        // loads landed

    if (t0 >= T) return;   // (work-group-uniform, before any barrier: a token tile past the call's tokens; its loads were harmless)
    {
        enc_f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};   // (two chains: a dependent MFMA waits out the one before it)
        if constexpr (BF) {
            // 32-blocks b = 0, 1, ...: even ones into c0, odd ones into c1, each block hi hi, hi lo, lo hi (two blocks interleaved)
            static_assert(ITER % 4 == 0, "pairs of 32-blocks");
#pragma unroll
            for (int i = 0; i < ITER; i += 4) {
                enc_bf16x8 ah0, al0, ah1, al1;
                enc_split8(areg[i], areg[i + 1], ah0, al0);
                enc_split8(areg[i + 2], areg[i + 3], ah1, al1);
                const enc_bf16x8 wh0 = enc_as_bf16x8(wreg[i]), wl0 = enc_as_bf16x8(wreg[i + 1]), wh1 = enc_as_bf16x8(wreg[i + 2]), wl1 = enc_as_bf16x8(wreg[i + 3]);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah0, wh0, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah1, wh1, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah0, wl0, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah1, wl1, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al0, wh0, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al1, wh1, c1, 0, 0, 0);
            }
        } else
#pragma unroll
        for (int i = 0; i < ITER; i += 2) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[i].x, wreg[i].x, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[i + 1].x, wreg[i + 1].x, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[i].y, wreg[i].y, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[i + 1].y, wreg[i + 1].y, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[i].z, wreg[i].z, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[i + 1].z, wreg[i + 1].z, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[i].w, wreg[i].w, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[i + 1].w, wreg[i + 1].w, c1, 0, 0, 0);
        }
        float mw = 0.f, qw = 0.f;
        // LayerNorm without touching the operand: the MFMAs above ran on the RAW rows against W diag(g); what is left is
        // rstd (acc - mean c1) + c2 per output (epilogue). The statistics: the work-group's waves hold rows t0 .. t0 + 15
        // whole between them (lane (r16, kq) of wave w: 48 values of row r16); every wave reduces ITS 192 columns of a row
        // to (mean, centred sum of squares) - two passes over registers, VALU work that runs beside the MFMAs - and the
        // pairs are combined behind the barrier the partial sums need anyway.
        if constexpr (LNPRO) enc_piece_stats<ITER>(areg, mw, qw);
        ENC_STAMP(3);    // MFMAs issued, statistics done
        // C[token 4 (lane >> 4) + j][column lane & 15] in register j
#pragma unroll
        for (int j = 0; j < 4; ++j) red[wave][j * 64 + lane] = c0[j] + c1[j];
        if constexpr (LNPRO) {
            if (kq == 0) { lnred[0][wave][r16] = mw; lnred[1][wave][r16] = qw; }
        }
        ENC_STAMP(4);    // partial sums in LDS (the writes wait for the MFMAs)
        __syncthreads();
        ENC_STAMP(5);    // barrier passed
        if (ecol) {
            float s = 0.f;
            for (int w = 0; w < nw; ++w) s += red[w][tid];
            float mean = 0.f, rstd = 0.f;
            if constexpr (LNPRO) {
                enc_combine_stats(&lnred[0][0][et], &lnred[1][0][et], 16, nw, 16 * ITER, a.K, a.ln_eps, mean, rstd);
                if (blockIdx.x == 0 && en == 0 && t0 + et < T) {
                    a.stats_out[2 * (t0 + et)] = mean;
                    a.stats_out[2 * (t0 + et) + 1] = rstd;
                }
            }
            s = enc_epilogue<EPI, LNPRO>(s, ks == 0, mean, rstd, c1_v, bias_v, rsrc, rmean, rrstd, rg, rb);   // (ks: work-group-uniform - bias and residual once, in slab 0)
            if (t0 + et < T) {
                float *yo = a.y + (size_t)ks * a.slab;
                if constexpr (OUT_PA) yo[enc_pa(t0 + et, n0 + en, a.N, KW)] = s;
                else yo[(size_t)(t0 + et) * a.N + n0 + en] = s;
            }
        }
        ENC_STAMP_DRAIN();
        ENC_STAMP(6);    // epilogue stored
    }
}

// one chunk of at most CH keys of the flash recurrence of ONE (token, head): q = the lane's scaled query component, kreg / vreg =
// the lane's component of the chunk's keys / values; the first Lc are live (contraction off: the roundings are the ones written here)
// e^x of the softmax as ONE v_exp_f32 (2^(x log2 e): the product's rounding moves the exponent by |x| 2^-24 - a weight e^x is
// off by ~1e-6 of ITSELF where it is ~e^-20 of the row's largest, by nothing where it matters; exp(-inf) = 0). ocml's expf is a
// dozen instructions per key, a third of what this kernel costs in the batch form, where it is bound by its VALU work.
__device__ __forceinline__ float enc_exp(float x) {
#pragma clang fp contract(off)
    return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
}
template <int CH>
__device__ __forceinline__ void enc_attn_chunk(float q, const float (&kreg)[CH], const float (&vreg)[CH], int Lc, float &m_run, float &l_run, float &o_run) {
#pragma clang fp contract(off)
    // (keys in groups of eight behind wave-uniform branches: a 18-token sequence pays for 24 keys, not for the chunk's 32 - the
    //  sums are VALU work, ~7 DPP steps per key and token, and the batch form is bound by it)
    float sc[CH];
    float m = m_run;
#pragma unroll
    for (int jb = 0; jb < CH; jb += 8) {
        if (jb < Lc) {
#pragma unroll
            for (int j = jb; j < jb + 8; ++j) {
                sc[j] = j < Lc ? att_wave_sum(q * kreg[j]) : -INFINITY;   // (wave-uniform)
                m = fmaxf(m, sc[j]);
            }
        }
    }
    const float carry = enc_exp(m_run - m);      // (first chunk: exp(-inf) = 0)
    float l = l_run * carry, o = o_run * carry;
#pragma unroll
    for (int jb = 0; jb < CH; jb += 8) {
        if (jb < Lc) {
#pragma unroll
            for (int j = jb; j < jb + 8; ++j) {
                const float e = j < Lc ? enc_exp(sc[j] - m) : 0.f;
                l += e;
                o = j < Lc ? fmaf(e, vreg[j], o) : o;   // (a masked key's value is never touched: whatever sits in a row past the sequence - stale, another call's, not even finite - cannot reach the output)
            }
        }
    }
    m_run = m; l_run = l; o_run = o;
}

struct EncAttnArgs {
    const int *meta;
    const float *qkv;   // [TMAX][3 H] row-major: Q | K | V of a token side by side
    float *out;         // [TMAX][H] operand order
    int H, heads, KW;
    float scale;        // 1 / sqrt(64)
};
// softmax(q K^T scale) V for ONE (token, head) per wave, lane = head dimension d: every key and value row of the sequence is
// one coalesced 256-byte load, the score of a key one wave-wide sum (DPP), the softmax runs on wave-uniform numbers.
// Keys in chunks of 32 (all loads of a chunk in flight together) with the flash recurrence across chunks.
// SINGLE: the call is ONE sequence (the reference's call shape): its keys are rows 0, 1, ... whatever the length, so the
// first chunk's loads are issued without waiting for the descriptor (one dependent memory round trip less: ~2 us of a 6-us
// kernel); rows past the length hold finite stale values and are masked.
template <bool SINGLE, typename M = EncMetaSmall>
__global__ __launch_bounds__(256) void enc_attention_kernel(EncAttnArgs a) {
    constexpr int CH = 32;
    const int lane = threadIdx.x & 63;
    // (measured and NOT kept, round 6: task / first row / length forced wave-uniform with readfirstlane, so that the row offsets
    //  become scalar arithmetic instead of 64-bit vector multiplies - the batch form's launch went 203 -> 245 us per layer)
    const int task = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int t = task / a.heads, h = task - t * a.heads;
    if (t >= M::T_MAX) return;
    const size_t ld = 3 * (size_t)a.H;
    int r0 = 0, L;
    float kreg[CH], vreg[CH];
    if constexpr (SINGLE) {
        const float *kb0 = a.qkv + a.H + (size_t)h * ATT_HEAD_DIM + lane;
#pragma unroll
        for (int j = 0; j < CH; ++j) { kreg[j] = kb0[(size_t)j * ld]; vreg[j] = kb0[(size_t)j * ld + a.H]; }
        L = a.meta[0];
        if (t >= L) return;
    } else {
        r0 = a.meta[M::TOK_R0 + t];
        L = a.meta[M::TOK_LEN + t];   // (0 past the call's tokens)
        if (L <= 0) return;   // wave-uniform, no barriers below
    }
    const float q = a.qkv[(size_t)t * ld + (size_t)h * ATT_HEAD_DIM + lane] * a.scale;
    const float *kb = a.qkv + (size_t)r0 * ld + a.H + (size_t)h * ATT_HEAD_DIM + lane;
    const float *vb = kb + a.H;
    float m_run = -INFINITY, l_run = 0.f, o_run = 0.f;
    for (int k0 = 0; k0 < L; k0 += CH) {
        const int Lc = min(CH, L - k0);
        if (!SINGLE || k0 > 0) {
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                // (measured and NOT kept, round 6: the rows k0 + j unclamped, masked like the SINGLE form's - no 64-bit multiply per
                //  key, but 32 distinct rows per chunk whatever the length: 206 -> 481 us per layer in the batch form. The kernel
                //  is bound by its loads, not by its VALU work: the clamped tail re-reads ONE row)
                const size_t row = (size_t)(k0 + min(j, Lc - 1)) * ld;
                kreg[j] = kb[row];
                vreg[j] = vb[row];
            }
        }
        enc_attn_chunk<CH>(q, kreg, vreg, Lc, m_run, l_run, o_run);
    }
    a.out[enc_pa(t, h * ATT_HEAD_DIM + lane, a.H, a.KW)] = o_run / l_run;
}

// The batch form's attention (encoder_big.hpp): ONE wave per (TPW consecutive tokens, head). The kernel above is bound by its loads - a
// sequence's keys and values once per TOKEN; here a wave keeps the chunk of a sequence of at most 32 tokens in registers while its
// tokens stay in that sequence (packed tokens: consecutive ones mostly do) and reloads only across a boundary. Every token gets the
// arithmetic the kernel above gives it (enc_attn_chunk on the same operands): the same bits. (One wave per whole (sequence, head) was
// measured too: 495 against 205 us per layer - 5 000 long waves do not balance over the SIMDs where 25 000 of four tokens do.)
template <typename M, int TPW>
__global__ __launch_bounds__(256) void enc_attention_group_kernel(EncAttnArgs a) {
    constexpr int CH = 32;
    const int lane = threadIdx.x & 63;
    const int task = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int g = task / a.heads, h = task - g * a.heads;
    const int T = a.meta[0];
    const int t_first = g * TPW;
    if (t_first >= T) return;   // wave-uniform, no barriers below
    const size_t ld = 3 * (size_t)a.H;
    float kreg[CH], vreg[CH];
    int have_r0 = -1;           // first row of the sequence whose (single) chunk sits in kreg / vreg
    for (int t = t_first; t < min(t_first + TPW, T); ++t) {
        const int r0 = a.meta[M::TOK_R0 + t], L = a.meta[M::TOK_LEN + t];
        if (L <= 0) continue;
        const float q = a.qkv[(size_t)t * ld + (size_t)h * ATT_HEAD_DIM + lane] * a.scale;
        const float *kb = a.qkv + (size_t)r0 * ld + a.H + (size_t)h * ATT_HEAD_DIM + lane;
        const float *vb = kb + a.H;
        float m_run = -INFINITY, l_run = 0.f, o_run = 0.f;
        for (int k0 = 0; k0 < L; k0 += CH) {
            const int Lc = min(CH, L - k0);
            if (!(L <= CH && have_r0 == r0)) {   // (a one-chunk sequence whose chunk is in the registers already: nothing to load)
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const size_t row = (size_t)(k0 + min(j, Lc - 1)) * ld;
                    kreg[j] = kb[row];
                    vreg[j] = vb[row];
                }
                have_r0 = L <= CH ? r0 : -1;
            }
            enc_attn_chunk<CH>(q, kreg, vreg, Lc, m_run, l_run, o_run);
        }
        a.out[enc_pa(t, h * ATT_HEAD_DIM + lane, a.H, a.KW)] = o_run / l_run;
    }
}

struct EncPoolArgs {
    const int *meta;
    const float *y;        // [TMAX][H] pre-norm output of the last layer, ROW-MAJOR (the last FFN-down GEMM is launched without OUT_PA: nothing
                           // reads its output as a GEMM operand, and a wave here wants a token's row in 1-KB pieces - in operand order
                           // every lane's 16 bytes came from another 256-byte group: 10.5 us for 15 tokens, 51 for 98)
    const float *g, *b;    // its LayerNorm
    float eps;
    int H, KW;             // H = 256 NV
    int pooling;           // 0: mean over the sequence's tokens, 1: its first token ([CLS])
    int normalize;         // 1: L2-normalise (torch.nn.functional.normalize, eps 1e-12)
    float *out;            // [BMAX][H]
    float *hidden;         // [TMAX][H] row-major: the last hidden state of every token (token-classification heads)
    long long slab;        // y is the sum of NSLAB slabs this many floats apart (EncLinearArgs)
};
// one work-group of 16 waves per sequence: wave w normalises tokens w, w + 16, ... (two passes over registers; a row is a
// dependent memory round trip, so the rows are dealt over as many waves as a work-group has), keeps their sum, and the
// sums meet in LDS. Column 4 (lane + 64 j) + c sits in element c of chunk j of lane `lane`.
constexpr int ENC_POOL_WAVES = 16;
template <int NV, int NSLAB, typename M = EncMetaSmall>
__global__ __launch_bounds__(ENC_POOL_WAVES * 64) void enc_pool_kernel(EncPoolArgs a) {
    __shared__ float4 part[ENC_POOL_WAVES][NV][64];
    const int b = blockIdx.x;
    if (b >= a.meta[1]) return;
    const int r0 = a.meta[M::STARTS + b], r1 = a.meta[M::STARTS + b + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 gg[NV], bb[NV], acc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        gg[j] = *reinterpret_cast<const float4 *>(a.g + 4 * (lane + 64 * j));
        bb[j] = *reinterpret_cast<const float4 *>(a.b + 4 * (lane + 64 * j));
        acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int r = r0 + wave; r < r1; r += ENC_POOL_WAVES) {
        float4 v[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {   // (every load of the row in flight before the first add)
            const size_t idx = (size_t)r * a.H + 4 * (lane + 64 * j);
            float4 u[NSLAB];
#pragma unroll
            for (int sl = 0; sl < NSLAB; ++sl) u[sl] = *reinterpret_cast<const float4 *>(a.y + idx + (size_t)sl * a.slab);
            v[j] = u[0];
#pragma unroll
            for (int sl = 1; sl < NSLAB; ++sl) { v[j].x += u[sl].x; v[j].y += u[sl].y; v[j].z += u[sl].z; v[j].w += u[sl].w; }
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        const float mean = enc_wave_sum(s) / (float)a.H;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const float dx = v[j].x - mean, dy = v[j].y - mean, dz = v[j].z - mean, dw = v[j].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
        const float rstd = 1.0f / sqrtf(enc_wave_sum(q) / (float)a.H + a.eps);
        const bool take = a.pooling == 0 || r == r0;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            float4 o;
            o.x = (v[j].x - mean) * rstd * gg[j].x + bb[j].x; o.y = (v[j].y - mean) * rstd * gg[j].y + bb[j].y;
            o.z = (v[j].z - mean) * rstd * gg[j].z + bb[j].z; o.w = (v[j].w - mean) * rstd * gg[j].w + bb[j].w;
            *reinterpret_cast<float4 *>(a.hidden + (size_t)r * a.H + 4 * (lane + 64 * j)) = o;
            if (take) { acc[j].x += o.x; acc[j].y += o.y; acc[j].z += o.z; acc[j].w += o.w; }
        }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) part[wave][j][lane] = acc[j];
    __syncthreads();
    if (wave != 0) return;   // (no barrier below)
    const float cnt = a.pooling == 0 ? fmaxf((float)(r1 - r0), 1e-9f) : 1.0f;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        float4 t = part[0][j][lane];
        for (int w = 1; w < ENC_POOL_WAVES; ++w) { const float4 u = part[w][j][lane]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        t.x /= cnt; t.y /= cnt; t.z /= cnt; t.w /= cnt;
        acc[j] = t;
        ss += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
    }
    const float scale = a.normalize ? 1.0f / fmaxf(sqrtf(enc_wave_sum(ss)), 1e-12f) : 1.0f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        float4 t = acc[j];
        t.x *= scale; t.y *= scale; t.z *= scale; t.w *= scale;
        *reinterpret_cast<float4 *>(a.out + (size_t)b * a.H + 4 * (lane + 64 * j)) = t;
    }
}

}  // namespace icd
