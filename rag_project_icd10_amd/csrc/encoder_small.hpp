// encoder_small.hpp — the BERT-style sentence encoder for SMALL inputs: the reference's own call shape of the embed step,
// EmbeddingService.encode_query / encode_single (services/embedding_service.py:97-120: ONE string per
// SentenceTransformer.encode call) and the handful of diagnoses of one /query request.
//
// At 10-100 tokens the forward is ~85 MFLOP per token against 340 MB of fp32 weights: what sets its latency is not
// arithmetic but the NUMBER of kernels and the dependent memory round trips inside each (the framework's forward is ~220
// kernels of ~5 us: 1.1 ms replayed from a graph; a kernel that loads its arguments, one operand and stores takes 3-5 us on
// this part however little it computes, and a boundary 1.2-1.5: MI355X_MICROARCH.md, price list - a grid barrier inside
// one launch costs MORE than a boundary, 4-7 us, so this is not a persistent kernel). This file is that forward as FIVE
// launches per layer, all fp32, the LayerNorms folded into their consumers:
//
//   enc_linear_kernel   Y = act(A W^T + b) (+ R): one work-group per NT (16, 8 or 4) output columns, its waves split K (192
//                       columns each). A wave issues ALL its loads up front - its NT x 192 slice of W (12 x 16 B per lane:
//                       the whole slice in flight at once), the first 16 tokens' operand rows, LayerNorm parameters - so
//                       that the kernel is one memory round trip deep, then takes the tokens 16 at a time through
//                       v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate: exact products; the sum's order differs from the
//                       vendor GEMM's like any two GEMMs differ); the waves' partial sums meet in LDS, where bias and
//                       erf-GELU are applied.
//       LNPRO           A = LayerNorm(X) computed by the work-group itself (its waves hold whole rows between them: two
//                       reductions through LDS, mean then centred squares): the QKV and FFN-up GEMMs read the PRE-norm sums
//                       of the previous sublayer. Work-group 0 leaves mean and 1/std per token for ...
//       EPI == 2        ... the residual of the next GEMM (BertSelfOutput / BertOutput: dense(x) + LayerNorm-ed input),
//                       rebuilt from the pre-norm row and those two numbers. No normalised activation is ever stored.
//   enc_attention_kernel  softmax(q K^T / 8) V, one wave per (token, head): K rows and V columns of the sequence in
//                       registers (the arithmetic of attention_kernel.hpp, one query per wave instead of a loop over them).
//   enc_embed_kernel    (word + type) + position, pre-norm.        enc_pool_kernel   the last LayerNorm, masked-mean (or
//                       [CLS]) pooling and L2 normalisation, one work-group per sequence.
//
// Token ids, positions and sequence bounds come from a small descriptor in device memory (layout below) that the host
// fills per call: the launches themselves depend only on the token BUCKET (16 / 32 / 64 / 128), so a forward is one replay
// of a captured graph (icd_encoder.hpp). Arithmetic restated from transformers' BertModel - the published architecture
// the reference reaches through sentence-transformers; the checkpoint's numerics are unpinned (no weights offline,
// DESIGN.md section 7): the tests compare with the framework's fp32 forward of the same weights (1e-5).
#pragma once
#include <hip/hip_runtime.h>
#include "attention_kernel.hpp"

namespace icd {

constexpr int ENC_TMAX = 128;   // packed tokens per call
constexpr int ENC_BMAX = 32;    // sequences per call
// descriptor (int32 words): [0] T, [1] B, then TMAX token ids, TMAX positions, TMAX sequence-of-token, BMAX + 1 sequence
// starts (starts[b] = T for b >= B)
constexpr int ENC_META_IDS = 2, ENC_META_POS = 2 + ENC_TMAX, ENC_META_SEQ = 2 + 2 * ENC_TMAX, ENC_META_STARTS = 2 + 3 * ENC_TMAX;
constexpr int ENC_META_WORDS = ENC_META_STARTS + ENC_BMAX + 1;

typedef float enc_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float enc_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

struct EncEmbedArgs {
    const int *meta;
    const float *word, *pos, *type0;   // embeddings [vocab][H], [max_pos][H], token type 0 [H]
    int H;
    float *y;   // [TMAX][H] pre-norm
};
// y[t] = (word[id] + type0) + pos[p] (BertEmbeddings in front of its LayerNorm): one wave per token
template <int NV>
__global__ __launch_bounds__(256) void enc_embed_kernel(EncEmbedArgs a) {
    const int lane = threadIdx.x & 63, t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= a.meta[0]) return;   // wave-uniform
    const size_t id = (size_t)a.meta[ENC_META_IDS + t], p = (size_t)a.meta[ENC_META_POS + t];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = 4 * (lane + 64 * j);
        const float4 w = *reinterpret_cast<const float4 *>(a.word + id * a.H + c);
        const float4 ty = *reinterpret_cast<const float4 *>(a.type0 + c);
        const float4 pe = *reinterpret_cast<const float4 *>(a.pos + p * a.H + c);
        float4 v;
        v.x = (w.x + ty.x) + pe.x; v.y = (w.y + ty.y) + pe.y; v.z = (w.z + ty.z) + pe.z; v.w = (w.w + ty.w) + pe.w;
        *reinterpret_cast<float4 *>(a.y + (size_t)t * a.H + c) = v;
    }
}

struct EncLinearArgs {
    const int *meta;
    const float *x;        // [TMAX][K]: the A operand, or (LNPRO) the pre-norm rows it is the LayerNorm of
    const float *ln_g, *ln_b;   // LNPRO: that LayerNorm (over K)
    float ln_eps;
    float *stats_out;      // LNPRO: [TMAX][2] mean, 1 / sqrt(var + eps) of every token (written by work-group 0)
    const float *w;        // [N][K] (torch.nn.Linear.weight)
    const float *bias;     // [N]
    // EPI == 2: + LayerNorm(res_src)[t][n], rebuilt from the pre-norm row and the statistics a LNPRO kernel left
    const float *res_src;  // [TMAX][N]
    const float *res_stats, *res_g, *res_b;
    float *y;              // [TMAX][N]
    int K, N;
    unsigned long long *stamps;   // diagnostic builds (ICD_ABLATE): 8 x (s_memtime, s_memrealtime) of wave 0 of work-group 1; nullptr = none
};
#ifdef ICD_ABLATE
#define ENC_STAMP(i) do { if (a.stamps && blockIdx.x == 1 && tid == 0) { __builtin_amdgcn_sched_barrier(0); \
    a.stamps[2 * (i)] = __builtin_amdgcn_s_memtime(); a.stamps[2 * (i) + 1] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define ENC_STAMP_DRAIN() do { if (a.stamps) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); } while (0)
#else
#define ENC_STAMP(i) do { } while (0)
#define ENC_STAMP_DRAIN() do { } while (0)
#endif
// ITER: 16-column k-steps per wave (K = 16 ITER x waves); NT: output columns per work-group (grid = N / NT);
// EPI: 0 bias, 1 bias + erf-GELU (BertIntermediate), 2 bias + LayerNorm-ed residual
template <int ITER, int NT, int EPI, bool LNPRO>
__global__ __launch_bounds__(LNPRO ? 256 : 1024) void enc_linear_kernel(EncLinearArgs a) {
    static_assert(NT == 16 || NT == 8 || NT == 4, "columns per work-group");
    __shared__ float red[16][256];
    __shared__ float lnred[2][16][16];   // LNPRO: per wave and row, sum and centred sum of squares
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = (int)blockDim.x >> 6;
    ENC_STAMP(0);
    const int n0 = blockIdx.x * NT;
    const int r16 = lane & 15, kq = lane >> 4;
    const int kbase = wave * (16 * ITER) + 4 * kq;
    // ---- everything this wave will read, issued before anything waits ------------------------------------------------------
    float4 wreg[ITER];   // W rows n0 .. n0 + NT - 1 (lanes r16 >= NT: zero columns of the MFMA's B operand)
    {
        const float *wp = a.w + (size_t)(n0 + (r16 < NT ? r16 : 0)) * a.K + kbase;
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            const float4 v = *reinterpret_cast<const float4 *>(wp + 16 * i);
            wreg[i] = r16 < NT ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float4 areg[ITER];   // rows t0 .. t0 + 15 of the operand (rows past T inside the last tile: finite stale values, never stored)
    {
        const float *xp = a.x + (size_t)r16 * a.K + kbase;
#pragma unroll
        for (int i = 0; i < ITER; ++i) areg[i] = *reinterpret_cast<const float4 *>(xp + 16 * i);
    }
    float4 greg[LNPRO ? ITER : 1], breg[LNPRO ? ITER : 1];
    if constexpr (LNPRO) {
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            greg[i] = *reinterpret_cast<const float4 *>(a.ln_g + kbase + 16 * i);
            breg[i] = *reinterpret_cast<const float4 *>(a.ln_b + kbase + 16 * i);
        }
    }
    // the epilogue's operands of thread tid < 256: output (token 4 (l >> 4) + j, column l & 15), l = tid & 63, j = tid >> 6
    const int ej = tid >> 6, el = tid & 63;
    const int en = el & 15, et = 4 * (el >> 4) + ej;
    const bool ecol = tid < 256 && en < NT;
    float bias_v = 0.f, rg = 0.f, rb = 0.f, rsrc = 0.f, rmean = 0.f, rrstd = 0.f;
    if (ecol) {
        bias_v = a.bias[n0 + en];
        if constexpr (EPI == 2) {
            rg = a.res_g[n0 + en]; rb = a.res_b[n0 + en];
            rsrc = a.res_src[(size_t)et * a.N + n0 + en]; rmean = a.res_stats[2 * et]; rrstd = a.res_stats[2 * et + 1];   // (tile 0)
        }
    }
    const int T = a.meta[0];
    __builtin_amdgcn_sched_barrier(0);   // (left alone, hipcc sinks the loads to their uses: one round trip per four MFMAs instead of one in all)
    ENC_STAMP(1);        // loads issued
    ENC_STAMP_DRAIN();
    ENC_STAMP(2);        // loads landed

    for (int t0 = 0; t0 < T; t0 += 16) {
        if (t0 > 0) {
            const float *xp = a.x + (size_t)(t0 + r16) * a.K + kbase;
#pragma unroll
            for (int i = 0; i < ITER; ++i) areg[i] = *reinterpret_cast<const float4 *>(xp + 16 * i);
            if constexpr (EPI == 2) {
                if (ecol) { rsrc = a.res_src[(size_t)(t0 + et) * a.N + n0 + en]; rmean = a.res_stats[2 * (t0 + et)]; rrstd = a.res_stats[2 * (t0 + et) + 1]; }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (LNPRO) {
            // the work-group's waves hold rows t0 .. t0 + 15 whole between them: lane (r16, kq) of wave w has 4 ITER values of row r16
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < ITER; ++i) s += (areg[i].x + areg[i].y) + (areg[i].z + areg[i].w);
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (t0 > 0) __syncthreads();   // (the previous tile's statistics have been read)
            if (kq == 0) lnred[0][wave][r16] = s;
            __syncthreads();
            float mean = 0.f;
            for (int w = 0; w < nw; ++w) mean += lnred[0][w][r16];
            mean /= (float)a.K;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < ITER; ++i) {
                const float dx = areg[i].x - mean, dy = areg[i].y - mean, dz = areg[i].z - mean, dw = areg[i].w - mean;
                q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
            q += __shfl_xor(q, 16);
            q += __shfl_xor(q, 32);
            if (kq == 0) lnred[1][wave][r16] = q;
            __syncthreads();
            float var = 0.f;
            for (int w = 0; w < nw; ++w) var += lnred[1][w][r16];
            const float rstd = 1.0f / sqrtf(var / (float)a.K + a.ln_eps);
#pragma unroll
            for (int i = 0; i < ITER; ++i) {
                areg[i].x = (areg[i].x - mean) * rstd * greg[i].x + breg[i].x; areg[i].y = (areg[i].y - mean) * rstd * greg[i].y + breg[i].y;
                areg[i].z = (areg[i].z - mean) * rstd * greg[i].z + breg[i].z; areg[i].w = (areg[i].w - mean) * rstd * greg[i].w + breg[i].w;
            }
            if (blockIdx.x == 0 && wave == 0 && kq == 0 && t0 + r16 < T) {
                a.stats_out[2 * (t0 + r16)] = mean;
                a.stats_out[2 * (t0 + r16) + 1] = rstd;
            }
        }
        ENC_STAMP(3);    // LayerNorm prologue done
        enc_f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[i].x, wreg[i].x, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[i].y, wreg[i].y, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[i].z, wreg[i].z, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[i].w, wreg[i].w, c, 0, 0, 0);
        }
        // C[token 4 (lane >> 4) + j][column lane & 15] in register j
        if (t0 > 0) __syncthreads();   // the previous tile's sums have been read
#pragma unroll
        for (int j = 0; j < 4; ++j) red[wave][j * 64 + lane] = c[j];
        ENC_STAMP(4);    // MFMAs done (the LDS writes wait for them)
        __syncthreads();
        ENC_STAMP(5);    // barrier passed
        if (ecol) {
            float s = 0.f;
            for (int w = 0; w < nw; ++w) s += red[w][tid];
            s += bias_v;
            if constexpr (EPI == 1) s = 0.5f * s * (1.0f + erff(s * 0.70710678118654752440f));
            if constexpr (EPI == 2) s += (rsrc - rmean) * rrstd * rg + rb;
            if (t0 + et < T) a.y[(size_t)(t0 + et) * a.N + n0 + en] = s;
        }
        ENC_STAMP_DRAIN();
        ENC_STAMP(6);    // epilogue stored
    }
}

struct EncAttnArgs {
    const int *meta;
    const float *qkv;   // [TMAX][3 H]: Q | K | V of a token side by side
    float *out;         // [TMAX][H]
    int H, heads;
    float scale;        // 1 / sqrt(64)
};
// softmax(q K^T scale) V for ONE (token, head) per wave: keys in chunks of 64 (one per lane; a string of up to 64 tokens is
// one chunk), the running maximum / sum / output of the flash recurrence in registers. Inner arithmetic:
// attention_kernel.hpp (att_dot16: the query sits four registers deep, sixteen lanes wide, broadcast by DPP).
__global__ __launch_bounds__(256) void enc_attention_kernel(EncAttnArgs a) {
    const int lane = threadIdx.x & 63;
    const int task = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int t = task / a.heads, h = task - t * a.heads;
    if (t >= a.meta[0]) return;   // wave-uniform, no barriers below
    const int s = a.meta[ENC_META_SEQ + t];
    const int r0 = a.meta[ENC_META_STARTS + s];
    const int L = a.meta[ENC_META_STARTS + s + 1] - r0;
    const size_t ld = 3 * (size_t)a.H;
    const float *base = a.qkv + (size_t)r0 * ld + (size_t)h * ATT_HEAD_DIM;
    const float *qp = a.qkv + (size_t)t * ld + (size_t)h * ATT_HEAD_DIM + (lane & 15);
    const float q0 = qp[0] * a.scale, q1 = qp[16] * a.scale, q2 = qp[32] * a.scale, q3 = qp[48] * a.scale;
    float m_run = -INFINITY, l_run = 0.f, o_run = 0.f;
    for (int k0 = 0; k0 < L; k0 += ATT_MAX_LEN) {
        const int Lc = min(ATT_MAX_LEN, L - k0);
        const float *kbase = base + (size_t)k0 * ld;
        float kreg[ATT_HEAD_DIM];               // row `lane` of the chunk's K
        {
            const float4 *kp = reinterpret_cast<const float4 *>(kbase + (size_t)min(lane, Lc - 1) * ld + a.H);
            const bool live = lane < Lc;
#pragma unroll
            for (int c = 0; c < ATT_HEAD_DIM / 4; ++c) {
                const float4 v = kp[c];
                kreg[4 * c + 0] = live ? v.x : 0.f; kreg[4 * c + 1] = live ? v.y : 0.f;
                kreg[4 * c + 2] = live ? v.z : 0.f; kreg[4 * c + 3] = live ? v.w : 0.f;
            }
        }
        float vreg[ATT_MAX_LEN];                // column `lane` of the chunk's V, one register per key
        {
            const float *vp = kbase + 2 * (size_t)a.H + lane;
#pragma unroll
            for (int jb = 0; jb < ATT_MAX_LEN; jb += 8) {
                if (jb < Lc) {                  // wave-uniform
#pragma unroll
                    for (int j = jb; j < jb + 8; ++j) vreg[j] = (j < Lc) ? vp[(size_t)j * ld] : 0.f;
                } else {
#pragma unroll
                    for (int j = jb; j < jb + 8; ++j) vreg[j] = 0.f;
                }
            }
        }
        float acc = 0.f;
        att_dot16(acc, q0, kreg);
        att_dot16(acc, q1, kreg + 16);
        att_dot16(acc, q2, kreg + 32);
        att_dot16(acc, q3, kreg + 48);
        const float sc = lane < Lc ? acc : -INFINITY;
        const float m_new = fmaxf(m_run, att_wave_max(sc));
        const float carry = expf(m_run - m_new);      // (first chunk: exp(-inf) = 0)
        const float e = lane < Lc ? expf(sc - m_new) : 0.f;
        l_run = l_run * carry + att_wave_sum(e);
        float o = o_run * carry;
#pragma unroll
        for (int jb = 0; jb < ATT_MAX_LEN; jb += 8) {
            if (jb < Lc) {
#pragma unroll
                for (int j = jb; j < jb + 8; ++j) o = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e), j)), vreg[j], o);
            }
        }
        o_run = o;
        m_run = m_new;
    }
    a.out[(size_t)t * a.H + (size_t)h * ATT_HEAD_DIM + lane] = o_run / l_run;
}

struct EncPoolArgs {
    const int *meta;
    const float *y;        // [TMAX][H] pre-norm output of the last layer
    const float *g, *b;    // its LayerNorm
    float eps;
    int H;                 // 768
    int pooling;           // 0: mean over the sequence's tokens, 1: its first token ([CLS])
    int normalize;         // 1: L2-normalise (torch.nn.functional.normalize, eps 1e-12)
    float *out;            // [BMAX][H]
    float *hidden;         // [TMAX][H]: the last hidden state of every token (token-classification heads)
};
// one work-group per sequence: wave w normalises tokens w, w + 4, ... (two passes over registers), keeps their sum, and the
// four sums meet in LDS. Column 4 (lane + 64 j) + c sits in element c of chunk j of lane `lane`.
template <int NV>
__global__ __launch_bounds__(256) void enc_pool_kernel(EncPoolArgs a) {
    __shared__ float4 part[4][NV][64];
    __shared__ float nrm[4];
    const int b = blockIdx.x;
    if (b >= a.meta[1]) return;
    const int r0 = a.meta[ENC_META_STARTS + b], r1 = a.meta[ENC_META_STARTS + b + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 gg[NV], bb[NV], acc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        gg[j] = *reinterpret_cast<const float4 *>(a.g + 4 * (lane + 64 * j));
        bb[j] = *reinterpret_cast<const float4 *>(a.b + 4 * (lane + 64 * j));
        acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int r = r0 + wave; r < r1; r += 4) {
        float4 v[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = *reinterpret_cast<const float4 *>(a.y + (size_t)r * a.H + 4 * (lane + 64 * j));
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
        for (int w = 1; w < 4; ++w) { const float4 u = part[w][j][lane]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
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
