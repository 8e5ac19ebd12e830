// attention_kernel.hpp — the self-attention of the packed BERT encoder (services/embedding_service.py _PackedBert): every
// Linear of the encoder runs over packed tokens [T, hidden], and this kernel is what keeps the attention packed too.
//
// Reference: the attention the reference reaches through sentence-transformers -> transformers' BertSelfAttention /
// XLMRobertaSelfAttention (softmax(Q K^T / sqrt(d_head)) V per sequence and head, no dropout in eval), restated for
//   qkv  [T][3 * hidden] fp32, one row per token, Q | K | V of a token side by side (the fused QKV GEMM's output)
//   out  [T][hidden]     fp32, heads side by side (what the attention-output Linear reads)
// over sequences given by their first rows, `starts[s] .. starts[s + 1]`, each at most 512 tokens long. Without it a layer
// gathers the packed QKV to a padded [n, L] view per group of similar length, runs SDPA, and scatters the result back: 585
// MB of traffic and 18 launches per layer for the 1 000 golden diagnosis strings, against 224 MB and one launch here.
//
// One wave per (sequence, head), head dimension 64:
//   K   lane j keeps row j of the head's K in 64 registers (rows >= L are zero),
//   V   lane d keeps column d of the head's V in 64 registers (one per key, zero beyond L),
//   per query i: the 64-term dot products q_i . k_j of ALL keys at once - the query sits four registers deep, sixteen lanes
//   wide, and v_fmac_f32_dpp row_newbcast hands element d to every lane of its row (no LDS, no scalar traffic) -, a wave-wide
//   max and sum for the softmax, then sum_j p_j v_j[d] with p_j read across lanes (v_readlane) eight keys per branch.
// A sequence longer than 64 tokens takes its keys in chunks of 64 with the flash-attention recurrence (running max / sum per
// query in LDS, the unnormalised output in `out`); up to 64 tokens - every diagnosis string - it is the single pass above.
// fp32 throughout; the summation order differs from SDPA's, the results agree to ~1e-7 (tests/test_encoder_gpu.py).
#pragma once
#include <hip/hip_runtime.h>

namespace icd {

constexpr int ATT_MAX_LEN = 64;   // keys per chunk (one per lane)
constexpr int ATT_MAX_SEQ = 512;  // tokens per sequence (the running softmax state of a wave lives in LDS: 4 KB)
constexpr int ATT_HEAD_DIM = 64;

struct PackedAttnArgs {
    const float *qkv;     // [T][ld]
    float *out;           // [T][out_ld]
    const int *starts;    // [nseq + 1] first packed row of every sequence
    int nseq, heads;
    long long ld, out_ld; // row strides in floats (3 * hidden, hidden)
    int hidden;           // heads * 64
    float scale;          // 1 / sqrt(64)
    int qsplit;           // waves per (sequence, head): wave p takes the queries p, p + qsplit, ... (0 / 1: one wave takes them all). The
                          // small-input encoder (encoder_small.hpp) deals a 16-token string's queries over four waves: latency, not throughput
};

// acc += q[16 g + n] * k[16 g + n] for n = 0..15, the query element broadcast from lane n of every 16-lane row
__device__ __forceinline__ void att_dot16(float &acc, float q, const float *k) {
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %0, %1, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %0, %1, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %0, %1, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %0, %1, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %0, %1, %10 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %0, %1, %11 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %0, %1, %12 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %0, %1, %13 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %0, %1, %14 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %0, %1, %15 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %0, %1, %16 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %0, %1, %17 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 : "+v"(acc)
                 : "v"(q), "v"(k[0]), "v"(k[1]), "v"(k[2]), "v"(k[3]), "v"(k[4]), "v"(k[5]), "v"(k[6]), "v"(k[7]), "v"(k[8]),
                   "v"(k[9]), "v"(k[10]), "v"(k[11]), "v"(k[12]), "v"(k[13]), "v"(k[14]), "v"(k[15]));
}

// wave-wide max / sum on the DPP network (row_shr 1, 2, 4, 8, then row_bcast 15 and 31: lane 63 ends up with the result) -
// six VALU instructions each; a __shfl_xor butterfly is six ds_bpermute round trips through the LDS, and with two
// reductions per query row that was 6 % of this kernel (116 -> 110 us per layer on the 1 000 golden strings; the next query row fetched one iteration ahead: 104 us. What remains is instruction issue: ~250 VALU instructions per query row).
#define ICD_ATT_DPP(v, ctrl, rmask, ident) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(ident)), __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false))
__device__ __forceinline__ float att_wave_max(float v) {
    v = fmaxf(v, ICD_ATT_DPP(v, 0x111, 0xf, -INFINITY));   // row_shr:1
    v = fmaxf(v, ICD_ATT_DPP(v, 0x112, 0xf, -INFINITY));   // row_shr:2
    v = fmaxf(v, ICD_ATT_DPP(v, 0x114, 0xf, -INFINITY));   // row_shr:4
    v = fmaxf(v, ICD_ATT_DPP(v, 0x118, 0xf, -INFINITY));   // row_shr:8
    v = fmaxf(v, ICD_ATT_DPP(v, 0x142, 0xa, -INFINITY));   // row_bcast:15
    v = fmaxf(v, ICD_ATT_DPP(v, 0x143, 0xc, -INFINITY));   // row_bcast:31
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float att_wave_sum(float v) {
    v += ICD_ATT_DPP(v, 0x111, 0xf, 0.f);
    v += ICD_ATT_DPP(v, 0x112, 0xf, 0.f);
    v += ICD_ATT_DPP(v, 0x114, 0xf, 0.f);
    v += ICD_ATT_DPP(v, 0x118, 0xf, 0.f);
    v += ICD_ATT_DPP(v, 0x142, 0xa, 0.f);
    v += ICD_ATT_DPP(v, 0x143, 0xc, 0.f);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
#undef ICD_ATT_DPP

__global__ __launch_bounds__(256) void packed_attention_kernel(PackedAttnArgs a) {
    // running softmax state of the queries of a sequence longer than one chunk of 64 keys: max and sum per query, per wave
    __shared__ float run_m[4][ATT_MAX_SEQ], run_l[4][ATT_MAX_SEQ];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qs = a.qsplit > 1 ? a.qsplit : 1;
    const int wtask = blockIdx.x * 4 + wave;
    const int task = wtask / qs, part = wtask - task * qs;
    if (task >= a.nseq * a.heads) return;   // wave-uniform (no work-group barriers below)
    const int s = task / a.heads, h = task - s * a.heads;
    const int r0 = a.starts[s];
    const int L = a.starts[s + 1] - r0;     // 1 .. ATT_MAX_SEQ (checked by the host)
    const float *base = a.qkv + (size_t)r0 * a.ld + (size_t)h * ATT_HEAD_DIM;
    float *op = a.out + (size_t)r0 * a.out_ld + (size_t)h * ATT_HEAD_DIM + lane;
    const float *qp = base + (lane & 15);
    float *wm = run_m[wave], *wl = run_l[wave];

    // keys in chunks of 64 (one per lane); a sequence of up to 64 tokens - every diagnosis string - is ONE chunk and never
    // touches the running state. Longer ones keep, per query, the running max / sum in LDS and the unnormalised output in
    // `out` itself (the flash-attention recurrence: o <- o * exp(m_old - m_new) + sum_j exp(s_j - m_new) v_j).
    for (int k0 = 0; k0 < L; k0 += ATT_MAX_LEN) {
        const int Lc = min(ATT_MAX_LEN, L - k0);   // keys of this chunk
        const bool first = k0 == 0, last = k0 + ATT_MAX_LEN >= L;
        const float *kbase = base + (size_t)k0 * a.ld;
        float kreg[ATT_HEAD_DIM];               // row `lane` of the chunk's K
        {
            const float4 *kp = reinterpret_cast<const float4 *>(kbase + (size_t)min(lane, Lc - 1) * a.ld + a.hidden);
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
            const float *vp = kbase + 2 * (size_t)a.hidden + lane;
#pragma unroll
            for (int jb = 0; jb < ATT_MAX_LEN; jb += 8) {
                if (jb < Lc) {                  // wave-uniform
#pragma unroll
                    for (int j = jb; j < jb + 8; ++j) vreg[j] = (j < Lc) ? vp[(size_t)j * a.ld] : 0.f;
                } else {
#pragma unroll
                    for (int j = jb; j < jb + 8; ++j) vreg[j] = 0.f;
                }
            }
        }
        // (the next query's four values are fetched while this one is worked on: a row's arithmetic is ~150 instructions,
        //  a global load's round trip several times that)
        const float *qfirst = qp + (size_t)min(part, L - 1) * a.ld;
        float n0 = qfirst[0], n1 = qfirst[16], n2 = qfirst[32], n3 = qfirst[48];
        for (int i = part; i < L; i += qs) {
            const float q0 = n0 * a.scale, q1 = n1 * a.scale, q2 = n2 * a.scale, q3 = n3 * a.scale;
            {
                const float *qn = qp + (size_t)min(i + qs, L - 1) * a.ld;
                n0 = qn[0]; n1 = qn[16]; n2 = qn[32]; n3 = qn[48];
            }
            float acc = 0.f;
            att_dot16(acc, q0, kreg);
            att_dot16(acc, q1, kreg + 16);
            att_dot16(acc, q2, kreg + 32);
            att_dot16(acc, q3, kreg + 48);
            const float sc = lane < Lc ? acc : -INFINITY;
            float m = att_wave_max(sc);
            float carry = 0.f, l_old = 0.f;     // exp(m_old - m_new) and the running sum so far
            if (!first) {                       // wave-uniform
                const float m_old = wm[i];
                l_old = wl[i];
                const float m_new = fmaxf(m_old, m);
                carry = expf(m_old - m_new);
                m = m_new;
            }
            const float e = lane < Lc ? expf(sc - m) : 0.f;
            const float l = l_old * carry + att_wave_sum(e);
            float o = first ? 0.f : op[(size_t)i * a.out_ld] * carry;
#pragma unroll
            for (int jb = 0; jb < ATT_MAX_LEN; jb += 8) {
                if (jb < Lc) {                  // wave-uniform: eight keys per branch (keys beyond Lc carry e = 0 and v = 0)
#pragma unroll
                    for (int j = jb; j < jb + 8; ++j) o = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e), j)), vreg[j], o);
                }
            }
            if (last) {
                op[(size_t)i * a.out_ld] = o / l;
            } else {
                op[(size_t)i * a.out_ld] = o;
                if (lane == 0) { wm[i] = m; wl[i] = l; }
            }
        }
    }
}

// ---- split-bf16 operand of the encoder's Linear layers ---------------------------------------------------------------------
// The packed encoder runs its four Linear layers per block as ONE bf16 GEMM each with fp32 accumulation and output
// (services/embedding_service.py _PackedBert: y = x_hi W_hi + x_hi W_lo + x_lo W_hi + b, the fp32 arithmetic of
// SentenceTransformer.encode - reference services/embedding_service.py:97-102 - to 1e-6). This kernel makes the A operand in
// one pass: row r of x [rows][cols] fp32, optionally through erf-GELU (the BertIntermediate activation), becomes
//   out[r] = [ hi(cols) | hi(cols) | lo(cols) | 1 1 0 ... 0 ]      bf16, row stride 3 cols + 64 (K stays a multiple of the
//   vendor GEMM's 64-deep K-step: with a tail of 8 the same GEMMs took 10.2 instead of 8.2 ms per forward)
// hi = bf16(x) (round to nearest even), lo = bf16(x - hi); the two ones meet the bias rows (b_hi, b_lo) of the weight
// operand. 16 bytes read, 3 x 16 bytes written per thread and step (bound: HBM): the torch form of the same thing (two
// casts, a subtraction, a concatenation, a bias add) moved 5 x the bytes in five launches.
constexpr int SPLIT_TAIL = 64;   // elements behind [hi | hi | lo]: the two ones of the bias rows, zeros up to a whole K-step
struct SplitArgs {
    const float *x;
    unsigned short *out;
    long long rows;
    int cols;      // multiple of 8
    int act;       // 0: none, 1: erf-GELU
    long long ld;  // elements per input row
};

__device__ __forceinline__ unsigned short split_bf16_bits(float v) {
    const __bf16 b = (__bf16)v;   // v_cvt_pk_bf16_f32: RNE, NaN stays NaN (MI355X_MICROARCH.md, correctness boundaries)
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float split_bf16_float(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

__global__ __launch_bounds__(256) void split_bf16x3_kernel(SplitArgs a) {
    const int c8 = a.cols >> 3;                       // 8-element pieces per row
    const long long total = a.rows * (long long)c8;
    const long long ostride = 3ll * a.cols + SPLIT_TAIL;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / c8;
        const int c = (int)(i - r * c8) * 8;
        const float4 v0 = *reinterpret_cast<const float4 *>(a.x + r * a.ld + c);
        const float4 v1 = *reinterpret_cast<const float4 *>(a.x + r * a.ld + c + 4);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        unsigned short hi[8], lo[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = v[j];
            if (a.act == 1) t = 0.5f * t * (1.0f + erff(t * 0.70710678118654752440f));
            hi[j] = split_bf16_bits(t);
            lo[j] = split_bf16_bits(t - split_bf16_float(hi[j]));
        }
        uint4 ph, pl;
        ph.x = hi[0] | ((unsigned)hi[1] << 16); ph.y = hi[2] | ((unsigned)hi[3] << 16); ph.z = hi[4] | ((unsigned)hi[5] << 16); ph.w = hi[6] | ((unsigned)hi[7] << 16);
        pl.x = lo[0] | ((unsigned)lo[1] << 16); pl.y = lo[2] | ((unsigned)lo[3] << 16); pl.z = lo[4] | ((unsigned)lo[5] << 16); pl.w = lo[6] | ((unsigned)lo[7] << 16);
        unsigned short *o = a.out + r * ostride + c;
        *reinterpret_cast<uint4 *>(o) = ph;
        *reinterpret_cast<uint4 *>(o + a.cols) = ph;
        *reinterpret_cast<uint4 *>(o + 2 * a.cols) = pl;
        if (c < SPLIT_TAIL) {   // the row's tail, 8 elements per thread: 1 1 0 0 ... (bf16 1.0 = 0x3F80)
            uint4 one; one.x = c == 0 ? 0x3F803F80u : 0u; one.y = 0u; one.z = 0u; one.w = 0u;
            *reinterpret_cast<uint4 *>(a.out + r * ostride + 3 * a.cols + c) = one;
        }
    }
}

}  // namespace icd
