// encoder_big.hpp — the BATCH form of the small-input sentence encoder (encoder_small.hpp): the same forward, the same bits,
// for thousands of tokens per pass (icd_encoder_encode_many: a corpus build, a batch of queries, /embed).
//
// The reference embeds corpus rows and queries through the same one-string call (tools/build_database.py:217-222,
// services/embedding_service.py:117-120): identical text, identical vector. A batch path with another arithmetic (round 5:
// split-bf16 GEMMs, 1.2e-6 off) flips near-tied hits. So the batch form keeps the small form's ARITHMETIC -
// per output the same MFMA chains over the same K pieces in the same order, the same LayerNorm statistics, the same epilogue
// (the shared functions of encoder_small.hpp) - and changes only the SHAPE of the work:
//
//   enc_linear_big_kernel   a wave owns TM x TN tiles of 16 tokens x 16 columns and walks the K pieces (192 columns each) in
//                           order, one after the other, where the small form gives each piece to its own wave (or work-group)
//                           and adds the partial sums up in LDS (or in the reader, slab by slab): piece sums p_w = c0 + c1 of
//                           the two interleaved chains; a slice of four pieces s = (((0 + p_0) + p_1) + p_2) + p_3; epilogue
//                           per slice; the slices of the FFN-down GEMM added in slab order. Operands come straight from
//                           global memory in operand order (enc_pa / enc_pw: 1 KB contiguous per wave instruction, the four
//                           waves of a work-group share their W fragments through the L1), the next step's loads in flight
//                           under this step's MFMAs: per step and wave (TM + TN) KB for 8 TM TN MFMAs - the small form's
//                           2 KB for 8. The products in the handle's arithmetic, like the small form: fp32 MFMAs (157 TFLOP/s is
//                           that arithmetic's roofline: the GEMMs reach 0.65-0.75 of it) or the split-bf16 form (BF: a fifth of
//                           the matrix time - the GEMMs are then bound by the operand supply through the L1).
//   enc_ln_stats_kernel     mean and 1 / std of every token's pre-norm row, by the lanes and in the order the small form's
//                           LNPRO kernels compute them inside the GEMM (enc_piece_stats per 192-column slice, enc_combine_stats).
//   embedding sum, pooling: encoder_small.hpp's kernels themselves, instantiated for the larger descriptor; the attention one wave per
//   (four consecutive tokens, head) instead of one per (token, head) - enc_attention_group_kernel, the same arithmetic per token.
#pragma once
#include <hip/hip_runtime.h>
#include "encoder_small.hpp"

namespace icd {

struct EncStatsArgs {
    const float *x;      // [T][K] operand order (one slab)
    float *stats;        // [T][2]
    int T, K;
    float eps;
};
// one wave per 16-token tile; K = 4 pieces of 16 ITER columns
template <int ITER>
__global__ __launch_bounds__(256) void enc_ln_stats_kernel(EncStatsArgs a) {
    __shared__ float red[4][2][4][16];   // [wave][mean | M2][piece][row]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x * 4 + wave;
    const int r16 = lane & 15, kq = lane >> 4;
    if (tile * 16 >= a.T) return;   // (wave-uniform; no barrier below: the LDS block is wave-private)
    const int nwk = a.K / (16 * ITER);
    for (int w = 0; w < nwk; ++w) {
        float4 areg[ITER];
        const float *xp = a.x + (((size_t)tile * nwk + w) * ITER * 64 + lane) * 4;
#pragma unroll
        for (int i = 0; i < ITER; ++i) areg[i] = *reinterpret_cast<const float4 *>(xp + (size_t)i * 256);
        float mw, qw;
        enc_piece_stats<ITER>(areg, mw, qw);
        if (kq == 0) { red[wave][0][w][r16] = mw; red[wave][1][w][r16] = qw; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (kq == 0) {
        float mean, rstd;
        enc_combine_stats(&red[wave][0][0][r16], &red[wave][1][0][r16], 16, nwk, 16 * ITER, a.K, a.eps, mean, rstd);
        const int t = tile * 16 + r16;
        if (t < a.T) { a.stats[2 * t] = mean; a.stats[2 * t + 1] = rstd; }
    }
}

struct EncBigLinearArgs {
    const float *x;        // [T][K] operand order, ONE slab: the A operand, or (LNPRO) the pre-norm rows it is the LayerNorm of
    const float *stats;    // LNPRO: [T][2] of x's rows (enc_ln_stats_kernel)
    const float *c1;       // LNPRO: [N]
    const float *w;        // [N][K] in NT-row tiles (enc_pw), as the small form's GEMM of the same layer reads it
    int NT;                // 16, or 8 (the attention-output weights)
    const float *bias;     // [N]; LNPRO: c2
    const float *res_src;  // EPI == 2: [T][N] operand order, one slab
    const float *res_stats, *res_g, *res_b;
    float *y;              // [T][N]: operand order (OUT_PA) or row-major
    int T, K, N;
    unsigned long long *stamps;   // diagnostic builds (ICD_ABLATE): 7 x (s_memtime, s_memrealtime) of wave 0 of a work-group in the middle of the grid; nullptr = none
    int pps;               // K pieces per slice: K = 16 ITER pps SLICES (4 where the small form's four waves cover K; inter / (64 ITER) for the FFN-down GEMM's four slabs)
};
// ITER: 16-column k-steps per K piece (12: hidden 768 / inter 3 072; 16: hidden 1 024 / inter 4 096); TM x TN: tiles of 16 tokens x
// 16 columns per wave; EPI / LNPRO / OUT_PA: as enc_linear_kernel; SLICES: K slices of `pps` pieces each whose sums the small
// form keeps in slabs (1, or ENC_SLABS = 4: the FFN-down GEMM)
// PF: pairs of k-steps in flight ahead of the MFMAs (1: a double buffer, 2: three buffers - where ITER / 2 is a multiple of three)
// BF: the split-bf16 arithmetic (encoder_small.hpp: three bf16 MFMAs per 32-block = per pair of k-steps; w holds w_hi / w_lo)
// (measured and NOT kept, round 6: the W fragments of a pair through LDS - each wave loads a quarter, all four read all of them
//  back, one barrier per pair: 394 against 228 us for the FFN-down GEMM; the four waves share their W fragments through the L1
//  well enough, and the barrier puts the LDS round trip on every pair's critical path: profiles/r06_encoder_big_bf_sweep.log)
#ifdef ICD_ABLATE
#define ENC_BIG_STAMP(i) do { if (a.stamps && blockIdx.x == 1 && blockIdx.y == gridDim.y / 2 && threadIdx.x == 0) { __builtin_amdgcn_sched_barrier(0); \
    a.stamps[2 * (i)] = __builtin_amdgcn_s_memtime(); a.stamps[2 * (i) + 1] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define ENC_BIG_DRAIN() do { if (a.stamps) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); } while (0)
#else
#define ENC_BIG_STAMP(i) do { } while (0)
#define ENC_BIG_DRAIN() do { } while (0)
#endif
template <int ITER, int TM, int TN, int EPI, bool LNPRO, bool OUT_PA, int SLICES, int PF = 1, bool BF = false>
__global__ __launch_bounds__(256) void enc_linear_big_kernel(EncBigLinearArgs a) {
    extern __shared__ char enc_big_occupancy_pin[];   // (never touched: the launch asks for more than half a CU's LDS so that a CU holds ONE work-group - see icd_encoder.hpp)
    constexpr int KW = 16 * ITER;
    constexpr int STEPS = ITER / 2;           // an iteration: two k-steps (one MFMA group of each chain)
    constexpr int NBUF = PF + 1;
    static_assert(STEPS % NBUF == 0, "the register ring returns to buffer 0 at every piece");
    ENC_BIG_STAMP(0);   // the first kernel argument is here
    const int NPIECE = a.pps * SLICES;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int tt0 = ((int)blockIdx.y * 4 + wave) * TM;      // first token tile of this wave
    const int ct0 = (int)blockIdx.x * TN;                    // first column tile
    if (tt0 * 16 >= a.T) return;                              // (wave-uniform; the kernel has no barrier)
    // operand addresses: A fragment (token tile tt, piece w, k-step i) = x + (((tt NPIECE + w) ITER + i) 64 + lane) 4;
    // B fragment (column tile ct, piece w, k-step i): the lane's row n = 16 ct + r16 of W in NT-row-tile order
    const float *ap[TM];
#pragma unroll
    for (int m = 0; m < TM; ++m) ap[m] = a.x + ((size_t)(tt0 + m) * NPIECE * ITER * 64 + lane) * 4;
    const float *bp[TN];
#pragma unroll
    for (int n = 0; n < TN; ++n) {
        const int col = (ct0 + n) * 16 + r16;
        bp[n] = a.w + (((size_t)(col / a.NT) * NPIECE) * ITER * 4 + kq) * (size_t)(a.NT * 4) + (size_t)(col % a.NT) * 4;
    }
    const size_t a_step = 256, a_piece = (size_t)ITER * 256;                        // floats between k-steps / pieces of an A tile
    const size_t b_step = (size_t)4 * a.NT * 4, b_piece = (size_t)ITER * b_step;    // ... of a W row tile
    float4 abuf[NBUF][2][TM], bbuf[NBUF][2][TN];   // [buffer][k-step of the pair][tile]
    auto load_pair = [&](const int buf, int piece, int st) {   // (buf is a constant at every call site: the buffers stay in registers)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int m = 0; m < TM; ++m) abuf[buf][h][m] = *reinterpret_cast<const float4 *>(ap[m] + piece * a_piece + (size_t)(2 * st + h) * a_step);
#pragma unroll
            for (int n = 0; n < TN; ++n) bbuf[buf][h][n] = *reinterpret_cast<const float4 *>(bp[n] + piece * b_piece + (size_t)(2 * st + h) * b_step);
        }
    };
    enc_f32x4 v[TM][TN];      // the output: the slices' epilogued sums added in slab order
    // per output element (register j of tile (m, n)): token 16 (tt0 + m) + 4 kq + j, column 16 (ct0 + n) + r16
    float bias_v[TN], c1_v[TN], rg[TN], rb[TN];
#pragma unroll
    for (int n = 0; n < TN; ++n) {
        const int col = (ct0 + n) * 16 + r16;
        bias_v[n] = a.bias[col];
        c1_v[n] = LNPRO ? a.c1[col] : 0.f;
        rg[n] = EPI == 2 ? a.res_g[col] : 0.f;
        rb[n] = EPI == 2 ? a.res_b[col] : 0.f;
    }
    load_pair(0, 0, 0);
    if constexpr (PF == 2) load_pair(1, 0, 1);
    // the epilogue's inputs, requested NOW: a work-group is one wave per SIMD, nothing fills a round trip to memory - the statistics and the
    // residual rows arrive under the K walk instead of after it (the chain arguments -> first fragments -> ... -> epilogue inputs, every link
    // a round trip, is ~8 us of each work-group's ~25: profiles/r06_encoder_big_ring_ablation.log)
    float ln_mean[TM][4], ln_rstd[TM][4], rs_mean[TM][4], rs_rstd[TM][4], rs_src[TM][4][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = (tt0 + m) * 16 + 4 * kq + j;
            const bool live = t < a.T;
            ln_mean[m][j] = ln_rstd[m][j] = rs_mean[m][j] = rs_rstd[m][j] = 0.f;
            if constexpr (LNPRO) { if (live) { ln_mean[m][j] = a.stats[2 * t]; ln_rstd[m][j] = a.stats[2 * t + 1]; } }
            if constexpr (EPI == 2) { if (live) { rs_mean[m][j] = a.res_stats[2 * t]; rs_rstd[m][j] = a.res_stats[2 * t + 1]; } }
#pragma unroll
            for (int n = 0; n < TN; ++n) {
                rs_src[m][j][n] = 0.f;
                if constexpr (EPI == 2) { if (live) rs_src[m][j][n] = a.res_src[enc_pa(t, (ct0 + n) * 16 + r16, a.N, KW)]; }
            }
        }
    ENC_BIG_STAMP(1);   // bias, first fragments, epilogue inputs requested
    ENC_BIG_DRAIN();
    ENC_BIG_STAMP(2);   // ... and here
#pragma unroll 1
    for (int sl = 0; sl < SLICES; ++sl) {
        enc_f32x4 s[TM][TN];
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n) s[m][n] = enc_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int pw = 0; pw < a.pps; ++pw) {
            const int piece = sl * a.pps + pw;
            if (piece == 1) ENC_BIG_STAMP(3);   // the first piece (six pairs of k-steps) done
            enc_f32x4 c0[TM][TN], c1[TM][TN];
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n) { c0[m][n] = enc_f32x4{0.f, 0.f, 0.f, 0.f}; c1[m][n] = enc_f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                // the next pair's loads (the next piece's first pair behind this piece's last) under this pair's MFMAs
                // the loads of the pair PF steps ahead (into the next pieces' first pairs behind this piece's last) under this pair's MFMAs
                {
                    const int ahead = st + PF;                       // (STEPS >= NBUF: at most one piece ahead)
                    const int pz = ahead >= STEPS ? piece + 1 : piece, sz = ahead >= STEPS ? ahead - STEPS : ahead;
                    if (pz < NPIECE) load_pair((st + PF) % NBUF, pz, sz);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (BF) {
                    // this pair of k-steps IS a 32-block (block st of the piece: even ones into c0, odd ones into c1, as in the small
                    // form); its A fragments are split once and meet every column tile; term by term over ALL tiles
                    enc_bf16x8 ah[TM], al[TM];
#pragma unroll
                    for (int m = 0; m < TM; ++m) enc_split8(abuf[st % NBUF][0][m], abuf[st % NBUF][1][m], ah[m], al[m]);
#define ENC_BIG_BF(AOP, BSLOT)                                                                                             \
                    _Pragma("unroll") for (int m = 0; m < TM; ++m)                                                         \
                        _Pragma("unroll") for (int n = 0; n < TN; ++n) {                                                   \
                            if ((st & 1) == 0) c0[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AOP[m], enc_as_bf16x8(bbuf[st % NBUF][BSLOT][n]), c0[m][n], 0, 0, 0); \
                            else c1[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AOP[m], enc_as_bf16x8(bbuf[st % NBUF][BSLOT][n]), c1[m][n], 0, 0, 0); \
                        }
                    ENC_BIG_BF(ah, 0) ENC_BIG_BF(ah, 1) ENC_BIG_BF(al, 0)
#undef ENC_BIG_BF
                } else {
                // (component by component over ALL tiles: consecutive MFMAs never touch the same accumulator)
#define ENC_BIG_MFMA(C)                                                                                                   \
                _Pragma("unroll") for (int m = 0; m < TM; ++m)                                                            \
                    _Pragma("unroll") for (int n = 0; n < TN; ++n) {                                                      \
                        c0[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(abuf[st % NBUF][0][m].C, bbuf[st % NBUF][0][n].C, c0[m][n], 0, 0, 0); \
                        c1[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(abuf[st % NBUF][1][m].C, bbuf[st % NBUF][1][n].C, c1[m][n], 0, 0, 0); \
                    }
                ENC_BIG_MFMA(x) ENC_BIG_MFMA(y) ENC_BIG_MFMA(z) ENC_BIG_MFMA(w)
#undef ENC_BIG_MFMA
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // the piece's sum joins the slice's: s += (c0 + c1), in piece order (the small form: s = 0; for w: s += red[w])
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n)
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[m][n][j] += c0[m][n][j] + c1[m][n][j];
        }
        if (sl == SLICES - 1) ENC_BIG_STAMP(4);   // the K walk done (the accumulators may still be in the pipe)
        // the slice's epilogue (slab `sl` of the small form), added to the output in slab order
#pragma unroll
        for (int m = 0; m < TM; ++m) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float mean = ln_mean[m][j], rstd = ln_rstd[m][j], rmean = rs_mean[m][j], rrstd = rs_rstd[m][j];   // (rows past T: zeros, never stored; slices past the first ignore the residual)
#pragma unroll
                for (int n = 0; n < TN; ++n) {
                    const float rsrc = rs_src[m][j][n];
                    const float e = enc_epilogue<EPI, LNPRO>(s[m][n][j], sl == 0, mean, rstd, c1_v[n], bias_v[n], rsrc, rmean, rrstd, rg[n], rb[n]);
                    if (sl == 0) v[m][n][j] = e;
                    else v[m][n][j] += e;
                }
            }
        }
    }
    ENC_BIG_STAMP(5);   // epilogue arithmetic done
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = (tt0 + m) * 16 + 4 * kq + j;
            if (t >= a.T) continue;
#pragma unroll
            for (int n = 0; n < TN; ++n) {
                const int col = (ct0 + n) * 16 + r16;
                if constexpr (OUT_PA) a.y[enc_pa(t, col, a.N, KW)] = v[m][n][j];
                else a.y[(size_t)t * a.N + col] = v[m][n][j];
            }
        }
    ENC_BIG_DRAIN();
    ENC_BIG_STAMP(6);   // stores acknowledged
}

}  // namespace icd
