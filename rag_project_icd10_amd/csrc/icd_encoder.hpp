// icd_encoder.hpp — host side of the small-input sentence encoder (include/icd_search.h, icd_encoder_*; kernels:
// encoder_small.hpp, attention_kernel.hpp). Included by icd_search.hip (its fail() / HIP_TRY).
//
// One forward = ONE graph launch: [descriptor H2D] -> embedding sum -> layers x (LayerNorm + QKV, attention, attention output
// + residual, LayerNorm + FFN up + GELU, FFN down + residual) -> LayerNorm + pooling -> [result D2H into a pinned block]. The
// graph depends on the token bucket (16 / 32 / 64 / 128 / 256: the grids of the per-token kernels) and the pooling flags only;
// token ids, positions and sequence bounds travel in the descriptor. Captured once per (bucket, pooling, normalize) on a
// private stream, replayed on the caller's (a call made while the caller's stream is itself being captured is refused: the
// token ids are read from host memory at call time).
#pragma once
#include "attention_kernel.hpp"
#include "encoder_small.hpp"
#include "encoder_big.hpp"

struct icd_encoder {
    uint64_t magic = 0;
    int device = 0;
    icd_encoder_desc d{};
    std::vector<const float *> b_qkv, b_ao, ln1_g, ln1_b, b_up, b_down, ln2_g, ln2_b;   // borrowed from the caller
    std::vector<float *> w_qkv, w_ao, w_up, w_down;   // OWNED copies of the four Linear weights in the order their GEMM reads them (encoder_small.hpp enc_pw);
                                                      // w_qkv / w_up with the LayerNorm in front of them folded in, c1 / c2 per output column (enc_fold_ln_kernel)
    std::vector<float *> c1_qkv, c2_qkv, c1_up, c2_up;
    // workspace: activations of at most ENC_TMAX tokens
    float *yb[3] = {nullptr, nullptr, nullptr};   // pre-norm sublayer outputs, rotating, ENC_SLABS slabs each (encoder_small.hpp: the FFN-down GEMM splits K over work-groups)
    float *x = nullptr, *qkv = nullptr, *ctx = nullptr, *mid = nullptr, *pooled = nullptr, *sA = nullptr, *sB = nullptr;
    int *d_meta = nullptr;
    int *h_meta = nullptr;       // pinned: the descriptor the graph's first node copies
    float *h_out = nullptr;      // pinned: the pooled rows the graph's last node fills
    hipStream_t cap_stream = nullptr;
    hipEvent_t ev_done = nullptr;   // behind every launch: the next call may rewrite h_meta only after the copy node has run
    bool ev_pending = false;
    // the workspaces (small and batch form) belong to the handle, not to a stream: a call that returned without waiting (device
    // output) leaves ev_tail behind its last launch, and a call on ANOTHER stream waits for it on the device before its first
    hipEvent_t ev_tail = nullptr;
    hipStream_t tail_stream = nullptr;
    bool tail_pending = false;
    static constexpr int NBUCKET = 6;   // 16, 32, 64, 128, 256, 512 tokens
    hipGraphExec_t exec[NBUCKET][2][2][2][2] = {};   // [bucket][pooling][normalize][one sequence][with the descriptor / result copy nodes]
    // icd_encoder_encode_many: the descriptors of consecutive calls come from a ring of pinned blocks, copied to d_meta on the
    // caller's stream in front of a graph WITHOUT copy nodes - the host fills call i + 1 ... i + RING - 1 while call i runs
    static constexpr int RING = 8;
    int *h_ring[RING] = {};
    hipEvent_t ev_ring[RING] = {};      // recorded behind the H2D copy of the slot: the slot may be refilled once it has run
    bool ring_pending[RING] = {};
    // the batch form's workspace (encoder_big.hpp): allocated at the first icd_encoder_encode_many of more than one small call
    struct Big {
        float *y[3] = {nullptr, nullptr, nullptr}, *x = nullptr, *qkv = nullptr, *ctx = nullptr, *mid = nullptr, *pooled = nullptr, *sA = nullptr, *sB = nullptr;
        int *d_meta = nullptr;
        int *h_meta[2] = {nullptr, nullptr};          // pinned descriptors of consecutive passes
        hipEvent_t ev[2] = {nullptr, nullptr};        // behind the H2D copy of each
        bool pending[2] = {false, false};
        bool ready = false;
    } big;
    unsigned long long *stamps = nullptr;   // diagnostic builds (ICD_ABLATE, env ICD_ENC_STAMPS=1): [4 GEMMs of layer 0][16] clock stamps
    std::mutex mu;
};

namespace icd {
constexpr uint64_t ENC_MAGIC = 0x49434445454e4331ull;
inline bool enc_valid(const icd_encoder *e) { return e && e->magic == ENC_MAGIC; }

// the launches of one forward on stream s (inside a capture): the descriptor H2D in front, the pooled rows' D2H behind
constexpr int ENC_SLABS = 4;   // K slices of the FFN-down GEMM = slabs of its output
// ITER: 16-column k-steps per wave (12: hidden 768 / inter 3 072, 16: hidden 1 024 / inter 4 096); NV = hidden / 256
template <int ITER, int NV, bool BF>
inline int enc_enqueue_t(icd_encoder *e, int bucket_tokens, int pooling, int normalize, bool single, bool copies, hipStream_t s) {
    constexpr int KW = 16 * ITER;
    const icd_encoder_desc &d = e->d;
    const int H = d.hidden, I = d.inter;
    const long long slab = (long long)ENC_TMAX * H;
    const int tiles = bucket_tokens / 16;   // grid.y of the GEMMs: the 16-token tiles of the bucket side by side
    if (copies) HIP_TRY(hipMemcpyAsync(e->d_meta, e->h_meta, ENC_META_WORDS * sizeof(int), hipMemcpyHostToDevice, s));
    {
        EncEmbedArgs a{};
        a.meta = e->d_meta; a.word = d.word_emb; a.pos = d.pos_emb; a.type0 = d.type_emb0; a.H = H; a.KW = KW; a.y = e->yb[0];
        hipLaunchKernelGGL(enc_embed_kernel<NV>, dim3((bucket_tokens + 3) / 4), dim3(256), 0, s, a);
    }
    // y[cur]: the PRE-norm output of the previous sublayer, (pg, pb) the LayerNorm it still has to go through
    int cur = 0;
    const float *pg = d.emb_ln_g, *pb = d.emb_ln_b;
    for (int l = 0; l < d.layers; ++l) {
        float *y0 = e->yb[cur], *y1 = e->yb[(cur + 1) % 3], *y2 = e->yb[(cur + 2) % 3];
        {   // Q | K | V = LayerNorm(y0) Wqkv^T + b; leaves LayerNorm's statistics in sA
            EncLinearArgs a{};
            a.meta = e->d_meta; a.x = y0; a.ln_eps = d.ln_eps; a.stats_out = e->sA;
            a.w = e->w_qkv[l]; a.c1 = e->c1_qkv[l]; a.bias = e->c2_qkv[l]; a.y = e->qkv; a.K = H; a.N = 3 * H; a.stamps = (e->stamps && l == d.layers - 1) ? e->stamps : nullptr;
            a.nwk = H / KW; a.slab = slab; a.res_nslab = 1;
            if (l == 0) hipLaunchKernelGGL((enc_linear_kernel<ITER, 16, 0, true, false, 4, 1, BF>), dim3(3 * H / 16, tiles), dim3(64 * (H / KW)), 0, s, a);   // (the embedding sum: one slab)
            else hipLaunchKernelGGL((enc_linear_kernel<ITER, 16, 0, true, false, 4, ENC_SLABS, BF>), dim3(3 * H / 16, tiles), dim3(64 * (H / KW)), 0, s, a);
        }
        {
            EncAttnArgs a{};
            a.meta = e->d_meta; a.qkv = e->qkv; a.out = e->ctx; a.H = H; a.heads = d.heads; a.KW = KW; a.scale = 0.125f;
            if (single) hipLaunchKernelGGL(enc_attention_kernel<true>, dim3((bucket_tokens * d.heads + 3) / 4), dim3(256), 0, s, a);
            else hipLaunchKernelGGL(enc_attention_kernel<false>, dim3((bucket_tokens * d.heads + 3) / 4), dim3(256), 0, s, a);
        }
        {   // y1 = ctx Wo^T + b + LayerNorm(y0)      (BertSelfOutput in front of its LayerNorm)
            EncLinearArgs a{};
            a.meta = e->d_meta; a.x = e->ctx; a.w = e->w_ao[l]; a.bias = e->b_ao[l];
            a.res_src = y0; a.res_stats = e->sA; a.res_g = pg; a.res_b = pb; a.y = y1; a.K = H; a.N = H; a.stamps = (e->stamps && l == d.layers - 1) ? e->stamps + 16 : nullptr;
            a.nwk = H / KW; a.slab = slab; a.res_nslab = l == 0 ? 1 : ENC_SLABS;
            hipLaunchKernelGGL((enc_linear_kernel<ITER, 8, 2, false, true, 4, 1, BF>), dim3(H / 8, tiles), dim3(64 * (H / KW)), 0, s, a);
        }
        {   // mid = GELU(LayerNorm1(y1) Wup^T + b); statistics of LayerNorm1 in sB
            EncLinearArgs a{};
            a.meta = e->d_meta; a.x = y1; a.ln_eps = d.ln_eps; a.stats_out = e->sB;
            a.w = e->w_up[l]; a.c1 = e->c1_up[l]; a.bias = e->c2_up[l]; a.y = e->mid; a.K = H; a.N = I; a.stamps = (e->stamps && l == d.layers - 1) ? e->stamps + 32 : nullptr;
            a.nwk = H / KW; a.slab = slab; a.res_nslab = 1;
            hipLaunchKernelGGL((enc_linear_kernel<ITER, 16, 1, true, true, 4, 1, BF>), dim3(I / 16, tiles), dim3(64 * (H / KW)), 0, s, a);
        }
        {   // y2 = mid Wdown^T + b + LayerNorm1(y1)   (BertOutput in front of its LayerNorm)
            EncLinearArgs a{};
            a.meta = e->d_meta; a.x = e->mid; a.w = e->w_down[l]; a.bias = e->b_down[l];
            a.res_src = y1; a.res_stats = e->sB; a.res_g = e->ln1_g[l]; a.res_b = e->ln1_b[l]; a.y = y2; a.K = I; a.N = H; a.stamps = (e->stamps && l == d.layers - 1) ? e->stamps + 48 : nullptr;
            // K = inter split over ENC_SLABS work-groups of (inter / 192 / ENC_SLABS) waves per 16 output columns: partial sums into the slabs of y2
            a.nwk = I / KW; a.slab = slab; a.res_nslab = 1;
            // (the last layer's output feeds the pooling kernel only: row-major)
            if (l == d.layers - 1) hipLaunchKernelGGL((enc_linear_kernel<ITER, 16, 2, false, false, 4, 1, BF>), dim3(H / 16 * ENC_SLABS, tiles), dim3(64 * (I / KW / ENC_SLABS)), 0, s, a);
            else hipLaunchKernelGGL((enc_linear_kernel<ITER, 16, 2, false, true, 4, 1, BF>), dim3(H / 16 * ENC_SLABS, tiles), dim3(64 * (I / KW / ENC_SLABS)), 0, s, a);
        }
        cur = (cur + 2) % 3;
        pg = e->ln2_g[l]; pb = e->ln2_b[l];
    }
    {   // the last LayerNorm, pooling, normalisation; the last hidden state of every token into x
        EncPoolArgs a{};
        a.meta = e->d_meta; a.y = e->yb[cur]; a.g = pg; a.b = pb; a.eps = d.ln_eps; a.H = H; a.KW = KW; a.pooling = pooling; a.normalize = normalize;
        a.out = e->pooled; a.hidden = e->x; a.slab = slab;
        hipLaunchKernelGGL((enc_pool_kernel<NV, ENC_SLABS>), dim3(ENC_BMAX), dim3(ENC_POOL_WAVES * 64), 0, s, a);
    }
    HIP_TRY(hipGetLastError());
    if (copies) HIP_TRY(hipMemcpyAsync(e->h_out, e->pooled, (size_t)ENC_BMAX * H * sizeof(float), hipMemcpyDeviceToHost, s));
    return ICD_OK;
}

inline int enc_enqueue(icd_encoder *e, int bucket_tokens, int pooling, int normalize, bool single, bool copies, hipStream_t s) {
    if (e->d.arithmetic == ICD_ENCODER_ARITH_BF16X3)
        return e->d.hidden == 1024 ? enc_enqueue_t<16, 4, true>(e, bucket_tokens, pooling, normalize, single, copies, s)
                                   : enc_enqueue_t<12, 3, true>(e, bucket_tokens, pooling, normalize, single, copies, s);
    return e->d.hidden == 1024 ? enc_enqueue_t<16, 4, false>(e, bucket_tokens, pooling, normalize, single, copies, s)
                               : enc_enqueue_t<12, 3, false>(e, bucket_tokens, pooling, normalize, single, copies, s);
}

// the token bucket of a call (index into icd_encoder::exec; its launches cover 16 << index tokens)
inline int enc_bucket(int T) {
    int bi = 0;
    while ((16 << bi) < T) ++bi;
    return bi;
}

// the graph of (bucket, pooling, normalize, one sequence, with / without the copy nodes): captured at first use on the private stream
inline int enc_graph(icd_encoder *e, int bi, int pooling, int normalize, bool single, bool copies, hipGraphExec_t *out) {
    hipGraphExec_t &gx = e->exec[bi][pooling][normalize][single ? 1 : 0][copies ? 1 : 0];
    if (!gx) {
        hipGraph_t g = nullptr;
        HIP_TRY(hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeThreadLocal));
        const int rc = enc_enqueue(e, 16 << bi, pooling, normalize, single, copies, e->cap_stream);
        const hipError_t ec = hipStreamEndCapture(e->cap_stream, &g);
        if (rc) { if (g) hipGraphDestroy(g); return rc; }
        HIP_TRY(ec);
        const hipError_t ei = hipGraphInstantiate(&gx, g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
        if (ei != hipSuccess) { gx = nullptr; return fail(ICD_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(ei)); }
    }
    *out = gx;
    return ICD_OK;
}

// a call on stream s after one that left work behind on another stream (see icd_encoder::ev_tail); enc_tail_mark: behind the last launch
inline int enc_tail_wait(icd_encoder *e, hipStream_t s) {
    if (e->tail_pending && e->tail_stream != s) HIP_TRY(hipStreamWaitEvent(s, e->ev_tail, 0));
    return ICD_OK;
}
inline int enc_tail_mark(icd_encoder *e, hipStream_t s, bool synchronised) {
    if (synchronised) { e->tail_pending = false; return ICD_OK; }   // (the stream was drained, and with it whatever it waited for)
    HIP_TRY(hipEventRecord(e->ev_tail, s));
    e->tail_stream = s; e->tail_pending = true;
    return ICD_OK;
}

// the descriptor of one call (encoder_small.hpp, EncMeta): `nb` sequences of `lengths`, their ids back to back at `ids`
template <typename M>
inline void enc_fill_meta(int *m, const icd_encoder_desc &d, const int32_t *ids, const int32_t *lengths, int nb, int T) {
    m[0] = T; m[1] = nb;
    int t = 0;
    for (int b = 0; b < nb; ++b) {
        m[M::STARTS + b] = t;
        const int first = t;
        for (int i = 0; i < lengths[b]; ++i, ++t) { m[M::IDS + t] = ids[t]; m[M::POS + t] = d.pos_offset + i; m[M::TOK_R0 + t] = first; m[M::TOK_LEN + t] = lengths[b]; }
    }
    for (int b = nb; b <= M::B_MAX; ++b) m[M::STARTS + b] = T;
    for (int u = T; u < M::T_MAX; ++u) { m[M::TOK_R0 + u] = 0; m[M::TOK_LEN + u] = 0; }
}

// ---- the batch form: one pass over T <= ENC_BIG_TMAX tokens / nb <= ENC_BIG_BMAX sequences whose descriptor is in big.d_meta ----
// ONE work-group per CU for the GEMMs: the four waves of a work-group share their W fragments through the CU's L1 (32 KB); a second
// work-group's fragments evict them (measured, per layer and 8 192 tokens: QKV 333 -> 254 us, FFN up 457 -> 365, FFN down 447 -> 354,
// attention output 141 -> 104, profiles/r06_encoder_big_kernel_stats.log). The kernels use no LDS: asking for more than half of a
// CU's 160 KB as dynamic LDS is what keeps the second work-group away.
constexpr size_t ENC_BIG_LDS_PIN = 84 * 1024;
// TM: 16-token tiles per wave; TNQ / TNU / TNO: 16-column tiles per wave of the QKV, FFN-up and the two hidden-sized-output GEMMs (the
// column groups per GEMM a multiple of the 8 XCDs: the hardware deals work-groups round-robin over them, so a column group's W
// tiles stay in ONE XCD's L2); PF: pairs of k-steps in flight
template <int ITER, int NV, int TM, int TNQ, int TNU, int TNO, int PF, bool BF = false>
inline int enc_big_enqueue_t(icd_encoder *e, int T, int nb, int pooling, int normalize, hipStream_t s) {
    constexpr int KW = 16 * ITER;
    using M = EncMetaBig;
    const icd_encoder_desc &d = e->d;
    icd_encoder::Big &g = e->big;
    const int H = d.hidden, I = d.inter;
    const int tiles = (T + 15) / 16;
    const unsigned gy = (unsigned)((tiles + 4 * TM - 1) / (4 * TM));
    size_t lds_pin = ENC_BIG_LDS_PIN;
#ifdef ICD_ABLATE
    if (const char *v = getenv("ICD_ENCBIG_LDS")) lds_pin = (size_t)atoi(v);   // A/B: 0 = as many work-groups per CU as the registers allow
#endif
    auto k_qkv = enc_linear_big_kernel<ITER, TM, TNQ, 0, true, false, 1, PF, BF>;
    auto k_ao = enc_linear_big_kernel<ITER, TM, TNO, 2, false, true, 1, PF, BF>;
    auto k_up = enc_linear_big_kernel<ITER, TM, TNU, 1, true, true, 1, PF, BF>;
    auto k_down = enc_linear_big_kernel<ITER, TM, TNO, 2, false, true, ENC_SLABS, PF, BF>;
    auto k_down_last = enc_linear_big_kernel<ITER, TM, TNO, 2, false, false, ENC_SLABS, PF, BF>;   // the last layer's: row-major, for the pooling kernel
    {
        static int c0[MAX_DEVICES] = {}, c1[MAX_DEVICES] = {}, c2[MAX_DEVICES] = {}, c3[MAX_DEVICES] = {}, c4[MAX_DEVICES] = {};   // (per instantiation and device; calls on a handle are serialised)
        HIP_TRY(ensure_dynamic_lds(k_qkv, e->device, lds_pin, c0)); HIP_TRY(ensure_dynamic_lds(k_ao, e->device, lds_pin, c1));
        HIP_TRY(ensure_dynamic_lds(k_up, e->device, lds_pin, c2)); HIP_TRY(ensure_dynamic_lds(k_down, e->device, lds_pin, c3));
        HIP_TRY(ensure_dynamic_lds(k_down_last, e->device, lds_pin, c4));
    }
    {
        EncEmbedArgs a{};
        a.meta = g.d_meta; a.word = d.word_emb; a.pos = d.pos_emb; a.type0 = d.type_emb0; a.H = H; a.KW = KW; a.y = g.y[0];
        hipLaunchKernelGGL((enc_embed_kernel<NV, M>), dim3((T + 3) / 4), dim3(256), 0, s, a);
    }
    int cur = 0;
    const float *pg = d.emb_ln_g, *pb = d.emb_ln_b;
    auto stats = [&](const float *x, float *out) {
        EncStatsArgs a{};
        a.x = x; a.stats = out; a.T = T; a.K = H; a.eps = d.ln_eps;
        hipLaunchKernelGGL((enc_ln_stats_kernel<ITER>), dim3((tiles + 3) / 4), dim3(256), 0, s, a);
    };
    for (int l = 0; l < d.layers; ++l) {
        float *y0 = g.y[cur], *y1 = g.y[(cur + 1) % 3], *y2 = g.y[(cur + 2) % 3];
        stats(y0, g.sA);
        {   // Q | K | V = LayerNorm(y0) Wqkv^T + b
            EncBigLinearArgs a{};
            a.x = y0; a.stats = g.sA; a.c1 = e->c1_qkv[l]; a.w = e->w_qkv[l]; a.NT = 16; a.bias = e->c2_qkv[l]; a.y = g.qkv;
            a.T = T; a.K = H; a.N = 3 * H; a.pps = H / KW; a.stamps = (e->stamps && l == d.layers - 1) ? e->stamps : nullptr;
            hipLaunchKernelGGL(k_qkv, dim3(3 * H / (16 * TNQ), gy), dim3(256), lds_pin, s, a);
        }
        {
            // (one wave per (token, head): the attention is VALU work - a wave-wide sum per key and token - and 100 000 short waves
            //  balance over the SIMDs where 5 000 long ones, one per (sequence, head) with the keys kept in registers, do not:
            //  205 against 495 us per layer and 8 192 tokens, profiles/r06_encoder_big_kernel_stats.log)
            EncAttnArgs a{};
            a.meta = g.d_meta; a.qkv = g.qkv; a.out = g.ctx; a.H = H; a.heads = d.heads; a.KW = KW; a.scale = 0.125f;
            int tpw = 4;
#ifdef ICD_ABLATE
            if (const char *v = getenv("ICD_ENCBIG_TPW")) tpw = atoi(v);   // A/B: tokens per wave of the attention (1: the small form's kernel)
#endif
            const size_t groups = ((size_t)T + tpw - 1) / (size_t)std::max(1, tpw);
            if (tpw == 4) hipLaunchKernelGGL((enc_attention_group_kernel<M, 4>), dim3((unsigned)((groups * d.heads + 3) / 4)), dim3(256), 0, s, a);
            else if (tpw == 2) hipLaunchKernelGGL((enc_attention_group_kernel<M, 2>), dim3((unsigned)((groups * d.heads + 3) / 4)), dim3(256), 0, s, a);
            else if (tpw == 8) hipLaunchKernelGGL((enc_attention_group_kernel<M, 8>), dim3((unsigned)((groups * d.heads + 3) / 4)), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((enc_attention_kernel<false, M>), dim3((unsigned)(((size_t)T * d.heads + 3) / 4)), dim3(256), 0, s, a);
        }
        {   // y1 = ctx Wo^T + b + LayerNorm(y0)
            EncBigLinearArgs a{};
            a.x = g.ctx; a.w = e->w_ao[l]; a.NT = 8; a.bias = e->b_ao[l];
            a.res_src = y0; a.res_stats = g.sA; a.res_g = pg; a.res_b = pb; a.y = y1; a.T = T; a.K = H; a.N = H; a.pps = H / KW; a.stamps = (e->stamps && l == d.layers - 1) ? e->stamps + 16 : nullptr;
            hipLaunchKernelGGL(k_ao, dim3(H / (16 * TNO), gy), dim3(256), lds_pin, s, a);
        }
        stats(y1, g.sB);
        {   // mid = GELU(LayerNorm1(y1) Wup^T + b)
            EncBigLinearArgs a{};
            a.x = y1; a.stats = g.sB; a.c1 = e->c1_up[l]; a.w = e->w_up[l]; a.NT = 16; a.bias = e->c2_up[l]; a.y = g.mid;
            a.T = T; a.K = H; a.N = I; a.pps = H / KW; a.stamps = (e->stamps && l == d.layers - 1) ? e->stamps + 32 : nullptr;
            hipLaunchKernelGGL(k_up, dim3(I / (16 * TNU), gy), dim3(256), lds_pin, s, a);
        }
        {   // y2 = mid Wdown^T + b + LayerNorm1(y1): the small form's ENC_SLABS K slices, added in slab order
            EncBigLinearArgs a{};
            a.x = g.mid; a.w = e->w_down[l]; a.NT = 16; a.bias = e->b_down[l];
            a.res_src = y1; a.res_stats = g.sB; a.res_g = e->ln1_g[l]; a.res_b = e->ln1_b[l]; a.y = y2; a.T = T; a.K = I; a.N = H; a.pps = I / KW / ENC_SLABS; a.stamps = (e->stamps && l == d.layers - 1) ? e->stamps + 48 : nullptr;
            hipLaunchKernelGGL(l == d.layers - 1 ? k_down_last : k_down, dim3(H / (16 * TNO), gy), dim3(256), lds_pin, s, a);
        }
        cur = (cur + 2) % 3;
        pg = e->ln2_g[l]; pb = e->ln2_b[l];
    }
    {
        EncPoolArgs a{};
        a.meta = g.d_meta; a.y = g.y[cur]; a.g = pg; a.b = pb; a.eps = d.ln_eps; a.H = H; a.KW = KW; a.pooling = pooling; a.normalize = normalize;
        a.out = g.pooled; a.hidden = g.x; a.slab = 0;
        hipLaunchKernelGGL((enc_pool_kernel<NV, 1, M>), dim3(nb), dim3(ENC_POOL_WAVES * 64), 0, s, a);
    }
    HIP_TRY(hipGetLastError());
    return ICD_OK;
}

inline int enc_big_enqueue(icd_encoder *e, int T, int nb, int pooling, int normalize, hipStream_t s) {
    // tiles of 16 tokens per wave: as many as still leave every CU a work-group of the narrowest GEMM (16 column groups x T / (64 TM) rows)
    const int tm = T >= 4096 ? 4 : (T >= 2048 ? 2 : 1);
#ifdef ICD_ABLATE
    if (const char *v = getenv("ICD_ENCBIG_VAR")) {   // A/B: tile shapes and prefetch depth of the batch form's GEMMs (profiles/r06_encoder_big_sweep.log)
        switch (atoi(v)) {
        case 0: return enc_big_enqueue_t<12, 3, 2, 3, 4, 2, 1>(e, T, nb, pooling, normalize, s);
        case 1: return enc_big_enqueue_t<12, 3, 2, 3, 4, 3, 1>(e, T, nb, pooling, normalize, s);
        case 2: return enc_big_enqueue_t<12, 3, 4, 3, 4, 3, 1>(e, T, nb, pooling, normalize, s);
        case 3: return enc_big_enqueue_t<12, 3, 4, 3, 4, 2, 1>(e, T, nb, pooling, normalize, s);
        case 4: return enc_big_enqueue_t<12, 3, 2, 3, 4, 2, 2>(e, T, nb, pooling, normalize, s);
        case 5: return enc_big_enqueue_t<12, 3, 4, 3, 3, 3, 2>(e, T, nb, pooling, normalize, s);
        case 6: return enc_big_enqueue_t<12, 3, 2, 3, 4, 3, 2>(e, T, nb, pooling, normalize, s);
        default: break;
        }
    }
#endif
    if (e->d.arithmetic == ICD_ENCODER_ARITH_BF16X3) {
        // The split-bf16 arithmetic: the matrix time is a fifth of the fp32 form's, the operand supply per MFMA cycle five times -
        // the GEMMs are bound by the L1 (every wave loads its A and W fragments itself). Shapes measured on the diagnostic build
        // (profiles/r06_encoder_big_bf_sweep.log): two token tiles x SIX column tiles per wave (24 / 32 / 8 column groups per GEMM)
        // 104.1 ms per 4 000 strings against 110.8 for the fp32 form's shapes (4 x 3 / 4 / 3); W through LDS: 146-174 (not kept).
        if (e->d.hidden == 1024) {   // (inter 4 096 = 256 column tiles: no groups of six)
            if (tm == 4) return enc_big_enqueue_t<16, 4, 4, 3, 4, 2, 1, true>(e, T, nb, pooling, normalize, s);
            if (tm == 2) return enc_big_enqueue_t<16, 4, 2, 3, 4, 2, 1, true>(e, T, nb, pooling, normalize, s);
            return enc_big_enqueue_t<16, 4, 1, 3, 4, 2, 1, true>(e, T, nb, pooling, normalize, s);
        }
#ifdef ICD_ABLATE
        if (const char *v = getenv("ICD_ENCBIG_BF_VAR")) {   // A/B: tile shapes of the split-bf16 form (hidden 768)
            switch (atoi(v)) {
            case 1: return enc_big_enqueue_t<12, 3, 2, 3, 4, 3, 1, true>(e, T, nb, pooling, normalize, s);
            case 2: return enc_big_enqueue_t<12, 3, 4, 3, 4, 3, 1, true>(e, T, nb, pooling, normalize, s);
            case 3: return enc_big_enqueue_t<12, 3, 2, 6, 4, 3, 1, true>(e, T, nb, pooling, normalize, s);
            default: break;
            }
        }
#endif
        if (tm >= 2) return enc_big_enqueue_t<12, 3, 2, 6, 6, 6, 1, true>(e, T, nb, pooling, normalize, s);
        return enc_big_enqueue_t<12, 3, 1, 6, 6, 6, 1, true>(e, T, nb, pooling, normalize, s);
    }
    // Tile shapes (profiles/r06_encoder_big_sweep.log: seven shapes and prefetch depths within 5 % of each other once a CU holds one
    // work-group - the GEMMs run at 0.65-0.75 of the 157 TFLOP/s fp32 MFMA peak, which is what the fp32-MFMA search kernel reaches too)
    if (e->d.hidden == 1024) {
        if (tm == 4) return enc_big_enqueue_t<16, 4, 4, 3, 4, 2, 1>(e, T, nb, pooling, normalize, s);
        if (tm == 2) return enc_big_enqueue_t<16, 4, 2, 3, 4, 2, 1>(e, T, nb, pooling, normalize, s);
        return enc_big_enqueue_t<16, 4, 1, 3, 4, 2, 1>(e, T, nb, pooling, normalize, s);
    }
    if (tm == 4) return enc_big_enqueue_t<12, 3, 4, 3, 4, 3, 1>(e, T, nb, pooling, normalize, s);
    if (tm == 2) return enc_big_enqueue_t<12, 3, 2, 3, 4, 3, 1>(e, T, nb, pooling, normalize, s);
    return enc_big_enqueue_t<12, 3, 1, 3, 4, 3, 1>(e, T, nb, pooling, normalize, s);
}

// the batch form's buffers: activations of ENC_BIG_TMAX tokens (+ the rows a last partial work-group reads), ~0.5 GB at hidden 768
inline int enc_big_alloc(icd_encoder *e) {
    icd_encoder::Big &g = e->big;
    if (g.ready) return ICD_OK;
    const size_t H = (size_t)e->d.hidden, I = (size_t)e->d.inter, T = (size_t)ENC_BIG_TMAX + 128;   // (+ the rows a last partial work-group of 64 TM tokens reads)
    struct { float **p; size_t n; } bufs[] = {{&g.y[0], T * H}, {&g.y[1], T * H}, {&g.y[2], T * H}, {&g.x, T * H}, {&g.qkv, T * 3 * H}, {&g.ctx, T * H},
                                              {&g.mid, T * I}, {&g.pooled, (size_t)ENC_BIG_BMAX * H}, {&g.sA, 2 * T}, {&g.sB, 2 * T}};
    for (auto &b : bufs) {
        if (*b.p) continue;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(b.p), b.n * sizeof(float)));
        HIP_TRY(hipMemset(*b.p, 0, b.n * sizeof(float)));   // (rows past a pass's tokens are read by its last tiles: finite, never stored)
    }
    if (!g.d_meta) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g.d_meta), EncMetaBig::WORDS * sizeof(int)));
    for (int r = 0; r < 2; ++r) {
        if (!g.h_meta[r]) HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&g.h_meta[r]), EncMetaBig::WORDS * sizeof(int), hipHostMallocDefault));
        if (!g.ev[r]) HIP_TRY(hipEventCreateWithFlags(&g.ev[r], hipEventDisableTiming));
    }
    HIP_TRY(hipDeviceSynchronize());
    g.ready = true;
    return ICD_OK;
}

inline void enc_free(icd_encoder *e) {
    for (auto &b : e->exec) for (auto &p : b) for (auto &n : p) for (auto &o : n) for (auto &g : o) if (g) hipGraphExecDestroy(g);
    for (float *p : {e->big.y[0], e->big.y[1], e->big.y[2], e->big.x, e->big.qkv, e->big.ctx, e->big.mid, e->big.pooled, e->big.sA, e->big.sB}) if (p) hipFree(p);
    if (e->big.d_meta) hipFree(e->big.d_meta);
    for (int *p : e->big.h_meta) if (p) hipHostFree(p);
    for (hipEvent_t ev : e->big.ev) if (ev) hipEventDestroy(ev);
    for (int *p : e->h_ring) if (p) hipHostFree(p);
    for (hipEvent_t ev : e->ev_ring) if (ev) hipEventDestroy(ev);
    for (float *p : {e->yb[0], e->yb[1], e->yb[2], e->x, e->qkv, e->ctx, e->mid, e->pooled, e->sA, e->sB}) if (p) hipFree(p);
    for (auto *v : {&e->w_qkv, &e->w_ao, &e->w_up, &e->w_down, &e->c1_qkv, &e->c2_qkv, &e->c1_up, &e->c2_up}) for (float *p : *v) if (p) hipFree(p);
    if (e->d_meta) hipFree(e->d_meta);
    if (e->stamps) hipFree(e->stamps);
    if (e->h_meta) hipHostFree(e->h_meta);
    if (e->h_out) hipHostFree(e->h_out);
    if (e->ev_done) hipEventDestroy(e->ev_done);
    if (e->ev_tail) hipEventDestroy(e->ev_tail);
    if (e->cap_stream) hipStreamDestroy(e->cap_stream);
    e->magic = 0;
    delete e;
}
}  // namespace icd

extern "C" {

int icd_encoder_create(int32_t device, const icd_encoder_desc *desc, icd_encoder **out) {
    using namespace icd;
    if (!out) return fail(ICD_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!desc) return fail(ICD_ERR_INVALID, "desc is NULL");
    const icd_encoder_desc &d = *desc;
    if (d.layers < 1 || d.layers > 48) return fail(ICD_ERR_INVALID, "layers=%d", d.layers);
    if (d.heads < 1 || d.hidden != d.heads * ATT_HEAD_DIM) return fail(ICD_ERR_UNSUPPORTED, "hidden=%d heads=%d (this encoder is written for 64-wide heads)", d.hidden, d.heads);
    // enc_linear_kernel: four waves cover the hidden size (192 or 256 columns of K each); embed / pool: 3 or 4 x 256 columns
    if (d.hidden != 768 && d.hidden != 1024) return fail(ICD_ERR_UNSUPPORTED, "hidden=%d (the small-input encoder is instantiated for 768 and 1024)", d.hidden);
    const int kw = d.hidden == 1024 ? 256 : 192;   // columns of K per wave (four waves cover the hidden size)
    if (d.inter < 4 * kw || d.inter % (4 * kw) != 0 || d.inter > 16 * kw) return fail(ICD_ERR_UNSUPPORTED, "inter=%d (a multiple of %d, at most %d: four K slices of at most four waves)", d.inter, 4 * kw, 16 * kw);
    if (d.vocab < 1 || d.max_pos < 1 || d.pos_offset < 0 || d.pos_offset >= d.max_pos) return fail(ICD_ERR_INVALID, "vocab=%d max_pos=%d pos_offset=%d", d.vocab, d.max_pos, d.pos_offset);
    if (!(d.ln_eps > 0.0f)) return fail(ICD_ERR_INVALID, "ln_eps=%g", (double)d.ln_eps);
    if (d.arithmetic != ICD_ENCODER_ARITH_FP32 && d.arithmetic != ICD_ENCODER_ARITH_BF16X3) return fail(ICD_ERR_INVALID, "arithmetic=%d (ICD_ENCODER_ARITH_FP32 or _BF16X3)", d.arithmetic);
    if (!d.word_emb || !d.pos_emb || !d.type_emb0 || !d.emb_ln_g || !d.emb_ln_b) return fail(ICD_ERR_INVALID, "an embedding pointer is NULL");
    const float *const *arrs[12] = {d.w_qkv, d.b_qkv, d.w_ao, d.b_ao, d.ln1_g, d.ln1_b, d.w_up, d.b_up, d.w_down, d.b_down, d.ln2_g, d.ln2_b};
    for (auto a : arrs) {
        if (!a) return fail(ICD_ERR_INVALID, "a per-layer pointer array is NULL");
        for (int l = 0; l < d.layers; ++l)
            if (!a[l] || (reinterpret_cast<uintptr_t>(a[l]) & 15) != 0) return fail(ICD_ERR_INVALID, "layer %d: a weight pointer is NULL or not 16-byte aligned", l);
    }
    for (const float *p : {d.word_emb, d.pos_emb, d.type_emb0, d.emb_ln_g, d.emb_ln_b})
        if ((reinterpret_cast<uintptr_t>(p) & 15) != 0) return fail(ICD_ERR_INVALID, "an embedding pointer is not 16-byte aligned");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(ICD_ERR_INVALID, "device=%d of %d", device, ndev);
    HIP_TRY(hipSetDevice(device));
    icd_encoder *e = new (std::nothrow) icd_encoder();
    if (!e) return fail(ICD_ERR_NOMEM, "out of host memory");
    e->magic = ENC_MAGIC; e->device = device; e->d = d;
    auto keep = [&](std::vector<const float *> &v, const float *const *a) { v.assign(a, a + d.layers); };
    keep(e->b_qkv, d.b_qkv); keep(e->b_ao, d.b_ao); keep(e->ln1_g, d.ln1_g); keep(e->ln1_b, d.ln1_b);
    keep(e->b_up, d.b_up); keep(e->b_down, d.b_down); keep(e->ln2_g, d.ln2_g); keep(e->ln2_b, d.ln2_b);
    for (auto *v : {&e->w_qkv, &e->w_ao, &e->w_up, &e->w_down, &e->c1_qkv, &e->c2_qkv, &e->c1_up, &e->c2_up}) v->assign(d.layers, nullptr);
    e->d.w_qkv = e->d.b_qkv = e->d.w_ao = e->d.b_ao = e->d.ln1_g = e->d.ln1_b = e->d.w_up = e->d.b_up = e->d.w_down = e->d.b_down = e->d.ln2_g = e->d.ln2_b = nullptr;   // (the caller's arrays need not outlive this call)
#define ENC_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { enc_free(e); return fail(e_ == hipErrorOutOfMemory ? ICD_ERR_NOMEM : ICD_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } } while (0)
    const size_t H = (size_t)d.hidden, I = (size_t)d.inter, T = ENC_TMAX;
    struct { float **p; size_t n; } bufs[] = {{&e->yb[0], ENC_SLABS * T * H}, {&e->yb[1], ENC_SLABS * T * H}, {&e->yb[2], ENC_SLABS * T * H}, {&e->x, T * H}, {&e->qkv, (T + 32) * 3 * H} /* (+ a chunk of keys past the last sequence: loaded, masked) */, {&e->ctx, T * H},
                                              {&e->mid, T * I}, {&e->pooled, (size_t)ENC_BMAX * H}, {&e->sA, 2 * T}, {&e->sB, 2 * T}};
    for (auto &b : bufs) {
        ENC_TRY(hipMalloc(reinterpret_cast<void **>(b.p), b.n * sizeof(float)));
        ENC_TRY(hipMemset(*b.p, 0, b.n * sizeof(float)));   // (rows past a call's tokens are read by the last 16-token tile: finite, never stored)
    }
    // the four Linear weights of every layer, copied ONCE into the order their GEMM's loads want (enc_pw: tiles of 16 / 8 / 16 / 16
    // rows); the two that read a LayerNorm's output (QKV: the previous sublayer's, FFN up: LayerNorm1) with it folded in
    {
        auto permute = [&](const float *src, const float *colscale, float **dst, int N, int K, int NT) -> hipError_t {
            hipError_t er = hipMalloc(reinterpret_cast<void **>(dst), (size_t)N * K * sizeof(float));
            if (er != hipSuccess) return er;
            if (d.arithmetic == ICD_ENCODER_ARITH_BF16X3) {   // w_hi / w_lo of every 32-block in the block's two 16-byte slots
                const size_t groups = (size_t)N * K / 8;
                hipLaunchKernelGGL(enc_permute_w_bf16_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, nullptr, src, colscale, *dst, N, K, NT, kw);
                return hipGetLastError();
            }
            const size_t groups = (size_t)N * K / 4;
            hipLaunchKernelGGL(enc_permute_w_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, nullptr, src, colscale, *dst, N, K, NT, kw);
            return hipGetLastError();
        };
        auto fold = [&](const float *w, const float *g, const float *b, const float *bias, float **c1, float **c2, int N, int K) -> hipError_t {
            hipError_t er = hipMalloc(reinterpret_cast<void **>(c1), (size_t)N * sizeof(float));
            if (er != hipSuccess) return er;
            er = hipMalloc(reinterpret_cast<void **>(c2), (size_t)N * sizeof(float));
            if (er != hipSuccess) return er;
            hipLaunchKernelGGL(enc_fold_ln_kernel, dim3((N + 3) / 4), dim3(256), 0, nullptr, w, g, b, bias, *c1, *c2, N, K);
            return hipGetLastError();
        };
        const float *pg = d.emb_ln_g, *pb = d.emb_ln_b;   // the LayerNorm in front of layer l's QKV
        for (int l = 0; l < d.layers; ++l) {
            ENC_TRY(permute(desc->w_qkv[l], pg, &e->w_qkv[l], 3 * d.hidden, d.hidden, 16));
            ENC_TRY(fold(desc->w_qkv[l], pg, pb, desc->b_qkv[l], &e->c1_qkv[l], &e->c2_qkv[l], 3 * d.hidden, d.hidden));
            ENC_TRY(permute(desc->w_ao[l], nullptr, &e->w_ao[l], d.hidden, d.hidden, 8));
            ENC_TRY(permute(desc->w_up[l], desc->ln1_g[l], &e->w_up[l], d.inter, d.hidden, 16));
            ENC_TRY(fold(desc->w_up[l], desc->ln1_g[l], desc->ln1_b[l], desc->b_up[l], &e->c1_up[l], &e->c2_up[l], d.inter, d.hidden));
            ENC_TRY(permute(desc->w_down[l], nullptr, &e->w_down[l], d.hidden, d.inter, 16));
            pg = desc->ln2_g[l]; pb = desc->ln2_b[l];
        }
    }
    ENC_TRY(hipMalloc(reinterpret_cast<void **>(&e->d_meta), ENC_META_WORDS * sizeof(int)));
    ENC_TRY(hipMemset(e->d_meta, 0, ENC_META_WORDS * sizeof(int)));
    ENC_TRY(hipHostMalloc(reinterpret_cast<void **>(&e->h_meta), ENC_META_WORDS * sizeof(int), hipHostMallocDefault));
    ENC_TRY(hipHostMalloc(reinterpret_cast<void **>(&e->h_out), (size_t)ENC_BMAX * H * sizeof(float), hipHostMallocDefault));
    memset(e->h_meta, 0, ENC_META_WORDS * sizeof(int));
    for (int r = 0; r < icd_encoder::RING; ++r) {
        ENC_TRY(hipHostMalloc(reinterpret_cast<void **>(&e->h_ring[r]), ENC_META_WORDS * sizeof(int), hipHostMallocDefault));
        memset(e->h_ring[r], 0, ENC_META_WORDS * sizeof(int));
        ENC_TRY(hipEventCreateWithFlags(&e->ev_ring[r], hipEventDisableTiming));
    }
#ifdef ICD_ABLATE
    if (getenv("ICD_ENC_STAMPS") && atoi(getenv("ICD_ENC_STAMPS")) == 1) {
        ENC_TRY(hipMalloc(reinterpret_cast<void **>(&e->stamps), 64 * sizeof(unsigned long long)));
        ENC_TRY(hipMemset(e->stamps, 0, 64 * sizeof(unsigned long long)));
    }
#endif
    ENC_TRY(hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking));
    ENC_TRY(hipEventCreateWithFlags(&e->ev_done, hipEventDisableTiming));
    ENC_TRY(hipEventCreateWithFlags(&e->ev_tail, hipEventDisableTiming));
    ENC_TRY(hipDeviceSynchronize());
#undef ENC_TRY
    *out = e;
    return ICD_OK;
}

#ifdef ICD_ABLATE
// diagnostic builds: the 64 clock stamps of layer 6's four GEMMs (encoder_small.hpp ENC_STAMP), after a synchronisation
int icd_debug_encoder_stamps(icd_encoder *e, unsigned long long *out) {
    if (!icd::enc_valid(e) || !e->stamps || !out) return fail(ICD_ERR_STATE, "no stamps (ICD_ENC_STAMPS=1 at create, ABLATE build)");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, e->stamps, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long first[8];
    HIP_TRY(hipMemcpyFromSymbol(first, HIP_SYMBOL(icd::g_enc_first), sizeof first));
    for (int g = 0; g < 4; ++g) { out[g * 16 + 14] = first[2 * g]; out[g * 16 + 15] = first[2 * g + 1]; }   // (slot 7 of every GEMM: its first instruction)
    return ICD_OK;
}
#endif

int icd_encoder_destroy(icd_encoder *e) {
    if (!icd::enc_valid(e)) return fail(ICD_ERR_STATE, "invalid encoder handle");
    hipSetDevice(e->device);
    hipDeviceSynchronize();
    icd::enc_free(e);
    return ICD_OK;
}

int icd_encoder_encode(icd_encoder *e, const int32_t *ids, const int32_t *lengths, int32_t nseq, int32_t pooling, int32_t normalize,
                       float *out, int32_t out_on_device, float *hidden_out, void *stream) {
    using namespace icd;
    if (!enc_valid(e)) return fail(ICD_ERR_STATE, "invalid encoder handle");
    std::lock_guard<std::mutex> guard(e->mu);
    if (nseq < 0 || nseq > ENC_BMAX) return fail(ICD_ERR_INVALID, "nseq=%d (at most %d sequences per call)", nseq, ENC_BMAX);
    if (nseq == 0) return ICD_OK;
    if (!ids || !lengths || !out) return fail(ICD_ERR_INVALID, "pointer is NULL");
    if ((pooling != 0 && pooling != 1) || (normalize != 0 && normalize != 1)) return fail(ICD_ERR_INVALID, "pooling=%d normalize=%d", pooling, normalize);
    const icd_encoder_desc &d = e->d;
    int T = 0;
    for (int b = 0; b < nseq; ++b) {
        if (lengths[b] < 1 || lengths[b] > d.max_pos - d.pos_offset) return fail(ICD_ERR_INVALID, "sequence %d: %d tokens (1 .. %d)", b, lengths[b], d.max_pos - d.pos_offset);
        T += lengths[b];
        if (T > ENC_TMAX) return fail(ICD_ERR_INVALID, "more than %d tokens per call", ENC_TMAX);
    }
    for (int t = 0; t < T; ++t)
        if (ids[t] < 0 || ids[t] >= d.vocab) return fail(ICD_ERR_INVALID, "token %d: id %d outside the vocabulary of %d", t, ids[t], d.vocab);
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = s != nullptr && hipStreamIsCapturing(s, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive;
    if (capturing && !out_on_device) return fail(ICD_ERR_INVALID, "a call with a host output synchronises: it cannot be captured into a graph");
    if (capturing) return fail(ICD_ERR_UNSUPPORTED, "icd_encoder_encode reads its token ids from host memory at call time: it cannot be captured into a graph");
    // the previous launch's copy node may not have read h_meta yet (device outputs: the call did not wait)
    if (e->ev_pending) { HIP_TRY(hipEventSynchronize(e->ev_done)); e->ev_pending = false; }
    enc_fill_meta<EncMetaSmall>(e->h_meta, d, ids, lengths, nseq, T);
    hipGraphExec_t gx = nullptr;
    { const int rc = enc_graph(e, enc_bucket(T), pooling, normalize, nseq == 1, true, &gx); if (rc) return rc; }
    { const int rc = enc_tail_wait(e, s); if (rc) return rc; }
    HIP_TRY(hipGraphLaunch(gx, s));
    const size_t H = (size_t)d.hidden;
    if (hidden_out) HIP_TRY(hipMemcpyAsync(hidden_out, e->x, (size_t)T * H * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (out_on_device) {
        HIP_TRY(hipMemcpyAsync(out, e->pooled, (size_t)nseq * H * sizeof(float), hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipEventRecord(e->ev_done, s));
        e->ev_pending = true;
        return enc_tail_mark(e, s, false);
    }
    HIP_TRY(hipStreamSynchronize(s));
    memcpy(out, e->h_out, (size_t)nseq * H * sizeof(float));
    return enc_tail_mark(e, s, true);
}

int icd_encoder_encode_many(icd_encoder *e, const int32_t *ids, const int32_t *lengths, int64_t nseq, int32_t pooling, int32_t normalize,
                            float *out, int32_t out_on_device, void *stream) {
    using namespace icd;
    if (!enc_valid(e)) return fail(ICD_ERR_STATE, "invalid encoder handle");
    std::lock_guard<std::mutex> guard(e->mu);
    if (nseq < 0) return fail(ICD_ERR_INVALID, "nseq=%lld", (long long)nseq);
    if (nseq == 0) return ICD_OK;
    if (!ids || !lengths || !out) return fail(ICD_ERR_INVALID, "pointer is NULL");
    if ((pooling != 0 && pooling != 1) || (normalize != 0 && normalize != 1)) return fail(ICD_ERR_INVALID, "pooling=%d normalize=%d", pooling, normalize);
    const icd_encoder_desc &d = e->d;
    const int max_len = d.max_pos - d.pos_offset < ENC_TMAX ? d.max_pos - d.pos_offset : ENC_TMAX;
    int64_t total = 0;
    for (int64_t b = 0; b < nseq; ++b) {
        if (lengths[b] < 1 || lengths[b] > max_len) return fail(ICD_ERR_INVALID, "sequence %lld: %d tokens (1 .. %d)", (long long)b, lengths[b], max_len);
        total += lengths[b];
    }
    for (int64_t t = 0; t < total; ++t)
        if (ids[t] < 0 || ids[t] >= d.vocab) return fail(ICD_ERR_INVALID, "token %lld: id %d outside the vocabulary of %d", (long long)t, ids[t], d.vocab);
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (s != nullptr && hipStreamIsCapturing(s, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive)
        return fail(ICD_ERR_UNSUPPORTED, "icd_encoder_encode_many reads its token ids from host memory at call time: it cannot be captured into a graph");
    // (a device-output call of icd_encoder_encode may still be reading h_meta: not touched here; d_meta and the activations are
    //  protected by stream order, across streams by ev_tail)
    { const int rc = enc_tail_wait(e, s); if (rc) return rc; }
    const size_t H = (size_t)d.hidden;
    int64_t b0 = 0, t0 = 0;
    if (total > ENC_TMAX || nseq > ENC_BMAX) {
        // more than one small call: the batch form (encoder_big.hpp) - the same arithmetic per token in large tiles, passes of
        // at most ENC_BIG_TMAX tokens / ENC_BIG_BMAX sequences cut greedily in the given order; the host fills pass i + 1's
        // descriptor while pass i runs
        { const int rc = enc_big_alloc(e); if (rc) return rc; }
        icd_encoder::Big &g = e->big;
        int slot = 0;
        while (b0 < nseq) {
            int nb = 0, T = 0;
            while (b0 + nb < nseq && nb < ENC_BIG_BMAX && T + lengths[b0 + nb] <= ENC_BIG_TMAX) { T += lengths[b0 + nb]; ++nb; }
            if (g.pending[slot]) { HIP_TRY(hipEventSynchronize(g.ev[slot])); g.pending[slot] = false; }
            enc_fill_meta<EncMetaBig>(g.h_meta[slot], d, ids + t0, lengths + b0, nb, T);
            HIP_TRY(hipMemcpyAsync(g.d_meta, g.h_meta[slot], EncMetaBig::WORDS * sizeof(int), hipMemcpyHostToDevice, s));
            HIP_TRY(hipEventRecord(g.ev[slot], s));
            g.pending[slot] = true;
            { const int rc = enc_big_enqueue(e, T, nb, pooling, normalize, s); if (rc) return rc; }
            HIP_TRY(hipMemcpyAsync(out + (size_t)b0 * H, g.pooled, (size_t)nb * H * sizeof(float), out_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
            b0 += nb; t0 += T;
            slot ^= 1;
        }
        if (!out_on_device) HIP_TRY(hipStreamSynchronize(s));
        return enc_tail_mark(e, s, !out_on_device);
    }
    int slot = 0;
    while (b0 < nseq) {
        // greedy, in the given order: as many sequences as fit the largest bucket
        int nb = 0, T = 0;
        while (b0 + nb < nseq && nb < ENC_BMAX && T + lengths[b0 + nb] <= ENC_TMAX) { T += lengths[b0 + nb]; ++nb; }
        if (e->ring_pending[slot]) { HIP_TRY(hipEventSynchronize(e->ev_ring[slot])); e->ring_pending[slot] = false; }
        enc_fill_meta<EncMetaSmall>(e->h_ring[slot], d, ids + t0, lengths + b0, nb, T);
        hipGraphExec_t gx = nullptr;
        { const int rc = enc_graph(e, enc_bucket(T), pooling, normalize, nb == 1, false, &gx); if (rc) return rc; }
        HIP_TRY(hipMemcpyAsync(e->d_meta, e->h_ring[slot], ENC_META_WORDS * sizeof(int), hipMemcpyHostToDevice, s));
        HIP_TRY(hipEventRecord(e->ev_ring[slot], s));
        e->ring_pending[slot] = true;
        HIP_TRY(hipGraphLaunch(gx, s));
        HIP_TRY(hipMemcpyAsync(out + (size_t)b0 * H, e->pooled, (size_t)nb * H * sizeof(float), out_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
        b0 += nb; t0 += T;
        slot = (slot + 1) % icd_encoder::RING;
    }
    if (!out_on_device) HIP_TRY(hipStreamSynchronize(s));
    return enc_tail_mark(e, s, !out_on_device);
}

}  // extern "C"
