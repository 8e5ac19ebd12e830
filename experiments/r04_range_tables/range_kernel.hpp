// range_kernel.hpp — canonical scores of 32 neighbouring queries against a dense range of corpus rows, on the fp32 MFMA.
//
// The rescoring step of the search behind MilvusClient.search (services/milvus_service.py:280-285) on a corpus whose rows
// come in families of near-identical neighbours (tools/build_database.py:156-171): in wide mode a query's rescoring window
// is its whole family, ~124 rows x 3 KB, the SAME rows for the ~33 queries of that family. finalize_kernel walks them once
// per query (3.7 GB per 10 000 queries through the CUs' L1: 0.18 of its 0.24 ms). Here the queries - already ordered family
// by family (order_keys_kernel / order_scatter_kernel, finalize.hpp) - are taken 32 at a time: the group's range starts 128
// rows below the lowest best-candidate row of its queries and ends 128 rows above the highest (at most RANGE_ROWS = 512), and one
// work-group computes the 32 x range score table with v_mfma_f32_32x32x2_f32 - bit for bit the fmaf chain over d ascending,
// as in exact_kernel.hpp - reading every row ONCE per group. finalize then looks a window row inside the range up in the
// table and walks only what lies outside it.
//
#pragma once
#include "topk_select.hpp"

namespace icd {

constexpr int RANGE_TILES = 16;                 // 32-row tiles of a group's range, one wave each
constexpr int RANGE_ROWS = RANGE_TILES * 32;    // 512: two 124-row families side by side and a margin either end
constexpr int RANGE_MARGIN = 128;               // rows below the lowest / above the highest best candidate (a family's size)

struct RangeArgs {
    const float *corpus;    // [n][dim]
    const float *queries;   // [*][dim]
    const int *pos_q;       // [nq] position of the family order -> query
    const int *best_row;    // [query] original row of the query's best coarse candidate (-1: none)
    int nq, n, dim;         // dim a multiple of 32
    int *tab_lo;            // [group] out
    int *tab_tiles;         // [group] out
    float *tab;             // [group][RANGE_ROWS][32] out
};

// Work-group b: tiles 4 (b & 3) ... + 3 of group b >> 2, one tile per wave: a tile's chain is 384 dependent MFMAs whatever
// is done, so the parallelism is across tiles (four or five waves per SIMD). Both operands go through LDS, register-staged and
// double-buffered, one barrier per 32-float stage: the group's 32 queries once per work-group, a wave's 32 rows in its own
// region (single-buffered); the global loads are coalesced (eight lanes per 128-byte line), the fragment reads conflict-free (rows padded to 33
// floats). (Forms measured before this one, per 10 000 queries of the family corpus, against 42 us of MFMA time: one
// work-group per group with up to four tiles per wave and the operands straight from global memory - lane = row, 64 lines
// per load instruction - 187 us; the same with one tile per wave 118 us; with two 137; without run-ahead at eight waves per
// SIMD 163: all bound by the texture-address path, a line per lane.)
__global__ __launch_bounds__(256, 6) void range_scores_kernel(RangeArgs a) {
    constexpr int LDT = 33;
    __shared__ float Qs[2][32 * LDT];
    __shared__ float Cs[4][32 * LDT];   // (a wave's own region: written behind its own reads, LDS serves one wave's operations in order - no second buffer, no barrier)
    const int g = blockIdx.x >> 2, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = (blockIdx.x & 3) * 4 + wave;
    const int c = lane & 31, h = lane >> 5;
    int lo = -1, tiles = 0;
    {
        const int p = g * 32 + c;
        const int q = a.pos_q[min(p, a.nq - 1)];
        const int br = p < a.nq ? a.best_row[q] : -1;
        int mn = br >= 0 ? br : 0x7fffffff, mx = br;
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) {
            mn = min(mn, __shfl_xor(mn, off));
            mx = max(mx, __shfl_xor(mx, off));
        }
        // (both halves of every wave of the group's four work-groups compute the same values)
        if (mx >= 0) {   // (best rows further apart than the range: it covers the lower ones, the others are walked)
            lo = max(0, mn - RANGE_MARGIN) & ~31;
            tiles = min(RANGE_TILES, (min(mx + RANGE_MARGIN, a.n) - lo + 31) >> 5);
        }
    }
    if ((blockIdx.x & 3) == 0 && tid == 0) { a.tab_lo[g] = lo; a.tab_tiles[g] = tiles; }
    if ((int)(blockIdx.x & 3) * 4 >= tiles) return;   // (work-group-uniform: none of its tiles is needed)
    const bool active = tile < tiles;
    const int nks = a.dim >> 5;
    // staging: the work-group's 256 threads load the 32 queries' stage (one float4 each), a wave its own tile's (four each)
    const int qr = tid >> 3, q4 = (tid & 7) * 4;
    const float *qsrc = a.queries + (size_t)a.pos_q[min(g * 32 + qr, a.nq - 1)] * a.dim + q4;
    const int qdst = qr * LDT + q4;
    const float *csrc[4];
    int cdst[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = lane + 64 * i, r = idx >> 3, c4 = (idx & 7) * 4;
        csrc[i] = a.corpus + (size_t)min(lo + tile * 32 + r, a.n - 1) * a.dim + c4;   // (rows past the corpus: a valid row, never looked up)
        cdst[i] = r * LDT + c4;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    float4 qreg, creg[4];
    auto put = [&](int buf) {
        float *d = &Qs[buf][qdst];
        d[0] = qreg.x; d[1] = qreg.y; d[2] = qreg.z; d[3] = qreg.w;
        if (active) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float *e = &Cs[wave][cdst[i]];
                e[0] = creg[i].x; e[1] = creg[i].y; e[2] = creg[i].z; e[3] = creg[i].w;
            }
        }
    };
    qreg = *reinterpret_cast<const float4 *>(qsrc);
    if (active) {
#pragma unroll
        for (int i = 0; i < 4; ++i) creg[i] = *reinterpret_cast<const float4 *>(csrc[i]);
    }
    put(0);
    __syncthreads();
    for (int ks = 0; ks < nks; ++ks) {
        const int cur = ks & 1;
        const bool more = ks + 1 < nks;
        if (more) {
            const int k0 = (ks + 1) * 32;
            qreg = *reinterpret_cast<const float4 *>(qsrc + k0);
            if (active) {
#pragma unroll
                for (int i = 0; i < 4; ++i) creg[i] = *reinterpret_cast<const float4 *>(csrc[i] + k0);
            }
        }
        if (active) {
            const float *qrow = &Qs[cur][c * LDT + h];
            const float *crow = &Cs[wave][c * LDT + h];
#pragma unroll
            for (int sidx = 0; sidx < 16; ++sidx)   // k = 2 s + h, ascending: the chain's order
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(crow[2 * sidx], qrow[2 * sidx], acc, 0, 0, 0);
        }
        if (more) put(cur ^ 1);
        __syncthreads();
    }
    if (active) {   // lane (c, h) holds, for query c, rows (r & 3) + 8 (r >> 2) + 4 h of the tile (topk_select.hpp)
        float *dst = a.tab + ((size_t)g * RANGE_ROWS + (size_t)tile * 32) * 32 + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(size_t)((r & 3) + 8 * (r >> 2) + 4 * h) * 32] = acc[r];
    }
}

}  // namespace icd
