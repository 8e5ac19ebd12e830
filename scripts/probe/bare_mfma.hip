// bare_mfma.hip — what a pure MFMA stream reaches on this chip, per shape, on random fp16 operands held in registers:
// one wave per SIMD (launch_bounds 256, big register footprint like the coarse kernel), every wave issues the MFMAs of
// NT 128 x 128 x 768 tiles for its 32 queries (384 v_mfma_f32_16x16x32_f16, or 192 v_mfma_f32_32x32x16_f16) and nothing
// else: no LDS, no loads in the loop. The reference line for DESIGN.md section 4.1 ("bare MFMA stream"): the clock the
// chip holds under a 100 % MFMA duty differs by shape (MI355X_MICROARCH.md, DVFS give-back item 7).
//   hipcc --offload-arch=gfx950 -O3 -o bare_mfma bare_mfma.hip && ./bare_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>   // 16: 16x16x32, 32: 32x32x16
__global__ __launch_bounds__(256, 1) void bare(const half8 *src, float *out, int ntiles, unsigned long long *clk) {
    const int tid = threadIdx.x + blockIdx.x * 256;
    // in-kernel clock (MI355X_MICROARCH.md, DVFS give-back item 6): one stamp pair around the whole loop, into a buffer of its own
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    half8 q[48], a[8];
#pragma unroll
    for (int i = 0; i < 48; ++i) q[i] = src[(tid * 48 + i) & 0xFFFF];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = src[(tid * 8 + i + 7777) & 0xFFFF];
    float keep = 0.f;
    for (int t = 0; t < ntiles; ++t) {
        if constexpr (SHAPE == 16) {
            f32x4 acc[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 24; ++ks)           // 24 k-steps of 32
#pragma unroll
                for (int rg = 0; rg < 8; ++rg)        // 8 row groups x 2 query groups
#pragma unroll
                    for (int g = 0; g < 2; ++g)
                        acc[2 * rg + g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rg], q[g * 24 + ks], acc[2 * rg + g], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) keep += acc[i][0] + acc[i][3];
        } else {
            f32x16 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 48; ++ks)           // 48 k-steps of 16
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)        // 4 row tiles of 32
                    acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rt + 4 * (ks & 1)], q[ks], acc[rt], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) keep += acc[i][0] + acc[i][15];
        }
        // rotate the A fragments so that consecutive tiles do not multiply identical operands
        const half8 t0 = a[0];
#pragma unroll
        for (int i = 0; i < 7; ++i) a[i] = a[i + 1];
        a[7] = t0;
    }
    out[tid] = keep;
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (clk && (threadIdx.x & 63) == 0) { clk[2 * (tid >> 6)] = c1 - c0; clk[2 * (tid >> 6) + 1] = r1 - r0; }
}

int main(int argc, char **argv) {
    // usage: bare_mfma [tiles per work-group] [seconds of settling per arm] [shape: 16 | 32 | 0 = both]
    const int nwg = 256, ntiles = argc > 1 ? atoi(argv[1]) : 90, iters = 30;   // 256 work-groups x 90 tiles ~ the bench launch (22 910 tiles)
    const double settle_s = argc > 2 ? atof(argv[2]) : 0.0;   // (clock / power samplers need seconds of one load: gemm_reference.py)
    const int only = argc > 3 ? atoi(argv[3]) : 0;
    std::vector<_Float16> h(65536 * 8);
    srand(1);
    for (auto &x : h) x = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
    half8 *d; float *o; unsigned long long *clk;
    hipMalloc(&d, h.size() * 2); hipMalloc(&o, nwg * 256 * 4); hipMalloc(&clk, nwg * 4 * 2 * 8);
    std::vector<unsigned long long> hclk(nwg * 4 * 2);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep)
        for (int shape : {16, 32}) {
            if (only && shape != only) continue;
            const int settle = settle_s > 0 ? (int)(settle_s / 0.33e-3) : 200;
            for (int w = 0; w < settle; ++w) {   // settle the clock under this load
                if (shape == 16) hipLaunchKernelGGL(bare<16>, dim3(nwg), dim3(256), 0, 0, d, o, ntiles, nullptr);
                else hipLaunchKernelGGL(bare<32>, dim3(nwg), dim3(256), 0, 0, d, o, ntiles, nullptr);
                if ((w & 255) == 255) hipDeviceSynchronize();   // (bounded launch queue)
            }
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int it = 0; it < iters; ++it) {
                if (shape == 16) hipLaunchKernelGGL(bare<16>, dim3(nwg), dim3(256), 0, 0, d, o, ntiles, clk);
                else hipLaunchKernelGGL(bare<32>, dim3(nwg), dim3(256), 0, 0, d, o, ntiles, clk);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = 2.0 * 128 * 128 * 768 * (double)nwg * ntiles * iters;   // 4 waves x 32 queries x 128 rows per tile
            printf("shape %s: %.4f ms per launch of %d tiles per work-group, %.0f TFLOP/s (%.3f of 2500)\n", shape == 16 ? "16x16x32" : "32x32x16",
                   ms / iters, ntiles, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 2500.0);
            hipMemcpy(hclk.data(), clk, hclk.size() * 8, hipMemcpyDeviceToHost);
            std::vector<double> ghz;
            for (size_t w = 0; w < hclk.size() / 2; ++w) if (hclk[2 * w + 1]) ghz.push_back((double)hclk[2 * w] / (double)hclk[2 * w + 1] * 0.1);
            std::sort(ghz.begin(), ghz.end());
            if (!ghz.empty()) printf("   in-kernel clock (s_memtime / s_memrealtime x 100 MHz, %zu waves): median %.3f GHz (min %.3f, max %.3f)\n",
                                     ghz.size(), ghz[ghz.size() / 2], ghz.front(), ghz.back());
        }
    return 0;
}
