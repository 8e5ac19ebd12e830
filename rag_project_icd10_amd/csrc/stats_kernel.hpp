// stats_kernel.hpp — SURVEY.md row N3: the confidence service's score statistics and its semantic-coherence cosine,
// for a whole batch of queries on the device.
//
// Reference (services/multidimensional_confidence_service.py):
//   :936-963    _assess_model_uncertainty      np.mean / np.std / max over the candidates' scores
//   :1087-1099  _calculate_prediction_variance  np.var over the candidates' scores (0.1 for fewer than two)
//   :273-280    semantic_coherence = sklearn cosine_similarity([encode_query(query)], [encode_query(title)])[0][0]
//
// The statistics are IEEE double in numpy's own summation order (float64 add-reduction: fewer than 8 elements left to
// right; 8..128 elements eight strided partial sums combined pairwise, the tail left to right), so the results are
// bit-identical to np.mean / np.var / np.std of the same list (tests/test_confidence_gpu.py); the formulas on top are in
// Python's evaluation order with contraction off. The cosine follows sklearn (each row divided by its Euclidean norm in
// double, then the dot product); its summation order is a wave's, sklearn's is the BLAS's: equal to 1e-14, not bitwise.
#pragma once
#include <hip/hip_runtime.h>

namespace icd {

constexpr int STATS_MAX_K = 128;   // numpy's single pairwise block
constexpr int STATS_OUT = 6;       // mean, std, var, max, model_uncertainty, prediction_variance

struct StatsArgs {
    const double *scores;   // [nq][k]
    const int *order;       // nullable [nq][k]: entry j of a query exists iff order[j] >= 0 (icd_hier_rescore's out_order)
    int nq, k, use;         // statistics over the first min(use, valid) entries of every query
    double *out;            // [nq][STATS_OUT]
};

#pragma clang fp contract(off)
template <typename F>
__device__ __forceinline__ double np_pairwise_sum(int n, F f) {   // n <= 128
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += f(i);
        return res;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = f(j);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += f(i + j);
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += f(i);
    return res;
}

__global__ __launch_bounds__(64) void score_stats_kernel(StatsArgs a) {
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q >= a.nq) return;
    const double *s = a.scores + (size_t)q * a.k;
    int n = min(a.use, a.k);
    if (a.order) {
        const int *o = a.order + (size_t)q * a.k;
        int c = 0;
        while (c < n && o[c] >= 0) ++c;
        n = c;
    }
    double *out = a.out + (size_t)q * STATS_OUT;
    if (n == 0) {   // _assess_model_uncertainty: 0.0 without candidates; _calculate_prediction_variance: 0.1
        out[0] = 0.0; out[1] = 0.0; out[2] = 0.0; out[3] = 0.0; out[4] = 0.0; out[5] = 0.1;
        return;
    }
    const double dn = (double)n;
    const double mean = (0.0 + np_pairwise_sum(n, [&](int i) { return s[i]; })) / dn;
    const double var = (0.0 + np_pairwise_sum(n, [&](int i) { const double d = s[i] - mean; return d * d; })) / dn;
    const double sd = sqrt(var);
    double mx = s[0];
    for (int i = 1; i < n; ++i) mx = s[i] > mx ? s[i] : mx;
    const double uncertainty_score = 1.0 - (sd < 0.5 ? sd : 0.5) / 0.5;
    const double p0 = uncertainty_score * 0.6;
    const double p1 = mx * 0.4;
    const double fin = p0 + p1;
    out[0] = mean; out[1] = sd; out[2] = var; out[3] = mx;
    out[4] = fin < 1.0 ? fin : 1.0;
    out[5] = n > 1 ? var : 0.1;
}

struct CosArgs {
    const float *x;        // [nq][dim]
    const float *y;        // [nq][dim] (y_stride = dim) or one row for every query (y_stride = 0)
    long long y_stride;
    int nq, dim;
    double *out;           // [nq]
};

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// one wave per row
__global__ __launch_bounds__(256) void cosine_rows_kernel(CosArgs a) {
    const int lane = threadIdx.x & 63, q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= a.nq) return;
    const float *x = a.x + (size_t)q * a.dim;
    const float *y = a.y + (size_t)q * a.y_stride;
    double sx = 0.0, sy = 0.0;
    for (int d = lane; d < a.dim; d += 64) {
        const double xv = (double)x[d], yv = (double)y[d];
        sx += xv * xv;
        sy += yv * yv;
    }
    double nx = sqrt(wave_sum_f64(sx)), ny = sqrt(wave_sum_f64(sy));
    if (nx == 0.0) nx = 1.0;   // (sklearn.preprocessing.normalize leaves a zero row alone)
    if (ny == 0.0) ny = 1.0;
    double dot = 0.0;
    for (int d = lane; d < a.dim; d += 64) dot += ((double)x[d] / nx) * ((double)y[d] / ny);
    dot = wave_sum_f64(dot);
    if (lane == 0) a.out[q] = dot;
}
#pragma clang fp contract(fast)

}  // namespace icd
