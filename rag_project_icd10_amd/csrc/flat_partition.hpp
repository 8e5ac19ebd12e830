// flat_partition.hpp — index arithmetic of the flat (query tile x corpus tile) partition of coarse_flat_kernel.hpp.
// Plain C++ (no HIP types): shared by the kernel, the host launcher and tests/test_flat_partition.py (g++).
//
// Units are numbered u = mtile * ctiles + tile; work-group w takes [w U, (w+1) U). Its run inside one query tile is cut
// into lists of at most `list_tiles` tiles; ordinals count the lists of a query tile in row order.
#pragma once
#if defined(__HIPCC__) || defined(__CUDACC__)
#define ICD_HD __host__ __device__
#else
#define ICD_HD
#endif

namespace icd {

// number of lists a run of `len` tiles is cut into
ICD_HD inline int flat_lists_of_run(int len, int list_tiles) { return (len + list_tiles - 1) / list_tiles; }

// ordinal (within query tile m) of the first list of work-group w's run: the lists of the earlier work-groups
ICD_HD inline int flat_first_ordinal(int m, int w, int ctiles, int U, int list_tiles) {
    const long long m0 = (long long)m * ctiles, m1 = m0 + ctiles;
    int ord = 0;
    for (long long wp = m0 / U; wp < w; ++wp) {
        const long long r0 = wp * U > m0 ? wp * U : m0;
        const long long r1 = (wp + 1) * U < m1 ? (wp + 1) * U : m1;
        if (r1 > r0) ord += flat_lists_of_run((int)(r1 - r0), list_tiles);
    }
    return ord;
}


// work-group index of hardware block `w` of a grid of G: every XCD (blockIdx mod 8) gets a contiguous stretch of the
// class-major order of the logical indices (class = index mod T: work-groups of a class start on the same corpus
// tile); T <= 0: identity. A bijection of [0, G).
ICD_HD inline int flat_workgroup_of_block(int w, int G, int T) {
    if (T <= 0) return w;
    const int xcd = w & 7, q8 = G >> 3, r8 = G & 7;
    const int jx = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (w >> 3);   // contiguous per XCD
    const int qT = G / T, rT = G - qT * T;
    if (qT == 0) return jx;
    if (jx < rT * (qT + 1)) return jx / (qT + 1) + (jx % (qT + 1)) * T;
    const int jj = jx - rT * (qT + 1);
    return rT + jj / qT + (jj % qT) * T;
}

}  // namespace icd
