// flat_partition.hpp — index arithmetic of the flat (query tile x corpus tile) partition of coarse_flat_kernel.hpp.
// Plain C++ (no HIP types): shared by the kernel, the host launcher and tests/test_flat_partition.py (g++).
//
// Units are numbered u = mtile * ctiles + tile; work-group w takes [w U, (w+1) U). Its run inside one query tile is cut
// into lists of at most `list_tiles` tiles; ordinals count the lists of a query tile in row order.
#pragma once
#if defined(__HIPCC__)
#define ICD_HD __host__ __device__
#else
#define ICD_HD
#endif

namespace icd {

constexpr int FLAT_SPARE_TILES = 7;   // zero tiles allocated behind the fp16 corpus image (plan_flat_tiles may sweep them)

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


// Work-groups l and l + T start on the same corpus tile, T = ctiles / gcd(U mod ctiles, ctiles): they stream the same
// tiles at the same time and, placed on one XCD (flat_workgroup_of_block), share its L2.
ICD_HD inline int flat_class_period(int U, int ctiles) {
    int g = U % ctiles, h = ctiles;   // gcd(0, c) = c
    while (g) { const int t = h % g; h = g; g = t; }
    return ctiles / h;
}

// Tiles per work-group U and the corpus tile count to sweep (ctiles_min .. ctiles_min + spare; the tiles past the corpus
// are zero rows that never pass the select). The smallest U leaves every CU the same number of tiles, but when it is
// coprime to the tile count no two work-groups ever stream the same tile together and every XCD fetches the corpus for
// itself (measured: 40 474 rows = 317 tiles, U = 98: 0.771 ms; 320 tiles, U = 100: 0.721 ms, the rate of the 37 000-row
// case). Cost model: time ~ U, +7 % when classes have fewer than two members, +3 % below four.
struct FlatPlan { int ctiles, U; };
ICD_HD inline FlatPlan plan_flat_tiles(int mtc, int ctiles_min, int spare, int num_cu) {
    FlatPlan best{ctiles_min, 1};
    double best_cost = 1e300;
    for (int ce = ctiles_min; ce <= ctiles_min + spare; ++ce) {
        const long long units = (long long)mtc * ce;
        const int umin = (int)((units + num_cu - 1) / num_cu) > 1 ? (int)((units + num_cu - 1) / num_cu) : 1;
        const int umax = umin + (umin / 16 > 1 ? umin / 16 : 1);
        for (int U = umin; U <= umax; ++U) {
            const int nwg = (int)((units + U - 1) / U);
            const int members = nwg / flat_class_period(U, ce);
            const double cost = (double)U * (members >= 4 ? 1.0 : (members >= 2 ? 1.03 : 1.07));
            if (cost < best_cost - 1e-9) { best_cost = cost; best.ctiles = ce; best.U = U; }
        }
    }
    return best;
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


// List counts after every reduction level of the streaming kernel's 4 * nwg per-wave lists: a reduce wave merges at most
// per_max lists, the last level must leave p_final lists (p_final = 0: as few as one more level gives), and the number
// of levels must be ODD because the levels ping-pong between the two list workspaces and finalize reads the second one.
// (Round 1 sized the sweep so that ONE level sufficed: 16 work-groups at k > 16, 4 at k > 64 - a 0.45 / 1.6 ms fallback
// for a single uncertified query at k = 64 / 100, profiles/r02_shapes_before.log. Now the sweep always fills the chip.)
ICD_HD inline int plan_reduce_levels(int nlists, int per_max, int p_final, int p_cap, int *plan /* [8] */) {
    const int target = p_final > 0 ? p_final : p_cap;
    int levels = 1;
    long long reach = (long long)per_max * target;          // lists that `levels` levels can bring down to `target`
    while (reach < nlists) { reach *= per_max; ++levels; }
    const bool extra = levels % 2 == 0;                      // an even count gets one more level, of fan-in 2, in front
    if (levels + (extra ? 1 : 0) > 8) return 0;
    int cnt = 0, cur = nlists;
    if (extra) { cur = (cur + 1) / 2; plan[cnt++] = cur; }
    for (int l = 1; l < levels; ++l) { cur = (cur + per_max - 1) / per_max; plan[cnt++] = cur; }
    plan[cnt++] = p_final > 0 ? p_final : ((cur + per_max - 1) / per_max > 1 ? (cur + per_max - 1) / per_max : 1);
    return cnt;
}

}  // namespace icd
