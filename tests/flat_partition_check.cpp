// Checker for rag_project_icd10_amd/csrc/flat_partition.hpp (built by tests/test_flat_partition.py with g++).
// For (query tiles, corpus tiles, units per work-group, list length): walks every work-group's range exactly like
// coarse_flat_kernel does and verifies
//   * every (query tile, corpus tile) unit is covered exactly once,
//   * the lists of a query tile get the ordinals 0, 1, 2, ... in row order, each used once, all below P,
//   * the hardware-block -> work-group map is a bijection for every class period.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "flat_partition.hpp"

using namespace icd;

static int check(int mtc, int ctiles, int U, int L) {
    const long long total = (long long)mtc * ctiles;
    const int G = (int)((total + U - 1) / U);
    int P = 0;
    for (int m = 0; m < mtc; ++m) {
        const long long m1 = (long long)(m + 1) * ctiles;
        const int wl = (int)((m1 - 1) / U);
        const int p = flat_first_ordinal(m, wl + 1, ctiles, U, L);
        if (p > P) P = p;
    }
    std::vector<int> covered((size_t)total, 0);
    std::vector<std::vector<int>> ord_first((size_t)mtc);   // [mtile][ordinal] = first tile of that list
    for (int m = 0; m < mtc; ++m) ord_first[m].assign((size_t)P, -1);
    for (int w = 0; w < G; ++w) {
        const long long u_begin = (long long)w * U, u_end = u_begin + U < total ? u_begin + U : total;
        long long u = u_begin;
        while (u < u_end) {
            const int mtile = (int)(u / ctiles), t0 = (int)(u - (long long)mtile * ctiles);
            const long long mb = (long long)mtile * ctiles;
            const int run0 = (int)(u_begin > mb ? u_begin - mb : 0);
            const int run1 = (int)(u_end - mb < ctiles ? u_end - mb : ctiles);
            const int j = (t0 - run0) / L;
            const int t1 = run1 < run0 + (j + 1) * L ? run1 : run0 + (j + 1) * L;
            const int ord = flat_first_ordinal(mtile, w, ctiles, U, L) + j;
            if (ord < 0 || ord >= P) { printf("ordinal %d out of range P=%d (m=%d w=%d)\n", ord, P, mtile, w); return 1; }
            if (ord_first[mtile][ord] != -1) { printf("ordinal %d used twice (m=%d)\n", ord, mtile); return 1; }
            ord_first[mtile][ord] = t0;
            if (t1 <= t0) { printf("empty list (m=%d w=%d)\n", mtile, w); return 1; }
            for (int t = t0; t < t1; ++t) covered[(size_t)(mb + t)] += 1;
            u += t1 - t0;
        }
    }
    for (long long i = 0; i < total; ++i)
        if (covered[(size_t)i] != 1) { printf("unit %lld covered %d times\n", i, covered[(size_t)i]); return 1; }
    for (int m = 0; m < mtc; ++m) {   // ordinals are dense and in row order
        int prev = -1;
        bool ended = false;
        for (int o = 0; o < P; ++o) {
            const int f = ord_first[m][o];
            if (f == -1) { ended = true; continue; }
            if (ended || f <= prev) { printf("ordinals of query tile %d not dense / not in row order\n", m); return 1; }
            prev = f;
        }
        if (ord_first[m][0] != 0) { printf("query tile %d: first list does not start at tile 0\n", m); return 1; }
    }
    return 0;
}

int main() {
    int cases = 0;
    const int mtcs[] = {1, 2, 8, 79, 128}, cts[] = {1, 2, 32, 290, 317, 1000};
    for (int mtc : mtcs)
        for (int ct : cts)
            for (int U : {1, 2, 3, 12, 90, 97, 145, 290, 500, 4883})
                for (int L : {1, 5, 24, 145, 290, 100000}) {
                    if (check(mtc, ct, U, L)) { printf("FAILED at mtc=%d ctiles=%d U=%d L=%d\n", mtc, ct, U, L); return 1; }
                    ++cases;
                }
    for (int G : {1, 7, 8, 9, 64, 255, 256, 300})   // block -> work-group map: a bijection
        for (int T : {0, 1, 2, 3, 29, 58, 290, 1 << 30}) {
            std::vector<int> seen((size_t)G, 0);
            for (int w = 0; w < G; ++w) {
                const int l = flat_workgroup_of_block(w, G, T);
                if (l < 0 || l >= G || seen[(size_t)l]++) { printf("not a bijection: G=%d T=%d w=%d -> %d\n", G, T, w, l); return 1; }
            }
            ++cases;
        }
    // reduction plan of the streaming kernel's per-wave lists: odd number of levels (the levels ping-pong between two
    // workspaces and finalize reads the second), fan-in <= per_max at every level, the requested final list count (or as few
    // as possible), at most 512 lists after the first of several levels (workspace sizing)
    for (int m : {32, 8, 4})
        for (int pf : {0, m})
            for (int n = 1; n <= 1024; ++n) {
                int plan[8];
                const int c = plan_reduce_levels(n, m, pf, m, plan);
                bool ok = c > 0 && c <= 8 && c % 2 == 1;
                int cur = n;
                for (int i = 0; i < c && ok; ++i) {
                    const int per = (cur + plan[i] - 1) / plan[i];
                    if (per > m || plan[i] < 1 || (i == 0 && c > 1 && plan[0] > 512)) ok = false;
                    cur = plan[i];
                }
                if (ok && pf > 0 && plan[c - 1] != pf) ok = false;
                if (ok && pf == 0 && plan[c - 1] > m) ok = false;
                if (!ok) { printf("bad reduction plan: nlists=%d per_max=%d p_final=%d\n", n, m, pf); return 1; }
                ++cases;
            }
    // tile planner: U never below the balanced minimum nor more than ~6 % (+1) above it, the swept tile count inside the
    // allocation, the class period consistent with gcd; the planned partition itself is a valid one; the two measured cases
    for (int mtc : {1, 2, 3, 8, 40, 79, 128, 782, 977})
        for (int ct : {1, 2, 3, 33, 100, 289, 290, 317, 320, 1000, 9766, 78125})
            for (int spare : {0, 7})
                for (int cus : {1, 8, 256}) {
                    const FlatPlan p = plan_flat_tiles(mtc, ct, spare, cus);
                    const long long units = (long long)mtc * p.ctiles;
                    const long long umin = (units + cus - 1) / cus > 1 ? (units + cus - 1) / cus : 1;
                    bool ok = p.ctiles >= ct && p.ctiles <= ct + spare && p.U >= umin && p.U <= umin + (umin / 16 > 1 ? umin / 16 : 1);
                    const int T = flat_class_period(p.U, p.ctiles);
                    ok = ok && T >= 1 && p.ctiles % T == 0 && ((long long)p.U * T) % p.ctiles == 0;
                    if (!ok) { printf("bad tile plan: mtc=%d ctiles=%d spare=%d cus=%d -> ctiles=%d U=%d\n", mtc, ct, spare, cus, p.ctiles, p.U); return 1; }
                    if (units <= 400000 && check(mtc, p.ctiles, p.U, (p.ctiles + 1) / 2)) { printf("planned partition invalid\n"); return 1; }
                    ++cases;
                }
    {
        const FlatPlan a = plan_flat_tiles(79, 290, 7, 256), b = plan_flat_tiles(79, 317, 7, 256);
        if (a.ctiles != 290 || a.U != 90 || b.ctiles != 319 || b.U != 99) {
            printf("tile plan changed: 37 000 rows -> (%d, %d), 40 474 rows -> (%d, %d)\n", a.ctiles, a.U, b.ctiles, b.U);
            return 1;
        }
    }
    printf("flat_partition: %d cases ok\n", cases);
    return 0;
}
