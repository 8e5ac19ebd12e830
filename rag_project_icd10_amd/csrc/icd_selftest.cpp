// icd_selftest.cpp — developer harness: parity of libicdsearch.so against the CPU oracle
// (oracle/libicd_oracle.so, loaded with dlopen: test infrastructure, never linked into the product)
// and kernel timing. Usage:
//   icd_selftest [--oracle path/to/libicd_oracle.so] [--quick] [--bench] [--nq N] [--n N] [--iters I]
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/icd_search.h"

typedef int (*oracle_topk_fn)(const float *, int64_t, int, const float *, int64_t, int, int64_t, int, float *, int64_t *);
typedef void (*oracle_reweight_fn)(const float *, const int64_t *, const int32_t *, int64_t, int64_t, int, double *,
                                   float *, int64_t *, int32_t *);
static oracle_topk_fn oracle_topk = nullptr;
static oracle_reweight_fn oracle_reweight = nullptr;

static int g_fail = 0, g_pass = 0;
#define CHECK_RC(expr)                                                          \
    do {                                                                        \
        int rc_ = (expr);                                                       \
        if (rc_ != 0) {                                                         \
            printf("[FAIL] %s -> %d: %s\n", #expr, rc_, icd_last_error());      \
            ++g_fail;                                                           \
            return;                                                             \
        }                                                                       \
    } while (0)

struct Data {
    std::vector<float> corpus, queries;
    std::vector<int32_t> levels;
    int64_t n, nq;
    int dim;
};

// kind 0: iid gaussian unit rows; 1: clustered (near ties); 2: with duplicate rows (exact ties)
static Data make_data(int64_t n, int64_t nq, int dim, int kind, uint64_t seed) {
    Data d;
    d.n = n; d.nq = nq; d.dim = dim;
    d.corpus.resize((size_t)n * dim);
    d.queries.resize((size_t)nq * dim);
    d.levels.resize((size_t)n);
    std::mt19937_64 rng(seed);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> cent;
    const int ncent = 64;
    if (kind == 1) {
        cent.resize((size_t)ncent * dim);
        for (auto &v : cent) v = nd(rng);
    }
    auto fill = [&](float *row, bool is_query) {
        double ss = 0;
        if (kind == 1) {
            const float *c = &cent[(size_t)(rng() % ncent) * dim];
            const float noise = is_query ? 0.3f : 0.05f;
            for (int j = 0; j < dim; ++j) { row[j] = c[j] + noise * nd(rng); ss += (double)row[j] * row[j]; }
        } else {
            for (int j = 0; j < dim; ++j) { row[j] = nd(rng); ss += (double)row[j] * row[j]; }
        }
        const float inv = (float)(1.0 / std::sqrt(ss));
        for (int j = 0; j < dim; ++j) row[j] *= inv;
    };
    for (int64_t i = 0; i < n; ++i) {
        fill(&d.corpus[(size_t)i * dim], false);
        const unsigned r = (unsigned)(rng() % 10000);
        d.levels[i] = r < 1243 ? 1 : (r < 4234 ? 2 : 3);  // real CSV histogram
    }
    if (kind == 2) {
        // every 5th row duplicates an earlier row: exact score ties, broken by row id
        for (int64_t i = 5; i < n; i += 5) memcpy(&d.corpus[(size_t)i * dim], &d.corpus[(size_t)(i / 2) * dim], sizeof(float) * dim);
    }
    for (int64_t i = 0; i < nq; ++i) fill(&d.queries[(size_t)i * dim], true);
    return d;
}

static void run_case(const char *name, const Data &d, int k, int mode, int64_t id_base = 0) {
    icd_index *idx = nullptr;
    CHECK_RC(icd_index_create(d.corpus.data(), d.n, d.dim, d.levels.data(), id_base, 0, (int)std::max<int64_t>(d.nq, 1), std::max(k, 1), 0, &idx));
    const size_t no = (size_t)d.nq * k;
    std::vector<float> s(no), ar(no), os(no), oar(no);
    std::vector<int64_t> ids(no), aids(no), oids(no), oaids(no);
    std::vector<double> adj(no), oadj(no);
    std::vector<int32_t> alv(no), oalv(no);
    int rc = icd_index_search(idx, d.queries.data(), d.nq, k, 0, mode, s.data(), ids.data(), 0, nullptr);
    if (rc) { printf("[FAIL] %s search rc=%d %s\n", name, rc, icd_last_error()); ++g_fail; icd_index_destroy(idx); return; }
    icd_stats st;
    icd_index_stats(idx, &st);
    rc = icd_index_search_reweighted(idx, d.queries.data(), d.nq, k, 0, mode, adj.data(), ar.data(), aids.data(), alv.data(), 0, nullptr);
    if (rc) { printf("[FAIL] %s reweighted rc=%d %s\n", name, rc, icd_last_error()); ++g_fail; icd_index_destroy(idx); return; }
    oracle_topk(d.corpus.data(), d.n, d.dim, d.queries.data(), d.nq, k, id_base, 0, os.data(), oids.data());
    oracle_reweight(os.data(), oids.data(), d.levels.data(), id_base, d.nq, k, oadj.data(), oar.data(), oaids.data(), oalv.data());
    size_t bad_id = 0, bad_s = 0, bad_adj = 0;
    long first_bad = -1;
    for (size_t i = 0; i < no; ++i) {
        if (ids[i] != oids[i]) { ++bad_id; if (first_bad < 0) first_bad = (long)i; }
        if (memcmp(&s[i], &os[i], 4) != 0) ++bad_s;
        if (aids[i] != oaids[i] || memcmp(&adj[i], &oadj[i], 8) != 0 || alv[i] != oalv[i] || memcmp(&ar[i], &oar[i], 4) != 0) ++bad_adj;
    }
    const bool ok = bad_id == 0 && bad_s == 0 && bad_adj == 0;
    printf("[%s] %-34s n=%lld nq=%lld dim=%d k=%d mode=%s chunks=%d fallback=%lld/%lld  id_mismatch=%zu score_bits_mismatch=%zu reweight_mismatch=%zu\n",
           ok ? "PASS" : "FAIL", name, (long long)d.n, (long long)d.nq, d.dim, k, st.last_mode == ICD_MODE_AUTO ? "auto" : "exact",
           st.last_chunks, (long long)st.last_fallback, (long long)d.nq, bad_id, bad_s, bad_adj);
    if (!ok && first_bad >= 0) {
        const size_t q = first_bad / k;
        printf("       first mismatch query %zu:\n        got :", q);
        for (int j = 0; j < k && j < 12; ++j) printf(" %lld(%.7f)", (long long)ids[q * k + j], s[q * k + j]);
        printf("\n        want:");
        for (int j = 0; j < k && j < 12; ++j) printf(" %lld(%.7f)", (long long)oids[q * k + j], os[q * k + j]);
        printf("\n");
    }
    ok ? ++g_pass : ++g_fail;
    icd_index_destroy(idx);
}

static void bench(int64_t n, int64_t nq, int dim, int k, int iters, int chunks, bool verify, bool auto_only = false) {
    printf("== bench n=%lld nq=%lld dim=%d k=%d iters=%d chunks=%d\n", (long long)n, (long long)nq, dim, k, iters, chunks);
    Data d = make_data(n, nq, dim, 0, 1234);
    icd_index *idx = nullptr;
    CHECK_RC(icd_index_create(d.corpus.data(), n, dim, d.levels.data(), 0, 0, (int)nq, k, 0, &idx));
    if (chunks > 0) icd_index_set_chunks(idx, chunks);
    float *dq; float *ds; int64_t *di;
    hipMalloc((void **)&dq, (size_t)nq * dim * 4);
    hipMalloc((void **)&ds, (size_t)nq * k * 4);
    hipMalloc((void **)&di, (size_t)nq * k * 8);
    hipMemcpy(dq, d.queries.data(), (size_t)nq * dim * 4, hipMemcpyHostToDevice);
    icd_index_set_profiling(idx, 1);
    for (int mode = 0; mode < (auto_only ? 1 : 2); ++mode) {
        const int it = mode == 0 ? iters : std::max(1, iters / 5);
        icd_profile acc{};
        double wall = 0;
        for (int i = -2; i < it; ++i) {
            auto t0 = std::chrono::steady_clock::now();
            CHECK_RC(icd_index_search(idx, dq, nq, k, 1, mode, ds, di, 1, nullptr));
            hipDeviceSynchronize();
            auto t1 = std::chrono::steady_clock::now();
            icd_profile p;
            CHECK_RC(icd_index_last_profile(idx, &p));
            if (i >= 0) {
                wall += std::chrono::duration<double, std::milli>(t1 - t0).count();
                acc.ms_prep += p.ms_prep; acc.ms_coarse += p.ms_coarse; acc.ms_finalize += p.ms_finalize;
                acc.ms_exact += p.ms_exact; acc.ms_exact_finalize += p.ms_exact_finalize; acc.ms_total += p.ms_total;
            }
        }
        icd_stats st;
        icd_index_stats(idx, &st);
        const double tot = acc.ms_total / it;
        const double flop = 2.0 * nq * n * dim;
        const double dom = (mode == 0 ? acc.ms_coarse : acc.ms_exact) / it;
        printf("   mode=%-5s total=%.4f ms (wall %.4f)  prep=%.4f coarse=%.4f finalize=%.4f exact=%.4f exact_fin=%.4f | %.3f Mq/s | dominant kernel %.1f TFLOP/s | chunks=%d fallback=%lld\n",
               mode == 0 ? "auto" : "exact", tot, wall / it, acc.ms_prep / it, acc.ms_coarse / it, acc.ms_finalize / it,
               acc.ms_exact / it, acc.ms_exact_finalize / it, nq / tot / 1e3, flop / (dom * 1e-3) / 1e12, st.last_chunks,
               (long long)st.last_fallback);
    }
    if (getenv("ICD_FLAT_VAR") && (atoi(getenv("ICD_FLAT_VAR")) & 1024)) {
        std::vector<unsigned long long> c(8192);
        CHECK_RC(icd_index_debug_counters(idx, c.data(), (int)c.size()));
        double vm = 0, bar = 0, body = 0, sel = 0, tiles = 0, nc = 0, cc = 0, thr = 0, boot = 0; int cnt = 0;
        for (size_t i = 0; i + 7 < c.size(); i += 8) if (c[i + 4]) {
            vm += c[i]; bar += c[i + 1]; body += c[i + 2]; sel += c[i + 3]; tiles += c[i + 4]; nc += c[i + 5]; cc += c[i + 6];
            thr += (double)(c[i + 7] >> 32); boot += (double)(c[i + 7] & 0xffffffffull); ++cnt;
        }
        printf("   stamps (avg per wave over %d waves, cycles per tile): dma wait=%.0f barrier=%.0f body=%.0f select=%.0f (thr exchange %.0f, bootstrap %.0f, compactions %.2f x %.0f = %.0f) tiles/wave=%.1f\n",
               cnt, vm / tiles, bar / tiles, body / tiles, sel / tiles, thr / tiles, boot / tiles, nc / tiles, nc ? cc / nc : 0.0, cc / tiles, tiles / cnt);
    }
    if (getenv("ICD_FLAT_VAR") && (atoi(getenv("ICD_FLAT_VAR")) & 33554432)) {
        // in-kernel clock of the coarse launch: d(s_memtime) / d(s_memrealtime) x 100 MHz per wave, median over the waves
        std::vector<unsigned long long> c(8192 * 8);
        CHECK_RC(icd_index_debug_counters(idx, c.data(), (int)c.size()));
        std::vector<double> ghz;
        for (size_t i = 0; i + 7 < c.size(); i += 8)
            if (c[i + 4] == 1 && c[i + 1] > 0) ghz.push_back((double)c[i] / (double)c[i + 1] * 0.1);
        std::sort(ghz.begin(), ghz.end());
        if (!ghz.empty())
            printf("   in-kernel clock of the coarse launch (s_memtime / s_memrealtime, %zu waves): median %.3f GHz (min %.3f, max %.3f)\n",
                   ghz.size(), ghz[ghz.size() / 2], ghz.front(), ghz.back());
    }
    if (verify) {
        // parity of a query sample of the big run against the oracle (AUTO mode)
        const int ns = 128;
        std::vector<float> s((size_t)nq * k), os((size_t)ns * k);
        std::vector<int64_t> ids((size_t)nq * k), oids((size_t)ns * k);
        CHECK_RC(icd_index_search(idx, dq, nq, k, 1, ICD_MODE_AUTO, ds, di, 1, nullptr));
        hipMemcpy(s.data(), ds, s.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(ids.data(), di, ids.size() * 8, hipMemcpyDeviceToHost);
        std::vector<float> qs((size_t)ns * dim);
        const int64_t stride = std::max<int64_t>(1, nq / ns);
        for (int i = 0; i < ns; ++i) memcpy(&qs[(size_t)i * dim], &d.queries[(size_t)(i * stride % nq) * dim], sizeof(float) * dim);
        oracle_topk(d.corpus.data(), n, dim, qs.data(), ns, k, 0, 0, os.data(), oids.data());
        size_t bad = 0;
        for (int i = 0; i < ns; ++i)
            for (int j = 0; j < k; ++j) {
                const size_t g = (size_t)(i * stride % nq) * k + j;
                if (ids[g] != oids[(size_t)i * k + j] || memcmp(&s[g], &os[(size_t)i * k + j], 4) != 0) ++bad;
            }
        printf("[%s] bench-sample parity (auto, %d queries): mismatches=%zu\n", bad ? "FAIL" : "PASS", ns, bad);
        bad ? ++g_fail : ++g_pass;
    }
    hipFree(dq); hipFree(ds); hipFree(di);
    icd_index_destroy(idx);
}

int main(int argc, char **argv) {
    std::string opath = "oracle/libicd_oracle.so";
    bool quick = false, do_bench = false, skip_cases = false, auto_only = false;
    int64_t bn = 37000, bnq = 10000;
    int iters = 20, chunks = 0;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--oracle" && i + 1 < argc) opath = argv[++i];
        else if (a == "--quick") quick = true;
        else if (a == "--bench") do_bench = true;
        else if (a == "--skip-cases") skip_cases = true;
        else if (a == "--auto-only") auto_only = true;   // (--bench: the AUTO-mode loop alone, e.g. seconds of one load for a clock sampler)
        else if (a == "--n" && i + 1 < argc) bn = atoll(argv[++i]);
        else if (a == "--nq" && i + 1 < argc) bnq = atoll(argv[++i]);
        else if (a == "--iters" && i + 1 < argc) iters = atoi(argv[++i]);
        else if (a == "--chunks" && i + 1 < argc) chunks = atoi(argv[++i]);
    }
    void *h = dlopen(opath.c_str(), RTLD_NOW);
    if (!h) { printf("cannot load oracle %s: %s\n", opath.c_str(), dlerror()); return 2; }
    oracle_topk = (oracle_topk_fn)dlsym(h, "icd_oracle_flat_ip_topk");
    oracle_reweight = (oracle_reweight_fn)dlsym(h, "icd_oracle_reweight");
    if (!oracle_topk || !oracle_reweight) { printf("oracle symbols missing\n"); return 2; }
    printf("icd_selftest: abi=%d devices=%d\n", icd_abi_version(), icd_device_count());

    if (!skip_cases) {
    // ---- exact path ----
    run_case("exact/tiny", make_data(100, 3, 768, 0, 1), 5, ICD_MODE_EXACT);
    run_case("exact/n<k", make_data(7, 2, 768, 0, 2), 10, ICD_MODE_EXACT);
    run_case("exact/ragged", make_data(1000, 37, 768, 0, 3), 10, ICD_MODE_EXACT);
    run_case("exact/dups(ties)", make_data(3001, 70, 768, 2, 4), 10, ICD_MODE_EXACT);
    run_case("exact/k=1", make_data(2049, 5, 768, 0, 5), 1, ICD_MODE_EXACT);
    run_case("exact/k=17(KP64)", make_data(2500, 33, 768, 0, 6), 17, ICD_MODE_EXACT);
    run_case("exact/k=100(KP128)", make_data(5000, 9, 768, 0, 7), 100, ICD_MODE_EXACT);
    run_case("exact/dim1024", make_data(1500, 20, 1024, 0, 8), 10, ICD_MODE_EXACT);
    run_case("exact/dim64", make_data(900, 11, 64, 0, 9), 5, ICD_MODE_EXACT);
    run_case("exact/id_base", make_data(700, 6, 768, 0, 10), 10, ICD_MODE_EXACT, 1000000);
    // ---- fast path ----
    run_case("auto/tiny", make_data(100, 3, 768, 0, 11), 5, ICD_MODE_AUTO);
    run_case("auto/n<k", make_data(7, 2, 768, 0, 12), 10, ICD_MODE_AUTO);
    run_case("auto/ragged", make_data(1000, 37, 768, 0, 13), 10, ICD_MODE_AUTO);
    run_case("auto/dups(ties)", make_data(3001, 70, 768, 2, 14), 10, ICD_MODE_AUTO);
    run_case("auto/clustered", make_data(6000, 200, 768, 1, 15), 10, ICD_MODE_AUTO);
    run_case("auto/k=1", make_data(2049, 5, 768, 0, 16), 1, ICD_MODE_AUTO);
    run_case("auto/k=12", make_data(4000, 129, 768, 0, 17), 12, ICD_MODE_AUTO);
    run_case("auto/dim1024", make_data(1500, 20, 1024, 0, 18), 10, ICD_MODE_AUTO);
    run_case("auto/id_base", make_data(700, 6, 768, 0, 19), 10, ICD_MODE_AUTO, 1000000);
    if (!quick) {
        run_case("exact/37k", make_data(37000, 130, 768, 0, 20), 10, ICD_MODE_EXACT);
        run_case("auto/37k", make_data(37000, 300, 768, 0, 21), 10, ICD_MODE_AUTO);
        run_case("auto/40474", make_data(40474, 130, 768, 0, 22), 10, ICD_MODE_AUTO);
        run_case("auto/37k-clustered", make_data(37000, 256, 768, 1, 23), 10, ICD_MODE_AUTO);
        run_case("auto/37k-k5", make_data(37000, 1, 768, 0, 24), 5, ICD_MODE_AUTO);
    }
    }
    if (do_bench) {
        const int var = getenv("ICD_FLAT_VAR") ? atoi(getenv("ICD_FLAT_VAR")) : 0;
        bench(bn, bnq, 768, 10, iters, chunks, (var & (64 | 256 | 512 | 8192)) == 0, auto_only);   // (timing-only variants compute garbage)
    }
    printf("icd_selftest: %d passed, %d failed\n", g_pass, g_fail);
    return g_fail ? 1 : 0;
}
