// icd_search.hip — host side of libicdsearch.so (C ABI declared in include/icd_search.h).
// gfx950 (MI355X) only. No allocation and no synchronisation inside a search whose inputs and
// outputs are device buffers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/icd_search.h"
#include "coarse_flat_kernel.hpp"
// The dropped coarse-kernel forms of rounds 1-4 (row groups, eight waves, 16-query groups, K-split wave pairs) live on the git
// branch archive/experiments-r01-r04 (directory experiments/); an A/B build that wants them checks that directory out and
// adds EXTRA=-DICD_WITH_EXPERIMENTS. The tree itself carries only the shipped kernels and the variants of the shipped stage.
#if defined(ICD_ABLATE) && !defined(ICD_FV_LIST) && defined(ICD_WITH_EXPERIMENTS)
#define ICD_ABLATE_EXPERIMENTS 1
#endif
#if defined(ICD_ABLATE) && defined(ICD_WITH_EXPERIMENTS)
#define ICD_ABLATE_KSPLIT 1
#include "../../experiments/r04_ksplit_kernel/coarse_ksplit_kernel.hpp"   // (wave pairs split K, 64 queries per wave)
#endif
#ifdef ICD_ABLATE_EXPERIMENTS
#include "../../experiments/r02_rg_kernel/coarse_rg_kernel.hpp"   // (A/B builds only: the row-group experiment)
#include "../../experiments/r02_w8_kernel/coarse_w8_kernel.hpp"   // (A/B builds only: eight waves, two per SIMD)
#include "../../experiments/r02_g16_kernel/coarse_g16_kernel.hpp" // (A/B builds only: the flat geometry on 16x16x32)
#endif
#include "attention_kernel.hpp"
#include "exact_kernel.hpp"
#include "finalize.hpp"
#include "hier_kernel.hpp"
#include "stats_kernel.hpp"
#include "stream_kernel.hpp"

using namespace icd;

namespace {

thread_local std::string g_err = "";
int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

}  // namespace

// (icd_group.cpp reports its errors through the same thread-local text)
// Query-sharded gather (icd_group.cpp): the all-gathered send buffers hold, per rank r, one PADDED slice of `width` queries
// in four arrays (adj f64 | ids i64 | raw f32 | levels i32, each [world][width * k]); rank r's valid part is the first
// (hi_r - lo_r) queries. ONE launch scatters every rank's valid hits to the contiguous [nq][k] outputs (it replaces four
// hipMemcpyAsync per rank: 32 copies at 8 ranks). shard_bounds' arithmetic restated: base = nq / world, rem = nq % world.
__global__ void unpack_query_slices_kernel(const double *r_adj, const long long *r_ids, const float *r_raw, const int *r_lv,
                                           int world, long long nq, int k, long long width, double *out_adj, float *out_raw,
                                           long long *out_ids, int *out_lv) {
    const long long total = nq * (long long)k;
    const long long base = nq / world, rem = nq % world;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long q = e / k;
        const int j = (int)(e - q * k);
        // rank of query q: the first `rem` ranks own base + 1 queries each
        const long long split = rem * (base + 1);
        const int r = q < split ? (int)(q / (base + 1)) : (int)(rem + (q - split) / (base > 0 ? base : 1));
        const long long lo = (long long)r * base + (r < rem ? r : rem);
        const long long src = (long long)r * width * k + (q - lo) * k + j;
        out_adj[e] = r_adj[src]; out_ids[e] = r_ids[src]; out_raw[e] = r_raw[src]; out_lv[e] = r_lv[src];
    }
}
extern "C" __attribute__((visibility("hidden"))) int icd_internal_unpack_query_slices(const void *r_adj, const void *r_ids, const void *r_raw,
        const void *r_lv, int world, long long nq, int k, long long width, void *out_adj, void *out_raw, void *out_ids, void *out_lv, void *stream) {
    const long long total = nq * (long long)k;
    if (total <= 0) return 0;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 2048);
    hipLaunchKernelGGL(unpack_query_slices_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       static_cast<const double *>(r_adj), static_cast<const long long *>(r_ids), static_cast<const float *>(r_raw),
                       static_cast<const int *>(r_lv), world, nq, k, width, static_cast<double *>(out_adj), static_cast<float *>(out_raw),
                       static_cast<long long *>(out_ids), static_cast<int *>(out_lv));
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

extern "C" __attribute__((visibility("hidden"))) int icd_internal_fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

namespace {

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? ICD_ERR_NOMEM : ICD_ERR_HIP, "%s: %s (%s:%d)", \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);                         \
    } while (0)

constexpr float EPS_REL = 1.2e-3f;
constexpr size_t PACE_WORDS = (size_t)1 << 18;   // arrival counters of a paced coarse sweep: classes x epochs (1 MB)
constexpr size_t COARSE_CACHED_IMAGE_BYTES = (size_t)96 << 20;   // fp16 images up to this size take CF_CACHED_VAR (both corpus copies then fit the 256 MiB Infinity Cache)
constexpr int FAST_MAX_K = 100;      // the rescoring window holds up to 256 candidates (four per lane): k + the rows inside 2 eps of the k-th
constexpr int COARSE_MAX_P = 32;     // P * KP <= FIN_MAX_CAND
constexpr int PASS2_CHUNKS = 20;     // second coarse pass (and wide_mode): about this many candidate lists per query
constexpr int PASS2_MAX_P = 24;      // ... at most this many (workspace); 24 x 16 candidates < FIN_MAX_CAND
constexpr int PASS2_SKIP = 24;       // ... and leaves this many flagged queries (or fewer) to the streaming kernel: 35 us per 8
constexpr int PASS2_BELOW = 320;     // the second pass runs when the first gave a query fewer candidates than this
constexpr int SPARSE_DISARM_AFTER = 96; // the streaming kernel's two fallback launches are dropped after this many consecutive searches with NO flagged query (doubles at every incident)
constexpr int PASS2_DISARM_AFTER = 4; // the second pass's two launches are dropped after this many consecutive searches that needed neither
constexpr int WIDE_MIN_NQ = 2048;    // "large batch": below it a query has 16+ lists anyway
constexpr int WIDE_REPROBE = 64;
constexpr int NUM_EV = 6;
constexpr int EV_RING = 128;         // profiled searches kept for icd_index_profile_summary

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, device): every launcher keeps, per device, the
// largest size it has configured and raises it when a launch needs more (two indexes of different dim share kernels).
constexpr int MAX_DEVICES = 64;
template <typename K>
hipError_t ensure_dynamic_lds(K kern, int device, size_t bytes, int *configured /* [MAX_DEVICES] */) {
    if (device < 0 || device >= MAX_DEVICES) return hipErrorInvalidDevice;
    if ((size_t)configured[device] >= bytes) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) configured[device] = (int)bytes;
    return e;
}

template <typename T>
hipError_t dmalloc(T **p, size_t count) {
    return hipMalloc(reinterpret_cast<void **>(p), std::max<size_t>(count, 1) * sizeof(T));
}

}  // namespace

struct icd_index {
    uint32_t magic = 0x1CD10A3Du;
    int device = 0;
    // A/B and test options, PER HANDLE (icd_index_create flags, icd_index_set_option): performance decisions only, results are
    // identical whatever they are set to. Read and written under `mu`.
    bool opt_permute = true;        // ICD_CREATE_ROW_ORDER clears it: row order of the fp16 corpus copy
    bool opt_probe = true;          // ICD_CREATE_NO_PROBE: the corpus-shape probe of icd_index_create
    bool opt_center = true;         // ICD_CREATE_NO_CENTER: the fp16 image is centred when the rows share a large common component
    int opt_pace_shift = 3, opt_pace_lead = 2;   // ICD_OPT_PACING_*: epochs of 2^shift tiles, classes kept within `lead` epochs; shift < 0: no pacing
    bool opt_exact_narrow = true;   // ICD_OPT_EXACT_NARROW: EXACT mode at k > 32 runs certified lists of 32 over row-strided chunks
    int opt_host_one = 3;           // ICD_OPT_HOST_ONE: a host caller's ONE query 1 = travels in the kernel arguments, 2 = completion by a polled word
    bool opt_stream_one = true;     // ICD_OPT_STREAM_ONE: one or two queries per call take the single-launch streaming kernel
    bool opt_family_order = true;   // ICD_OPT_FAMILY_ORDER: the wide-window finalize visits the queries in family order
    int opt_wide_from = 48;         // ICD_OPT_WIDE_FROM: k above this keeps 24 candidates per coarse list instead of 16 (>= ICD_MAX_K: never). 64 until round 5.
                                    // Lists of 16 leave 1-5 of 10 000 queries to the exact re-search at k = 40 ... 64: 0.18-0.20 ms per batch while the streaming
                                    // re-search started its lists at -inf, 0.06-0.07 since it takes finalize's threshold (round 6). With that the two list
                                    // widths meet at k = 50 ... 64 and 16 wins below (profiles/r06_k100_lists.log)
    int64_t n = 0, id_base = 0;
    int n_pad = 0;
    int dim = 0;
    int max_nq = 0, max_nq_pad = 0, max_k = 0;
    bool fast = false;
    float rmax = 0.f;        // largest row norm (unscaled; reported by icd_index_stats)
    float rmax_scaled = 0.f; // largest norm of the scaled fp16 rows
    int cexp = 0;            // fp16 corpus = fp32 corpus * 2^cexp
    float *cmean = nullptr;  // [dim] column mean subtracted from the fp16 image (nullptr: not centred; coarse_common.hpp)
    float rmax_unc_scaled = 0.f;   // centred image: largest norm of the UNcentred rows, in the image's scale
    float mean_share = 0.f;  // |mu|^2 / mean |row|^2 (the mean pairwise cosine of unit rows): reported by icd_index_stats
    int num_cu = 256;
    // corpus
    float *corpus = nullptr;
    _Float16 *c16 = nullptr;
    int *levels = nullptr;
    // workspace
    float *qdev = nullptr;  // staging for host queries
    _Float16 *q16 = nullptr;
    float *qnorm = nullptr;
    int *qexp = nullptr;
    unsigned char *qbad = nullptr;
    float *thr0 = nullptr;   // [query] threshold an uncertified query hands to its exact re-search (finalize.hpp)
    unsigned int *shared_thr = nullptr;   // [max_nq_pad] coarse pass: per-query threshold shared by its lists
    long long perm_mul = 0; int perm_mod = 0;   // row order of the fp16 corpus: position p holds row (p * perm_mul) mod perm_mod
    float *partc_s = nullptr; int *partc_r = nullptr; float *partc_b = nullptr; size_t partc_cap = 0;   // coarse lists + bounds
    float *part2_s = nullptr; int *part2_r = nullptr; float *part2_b = nullptr; size_t part2_cap = 0;   // lists of the second coarse pass
    float *partx_s = nullptr; int *partx_r = nullptr; size_t partx_cap = 0;
    float *lists_s = nullptr; int *lists_r = nullptr; size_t lists_cap = 0;   // streaming kernel: [slot][4 nwg][KP]
    int *nflag = nullptr; int *flagged = nullptr;   // [8]: fallback counters [0..3], [4] = consecutive fast-path searches whose exact re-search had nothing to do (kept by the device); lists [3][max_nq_pad] and lists [3][max_nq_pad]: after the first finalize, after its wide-window retry, after the second coarse pass
    int fallback_word = 0;                           // which counter / list the last search's exact re-search read
    int *h_nflag = nullptr;                          // pinned (mapped) host copy of nflag[4], written by the last kernel of every search
    int *h_nflag_dev = nullptr;                      // its device-side address
    hipEvent_t ev_nflag = nullptr;                   // recorded behind that copy: icd_index_stats waits for it and nothing else
    unsigned int *scratch_u32 = nullptr;  // [0]=rmax bits, [1]=any_bad
    int *order = nullptr, *order_key = nullptr; unsigned int *order_hist = nullptr;   // wide-window finalize in family order (finalize.hpp, order_*_kernel)
    // output staging (used when the caller's buffers are host memory)
    float *o_scores = nullptr; long long *o_ids = nullptr;
    double *o_adj = nullptr; float *o_adj_raw = nullptr; long long *o_adj_ids = nullptr; int *o_adj_lv = nullptr;
    // small host calls (the reference's one-query-per-call shape): ONE pinned, mapped block - the query goes through its first
    // part (CPU copy + a true asynchronous H2D copy; a pageable source makes hipMemcpyAsync stage and wait), the kernels write
    // the results straight into its second part (zero-copy stores over PCIe) and the host copies them out after the stream
    // synchronisation: no D2H memcpy call at all (six of them to pageable memory cost ~60 us of a 113-us call)
    char *h_pin = nullptr, *h_pin_dev = nullptr;
    // ... and for ONE query per call (services/milvus_service.py:280-285) two commands less: the vector travels IN the kernel
    // arguments of the single-launch kernel (no H2D copy in front of it: host_q is set for the duration of that call), and the
    // kernel's last work-group stores a sequence number behind its outputs (the block's last 64 bytes) that the host polls
    // instead of waiting for the stream's completion signal (done_seq counts the calls that asked for it)
    const float *host_q = nullptr;
    bool host_one_call = false;    // the current call is ONE query from host memory with host outputs in the mapped block: it may poll
    unsigned long long done_seq = 0;
    bool done_armed = false;       // the current call's launch carries the word's address
    size_t bytes_ws = 0;
    // knobs / counters
    int chunks_override = 0;
    // adaptive list count of LARGE batches (>= WIDE_MIN_NQ queries): when the second coarse pass had to rescue most of the
    // previous large batch (a corpus of tight families), the first pass is planned with its list count right away
    bool wide_mode = false;        // plan the first pass with PASS2_CHUNKS lists per query
    int wide_runs = 0;             // large searches since wide_mode was entered (every WIDE_REPROBE-th runs narrow again)
    bool last_narrow_large = false; // the last search was a large batch with the narrow plan and the second pass behind it
    int64_t last_narrow_nq = 0;     // ... and its size: the flagged counters read at the NEXT search are fractions of THIS batch
    // The second pass costs a search that flags nothing two launches that read a counter and leave (~9 us per 10 000-query
    // step). They are armed while nothing is known about the corpus (the first searches of an index) and whenever a recent
    // search flagged more queries than the streaming kernel takes cheaply; after PASS2_DISARM_AFTER consecutive searches
    // that flagged fewer they are left out (a batch that then flags many takes the exact re-search once and re-arms them).
    int p2_clean = 0;              // consecutive evaluated searches with <= PASS2_SKIP queries flagged by the first finalize
    bool p2_eval_pending = false;  // the last search's counters have not been looked at yet
    // The exact re-search behind every search is four launches: streaming kernel + list reduction (few flagged queries:
    // one corpus sweep per 8) and fp32-MFMA kernel + finalize (many). With nothing flagged each reads a counter and leaves.
    // After `sparse_need` consecutive searches whose re-search had nothing to do (counted on the device, so searches the
    // host never looked at count too) the streaming pair is left out and the MFMA kernel takes any count >= 1. Measured at
    // 10 000 x 37 000 (profiles/r03_sparse_fallback_policy.log): the pair costs a clean search ~3.4 us (0.5 %); without it
    // a search with 1-8 flagged queries takes 1.25 instead of 0.72 ms, once - the run restarts and the requirement
    // doubles. The break-even of the two is ~150 clean searches (ski rental); 96 stays below it (at most 2.6 x the
    // cost of the best fixed choice in hindsight, 2 x at the break-even).
    int sparse_need = SPARSE_DISARM_AFTER;
    int sparse_run_seen = 0;       // the device's run length when the host last saw a completed search
    bool sparse_disarmed = false;  // the last fast-path search went without the streaming pair
    std::mutex mu;   // host-side state of the handle (plans, adaptive counters, profiling ring): calls on one handle are serialised; the DEVICE workspace still belongs to one stream at a time (header)
    int probe_flagged = -1, probe_left = -1;   // the corpus-shape probe of icd_index_create: queries its first finalize flagged / its second pass left (-1: not run)
    bool pass2_enabled = true;     // test hook (icd_index_set_second_pass)
    bool adapt_enabled = true;     // ... 2 = second pass without the adaptive list count
    bool profiling = false;
    int prof_every = 1;            // events on every prof_every-th search (a recorded event keeps the next kernel from
    long prof_tick = 0;            // overlapping the previous one's tail: sampling keeps that cost out of a timed region)
    bool prof_now = false;
    bool capturing = false;        // the current search is being captured into a HIP graph (search_common): no event queries, no host-side adaptation
    hipEvent_t evring[EV_RING][NUM_EV + 1] = {};
    bool evring_valid[EV_RING][NUM_EV + 1] = {};
    long prof_count = 0;           // profiled searches since the last summary
    hipEvent_t *ev = evring[0];    // event set of the current / most recent search
    bool *ev_valid = evring_valid[0];
    int64_t last_nq = 0;
    int last_chunks = 0, last_mode = 0;
    int last_p2 = 0, last_p2_word = 0;   // lists per query of the last search's second pass (0: none) and the counter that fed it
    unsigned long long *dbg = nullptr;  // diagnostic cycle counters [8192][4][4]
    unsigned int *pace = nullptr;       // [PACE_WORDS] arrival counters of the paced coarse sweep (coarse_flat_kernel.hpp, 67108864)
};

namespace {

bool valid(icd_index *idx) { return idx && idx->magic == 0x1CD10A3Du; }

void free_all(icd_index *x) {
    if (!x) return;
    hipFree(x->corpus); hipFree(x->c16); hipFree(x->levels); hipFree(x->qdev); hipFree(x->q16);
    hipFree(x->qnorm); hipFree(x->qexp); hipFree(x->qbad); hipFree(x->thr0); hipFree(x->shared_thr); hipFree(x->partc_s); hipFree(x->partc_r); hipFree(x->partc_b); hipFree(x->partx_s);
    hipFree(x->part2_s); hipFree(x->part2_r); hipFree(x->part2_b);
    hipFree(x->partx_r); hipFree(x->lists_s); hipFree(x->lists_r); hipFree(x->nflag); hipFree(x->flagged); hipFree(x->scratch_u32);
    hipFree(x->order); hipFree(x->order_key); hipFree(x->order_hist); hipFree(x->cmean);
    hipFree(x->o_scores); hipFree(x->o_ids); hipFree(x->o_adj); hipFree(x->o_adj_raw);
    hipFree(x->o_adj_ids); hipFree(x->o_adj_lv);
    hipFree(x->dbg);
    hipFree(x->pace);
    if (x->h_nflag) hipHostFree(x->h_nflag);
    if (x->h_pin) hipHostFree(x->h_pin);
    if (x->ev_nflag) hipEventDestroy(x->ev_nflag);
    for (int r = 0; r < EV_RING; ++r)
        for (int i = 0; i <= NUM_EV; ++i)
            if (x->evring[r][i]) hipEventDestroy(x->evring[r][i]);
    x->magic = 0;
    delete x;
}

// candidates per exact list: the smallest of 16 / 32 / 64 / 128 that holds k. (32 is round 4's: the serving path searches
// top_k * 2 = 20, and finalize merges at most 512 candidates per query - 8 lists of 64 gave a few hundred flagged queries 80
// two-wave work-groups, 16 lists of 32 give them 160.)
int exact_kp_for(int k) { return k <= 16 ? 16 : (k <= 32 ? 32 : (k <= 64 ? 64 : 128)); }

// chunk count heuristic: enough work-groups to fill the chip, few enough lists to merge. `slots` = the work-groups the
// chip holds at once: ONE per CU for every instantiation of exact_topk (its LDS - two 33-KB operand stages + the candidate
// buffers - is 90 - 130 KB). Round 3 planned against two per CU: 79 query tiles x 4 chunks = 316 work-groups ran as a full
// round of 256 and a second one of 60 (MFMA pipe busy 56 % of the launch); against the real count 79 x 3 = 237 fill 93 % of
// ONE round.
int pick_chunks(int mtiles, int row_tiles, int pmax, int slots) {
    int p = 1;
    if (mtiles * 2 <= slots) {
        p = std::max(1, slots / mtiles);
    } else {
        // the smallest p whose last round of work-groups is at least 90 % full (else the fullest)
        double best = 0;
        int bestp = 1;
        const double want = 0.9;
        for (int c = 1; c <= std::min(pmax, 8); ++c) {
            const long items = (long)mtiles * c;
            const double util = (double)items / (double)(((items + slots - 1) / slots) * slots);
            if (util > best + 1e-9) { best = util; bestp = c; }
            if (util >= want) { bestp = c; break; }
        }
        p = bestp;
    }
    p = std::min(p, pmax);
    p = std::min(p, std::max(1, row_tiles));
    return std::max(1, p);
}

template <int KP, int E, int NW, int CAPV = 64 * E, int BK = 32, int OCC = 1, int GROUP = 32>
int launch_exact(icd_index *x, const ExactArgs &a, int mtiles, hipStream_t s) {
    auto kern = exact_topk_kernel<KP, E, NW, CAPV, BK, OCC, GROUP>;
    const size_t lds = exact_lds_bytes<KP, E, NW, CAPV, BK>();
    static int configured[MAX_DEVICES] = {};   // (guarded by the caller's one-stream-per-handle contract; worst case a repeated call)
    HIP_TRY(ensure_dynamic_lds(kern, x->device, (size_t)((int)lds), configured));
    hipLaunchKernelGGL(kern, dim3(mtiles * a.P), dim3(NW * 64), lds, s, a);
    HIP_TRY(hipGetLastError());
    return ICD_OK;
}

template <int D, int VAR = CF_PRODUCT_VAR, int KP = CO_KP, bool PERSIST = false>
int launch_coarse_flat(icd_index *x, const CoarseFlatArgs &a, int nwg, hipStream_t s) {
    auto kern = coarse_flat_kernel<D, VAR, KP, PERSIST>;
    constexpr int lds = cf_lds_bytes(VAR);
    static int configured[MAX_DEVICES] = {};   // (guarded by the caller's one-stream-per-handle contract; worst case a repeated call)
    HIP_TRY(ensure_dynamic_lds(kern, x->device, (size_t)(lds), configured));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, s, a);
    HIP_TRY(hipGetLastError());
    return ICD_OK;
}

#ifdef ICD_ABLATE_KSPLIT
template <int D, int KP = CO_KP, int TV = 0>
int launch_coarse_ksplit(icd_index *x, const CoarseFlatArgs &a, int nwg, hipStream_t s) {
    auto kern = coarse_ksplit_kernel<D, KP, TV>;
    constexpr int lds = ks_lds_bytes();
    static int configured[MAX_DEVICES] = {};
    HIP_TRY(ensure_dynamic_lds(kern, x->device, (size_t)(lds), configured));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, s, a);
    HIP_TRY(hipGetLastError());
    return ICD_OK;
}
#endif
#ifdef ICD_ABLATE_EXPERIMENTS
template <int D, int KP = CO_KP, int VAR = 0>
int launch_coarse_rg(icd_index *x, const CoarseFlatArgs &a, int nwg, hipStream_t s) {
    auto kern = coarse_rg_kernel<D, KP, VAR>;
    constexpr int lds = rg_lds_bytes();
    static int configured[MAX_DEVICES] = {};   // (guarded by the caller's one-stream-per-handle contract; worst case a repeated call)
    HIP_TRY(ensure_dynamic_lds(kern, x->device, (size_t)(lds), configured));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, s, a);
    HIP_TRY(hipGetLastError());
    return ICD_OK;
}
template <int D, int KP = CO_KP, int VAR = 0>
int launch_coarse_g16(icd_index *x, const CoarseFlatArgs &a, int nwg, hipStream_t s) {
    auto kern = coarse_g16_kernel<D, KP, VAR>;
    constexpr int lds = w8_lds_bytes();
    static int configured[MAX_DEVICES] = {};
    HIP_TRY(ensure_dynamic_lds(kern, x->device, (size_t)(lds), configured));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, s, a);
    HIP_TRY(hipGetLastError());
    return ICD_OK;
}

template <int D, int KP = CO_KP, int VAR = 0>
int launch_coarse_g16r(icd_index *x, const CoarseFlatArgs &a, int nwg, hipStream_t s) {
    auto kern = coarse_g16r_kernel<D, KP, VAR>;
    constexpr int lds = g16_lds_bytes();
    static int configured[MAX_DEVICES] = {};
    HIP_TRY(ensure_dynamic_lds(kern, x->device, (size_t)(lds), configured));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, s, a);
    HIP_TRY(hipGetLastError());
    return ICD_OK;
}

template <int D, int KP = CO_KP, int VAR = 0>
int launch_coarse_w8rg(icd_index *x, const CoarseFlatArgs &a, int nwg, hipStream_t s) {
    auto kern = coarse_w8rg_kernel<D, KP, VAR>;
    constexpr int lds = w8_lds_bytes();
    static int configured[MAX_DEVICES] = {};
    HIP_TRY(ensure_dynamic_lds(kern, x->device, (size_t)(lds), configured));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), lds, s, a);
    HIP_TRY(hipGetLastError());
    return ICD_OK;
}

template <int D, int KP = CO_KP, int VAR = 0>
int launch_coarse_w8(icd_index *x, const CoarseFlatArgs &a, int nwg, hipStream_t s) {
    auto kern = coarse_w8_kernel<D, KP, VAR>;
    constexpr int lds = w8_lds_bytes();
    static int configured[MAX_DEVICES] = {};
    HIP_TRY(ensure_dynamic_lds(kern, x->device, (size_t)(lds), configured));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), lds, s, a);
    HIP_TRY(hipGetLastError());
    return ICD_OK;
}
#endif

// streaming exact kernel + list reduction for a sparse query set (device-side gated when nq_ptr is given).
// p_out = 0: choose the smallest number of output lists (direct tiny-batch path); returns it in *p_used.
constexpr int LDS_LIMIT = 160 * 1024;   // LDS per CU (MI355X_MICROARCH.md)
constexpr size_t PIN_Q_BYTES = 64 * 1024;     // queries of a small host call (16 x 1024 floats)
constexpr size_t PIN_DONE_BYTES = 64;          // the completion word of a ONE-query host call (its own cache line)
constexpr size_t PIN_OUT_BYTES = 96 * 1024;   // their results: nq * k * 36 bytes over the six output arrays (nq * k <= 2730)

// does a pass of qb queries fit LDS with the minimum ring of two stages per wave?
inline bool stream_fits(int kp, int qb, int dim) {
    const int e = kp <= 32 ? 2 : (kp == 64 ? 3 : 4);
    return (size_t)qb * dim * 4 + (size_t)4 * 2 * ST_STAGE_BYTES + (size_t)4 * qb * 64 * e * 8 <= (size_t)LDS_LIMIT;
}

template <int KP, int E, int QB>
int launch_stream(icd_index *x, const float *dq, const int *qlist, const int *nq_ptr, int nq, int max_active,
                  int p_out, int *p_used, hipStream_t s, const float *thr0 = nullptr) {
    const int n = (int)x->n;
    const int per_max = FIN_MAX_CAND / KP;                       // lists one reduce wave can merge
    const int p_cap = FIN_MAX_CAND / KP;                         // lists finalize<false> can merge
    const int nwg_max = std::max(1, std::min(x->num_cu, 256));
    int rows_per_wg = ((n + nwg_max - 1) / nwg_max + 255) / 256 * 256;
    const int nwg = (n + rows_per_wg - 1) / rows_per_wg;
    int plan[8];
    const int levels = plan_reduce_levels(4 * nwg, per_max, p_out, p_cap, plan);
    if (levels <= 0) return fail(ICD_ERR_INVALID, "stream kernel: no reduction plan for %d lists (KP=%d, P=%d)", 4 * nwg, KP, p_out);
    p_out = plan[levels - 1];
    if (p_used) *p_used = p_out;
    StreamArgs a{};
    a.corpus = x->corpus; a.queries = dq; a.qlist = qlist; a.nq_ptr = nq_ptr; a.nq = std::min(nq, max_active);
    a.max_active = max_active; a.n = n; a.dim = x->dim; a.rows_per_wg = rows_per_wg; a.nwg = nwg;
    a.list_scores = x->lists_s; a.list_rows = x->lists_r; a.thr0 = thr0;
    // wave-private LDS ring: as many 8-KB stages per wave as fit next to the queries and candidate buffers
    int stages = 4;
    while (stages > 2 && stream_lds_bytes<KP, E, QB>(x->dim, stages) > (size_t)LDS_LIMIT) --stages;
    a.ring_stages = stages;
    if ((size_t)a.nq * 4 * nwg * KP > x->lists_cap) return fail(ICD_ERR_INVALID, "stream workspace too small");
    for (int l = 0; l < levels; ++l) {
        const size_t cap = (l % 2 == 0) ? x->partx_cap + (size_t)128 * FIN_MAX_CAND_X : x->lists_cap;
        if ((size_t)a.nq * plan[l] * KP > cap) return fail(ICD_ERR_INVALID, "stream workspace too small for reduction level %d", l);
    }
    auto kern = stream_topk_kernel<KP, E, QB>;
    const size_t lds = stream_lds_bytes<KP, E, QB>(x->dim, stages);
    if (lds > (size_t)LDS_LIMIT) return fail(ICD_ERR_INVALID, "stream kernel: dim=%d does not fit LDS with %d queries per pass", x->dim, QB);
    static int configured[MAX_DEVICES] = {};   // (guarded by the caller's one-stream-per-handle contract; worst case a repeated call)
    HIP_TRY(ensure_dynamic_lds(kern, x->device, (size_t)((int)lds), configured));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, s, a);
    HIP_TRY(hipGetLastError());
    int nlists = 4 * nwg;
    for (int l = 0; l < levels; ++l) {   // lists -> partx -> lists -> ... -> partx
        ReduceArgs r{};
        r.list_scores = (l % 2 == 0) ? x->lists_s : x->partx_s; r.list_rows = (l % 2 == 0) ? x->lists_r : x->partx_r;
        r.part_scores = (l % 2 == 0) ? x->partx_s : x->lists_s; r.part_rows = (l % 2 == 0) ? x->partx_r : x->lists_r;
        r.nlists = nlists; r.KP = KP; r.P_out = plan[l];
        r.nq_ptr = nq_ptr; r.nq = a.nq; r.max_active = max_active;
        hipLaunchKernelGGL(reduce_lists_kernel, dim3((a.nq * plan[l] + 3) / 4), dim3(256), 4 * 512 * 8, s, r);
        HIP_TRY(hipGetLastError());
        nlists = plan[l];
    }
    return ICD_OK;
}

// The reference's call shape - ONE query per MilvusService.search call (services/milvus_service.py:280-285), or two - as ONE
// launch: stream_topk_kernel<..., ONE = true> (stream_kernel.hpp) spreads the rows over every CU, merges the lists in the
// kernel (last-arriver ticket) and writes the final outputs itself. Geometry: rows per wave and step = the CU's share / 4,
// rounded up to whole 8-row LDS-DMA pieces (at most 64); the ring takes the stages that fit next to the candidate buffers.
struct StreamOnePlan { int rps, rows_per_wg, nwg, stages; };
inline bool plan_stream_one(int n, int dim, int num_cu, int qb, int cap_entries, StreamOnePlan *p) {
    const int ncu = std::max(1, std::min(num_cu, 256));
    const int per_cu = (n + ncu - 1) / ncu;
    const int rps = std::min(64, ((per_cu + 3) / 4 + 7) / 8 * 8);
    const int steps = (per_cu + 4 * rps - 1) / (4 * rps);
    p->rps = rps; p->rows_per_wg = 4 * rps * steps; p->nwg = (n + p->rows_per_wg - 1) / p->rows_per_wg;
    const size_t fixed = (size_t)qb * dim * 4 + (size_t)4 * qb * cap_entries * 8;
    if (fixed + (size_t)4 * 2 * rps * 128 > (size_t)LDS_LIMIT) return false;
    p->stages = (int)std::min<size_t>(8, ((size_t)LDS_LIMIT - fixed) / ((size_t)4 * rps * 128));
    return p->nwg >= 1 && p->nwg <= 256 && p->stages >= 2;
}

template <int KP, int E, int QB>
int launch_stream_one(icd_index *x, const float *dq, int nq, const FinArgs &f, hipStream_t s, const float *host_q = nullptr, bool poll_done = false) {
    StreamOnePlan pl;
    if (!plan_stream_one((int)x->n, x->dim, x->num_cu, QB, 64 * E, &pl)) return fail(ICD_ERR_INVALID, "single-launch stream kernel: no plan for n=%lld dim=%d", (long long)x->n, x->dim);
    if ((size_t)nq * pl.nwg * KP * 2 > x->lists_cap) return fail(ICD_ERR_INVALID, "stream workspace too small");
    StreamArgs a{};
    a.corpus = x->corpus; a.queries = dq; a.qlist = nullptr; a.nq_ptr = nullptr; a.nq = nq; a.max_active = nq;
    a.n = (int)x->n; a.dim = x->dim; a.rows_per_wg = pl.rows_per_wg; a.nwg = pl.nwg; a.ring_stages = pl.stages;
    a.rows_per_step = pl.rps;
    a.wg_keys = reinterpret_cast<u64 *>(x->lists_s);                    // (the per-wave lists' workspace: unused by this form)
    a.ticket = reinterpret_cast<u64 *>(x->nflag + 6);                   // nflag[6..7]: 8-byte aligned, zeroed at create, only ever counts up
    a.fin = f;
    a.fin.counters = x->nflag; a.fin.host_counters = x->h_nflag_dev;
    if (poll_done) {
        a.done = reinterpret_cast<u64 *>(x->h_pin_dev + PIN_Q_BYTES + PIN_OUT_BYTES);
        a.done_value = ++x->done_seq;
        x->done_armed = true;
    }
    auto kern = stream_topk_kernel<KP, E, QB, true>;
    // (at least 84 KB: ONE work-group per CU whatever the corpus size - the fence-free sc1 hand-off of the kernel's tail is the
    //  form measured for one work-group per CU, MI355X_MICROARCH.md visibility table)
    const size_t lds = std::max<size_t>(stream_one_lds_bytes<KP, E, QB>(x->dim, pl.stages, pl.rps), (size_t)84 * 1024);
    static int configured[MAX_DEVICES] = {};
    if constexpr (QB == 1) {
        if (host_q) {   // the vector in the kernel arguments: no copy command in front of the launch
            auto kern_in = stream_one_inline_kernel<KP, E>;
            static int configured_in[MAX_DEVICES] = {};
            HIP_TRY(ensure_dynamic_lds(kern_in, x->device, (size_t)LDS_LIMIT, configured_in));
            StreamInlineQuery iq;
            memcpy(iq.v, host_q, (size_t)x->dim * sizeof(float));
            a.queries = nullptr;
            hipLaunchKernelGGL(kern_in, dim3(pl.nwg), dim3(256), lds, s, a, iq);
            HIP_TRY(hipGetLastError());
            return ICD_OK;
        }
    }
    HIP_TRY(ensure_dynamic_lds(kern, x->device, (size_t)LDS_LIMIT, configured));
    hipLaunchKernelGGL(kern, dim3(pl.nwg), dim3(256), lds, s, a);
    HIP_TRY(hipGetLastError());
    return ICD_OK;
}

template <bool RESCORE, bool DEEP, int EWM>
int launch_finalize_t(icd_index *x, const FinArgs &a, hipStream_t s) {
    auto kern = finalize_kernel<RESCORE, DEEP, EWM>;
    const size_t lds = 4 * fin_wave_lds_bytes(RESCORE, x->dim, a.lds_cand > 0 ? a.lds_cand : a.P * a.KP, EWM);
    static int configured[MAX_DEVICES] = {};   // (guarded by the caller's one-stream-per-handle contract; worst case a repeated call)
    HIP_TRY(ensure_dynamic_lds(kern, x->device, (size_t)((int)(4 * fin_wave_lds_bytes(RESCORE, x->dim, (!RESCORE && EWM >= 4) ? FIN_MAX_CAND_X : FIN_MAX_CAND, EWM))), configured));
    hipLaunchKernelGGL(kern, dim3((a.nq + 3) / 4), dim3(256), lds, s, a);
    HIP_TRY(hipGetLastError());
    return ICD_OK;
}

template <bool RESCORE>
int launch_finalize(icd_index *x, const FinArgs &a, hipStream_t s) {
    // one wave per query: up to ~28 waves per CU are resident, so a launch of a few thousand queries is a single
    // round of waves and its duration is one wave's latency: prefetch the rescoring rows deeper there
    const bool deep = RESCORE && a.nq <= 8 * x->num_cu;
    if (!RESCORE && std::max(a.lds_cand, a.P * a.KP) > FIN_MAX_CAND) return launch_finalize_t<false, false, 4>(x, a, s);   // (32 keys per lane)
    if (a.k <= 32) return deep ? launch_finalize_t<RESCORE, true, 1>(x, a, s) : launch_finalize_t<RESCORE, false, 1>(x, a, s);
    return deep ? launch_finalize_t<RESCORE, true, 4>(x, a, s) : launch_finalize_t<RESCORE, false, 4>(x, a, s);
}

void rec(icd_index *x, int i, hipStream_t s) {
    if (x->prof_now) {
        hipEventRecord(x->ev[i], s);
        x->ev_valid[i] = true;
    }
}

struct Outs {
    float *scores; long long *ids;
    double *adj; float *adj_raw; long long *adj_ids; int *adj_lv;
};

// rows 0, stride, 2 stride, ... of the fp32 corpus as a query batch (the corpus-shape probe of icd_index_create)
__global__ void gather_rows_kernel(const float *src, float *dst, long long stride, int dim, int rows) {
    const int r = blockIdx.x;
    if (r >= rows) return;
    const float4 *s4 = reinterpret_cast<const float4 *>(src + (size_t)r * stride * dim);
    float4 *d4 = reinterpret_cast<float4 *>(dst + (size_t)r * dim);
    for (int i = threadIdx.x; i < dim / 4; i += blockDim.x) d4[i] = s4[i];
}

// Enqueue a search whose queries and outputs are device pointers.
int search_device(icd_index *x, const float *dq, int nq, int k, int mode, const Outs &o, hipStream_t s) {
    const int row_tiles = (int)((x->n + 127) / 128);
    const bool use_fast = (mode == ICD_MODE_AUTO) && x->fast && k <= FAST_MAX_K && (x->dim == 768 || x->dim == 1024);
    x->prof_now = !x->capturing && x->profiling && (x->prof_tick++ % x->prof_every == 0);
    if (x->prof_now) {
        const int slot = (int)(x->prof_count % EV_RING);
        x->ev = x->evring[slot];
        x->ev_valid = x->evring_valid[slot];
        ++x->prof_count;
        for (int i = 0; i <= NUM_EV; ++i) x->ev_valid[i] = false;
    }
    x->last_nq = nq;
    x->fallback_word = 0;
    x->last_p2 = 0;
    x->last_mode = use_fast ? ICD_MODE_AUTO : ICD_MODE_EXACT;
    rec(x, 0, s);

    FinArgs f{};
    f.k = k; f.queries = dq; f.corpus = x->corpus; f.dim = x->dim; f.qnorm = x->qnorm; f.qbad = x->qbad;
    f.qexp = x->qexp; f.rmax = x->rmax_scaled; f.cexp = x->cexp; f.eps_rel = EPS_REL; f.nflag = x->nflag; f.flagged = x->flagged;
    if (use_fast) f.thr0 = x->thr0;
    if (x->cmean) { f.rmax_unc = x->rmax_unc_scaled; f.eps_f32 = 2.0f * (float)x->dim * 5.9604645e-8f; }
    f.levels = x->levels; f.id_base = x->id_base;
    f.out_scores = o.scores; f.out_ids = o.ids; f.out_adj = o.adj; f.out_adj_raw = o.adj_raw;
    f.out_adj_ids = o.adj_ids; f.out_adj_levels = o.adj_lv;

    // exact-kernel configuration (full run, or fallback over the flagged list)
    const int kpx = exact_kp_for(k);
    const int nwx = kpx <= 64 ? 4 : 2;
    int wg_per_cu_x = kpx <= 32 ? 2 : 1;   // work-groups of exact_topk a CU holds (its LDS)
#ifdef ICD_ABLATE
    if (kpx == 32 && getenv("ICD_EXACT_CFG") && atoi(getenv("ICD_EXACT_CFG")) == 3) wg_per_cu_x = 1;
#endif
    const int ex = kpx == 16 ? 1 : (kpx <= 64 ? 2 : 3);
    const int bmq = nwx * 32;
    const int mtx = (nq + bmq - 1) / bmq;
    // lists per query finalize<false> merges: 2048 candidates at k <= 64 (its top-k prefilter keeps that cheap), 512 above
    const int cand_x = kpx <= 64 ? FIN_MAX_CAND_X : FIN_MAX_CAND;
    const int pmax_x = std::min(64, cand_x / kpx);
    const int p_dense_max = cand_x / kpx;   // chunk limit of the device-sized partition of a flagged list
    // sparse: streaming kernel (+ list reduction) instead of / next to the MFMA exact kernel
    // queries per pass: host-known for direct calls (1, 2, 4 or 8), 8 for the device-gated fallback
    const int p_sparse = FIN_MAX_CAND / kpx;   // lists per slot that finalize<false> merges (32 / 8 / 4)
    // How many flagged queries still go to the streaming kernel: it re-reads the corpus once per 8 queries (~37 us at
    // 37k rows); the fp32-MFMA kernel, its short list cut into up to 2048 / KP chunks, needs ~0.22 ms for any count up to a
    // few hundred: they meet near 45 (ST_FALLBACK_MAX_ACTIVE, stream_kernel.hpp); the list workspace caps it by k.
    // (the first reduction level leaves at most 512 lists per slot in the second workspace)
    const int sparse_max = (int)std::min<size_t>(std::min<size_t>(ST_FALLBACK_MAX_ACTIVE, x->lists_cap / ((size_t)1024 * kpx)),
                                                 kpx == 16 ? (size_t)ST_FALLBACK_MAX_ACTIVE : x->partx_cap / ((size_t)512 * kpx));
    auto run_stream = [&](const int *qlist, const int *nq_ptr, int nqs, int p_out, int *p_used, const float *thr0 = nullptr) -> int {
        int qb = nq_ptr ? 8 : (nqs <= 1 ? 1 : (nqs <= 2 ? 2 : (nqs <= 4 ? 4 : 8)));
        const int max_act = nq_ptr ? sparse_max : std::max(sparse_max, nqs);   // (a direct call - up to ST_MAX_ACTIVE queries - is not gated)
        while (qb > 1 && !stream_fits(kpx, qb, x->dim)) qb >>= 1;
#define ICD_ST(KPV, EV) \
        (qb == 1 ? launch_stream<KPV, EV, 1>(x, dq, qlist, nq_ptr, nqs, max_act, p_out, p_used, s, thr0) : \
         qb == 2 ? launch_stream<KPV, EV, 2>(x, dq, qlist, nq_ptr, nqs, max_act, p_out, p_used, s, thr0) : \
         qb == 4 ? launch_stream<KPV, EV, 4>(x, dq, qlist, nq_ptr, nqs, max_act, p_out, p_used, s, thr0) : \
                   launch_stream<KPV, EV, 8>(x, dq, qlist, nq_ptr, nqs, max_act, p_out, p_used, s, thr0))
        if (kpx == 16) return ICD_ST(16, 2);
        if (kpx == 32) return ICD_ST(32, 2);
        if (kpx == 64) return ICD_ST(64, 3);
        return ICD_ST(128, 4);
#undef ICD_ST
    };
    auto fit_p = [&](int p, size_t cap, int kp) {
        while (p > 1 && (size_t)nq * p * kp > cap) --p;
        return p;
    };
    auto run_exact = [&](const int *qlist, const int *nq_ptr, int px, bool mfma, bool stream, bool stream_launch = true, bool track_run = false, const float *thr0 = nullptr) -> int {
        const int sparse_here = stream_launch ? sparse_max : 0;   // (streaming pair left out: the MFMA kernel takes every count >= 1)
        if (!nq_ptr) x->last_chunks = px;   // (the fallback keeps the coarse pass's chunk count)
        int rc = ICD_OK;
        if (stream && stream_launch) {
            int used = px;
            rc = run_stream(qlist, nq_ptr, nq, mfma ? px : 0, &used, thr0);   // alone: fewest output lists
            if (rc) return rc;
            px = used;
            if (!nq_ptr) x->last_chunks = px;
        }
        int px_dense = 0;
        if (mfma) {
            // The MFMA kernel may have to take EVERY query (all of them flagged): its list count is sized against
            // the workspace like a full exact run; only the streaming kernel (<= ST_MAX_ACTIVE slots) uses px.
            int pm = px;
            if (stream) {
                pm = pick_chunks(mtx, row_tiles, pmax_x, x->num_cu * wg_per_cu_x);
                pm = fit_p(pm, x->partx_cap, kpx);
                if ((size_t)nq * pm * kpx > x->partx_cap) return fail(ICD_ERR_INVALID, "workspace too small for nq=%d k=%d", nq, k);
                const int tiles_per = (row_tiles + pm - 1) / pm;
                pm = (row_tiles + tiles_per - 1) / tiles_per;
                px_dense = pm;
            }
            ExactArgs a{};
            a.corpus = x->corpus; a.queries = dq; a.qlist = qlist; a.nq_ptr = nq_ptr; a.nq = nq;
            a.min_active = stream ? sparse_here : 0;
            a.adaptive_one_per_cu = wg_per_cu_x > 1 ? x->num_cu : 0;
            a.adaptive_max_p = stream ? p_dense_max : 0;   // fallback: the chunk count follows the actual flagged count
            a.n = (int)x->n; a.dim = x->dim; a.P = pm;
            a.rows_per_chunk = ((row_tiles + pm - 1) / pm) * 128;
            a.part_scores = x->partx_s; a.part_rows = x->partx_r; a.thr0 = thr0;
            // (LDS: k <= 16 two work-groups per CU - 17 KB of stage + 60-entry buffers; k <= 64 one of four waves - 34 KB +
            //  112-entry buffers; larger k two waves. Eight waves on 64-entry buffers at k <= 32 - two per SIMD - were built and
            //  are slower, 9.2 against 7.2 ms per 10 000 queries: a 64-entry buffer with 32 kept is compacted after every append)
#ifdef ICD_ABLATE
            if (kpx == 16 && getenv("ICD_EXACT_CFG") && atoi(getenv("ICD_EXACT_CFG")) == 1) rc = launch_exact<16, 1, 4>(x, a, mtx, s);
            else if (kpx == 16 && getenv("ICD_EXACT_CFG") && atoi(getenv("ICD_EXACT_CFG")) == 2) rc = launch_exact<16, 1, 4, 64, 16, 1>(x, a, mtx, s);
            else
#endif
            if (kpx == 16) rc = launch_exact<16, 1, 4, 60, 16, 2>(x, a, mtx, s);
#ifdef ICD_ABLATE
            else if (kpx == 32 && getenv("ICD_EXACT_CFG") && atoi(getenv("ICD_EXACT_CFG")) == 3) rc = launch_exact<32, 2, 4, 112>(x, a, mtx, s);
#endif
            else if (kpx == 32) rc = launch_exact<32, 1, 4, 62, 16, 2, 16>(x, a, mtx, s);
            else if (kpx == 64) rc = launch_exact<64, 2, 4, 112>(x, a, mtx, s);
            else rc = launch_exact<128, 3, 2>(x, a, mtx, s);
            if (rc) return rc;
        }
        rec(x, 4, s);
        FinArgs g = f;
        g.part_scores = x->partx_s; g.part_rows = x->partx_r; g.P = px; g.KP = kpx;
        g.P_dense = px_dense; g.sparse_max = sparse_here; g.track_run = track_run ? 1 : 0; g.lds_cand = std::max(px, (stream && mfma) ? p_dense_max : px_dense) * kpx;
        g.dense_grid = mtx * std::max(1, px_dense); g.dense_bmq = bmq; g.dense_max_p = (stream && mfma) ? p_dense_max : 0; g.dense_one_per_cu = wg_per_cu_x > 1 ? x->num_cu : 0; g.n_rows = (int)x->n;
        g.nq = nq; g.nq_ptr = nq_ptr; g.qlist = qlist;
        g.counters = x->nflag; g.host_counters = x->h_nflag_dev;   // (the last launch of every search: no separate copy)
        rc = launch_finalize<false>(x, g, s);
        rec(x, 5, s);
        (void)ex;
        return rc;
    };

    // a host caller's ONE query whose copy search_common left out (host_q): every path but the single-launch kernel wants it
    // in device memory after all - the copy it would have got, enqueued before anything that reads dq
    auto stage_host_query = [&]() -> int {
        if (!x->host_q) return ICD_OK;
        const size_t qbytes = (size_t)x->dim * sizeof(float);
        memcpy(x->h_pin, x->host_q, qbytes);
        HIP_TRY(hipMemcpyAsync(x->qdev, x->h_pin, qbytes, hipMemcpyHostToDevice, s));
        x->host_q = nullptr;
        return ICD_OK;
    };
    // small batches (the reference's one-query-per-call shape): stream the corpus once, exact, no coarse pass
    const bool stream_ok = x->dim % (32 * ST_PF) == 0 && stream_fits(kpx, 1, x->dim);
    // (k > 16 needs the 64-entry lists of the streaming kernel, ~0.9 ms per 16 queries: the coarse pass is faster there)
    const bool tiny = stream_ok && nq <= (use_fast ? (k <= 16 ? 16 : 0) : ST_MAX_ACTIVE);
    // (the fallback counter starts every search at zero: the exact-only paths clear it here, the fast path in its
    //  query-prep launch - one launch less per step)
    if (tiny) {
        x->last_mode = ICD_MODE_EXACT;
        x->last_chunks = p_sparse;
        StreamOnePlan pl;
        // (up to FOUR queries: at eight the sweep is bound by the lanes' fmaf chains, not by memory - 40 of a wave's 64 lanes hold
        //  rows in this form - and the general path is 12 us faster: profiles/r05_single_query.log)
        const int qb1 = nq <= 1 ? 1 : (nq <= 2 ? 2 : 4);
        bool one = nq <= 4 && kpx == 16 && x->opt_stream_one && plan_stream_one((int)x->n, x->dim, x->num_cu, qb1, 64 * 2, &pl) &&
                   (size_t)nq * pl.nwg * kpx * 2 <= x->lists_cap;
        if (one) {   // up to four queries (the reference's call shape is ONE; a /query request batches its D diagnoses): ONE launch,
                     // no memset, no reduction, no finalize
            rec(x, 3, s);
            const float *hq = qb1 == 1 ? x->host_q : nullptr;   // (a host caller's ONE query: search_common left the copy out)
            if (!hq) { const int rcq = stage_host_query(); if (rcq) return rcq; }
            const bool poll = x->host_one_call && (x->opt_host_one & 2) != 0;
            const int rc1 = qb1 == 1 ? launch_stream_one<16, 2, 1>(x, dq, 1, f, s, hq, poll) : qb1 == 2 ? launch_stream_one<16, 2, 2>(x, dq, 2, f, s)
                          : launch_stream_one<16, 2, 4>(x, dq, (int)nq, f, s);
            rec(x, 4, s);
            rec(x, 5, s);
            return rc1;
        }
        { const int rcq = stage_host_query(); if (rcq) return rcq; }
        HIP_TRY(hipMemsetAsync(x->nflag, 0, sizeof(int), s));
        rec(x, 3, s);
        return run_exact(nullptr, nullptr, p_sparse, false, true);
    }
    { const int rcq = stage_host_query(); if (rcq) return rcq; }
    if (!use_fast && kpx > 32 && x->opt_exact_narrow && nq > ST_MAX_ACTIVE && row_tiles >= 64) {
        // ---- k > 32: NARROW certified lists -----------------------------------------------------------------------------------
        // Lists of KP >= k need 64- or 128-entry candidate buffers: one work-group of four waves per CU at k <= 64, of two above
        // (0.39 / 0.15 of the fp32 MFMA peak at k = 50 / 100, against 0.59 at k = 20). Instead: the k <= 32 configuration (lists
        // of 32, two work-groups per CU) over ROW-STRIDED chunks - chunk c holds rows c, c + P, ... so that neighbouring rows
        // (an ICD family in code order) spread evenly - with P large enough that a list holds ~k / P << 32 members of the top-k,
        // and finalize CHECKS it: a full list whose worst kept key beats the k-th best of the merge may have dropped a member
        // of the top-k; such a query is re-searched with KP >= k lists (the flagged-list machinery of the AUTO fallback).
        // Results are the exact top-k either way.
        HIP_TRY(hipMemsetAsync(x->nflag, 0, sizeof(int), s));
        const int slots = x->num_cu * 2;
        const int mtx_n = (nq + 127) / 128;                   // query tiles of THIS configuration (four waves x 32 queries; `mtx` above counts the KP >= k configuration's tiles, 64 queries at k > 64: ADVICE r5)
        const int p_need = std::max(2, (k + 7) / 8);          // ~8 members of the top-k per list when they spread evenly
        int pn = p_need;
        {   // the count at or above p_need whose last round of work-groups is fullest
            double best = 0;
            for (int c = p_need; c <= std::min(FIN_MAX_CAND_X / 32, std::max(p_need + 8, slots / std::max(1, mtx_n))); ++c) {   // (a short batch: enough chunks to fill the chip)
                const long items = (long)mtx_n * c;
                const double util = (double)items / (double)(((items + slots - 1) / slots) * slots);
                if (util > best + 1e-9) { best = util; pn = c; }
            }
        }
        if ((size_t)nq * pn * 32 <= x->partx_cap && pn * 32 <= FIN_MAX_CAND_X) {
            x->last_chunks = pn;
            x->fallback_word = 0;
            rec(x, 3, s);
            ExactArgs a{};
            a.corpus = x->corpus; a.queries = dq; a.nq = nq; a.n = (int)x->n; a.dim = x->dim; a.P = pn; a.strided = 1;
            a.rows_per_chunk = 128;   // (unused by the strided form)
            a.part_scores = x->partx_s; a.part_rows = x->partx_r;
            int rc = launch_exact<32, 1, 4, 62, 16, 2, 16>(x, a, mtx_n, s);
            if (rc) return rc;
            rec(x, 4, s);
            FinArgs g = f;
            g.part_scores = x->partx_s; g.part_rows = x->partx_r; g.P = pn; g.KP = 32; g.nq = nq; g.lds_cand = pn * 32;
            g.narrow_check = 1; g.nflag = x->nflag; g.flagged = x->flagged;
            rc = launch_finalize<false>(x, g, s);
            if (rc) return rc;
            rec(x, 5, s);
            // the queries the check could not clear (none on data whose rows spread): KP >= k lists over the flagged list
            int px = std::min(p_sparse, row_tiles);
            return run_exact(x->flagged, x->nflag, px, true, stream_ok, true, false, nullptr);
        }
    }
    if (!use_fast) {
        HIP_TRY(hipMemsetAsync(x->nflag, 0, sizeof(int), s));
        int px = pick_chunks(mtx, row_tiles, pmax_x, x->num_cu * wg_per_cu_x);
        px = fit_p(px, x->partx_cap, kpx);
        if ((size_t)nq * px * kpx > x->partx_cap) return fail(ICD_ERR_INVALID, "workspace too small for nq=%d k=%d", nq, k);
        {
            const int tiles_per = (row_tiles + px - 1) / px;
            px = (row_tiles + tiles_per - 1) / tiles_per;
        }
        x->last_chunks = px;
        rec(x, 3, s);
        return run_exact(nullptr, nullptr, px, true, false);
    }

    // ---- AUTO: prep -> coarse -> finalize(certify + rescore) -> exact fallback ------------------
    const int nq_pad = ((nq + 127) / 128) * 128;
    ConvertArgs cv{};
    cv.src = dq; cv.dst = x->q16; cv.rows = nq; cv.rows_pad = nq_pad; cv.dim = x->dim; cv.mode = 0;
    cv.norm = x->qnorm; cv.scale_exp = x->qexp; cv.bad = x->qbad; cv.zero_u32 = x->shared_thr; cv.zero_i32 = x->nflag;
    cv.zero_u32b = x->shared_thr + x->max_nq_pad;
    hipLaunchKernelGGL(convert_rows_kernel, dim3((nq_pad + 3) / 4), dim3(256), 0, s, cv);
    HIP_TRY(hipGetLastError());
    rec(x, 1, s);

    const int mtc = nq_pad / 128;
    // corpus tiles to sweep and tiles per work-group (flat_partition.hpp: a few zero tiles behind the corpus buy a tile
    // count that shares a factor with U, i.e. work-groups that stream the same tiles together)
    const int ctiles_min = (int)((x->n + 127) / 128);
    const FlatPlan plan = plan_flat_tiles(mtc, ctiles_min, x->n_pad / 128 - ctiles_min, x->num_cu);
    const int ctiles = plan.ctiles;
    int pc = 0;
    const bool wide_lists = k > x->opt_wide_from && x->dim == 768;
    // ---- adaptive list count of large batches -------------------------------------------------------------------------
    // The counters of the previous search sit in pinned host memory once its copy event has completed (hipEventQuery: no
    // wait). If that search was a large batch on the narrow plan and its second coarse pass had to take in a quarter of
    // the queries and certified most of them, the corpus is one of tight families (ICD sibling codes): large batches then
    // start with the second pass's list count (1.5 instead of 2.2 ms per 10 000 queries there; 7 % slower on Gaussian data,
    // which is why it is not the default). Every WIDE_REPROBE-th large search runs narrow again and decides anew.
    const bool large = nq >= WIDE_MIN_NQ;
    // (while the stream is being captured into a HIP graph nothing may be queried: the plan is the one the host state gives now,
    //  and it is frozen into the graph - a performance decision only, the replayed search returns the exact top-k either way)
    const bool counters_in = !x->capturing && hipEventQuery(x->ev_nflag) == hipSuccess;   // (every search enqueued so far has completed)
    if (counters_in) {
        const int run = x->h_nflag[4];
        if (x->sparse_disarmed && run < x->sparse_need) x->sparse_need = std::min(x->sparse_need * 2, 1 << 16);   // an incident
        x->sparse_run_seen = run;
    }
    const bool sparse_off = x->adapt_enabled && x->sparse_run_seen >= x->sparse_need;
    x->sparse_disarmed = sparse_off;
    if (x->p2_eval_pending && counters_in) {
        x->p2_clean = x->h_nflag[0] > PASS2_SKIP ? 0 : std::min(x->p2_clean + 1, 1 << 20);
        x->p2_eval_pending = false;
    }
    if (x->last_narrow_large && counters_in) {
        const int f0 = x->h_nflag[0], f2 = x->h_nflag[2];
        const long long n_prev = x->last_narrow_nq;   // (x->last_nq already holds the CURRENT batch's size: ADVICE r3)
        if (n_prev > 0) {
            if (!x->wide_mode && (long long)f0 * 4 > n_prev && (long long)f2 * 2 < f0) { x->wide_mode = true; x->wide_runs = 0; }
            else if (x->wide_mode && (long long)f0 * 20 <= n_prev) x->wide_mode = false;   // (a narrow re-probe that certified 95 %)
        }
        x->last_narrow_large = false;
    }
    bool wide_now = false;
    if (large && x->wide_mode && x->chunks_override == 0 && k <= 32) wide_now = (++x->wide_runs % WIDE_REPROBE) != 0;
    const int kp_c = wide_lists ? CO_KP_WIDE : CO_KP;
    CoarseFlatArgs a2{};      // the second pass's arguments (filled next to the first pass's)
    int nwg2 = 0, p2 = 0;
    {
        // ---- product: flat partition of the (query tile x corpus tile) grid over the CUs (coarse_flat_kernel.hpp) ----
        CoarseFlatArgs a{};
        a.q16 = x->q16; a.c16 = x->c16; a.nq = nq; a.n = (int)x->n; a.n_pad = ctiles * 128; a.ctiles = ctiles;
        a.total_units = mtc * ctiles;
        // A query's lists should number at least two of comparable length: the certificate compares against the
        // largest score any list may have dropped, and with one list that is the query's own 16th best (8 % of
        // Gaussian queries then fail, profiles/r01_sizes_before_pmin2.log); with two or more it is about rank 32.
        a.list_tiles = std::max(1, (ctiles + 1) / 2);
        // Larger k: every list keeps KP candidates and ends on its own KP-th best, so the bound the certificate
        // compares the k-th best against sits near rank KP P / 2 of the whole corpus: ask for about k / 4 lists of 16.
        // Above k = 64 (dim 768) the lists keep 24: a query fails the certificate when ONE list holds more than KP of
        // the ~1.3 k rows around its top-k - with 16 that happens to 3-5 of 10 000 queries at k = 100 and costs an exact
        // corpus sweep per batch (0.34 ms); with 24 per list and about k / 6 lists it did not happen. The wider lists make
        // the coarse pass ~20 % slower (lower thresholds, more appends), so they only pay where that sweep is the larger
        // cost: k = 100 1.69 -> 1.54 ms, k = 32 would go 0.87 -> 0.99 (profiles/r02_tile_planner_and_shapes.log).
        if (wide_lists) a.list_tiles = std::max(1, std::min(a.list_tiles, ctiles / ((k + 5) / 6)));
        else if (k > 8) a.list_tiles = std::max(1, std::min(a.list_tiles, ctiles / ((k + 3) / 4)));
        // The bootstrap level (6th best of the first boot_tiles * 128 rows of a list) must stay far below the k-th best
        // of the whole corpus or the list's bound lands inside the window: fewer tiles for larger k (measured at
        // k = 48 with 8 tiles: 1.6 % of the queries uncertified).
        a.boot_tiles = std::max(1, std::min(CO_BOOT_TILES, 96 / std::max(1, k)));
        a.sparse_from = CO_SPARSE_FROM;
        int U = plan.U;
#ifdef ICD_ABLATE
        if (const char *e = getenv("ICD_FLAT_U")) U = std::max(U, atoi(e));   // A/B: tiles per work-group
        if (const char *e = getenv("ICD_FLAT_LIST")) a.list_tiles = std::max(1, atoi(e));   // A/B: tiles per list
        if (const char *e = getenv("ICD_FLAT_BOOT")) a.boot_tiles = std::max(0, atoi(e));   // A/B: bootstrap tiles
        if (const char *e = getenv("ICD_FLAT_SPARSE")) a.sparse_from = std::max(0, atoi(e));   // A/B: first tile of a list with the group pre-filter
#endif
        if (x->chunks_override > 0) {   // test hook: about `chunks` lists per query, every work-group's run one list
            U = std::max(1, (ctiles + x->chunks_override - 1) / x->chunks_override);
            a.list_tiles = ctiles;
        } else if (wide_now) {
            // Wide mode: about PASS2_CHUNKS lists per query, cut out of the SAME long sweeps as the narrow plan (a work-group
            // keeps its queries in registers and its ring running over ~100 tiles and closes a list every 16). Until round 4
            // every list was its own work-group (1 576 of them, six rounds on 256 CUs, each loading its queries and refilling
            // the ring): 0.67 against 0.60 ms on the family corpus, same lists (profiles/r04_wide_long_sweeps.log).
            a.list_tiles = std::max(1, (ctiles + PASS2_CHUNKS - 1) / PASS2_CHUNKS);
#ifdef ICD_ABLATE
            if (getenv("ICD_WIDE_SHORT_SWEEP")) { U = a.list_tiles; a.list_tiles = ctiles; }   // A/B: round 3's partition
#endif
        }
        auto lists_needed_lt = [&](int u, int list_tiles) {   // the largest number of lists of any query tile (same rule as the kernel)
            int worst = 0;
            for (int m = 0; m < mtc; ++m) {
                const long long m1 = (long long)(m + 1) * ctiles;
                const int wl = (int)((m1 - 1) / u);   // last work-group touching the query tile
                worst = std::max(worst, flat_first_ordinal(m, wl + 1, ctiles, u, list_tiles));
            }
            return worst;
        };
        auto lists_needed = [&](int u) { return lists_needed_lt(u, a.list_tiles); };
        int P = lists_needed(U);
        // too many lists for the workspace or for finalize's candidate window: longer lists first (the balance of the
        // partition is untouched), more tiles per work-group only when a list already spans the corpus
        while (P > COARSE_MAX_P || P * kp_c > FIN_MAX_CAND || (size_t)nq * P * kp_c > x->partc_cap) {
            if (a.list_tiles < ctiles) a.list_tiles = std::min(ctiles, a.list_tiles + std::max(1, a.list_tiles / 8));
            else if (U < ctiles) U = std::min(ctiles, U + std::max(1, U / 4));
            else break;
            P = lists_needed(U);
        }
        if (P > COARSE_MAX_P || P * kp_c > FIN_MAX_CAND || (size_t)nq * P * kp_c > x->partc_cap)
            return fail(ICD_ERR_INVALID, "coarse workspace too small for nq=%d (lists per query %d)", nq, P);
        a.units_per_wg = U; a.P = P;
        {
            a.pos_period = flat_class_period(U, ctiles);
            // A class of work-groups that stream the same tiles should have about a dozen members per XCD-local
            // group: many more and they all hit the same L2 channel at the same time (measured slower), so large
            // classes are split into sub-classes (l mod T s also start on the same tile).
            const int nwg_ = (a.total_units + U - 1) / U;
            const int members = nwg_ / std::max(1, a.pos_period);
            int split = (members + 11) / 12;
#ifdef ICD_ABLATE
            if (const char *e = getenv("ICD_XCD_MODE")) {   // A/B switch: 0 identity, 1 XCD swizzle only, 2 classes unsplit
                const int m = atoi(e);
                if (m == 0) a.pos_period = 0;
                else if (m == 1) a.pos_period = 1 << 30;
                else if (m == 2) split = 1;
            }
#endif
            if (a.pos_period > 0 && a.pos_period < (1 << 20)) a.pos_period *= std::max(1, split);
        }
        a.part_scores = x->partc_s; a.part_rows = x->partc_r; a.bounds = x->partc_b; a.shared_thr = x->shared_thr;
        a.dbg = x->dbg;
        pc = P;
        x->last_chunks = P;
        const int nwg = (a.total_units + U - 1) / U;
        a.nwg_virtual = nwg;
        // ---- the second pass's plan: the same sweep cut into about PASS2_CHUNKS lists per query, every work-group's run
        // one list (the shape icd_index_set_chunks asks for). Everything the host must know is independent of how many
        // queries will be flagged: U2, the list slots per query (worst case over all query tiles of the full batch), the
        // logical work-group count of a full batch. The kernel sizes the sweep from the flagged count on the device.
        if (x->pass2_enabled && x->p2_clean < PASS2_DISARM_AFTER && P * kp_c < PASS2_BELOW && x->part2_s) {
            int U2 = std::max(1, (ctiles + PASS2_CHUNKS - 1) / PASS2_CHUNKS);
            p2 = lists_needed_lt(U2, ctiles);
            while ((p2 > PASS2_MAX_P || (size_t)nq * p2 * CO_KP > x->part2_cap) && U2 < ctiles) {
                U2 = std::min(ctiles, U2 + std::max(1, U2 / 8));
                p2 = lists_needed_lt(U2, ctiles);
            }
            if (p2 <= PASS2_MAX_P && (size_t)nq * p2 * CO_KP <= x->part2_cap && p2 * CO_KP > P * kp_c) {
                a2 = a;
                a2.nq_ptr = nullptr; a2.qlist = nullptr;   // (set at the launch: which flag word / list feeds it)
                a2.units_per_wg = U2; a2.list_tiles = ctiles; a2.P = p2;
                a2.boot_tiles = a.boot_tiles;
                a2.pos_period = flat_class_period(U2, ctiles);
                nwg2 = (int)(((long long)mtc * ctiles + U2 - 1) / U2);
                {
                    const int members = nwg2 / std::max(1, a2.pos_period);
                    const int split = (members + 11) / 12;
                    if (a2.pos_period > 0 && a2.pos_period < (1 << 20)) a2.pos_period *= std::max(1, split);
                }
                a2.nwg_virtual = nwg2;
                a2.part_scores = x->part2_s; a2.part_rows = x->part2_r; a2.bounds = x->part2_b;
                a2.shared_thr = x->shared_thr + x->max_nq_pad;
            } else {
                p2 = 0;
            }
        }
        int rc;
        // (dim 1024: the query fragments alone are 256 registers: no pinning, no deeper fragment prefetch and the 32x32x16
        //  shape there; the quad select and the synchronised compaction apply)
#ifdef ICD_ABLATE
        if (x->dim == 1024 && getenv("ICD_1024_FULL")) rc = launch_coarse_flat<1024, CF_PRODUCT_VAR>(x, a, nwg, s);   // A/B: the 768 form
        else
#endif
        if (x->dim == 1024) rc = launch_coarse_flat<1024, (CF_PRODUCT_VAR & (3 | 16 | 2048))>(x, a, nwg, s);
#ifdef ICD_ABLATE_EXPERIMENTS
        else if (const char *wv = getenv("ICD_W8_VAR")) {   // A/B builds: the eight-wave kernel
            const int v = atoi(wv);
            if (v == 0 && !wide_lists) rc = launch_coarse_w8<768>(x, a, nwg, s);
            else if (v == 1) rc = launch_coarse_w8<768, CO_KP, 1>(x, a, nwg, s);
            else if (v == 20 && !wide_lists) rc = launch_coarse_g16<768>(x, a, nwg, s);
            else if (v == 21) rc = launch_coarse_g16<768, CO_KP, 1>(x, a, nwg, s);
            else if (v == 30 && !wide_lists) rc = launch_coarse_g16r<768>(x, a, nwg, s);
            else if (v == 10 && !wide_lists) rc = launch_coarse_w8rg<768>(x, a, nwg, s);
            else if (v == 11) rc = launch_coarse_w8rg<768, CO_KP, 1>(x, a, nwg, s);
            else if (v == 12) rc = launch_coarse_w8rg<768, CO_KP, 2>(x, a, nwg, s);
            else if (v == 14) rc = launch_coarse_w8rg<768, CO_KP, 4>(x, a, nwg, s);
            else return fail(ICD_ERR_INVALID, "ICD_W8_VAR=%d is not built", v);
        }
        else if (const char *rv = getenv("ICD_RG_VAR")) {   // A/B builds: the row-group kernel and its timing variants
            const int v = atoi(rv);
            if (wide_lists) rc = launch_coarse_rg<768, CO_KP_WIDE>(x, a, nwg, s);
            else if (v == 0) rc = launch_coarse_rg<768>(x, a, nwg, s);
            else if (v == 1) rc = launch_coarse_rg<768, CO_KP, 1>(x, a, nwg, s);
            else if (v == 4) rc = launch_coarse_rg<768, CO_KP, 4>(x, a, nwg, s);
            else if (v == 8) rc = launch_coarse_rg<768, CO_KP, 8>(x, a, nwg, s);
            else if (v == 16) rc = launch_coarse_rg<768, CO_KP, 16>(x, a, nwg, s);
            else if (v == 36) rc = launch_coarse_rg<768, CO_KP, 36>(x, a, nwg, s);
            else if (v == 64) rc = launch_coarse_rg<768, CO_KP, 64>(x, a, nwg, s);
            else if (v == 68) rc = launch_coarse_rg<768, CO_KP, 68>(x, a, nwg, s);
            else if (v == 100) rc = launch_coarse_rg<768, CO_KP, 100>(x, a, nwg, s);
            else return fail(ICD_ERR_INVALID, "ICD_RG_VAR=%d is not built", v);
        }
#endif
#ifdef ICD_ABLATE_KSPLIT
        else if (getenv("ICD_KS_VAR") && !wide_lists) {   // A/B builds: the K-split pair kernel (experiments/r04_ksplit_kernel)
            const int v = atoi(getenv("ICD_KS_VAR"));
            if (v == 0) rc = launch_coarse_ksplit<768, CO_KP, 0>(x, a, nwg, s);
            else if (v == 1) rc = launch_coarse_ksplit<768, CO_KP, 1>(x, a, nwg, s);
            else if (v == 2) rc = launch_coarse_ksplit<768, CO_KP, 2>(x, a, nwg, s);
            else if (v == 5) rc = launch_coarse_ksplit<768, CO_KP, 5>(x, a, nwg, s);
            else return fail(ICD_ERR_INVALID, "ICD_KS_VAR=%d is not built", v);
        }
#endif
#ifdef ICD_ABLATE
        else if (const char *fv = getenv("ICD_FLAT_VAR")) {   // A/B builds: stage / select variants of the flat kernel
            const int v = atoi(fv);
            if (false) {}
#define ICD_FV_CASE(V) else if (v == V) rc = launch_coarse_flat<768, V>(x, a, nwg, s);
#ifdef ICD_FV_LIST
            ICD_FV_LIST
#else
            ICD_FV_CASE(0) ICD_FV_CASE(139) ICD_FV_CASE(143) ICD_FV_CASE(155) ICD_FV_CASE(171) ICD_FV_CASE(187) ICD_FV_CASE(2187) ICD_FV_CASE(2203)
            ICD_FV_CASE(2235) ICD_FV_CASE(1163) ICD_FV_CASE(4235) ICD_FV_CASE(4251)
            ICD_FV_CASE(34971) ICD_FV_CASE(35995) ICD_FV_CASE(39067) ICD_FV_CASE(100507) ICD_FV_CASE(104603) ICD_FV_CASE(32923) ICD_FV_CASE(32955) ICD_FV_CASE(34843) ICD_FV_CASE(166043) ICD_FV_CASE(297115) ICD_FV_CASE(559259) ICD_FV_CASE(1083547)
#endif
#undef ICD_FV_CASE
            else return fail(ICD_ERR_INVALID, "ICD_FLAT_VAR=%d is not built", v);
        }
#endif
        else if (wide_lists) rc = launch_coarse_flat<768, CF_PRODUCT_VAR, CO_KP_WIDE>(x, a, nwg, s);
        else if ((size_t)x->n_pad * x->dim * 2 <= COARSE_CACHED_IMAGE_BYTES) rc = launch_coarse_flat<768, CF_CACHED_VAR>(x, a, nwg, s);   // (the image stays in the Infinity Cache)
        else {
            // an image streamed from HBM (a row shard): the classes of work-groups that sweep the same tiles are paced so that a
            // tile is fetched once per XCD instead of once per work-group (coarse_flat_kernel.hpp, VAR 67108864)
            const int t0 = flat_class_period(U, ctiles);
            const int members0 = t0 > 0 ? nwg / t0 : 0;
            const int epochs = x->opt_pace_shift >= 0 ? (U >> x->opt_pace_shift) + 1 : 0;
            if (x->opt_pace_shift >= 0 && members0 >= 4 && (size_t)t0 * epochs <= PACE_WORDS) {
                a.pace = x->pace; a.pace_period = t0; a.pace_epochs = epochs; a.pace_shift = x->opt_pace_shift; a.pace_lead = std::max(1, x->opt_pace_lead);
                HIP_TRY(hipMemsetAsync(x->pace, 0, (size_t)t0 * epochs * sizeof(unsigned int), s));
            }
            rc = launch_coarse_flat<768, CF_PACED_VAR>(x, a, nwg, s);
        }
        if (rc) return rc;
    }
    rec(x, 2, s);
    {
        FinArgs g = f;
        g.part_scores = x->partc_s; g.part_rows = x->partc_r; g.bounds = x->partc_b; g.P = pc; g.KP = wide_lists ? CO_KP_WIDE : CO_KP; g.nq = nq;
        g.perm_mul = x->perm_mul; g.perm_mod = x->perm_mod;
        g.perm_inv = (x->perm_mod > 0 && (double)x->perm_mod * (double)x->perm_mod < 9007199254740992.0) ? 1.0 / (double)x->perm_mod : 0.0;
        // Wide mode (a corpus of tight families, decided from earlier counters): the window sized for k would certify
        // next to nothing and its rescoring be thrown away (0.13 of 1.5 ms per 10 000 queries): the widest window takes
        // every query at once, and the retry below has nothing to add.
        const bool wide_fin = wide_now && k <= 32 && pc * g.KP >= 128;
        if (wide_fin) g.wide_window = 1;
#ifdef ICD_ABLATE
        if (getenv("ICD_FIN_SKIP_WALK")) g.skip_walk = 1;
#endif
        if (wide_fin && x->opt_family_order && nq >= 1024) {
            // every query's window is its family (the corpus is in code order): visit the queries family by family, XCD by XCD
            OrderArgs o{};
            o.part_scores = g.part_scores; o.part_rows = g.part_rows; o.P = g.P; o.KP = g.KP; o.nq = nq;
            o.perm_mul = g.perm_mul; o.perm_mod = g.perm_mod; o.perm_inv = g.perm_inv;
            o.shift = 6;
            while (((long long)x->n >> o.shift) >= ORDER_BUCKETS) ++o.shift;
            o.key = x->order_key; o.hist = x->order_hist; o.order = x->order;
            // (the scatter kernel leaves the histogram zero again; a search that failed between the two launches would not:
            //  cleared here, 4 KB, so that a stale count can never push a position past the order buffer)
            HIP_TRY(hipMemsetAsync(x->order_hist, 0, 2 * ORDER_BUCKETS * sizeof(unsigned int), s));
            hipLaunchKernelGGL(order_keys_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, o);
            hipLaunchKernelGGL(order_scatter_kernel, dim3((nq + 1023) / 1024), dim3(1024), 0, s, o);
            HIP_TRY(hipGetLastError());
            g.qlist = x->order; g.lists_by_query = 1; g.nq = 4 * ((nq + 3) / 4);
        }
        int rc = wide_fin ? launch_finalize_t<true, false, 4>(x, g, s) : launch_finalize<true>(x, g, s);
        if (rc) return rc;
        g.qlist = nullptr; g.lists_by_query = 0; g.nq = nq;
        // Second chance before the exact re-search. A query fails the first pass when more candidates lie within 2 eps of
        // its k-th best than the window sized for k holds (32 or 64 for k <= 32): a family of near-identical rows, the
        // shape ICD sibling codes have. Its lists may hold 128-512 candidates: the same kernel with the widest window
        // (256) certifies it from them in microseconds where the exact sweep costs 2.5-5 us per query
        // (profiles/r02_family_corpus_probe.log: 1 000 queries 2.7 / 5.2 ms -> see there). Only launched when the lists can
        // fill a wider window (mid-size batches; 10 000 queries have 5 lists of 16); gated on the flagged count on the device.
        x->fallback_word = 0;
        if (!wide_fin && k <= 32 && pc * g.KP >= 128) {
            FinArgs g2 = g;
            g2.qlist = x->flagged; g2.nq_ptr = x->nflag; g2.lists_by_query = 1; g2.wide_window = 1;
            g2.nflag = x->nflag + 1; g2.flagged = x->flagged + x->max_nq_pad;
            rc = launch_finalize_t<true, false, 4>(x, g2, s);
            if (rc) return rc;
            x->fallback_word = 1;
        }
        // Second coarse pass, over the queries that are still flagged only. A large batch gives a query 5-8 lists of 16
        // candidates: a family of near-identical rows larger than that (ICD sibling codes repeat their ancestors' names:
        // 124-row families of mutual cosine 0.99 sit inside 2 eps of each other) cannot be certified from them, whatever
        // the window. The same kernel sweeps the corpus again for the flagged queries with about PASS2_CHUNKS lists each
        // (the partition sized on the device from the flagged count, one block per CU looping over the logical
        // work-groups), finalize certifies from those lists with the widest window, and only what still fails goes to the
        // exact re-search. A batch with nothing flagged pays two launches that read one counter and leave.
        if (p2 > 0) {
            const int w_in = x->fallback_word;
            a2.nq_ptr = x->nflag + w_in;
            a2.qlist = x->flagged + (size_t)w_in * x->max_nq_pad;
            a2.skip_below = PASS2_SKIP;
            const int grid2 = std::max(8, std::min(nwg2, x->num_cu) & ~7);
            if (x->dim == 1024) rc = launch_coarse_flat<1024, (CF_PRODUCT_VAR & (3 | 16 | 2048)), CO_KP, true>(x, a2, grid2, s);
            else rc = launch_coarse_flat<768, CF_PRODUCT_VAR, CO_KP, true>(x, a2, grid2, s);
            if (rc) return rc;
            FinArgs g3 = g;
            g3.part_scores = x->part2_s; g3.part_rows = x->part2_r; g3.bounds = x->part2_b; g3.P = p2; g3.KP = CO_KP;
            g3.qlist = a2.qlist; g3.nq_ptr = a2.nq_ptr; g3.lists_by_query = 0; g3.wide_window = 1; g3.skip_below = PASS2_SKIP;
            g3.nflag = x->nflag + 2; g3.flagged = x->flagged + (size_t)2 * x->max_nq_pad;
            rc = launch_finalize_t<true, false, 4>(x, g3, s);
            if (rc) return rc;
            x->fallback_word = 2;
            x->last_p2 = p2; x->last_p2_word = w_in;
        }
        x->last_narrow_large = x->adapt_enabled && large && p2 > 0 && !wide_now && x->chunks_override == 0;
        if (x->last_narrow_large) x->last_narrow_nq = nq;
        x->p2_eval_pending = x->pass2_enabled && pc * kp_c < PASS2_BELOW;   // (a search the second pass applies to, armed or not)
    }
    rec(x, 3, s);
    const int *fl_list = x->flagged + (size_t)x->fallback_word * x->max_nq_pad;
    const int *fl_count = x->nflag + x->fallback_word;
    // fallback over the flagged list, both device-side gated on its length: the streaming kernel for
    // a sparse list (<= ST_MAX_ACTIVE queries), the fp32-MFMA kernel for a dense one. Both leave
    // [slot][p_sparse][KP] lists for the same finalize launch.
    int px = std::min(p_sparse, row_tiles);
    if (px < p_sparse) {   // tiny corpus: MFMA kernel only. EVERY query may be flagged (all-zero queries, duplicate rows):
        // the list count is sized against the workspace like a full exact run
        px = fit_p(px, x->partx_cap, kpx);
        if ((size_t)nq * px * kpx > x->partx_cap) return fail(ICD_ERR_INVALID, "workspace too small for nq=%d k=%d", nq, k);
        const int tiles_per = (row_tiles + px - 1) / px;
        px = (row_tiles + tiles_per - 1) / tiles_per;
        return run_exact(fl_list, fl_count, px, true, false, true, true, f.thr0);
    }
    return run_exact(fl_list, fl_count, p_sparse, true, stream_ok, !sparse_off, true, f.thr0);
}

}  // namespace

extern "C" {

int icd_abi_version(void) { return ICD_ABI_VERSION; }

const char *icd_last_error(void) { return g_err.c_str(); }

int icd_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(ICD_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

int icd_index_create(const float *corpus, int64_t n, int32_t dim, const int32_t *levels, int64_t id_base,
                     int32_t device, int32_t max_nq, int32_t max_k, int32_t flags,
                     icd_index **out) {
    if (!out) return fail(ICD_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (flags & ~(ICD_CREATE_CORPUS_ON_DEVICE | ICD_CREATE_ROW_ORDER | ICD_CREATE_NO_PROBE | ICD_CREATE_NO_CENTER)) return fail(ICD_ERR_INVALID, "flags=0x%x: unknown bits", flags);
    const int32_t corpus_on_device = flags & ICD_CREATE_CORPUS_ON_DEVICE;
    if (!corpus || n <= 0 || n > 0x7FFFFF00ll) return fail(ICD_ERR_INVALID, "corpus NULL or n=%lld out of range", (long long)n);
    if (dim <= 0 || dim % 32 != 0 || dim > 4096) return fail(ICD_ERR_UNSUPPORTED, "dim=%d: must be a multiple of 32, <= 4096", dim);
    if (max_nq <= 0 || max_k <= 0 || max_k > ICD_MAX_K) return fail(ICD_ERR_INVALID, "max_nq=%d max_k=%d (max_k <= %d)", max_nq, max_k, ICD_MAX_K);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(ICD_ERR_INVALID, "device %d of %d", device, ndev);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(ICD_ERR_UNSUPPORTED, "device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);

    icd_index *x = new (std::nothrow) icd_index();
    if (!x) return fail(ICD_ERR_NOMEM, "host allocation failed");
    x->device = device; x->n = n; x->id_base = id_base; x->dim = dim;
    x->opt_permute = !(flags & ICD_CREATE_ROW_ORDER); x->opt_probe = !(flags & ICD_CREATE_NO_PROBE); x->opt_center = !(flags & ICD_CREATE_NO_CENTER);
    x->n_pad = (int)(((n + 127) / 128 + FLAT_SPARE_TILES) * 128);   // (zero tiles behind the fp16 image: plan_flat_tiles)
    x->max_nq = max_nq; x->max_nq_pad = ((max_nq + 127) / 128) * 128; x->max_k = max_k;
    x->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const bool fast_dim = (dim == 768 || dim == 1024);

#define CR_TRY(expr)                                                                                 \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            free_all(x);                                                                             \
            return fail(e_ == hipErrorOutOfMemory ? ICD_ERR_NOMEM : ICD_ERR_HIP, "%s: %s", #expr,    \
                        hipGetErrorString(e_));                                                      \
        }                                                                                            \
    } while (0)

    const size_t nelem = (size_t)n * dim;
    // ST_PAD_ROWS zero rows behind the corpus: the streaming kernel's stages run to the next 256-row boundary
    CR_TRY(dmalloc(&x->corpus, nelem + (size_t)ST_PAD_ROWS * dim));
    CR_TRY(hipMemset(x->corpus + nelem, 0, (size_t)ST_PAD_ROWS * dim * sizeof(float)));
    CR_TRY(hipMemcpy(x->corpus, corpus, nelem * sizeof(float), corpus_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    if (levels) {
        CR_TRY(dmalloc(&x->levels, (size_t)n));
        CR_TRY(hipMemcpy(x->levels, levels, (size_t)n * sizeof(int), corpus_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    }
    CR_TRY(dmalloc(&x->scratch_u32, 4));
    CR_TRY(hipMemset(x->scratch_u32, 0, 4 * sizeof(unsigned)));
    size_t ws = 0;
    auto wsalloc = [&](auto **p, size_t count) -> hipError_t {
        hipError_t e = dmalloc(p, count);
        if (e == hipSuccess) ws += std::max<size_t>(count, 1) * sizeof(**p);
        return e;
    };
    if (fast_dim) {
        CR_TRY(dmalloc(&x->c16, (size_t)x->n_pad * dim));
        // Row order of the fp16 copy: a corpus in code order keeps families of near-identical rows next to each
        // other, so ONE candidate list would collect a query's whole family, end on a bound inside it and fail the
        // certificate. An affine permutation with a golden-ratio stride spreads neighbours evenly over the lists;
        // finalize maps list positions back with the same formula (no table). (ICD_CREATE_ROW_ORDER keeps the order:
        // a process-wide test switch, read at create; results are identical either way.)
        if (n > 2 && x->opt_permute) {
            long long a_ = (long long)(0.6180339887498949 * (double)n) | 1;
            auto gcd = [](long long u, long long v) { while (v) { const long long t = u % v; u = v; v = t; } return u; };
            while (gcd(a_, n) != 1) a_ += 2;
            x->perm_mul = a_ % n; x->perm_mod = (int)n;
        }
        // pass 0: the column mean; the image is centred when the rows share a large common component (coarse_common.hpp)
        if (n >= 64) {
            const int nb = (int)std::min<int64_t>(256, n);
            // the three temporaries of this pass come out of ONE allocation that a scope guard frees on every way out: a failure
            // between the allocations and the end of the pass (the next dmalloc, a launch, the copy) used to leak them, on the
            // 1.25 M-row shards exactly when memory is tightest (CR_TRY only frees the index)
            float *tmp3 = nullptr;
            struct Guard { float *&p; ~Guard() { if (p) hipFree(p); p = nullptr; } } tmp3_guard{tmp3};
            CR_TRY(dmalloc(&tmp3, (size_t)nb * dim + (size_t)nb + 2));
            float *part = tmp3, *sqp = tmp3 + (size_t)nb * dim, *out2 = sqp + nb;
            CR_TRY(dmalloc(&x->cmean, (size_t)dim));
            hipLaunchKernelGGL(column_sum_partial_kernel, dim3(nb), dim3(256), 0, 0, x->corpus, (int)n, dim, part, sqp);
            hipLaunchKernelGGL(column_sum_final_kernel, dim3(1), dim3(256), 0, 0, part, sqp, nb, (int)n, dim, x->cmean, out2);
            CR_TRY(hipGetLastError());
            float h2[2] = {0.f, 0.f};
            CR_TRY(hipMemcpy(h2, out2, sizeof h2, hipMemcpyDeviceToHost));
            x->mean_share = h2[1] > 0.f ? h2[0] / h2[1] : 0.f;
            if (!x->opt_center || !(x->mean_share >= CENTER_MIN_SHARE) || !std::isfinite(h2[0]) || !std::isfinite(h2[1])) { hipFree(x->cmean); x->cmean = nullptr; }
        }
        // pass 1: largest component of the corpus -> ONE power-of-two scale for its fp16 image; pass 2: convert
        ConvertArgs cv{};
        cv.src = x->corpus; cv.dst = x->c16; cv.rows = (int)n; cv.rows_pad = x->n_pad; cv.dim = dim; cv.mu = x->cmean;
        cv.amax_bits = x->scratch_u32 + 2; cv.any_bad = x->scratch_u32 + 1; cv.mode = 2;
        hipLaunchKernelGGL(convert_rows_kernel, dim3((x->n_pad + 3) / 4), dim3(256), 0, 0, cv);
        CR_TRY(hipGetLastError());
        unsigned hv[4] = {0, 0, 0, 0};
        CR_TRY(hipMemcpy(hv, x->scratch_u32, sizeof hv, hipMemcpyDeviceToHost));
        float amax = 0.f;
        memcpy(&amax, &hv[2], 4);
        x->cexp = scale_exp_for(amax);
        cv.mode = 1; cv.fixed_exp = x->cexp; cv.amax_bits = nullptr;
        cv.rmax_bits = x->scratch_u32;
        if (x->cmean) cv.norm_raw_max = reinterpret_cast<float *>(x->scratch_u32 + 3);
        cv.perm_mul = x->perm_mul; cv.perm_mod = x->perm_mod;
        hipLaunchKernelGGL(convert_rows_kernel, dim3((x->n_pad + 3) / 4), dim3(256), 0, 0, cv);
        CR_TRY(hipGetLastError());
        CR_TRY(hipMemcpy(hv, x->scratch_u32, sizeof hv, hipMemcpyDeviceToHost));
        memcpy(&x->rmax_scaled, &hv[0], 4);
        x->rmax = ldexpf(x->rmax_scaled, -x->cexp);
        if (x->cmean) {   // (rmax as reported stays the largest UNcentred norm; the certificate uses both)
            float raw = 0.f;
            memcpy(&raw, &hv[3], 4);
            x->rmax = raw;
            x->rmax_unc_scaled = ldexpf(raw, x->cexp);
        }
        x->fast = (hv[1] == 0) && std::isfinite(x->rmax_scaled);
        if (!x->fast) { hipFree(x->c16); x->c16 = nullptr; }
    }
    if (x->fast) {
        CR_TRY(wsalloc(&x->q16, (size_t)x->max_nq_pad * dim));
        // candidate lists of the coarse pass: about k / 4 lists of 16 up to k = 64, about k / 6 lists of 24 above (+ the
        // lists that work-group boundaries add)
        const int kcap = std::min(max_k, FAST_MAX_K);
        const int lists_for_max_k = std::min(COARSE_MAX_P, std::max(6, (std::min(kcap, 64) + 3) / 4 + 4));
        const int wide_for_max_k = kcap > 64 ? std::min(FIN_MAX_CAND / CO_KP_WIDE, (kcap + 5) / 6 + 4) : 0;
        // (+ the wide partition of large batches on family-shaped corpora, wide_mode: PASS2_MAX_P lists of CO_KP)
        x->partc_cap = std::max<size_t>((size_t)x->max_nq_pad * std::max(std::max(lists_for_max_k, PASS2_MAX_P) * CO_KP, wide_for_max_k * CO_KP_WIDE),
                                        (size_t)1 << 20);
        CR_TRY(wsalloc(&x->partc_s, x->partc_cap));
        CR_TRY(wsalloc(&x->partc_r, x->partc_cap));
        CR_TRY(wsalloc(&x->partc_b, x->partc_cap / CO_KP));
        x->part2_cap = (size_t)x->max_nq_pad * PASS2_MAX_P * CO_KP;
        CR_TRY(wsalloc(&x->part2_s, x->part2_cap));
        CR_TRY(wsalloc(&x->part2_r, x->part2_cap));
        CR_TRY(wsalloc(&x->part2_b, x->part2_cap / CO_KP));
    }
    CR_TRY(wsalloc(&x->qnorm, (size_t)x->max_nq_pad));
    CR_TRY(wsalloc(&x->qexp, (size_t)x->max_nq_pad));
    CR_TRY(wsalloc(&x->qbad, (size_t)x->max_nq_pad));
    CR_TRY(wsalloc(&x->thr0, (size_t)x->max_nq_pad));
    CR_TRY(wsalloc(&x->shared_thr, (size_t)2 * x->max_nq_pad));   // [0]: first coarse pass, [1]: second
    CR_TRY(wsalloc(&x->qdev, (size_t)max_nq * dim));
    x->partx_cap = std::max<size_t>((size_t)x->max_nq_pad * 2 * exact_kp_for(max_k), (size_t)1 << 21);
    if (exact_kp_for(max_k) > 16)   // multi-level reduction of the streaming kernel's lists: ST_MAX_ACTIVE slots x 512 lists
        x->partx_cap = std::max<size_t>(x->partx_cap, (size_t)ST_MAX_ACTIVE * 512 * exact_kp_for(max_k));
    // (+ one query tile of the widest layout: the device-chosen chunk count of the fallback rounds the slot count up)
    CR_TRY(wsalloc(&x->partx_s, x->partx_cap + (size_t)128 * FIN_MAX_CAND_X));
    CR_TRY(wsalloc(&x->partx_r, x->partx_cap + (size_t)128 * FIN_MAX_CAND_X));
    x->lists_cap = std::max((size_t)ST_MAX_ACTIVE * 1024 * exact_kp_for(max_k), (size_t)ST_FALLBACK_MAX_ACTIVE * 1024 * 16);
    CR_TRY(wsalloc(&x->lists_s, x->lists_cap));
    CR_TRY(wsalloc(&x->lists_r, x->lists_cap));
    CR_TRY(wsalloc(&x->nflag, 8));
    CR_TRY(hipMemset(x->nflag, 0, 8 * sizeof(int)));
    CR_TRY(hipHostMalloc(reinterpret_cast<void **>(&x->h_nflag), 8 * sizeof(int), hipHostMallocMapped));
    memset(x->h_nflag, 0, 8 * sizeof(int));
    CR_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&x->h_nflag_dev), x->h_nflag, 0));
    CR_TRY(hipEventCreateWithFlags(&x->ev_nflag, hipEventDisableTiming));
    CR_TRY(hipHostMalloc(reinterpret_cast<void **>(&x->h_pin), PIN_Q_BYTES + PIN_OUT_BYTES + PIN_DONE_BYTES, hipHostMallocMapped));
    CR_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&x->h_pin_dev), x->h_pin, 0));
    memset(x->h_pin + PIN_Q_BYTES + PIN_OUT_BYTES, 0, PIN_DONE_BYTES);   // (the completion word: sequence numbers start at 1)
    CR_TRY(wsalloc(&x->pace, PACE_WORDS));
    CR_TRY(wsalloc(&x->dbg, (size_t)8192 * 16));
    CR_TRY(hipMemset(x->dbg, 0, (size_t)8192 * 16 * 8));
    CR_TRY(wsalloc(&x->flagged, (size_t)3 * x->max_nq_pad));
    CR_TRY(wsalloc(&x->order, (size_t)x->max_nq_pad + 8));
    CR_TRY(wsalloc(&x->order_key, (size_t)x->max_nq_pad));
    CR_TRY(wsalloc(&x->order_hist, (size_t)2 * ORDER_BUCKETS));
    CR_TRY(hipMemset(x->order_hist, 0, 2 * ORDER_BUCKETS * sizeof(unsigned int)));
    const size_t no = (size_t)max_nq * max_k;
    CR_TRY(wsalloc(&x->o_scores, no)); CR_TRY(wsalloc(&x->o_ids, no)); CR_TRY(wsalloc(&x->o_adj, no));
    CR_TRY(wsalloc(&x->o_adj_raw, no)); CR_TRY(wsalloc(&x->o_adj_ids, no)); CR_TRY(wsalloc(&x->o_adj_lv, no));
    for (int r = 0; r < EV_RING; ++r)
        for (int i = 0; i <= NUM_EV; ++i) CR_TRY(hipEventCreate(&x->evring[r][i]));
    CR_TRY(hipDeviceSynchronize());
    x->bytes_ws = ws;
    // ---- corpus-shape probe ---------------------------------------------------------------------------------------------
    // Whether large batches should start with the wide partition (wide_mode) is a property of the CORPUS: families of
    // near-identical rows (ICD sibling codes repeat their ancestors' names) put more rows inside 2 eps of a query's k-th
    // best than the narrow plan's 5-8 lists can hold. The counters of a first user batch would tell (and later ones do,
    // search_device) - but the corpus can be asked now: WIDE_MIN_NQ of its own rows, evenly spaced, are searched as one
    // large batch (each finds itself and its family), and the same rule decides. A fresh index on a family corpus then
    // answers its FIRST large batch in 1.5 ms per 10 000 queries instead of 2.0-2.2; a Gaussian corpus flags nothing
    // and stays narrow. Costs one 2 048-query search at create; state and counters are reset afterwards.
    if (x->fast && max_nq >= WIDE_MIN_NQ && n >= (int64_t)4 * WIDE_MIN_NQ && dim % 4 == 0 && x->opt_probe) {
        const int pq = WIDE_MIN_NQ, pk = std::min(10, max_k);
        hipLaunchKernelGGL(gather_rows_kernel, dim3(pq), dim3(192), 0, 0, x->corpus, x->qdev, (long long)(n / pq), dim, pq);
        CR_TRY(hipGetLastError());
        Outs o{};
        o.adj = x->o_adj; o.adj_raw = x->o_adj_raw; o.adj_ids = x->o_adj_ids; o.adj_lv = x->o_adj_lv;
        const int prc = search_device(x, x->qdev, pq, pk, ICD_MODE_AUTO, o, 0);
        if (prc) { free_all(x); return prc; }
        CR_TRY(hipDeviceSynchronize());
        const int f0 = x->h_nflag[0], f2 = x->h_nflag[2];
        x->probe_flagged = f0; x->probe_left = f2;
        x->wide_mode = x->last_narrow_large && (long long)f0 * 4 > pq && (long long)f2 * 2 < f0;
        x->wide_runs = 0; x->last_narrow_large = false; x->p2_clean = 0; x->p2_eval_pending = false;
        x->sparse_run_seen = 0; x->sparse_disarmed = false;
        x->last_nq = 0; x->last_p2 = 0; x->fallback_word = 0; x->last_chunks = 0; x->last_mode = ICD_MODE_AUTO;
        CR_TRY(hipMemset(x->nflag, 0, 8 * sizeof(int)));
        memset(x->h_nflag, 0, 8 * sizeof(int));
    }
#undef CR_TRY
    *out = x;
    return ICD_OK;
}

int icd_index_destroy(icd_index *idx) {
    if (!valid(idx)) return fail(ICD_ERR_STATE, "invalid handle");
    hipSetDevice(idx->device);
    hipDeviceSynchronize();
    free_all(idx);
    return ICD_OK;
}

constexpr long DONE_POLL_US = 500;   // how long a ONE-query host call polls its completion word before it waits for the stream instead
// results of a small host call: out of the mapped block (the six arrays back to back, 8-byte ones first)
static void copy_pinned_out(icd_index *x, const Outs &user, size_t no_small) {
    const char *h = x->h_pin + PIN_Q_BYTES;
    if (user.adj) memcpy(user.adj, h, no_small * 8);              h += no_small * 8;
    if (user.ids) memcpy(user.ids, h, no_small * 8);              h += no_small * 8;
    if (user.adj_ids) memcpy(user.adj_ids, h, no_small * 8);      h += no_small * 8;
    if (user.scores) memcpy(user.scores, h, no_small * 4);        h += no_small * 4;
    if (user.adj_raw) memcpy(user.adj_raw, h, no_small * 4);      h += no_small * 4;
    if (user.adj_lv) memcpy(user.adj_lv, h, no_small * 4);
}

static int search_common(icd_index *x, const float *queries, int64_t nq, int32_t k, int32_t q_on_device,
                         int32_t mode, Outs user, int32_t out_on_device, void *stream) {
    if (!valid(x)) return fail(ICD_ERR_STATE, "invalid handle");
    std::lock_guard<std::mutex> guard(x->mu);
    if (nq < 0 || nq > x->max_nq) return fail(ICD_ERR_INVALID, "nq=%lld exceeds max_nq=%d", (long long)nq, x->max_nq);
    if (k <= 0 || k > x->max_k) return fail(ICD_ERR_INVALID, "k=%d exceeds max_k=%d", k, x->max_k);
    if (mode != ICD_MODE_AUTO && mode != ICD_MODE_EXACT) return fail(ICD_ERR_INVALID, "mode=%d", mode);
    if (nq == 0) return ICD_OK;
    if (!queries) return fail(ICD_ERR_INVALID, "queries is NULL");
    HIP_TRY(hipSetDevice(x->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    {   // device-in / device-out searches are graph-capturable (no allocation, no synchronisation, no query while capturing)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        x->capturing = s != nullptr && hipStreamIsCapturing(s, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive;
        if (x->capturing && (!q_on_device || !out_on_device)) return fail(ICD_ERR_INVALID, "a search with host buffers synchronises: it cannot be captured into a graph");
    }
    const float *dq = queries;
    if (!q_on_device) {
        const size_t qbytes = (size_t)nq * x->dim * sizeof(float);
        if (nq == 1 && (x->opt_host_one & 1) && qbytes <= sizeof(StreamInlineQuery) && !x->capturing) {
            x->host_q = queries;   // ONE query: search_device puts it into the single-launch kernel's arguments (or copies it after all)
        } else if (qbytes <= PIN_Q_BYTES) {
            memcpy(x->h_pin, queries, qbytes);
            HIP_TRY(hipMemcpyAsync(x->qdev, x->h_pin, qbytes, hipMemcpyHostToDevice, s));
        } else {
            HIP_TRY(hipMemcpyAsync(x->qdev, queries, qbytes, hipMemcpyHostToDevice, s));
        }
        dq = x->qdev;
    }
    Outs dev = user;
    const size_t no_small = (size_t)nq * k;
    const bool pinned_out = !out_on_device && no_small * 36 <= PIN_OUT_BYTES;
    if (pinned_out) {   // the six arrays back to back in the mapped block, 8-byte ones first
        char *d = x->h_pin_dev + PIN_Q_BYTES;
        dev.adj = user.adj ? reinterpret_cast<double *>(d) : nullptr;                          d += no_small * 8;
        dev.ids = user.ids ? reinterpret_cast<long long *>(d) : nullptr;                       d += no_small * 8;
        dev.adj_ids = user.adj_ids ? reinterpret_cast<long long *>(d) : nullptr;               d += no_small * 8;
        dev.scores = user.scores ? reinterpret_cast<float *>(d) : nullptr;                     d += no_small * 4;
        dev.adj_raw = user.adj_raw ? reinterpret_cast<float *>(d) : nullptr;                   d += no_small * 4;
        dev.adj_lv = user.adj_lv ? reinterpret_cast<int *>(d) : nullptr;
    } else if (!out_on_device) {
        dev.scores = user.scores ? x->o_scores : nullptr;
        dev.ids = user.ids ? x->o_ids : nullptr;
        dev.adj = user.adj ? x->o_adj : nullptr;
        dev.adj_raw = user.adj_raw ? x->o_adj_raw : nullptr;
        dev.adj_ids = user.adj_ids ? x->o_adj_ids : nullptr;
        dev.adj_lv = user.adj_lv ? x->o_adj_lv : nullptr;
    }
    x->done_armed = false;
    x->host_one_call = !q_on_device && nq == 1 && pinned_out && !x->capturing;
    int rc = search_device(x, dq, (int)nq, k, mode, dev, s);
    x->host_q = nullptr;
    x->host_one_call = false;
    if (rc) return rc;
    rec(x, NUM_EV, s);
    if (x->done_armed && pinned_out) {
        // ONE query through the single-launch kernel: its last work-group stores the call's sequence number behind the outputs
        // (system-scope release). Poll it for a bounded time - no event record, no wait for the queue's completion signal -
        // and fall back to the stream synchronisation (always correct) when the word does not arrive (a busy GPU, a failed launch).
        const volatile unsigned long long *done = reinterpret_cast<const volatile unsigned long long *>(x->h_pin + PIN_Q_BYTES + PIN_OUT_BYTES);
        const auto t0 = std::chrono::steady_clock::now();
        bool seen = false;
        for (unsigned spin = 0;; ++spin) {
            if (__atomic_load_n(done, __ATOMIC_ACQUIRE) == x->done_seq) { seen = true; break; }
            if ((spin & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(DONE_POLL_US)) break;
        }
        if (!seen) HIP_TRY(hipStreamSynchronize(s));
        copy_pinned_out(x, user, no_small);
        return ICD_OK;
    }
    // the search's last kernel has written the fallback counters to pinned host memory: icd_index_stats reads them after
    // waiting for THIS event only (no device-wide synchronisation: other streams - an encoder - keep running)
    if (!x->capturing) HIP_TRY(hipEventRecord(x->ev_nflag, s));   // (an event recorded inside a capture could not be waited for by icd_index_stats)
    if (pinned_out) {
        HIP_TRY(hipStreamSynchronize(s));   // (the kernels' stores to the mapped block are visible behind it, like the counters')
        copy_pinned_out(x, user, no_small);
        return ICD_OK;
    }
    if (!out_on_device) {
        const size_t no = (size_t)nq * k;
        if (user.scores) HIP_TRY(hipMemcpyAsync(user.scores, dev.scores, no * sizeof(float), hipMemcpyDeviceToHost, s));
        if (user.ids) HIP_TRY(hipMemcpyAsync(user.ids, dev.ids, no * sizeof(long long), hipMemcpyDeviceToHost, s));
        if (user.adj) HIP_TRY(hipMemcpyAsync(user.adj, dev.adj, no * sizeof(double), hipMemcpyDeviceToHost, s));
        if (user.adj_raw) HIP_TRY(hipMemcpyAsync(user.adj_raw, dev.adj_raw, no * sizeof(float), hipMemcpyDeviceToHost, s));
        if (user.adj_ids) HIP_TRY(hipMemcpyAsync(user.adj_ids, dev.adj_ids, no * sizeof(long long), hipMemcpyDeviceToHost, s));
        if (user.adj_lv) HIP_TRY(hipMemcpyAsync(user.adj_lv, dev.adj_lv, no * sizeof(int), hipMemcpyDeviceToHost, s));
    }
    if (!out_on_device || !q_on_device) HIP_TRY(hipStreamSynchronize(s));
    return ICD_OK;
}

int icd_index_search(icd_index *idx, const float *queries, int64_t nq, int32_t k, int32_t queries_on_device,
                     int32_t mode, float *out_scores, int64_t *out_ids, int32_t out_on_device, void *stream) {
    if (!out_scores || !out_ids) return fail(ICD_ERR_INVALID, "output pointer is NULL");
    Outs o{};
    o.scores = out_scores;
    o.ids = reinterpret_cast<long long *>(out_ids);
    return search_common(idx, queries, nq, k, queries_on_device, mode, o, out_on_device, stream);
}

int icd_index_search_reweighted(icd_index *idx, const float *queries, int64_t nq, int32_t k,
                                int32_t queries_on_device, int32_t mode, double *out_adj, float *out_raw,
                                int64_t *out_ids, int32_t *out_levels, int32_t out_on_device, void *stream) {
    if (!out_adj || !out_ids) return fail(ICD_ERR_INVALID, "output pointer is NULL");
    Outs o{};
    o.adj = out_adj;
    o.adj_raw = out_raw;
    o.adj_ids = reinterpret_cast<long long *>(out_ids);
    o.adj_lv = out_levels;
    return search_common(idx, queries, nq, k, queries_on_device, mode, o, out_on_device, stream);
}

int icd_merge_topk(int32_t device, const float *scores, const int64_t *ids, const int32_t *levels, int32_t G,
                   int64_t nq, int32_t k, double *out_adj, float *out_raw, int64_t *out_ids,
                   int32_t *out_levels, void *stream) {
    if (!scores || !ids || !levels) return fail(ICD_ERR_INVALID, "input pointer is NULL");
    if (G <= 0 || k <= 0 || k > ICD_MAX_K || (int64_t)G * k > 1024) return fail(ICD_ERR_INVALID, "G=%d k=%d: need G*k <= 1024", G, k);
    if (nq < 0 || nq > 0x7FFFFFFF) return fail(ICD_ERR_INVALID, "nq=%lld", (long long)nq);
    if (nq == 0) return ICD_OK;
    HIP_TRY(hipSetDevice(device));
    MergeArgs a{};
    a.scores = scores; a.ids = reinterpret_cast<const long long *>(ids); a.levels = levels;
    a.G = G; a.nq = (int)nq; a.k = k;
    a.out_adj = out_adj; a.out_raw = out_raw; a.out_ids = reinterpret_cast<long long *>(out_ids); a.out_levels = out_levels;
    const size_t lds = 4 * (1024 * 16 + 128 * 24);
    static int configured[MAX_DEVICES] = {};   // (guarded by the caller's one-stream-per-handle contract; worst case a repeated call)
    HIP_TRY(ensure_dynamic_lds(merge_topk_kernel, device, (size_t)((int)lds), configured));
    hipLaunchKernelGGL(merge_topk_kernel, dim3(((int)nq + 3) / 4), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), a);
    HIP_TRY(hipGetLastError());
    return ICD_OK;
}

int icd_hier_rescore(int32_t device, const double *adj, const int64_t *ids, int64_t nq, int32_t k, int64_t id_base,
                     int64_t n_rows, const uint8_t *row_tags, const double *q_params, const double *weights,
                     int32_t *out_order, double *out_enhanced, double *out_score, double *out_vs, double *out_hb,
                     double *out_boost, void *stream) {
    if (!adj || !ids || !row_tags || !q_params || !weights) return fail(ICD_ERR_INVALID, "input pointer is NULL");
    if (!out_order || !out_enhanced || !out_score || !out_vs || !out_hb || !out_boost) return fail(ICD_ERR_INVALID, "output pointer is NULL");
    if (k <= 0 || k > HIER_MAX_K) return fail(ICD_ERR_INVALID, "k=%d (1..%d)", k, HIER_MAX_K);
    if (nq < 0 || nq > 0x7FFFFFFF || n_rows < 0) return fail(ICD_ERR_INVALID, "nq=%lld n_rows=%lld", (long long)nq, (long long)n_rows);
    if (nq == 0) return ICD_OK;
    HIP_TRY(hipSetDevice(device));
    HierArgs a{};
    a.adj = adj; a.ids = reinterpret_cast<const long long *>(ids); a.nq = (int)nq; a.k = k; a.id_base = id_base; a.n_rows = n_rows;
    a.row_tags = row_tags; a.q_params = q_params;
    a.w_hb = weights[0]; a.w_em = weights[1]; a.w_sc = weights[2]; a.w_ca = weights[3]; a.w_cr = weights[4];
    a.sc_value = weights[5]; a.level_term = weights[6];
    a.out_order = out_order; a.out_enhanced = out_enhanced; a.out_score = out_score; a.out_vs = out_vs; a.out_hb = out_hb;
    a.out_boost = out_boost;
    hipLaunchKernelGGL(hier_rescore_kernel, dim3(((int)nq + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    HIP_TRY(hipGetLastError());
    return ICD_OK;
}

int icd_pack_winners(int32_t device, const int32_t *order, const int64_t *ids, const float *raw, const double *adj, const double *enhanced,
                     const double *vs, const double *hb, const double *boost, int64_t nq, int32_t k, int32_t kk, double *out, void *stream) {
    if (!order || !ids || !raw || !adj || !enhanced || !vs || !hb || !boost || !out) return fail(ICD_ERR_INVALID, "pointer is NULL");
    if (k <= 0 || kk <= 0 || kk > k || nq < 0 || nq > 0x7FFFFFFF) return fail(ICD_ERR_INVALID, "nq=%lld k=%d kk=%d", (long long)nq, k, kk);
    if (nq == 0) return ICD_OK;
    HIP_TRY(hipSetDevice(device));
    PackWinnersArgs a{};
    a.order = order; a.ids = reinterpret_cast<const long long *>(ids); a.raw = raw; a.adj = adj; a.enh = enhanced; a.vs = vs; a.hb = hb; a.boost = boost;
    a.nq = (int)nq; a.k = k; a.kk = kk; a.out = out;
    const long long per = (long long)nq * kk;
    hipLaunchKernelGGL(pack_winners_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    HIP_TRY(hipGetLastError());
    return ICD_OK;
}

int icd_score_stats(int32_t device, const double *scores, const int32_t *order, int64_t nq, int32_t k, int32_t use,
                    double *out, void *stream) {
    if (!scores || !out) return fail(ICD_ERR_INVALID, "pointer is NULL");
    if (k <= 0 || k > STATS_MAX_K || use <= 0) return fail(ICD_ERR_INVALID, "k=%d (1..%d) use=%d", k, STATS_MAX_K, use);
    if (nq < 0 || nq > 0x7FFFFFFF) return fail(ICD_ERR_INVALID, "nq=%lld", (long long)nq);
    if (nq == 0) return ICD_OK;
    HIP_TRY(hipSetDevice(device));
    StatsArgs a{};
    a.scores = scores; a.order = order; a.nq = (int)nq; a.k = k; a.use = use; a.out = out;
    hipLaunchKernelGGL(score_stats_kernel, dim3(((int)nq + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), a);
    HIP_TRY(hipGetLastError());
    return ICD_OK;
}

int icd_cosine_rows(int32_t device, const float *x, const float *y, int64_t y_stride, int64_t nq, int32_t dim,
                    double *out, void *stream) {
    if (!x || !y || !out) return fail(ICD_ERR_INVALID, "pointer is NULL");
    if (dim <= 0 || (y_stride != 0 && y_stride != dim)) return fail(ICD_ERR_INVALID, "dim=%d y_stride=%lld (0 or dim)", dim, (long long)y_stride);
    if (nq < 0 || nq > 0x7FFFFFFF) return fail(ICD_ERR_INVALID, "nq=%lld", (long long)nq);
    if (nq == 0) return ICD_OK;
    HIP_TRY(hipSetDevice(device));
    CosArgs a{};
    a.x = x; a.y = y; a.y_stride = y_stride; a.nq = (int)nq; a.dim = dim; a.out = out;
    hipLaunchKernelGGL(cosine_rows_kernel, dim3(((int)nq + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    HIP_TRY(hipGetLastError());
    return ICD_OK;
}

int icd_packed_attention(int32_t device, const float *qkv, int64_t ld, const int32_t *starts, int32_t nseq, int32_t heads,
                         int32_t head_dim, int32_t max_len, float *out, int64_t out_ld, void *stream) {
    if (!qkv || !starts || !out) return fail(ICD_ERR_INVALID, "pointer is NULL");
    if (head_dim != ATT_HEAD_DIM) return fail(ICD_ERR_UNSUPPORTED, "head_dim=%d (this kernel is written for %d)", head_dim, ATT_HEAD_DIM);
    if (max_len < 1 || max_len > ATT_MAX_SEQ) return fail(ICD_ERR_UNSUPPORTED, "max_len=%d (1..%d tokens per sequence)", max_len, ATT_MAX_SEQ);
    if (nseq < 0 || heads <= 0 || (int64_t)nseq * heads > 0x7FFFFFF0LL) return fail(ICD_ERR_INVALID, "nseq=%d heads=%d", nseq, heads);
    const int64_t hidden = (int64_t)heads * head_dim;
    if (ld < 3 * hidden || out_ld < hidden || ld % 4 != 0) return fail(ICD_ERR_INVALID, "ld=%lld out_ld=%lld for hidden=%lld", (long long)ld, (long long)out_ld, (long long)hidden);
    if (nseq == 0) return ICD_OK;
    HIP_TRY(hipSetDevice(device));
    PackedAttnArgs a{};
    a.qkv = qkv; a.out = out; a.starts = starts; a.nseq = nseq; a.heads = heads; a.ld = ld; a.out_ld = out_ld; a.hidden = (int)hidden;
    a.scale = 0.125f;   // 1 / sqrt(64), exact
    const int tasks = nseq * heads;
    hipLaunchKernelGGL(packed_attention_kernel, dim3((tasks + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    HIP_TRY(hipGetLastError());
    return ICD_OK;
}

int icd_split_bf16x3(int32_t device, const float *x, int64_t rows, int32_t cols, int64_t ld, int32_t act, void *out, void *stream) {
    if (!x || !out) return fail(ICD_ERR_INVALID, "pointer is NULL");
    if (rows < 0 || cols < SPLIT_TAIL || cols % 8 != 0 || ld < cols || ld % 4 != 0) return fail(ICD_ERR_INVALID, "rows=%lld cols=%d ld=%lld (cols a multiple of 8, ld >= cols and a multiple of 4)", (long long)rows, cols, (long long)ld);
    if (act != 0 && act != 1) return fail(ICD_ERR_INVALID, "act=%d (0 none, 1 erf-GELU)", act);
    if ((reinterpret_cast<uintptr_t>(x) & 15) != 0 || (reinterpret_cast<uintptr_t>(out) & 15) != 0) return fail(ICD_ERR_INVALID, "x and out must be 16-byte aligned");
    if (rows == 0) return ICD_OK;
    HIP_TRY(hipSetDevice(device));
    SplitArgs a{};
    a.x = x; a.out = static_cast<unsigned short *>(out); a.rows = rows; a.cols = cols; a.act = act; a.ld = ld;
    const long long total = rows * (long long)(cols / 8);
    const int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    HIP_TRY(hipGetLastError());
    return ICD_OK;
}

int icd_index_lookup_levels(icd_index *idx, const int64_t *ids, int64_t count, int32_t *out_levels, void *stream) {
    if (!valid(idx)) return fail(ICD_ERR_STATE, "invalid handle");
    std::lock_guard<std::mutex> guard(idx->mu);
    if (!ids || !out_levels || count < 0) return fail(ICD_ERR_INVALID, "bad arguments");
    if (count == 0) return ICD_OK;
    HIP_TRY(hipSetDevice(idx->device));
    hipLaunchKernelGGL(lookup_levels_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const long long *>(ids),
                       (long long)count, idx->levels, (long long)idx->id_base, (long long)idx->n, out_levels);
    HIP_TRY(hipGetLastError());
    return ICD_OK;
}

int icd_index_stats(icd_index *idx, icd_stats *out) {
    if (!valid(idx)) return fail(ICD_ERR_STATE, "invalid handle");
    std::lock_guard<std::mutex> guard(idx->mu);
    if (!out) return fail(ICD_ERR_INVALID, "out is NULL");
    memset(out, 0, sizeof *out);
    out->n = idx->n; out->dim = idx->dim; out->device = idx->device; out->id_base = idx->id_base;
    out->bytes_corpus_f32 = (int64_t)idx->n * idx->dim * 4;
    out->bytes_corpus_f16 = idx->c16 ? (int64_t)idx->n_pad * idx->dim * 2 : 0;
    out->bytes_workspace = (int64_t)idx->bytes_ws;
    out->max_nq = idx->max_nq; out->max_k = idx->max_k; out->fast_path = idx->fast ? 1 : 0;
    out->rmax = idx->rmax;
    out->last_nq = idx->last_nq;
    // the last search's fallback count: every search copies its counters to pinned host memory behind itself; this call
    // waits for that copy's event - the last search of THIS handle - and for nothing else on the device
    int nf = 0;
    if (idx->h_nflag && idx->last_nq > 0) {
        HIP_TRY(hipSetDevice(idx->device));
        HIP_TRY(hipEventSynchronize(idx->ev_nflag));
        nf = idx->h_nflag[idx->fallback_word];
    }
    out->last_fallback = nf;
    out->last_second_pass = (idx->last_p2 > 0 && idx->h_nflag && idx->last_nq > 0) ? idx->h_nflag[idx->last_p2_word] : 0;
    out->last_second_pass_lists = idx->last_p2;
    out->wide_mode = idx->wide_mode ? 1 : 0;
    out->second_pass_armed = (idx->pass2_enabled && idx->p2_clean < PASS2_DISARM_AFTER) ? 1 : 0;
    out->sparse_fallback_armed = idx->sparse_disarmed ? 0 : 1;
    out->centered = idx->cmean ? 1 : 0; out->mean_share = idx->mean_share;
    out->last_chunks = idx->last_chunks; out->last_mode = idx->last_mode;
    return ICD_OK;
}

int icd_unpack_query_slices(int32_t device, const void *gathered, int32_t world, int64_t nq, int32_t k, double *out_adj,
                                  float *out_raw, int64_t *out_ids, int32_t *out_levels, void *stream) {
    if (!gathered || !out_adj || !out_raw || !out_ids || !out_levels) return fail(ICD_ERR_INVALID, "pointer is NULL");
    if (world < 1 || nq < 0 || k <= 0) return fail(ICD_ERR_INVALID, "world=%d nq=%lld k=%d", world, (long long)nq, k);
    HIP_TRY(hipSetDevice(device));
    const size_t width = ((size_t)nq + world - 1) / world, per = width * k;
    const char *rb = static_cast<const char *>(gathered);
    if (icd_internal_unpack_query_slices(rb, rb + per * world * 8, rb + per * world * 16, rb + per * world * 20, world, nq, k,
                                         (long long)width, out_adj, out_raw, out_ids, out_levels, stream))
        return fail(ICD_ERR_HIP, "the unpack launch failed");
    return ICD_OK;
}

int icd_index_set_option(icd_index *idx, int32_t option, int32_t value) {
    if (!valid(idx)) return fail(ICD_ERR_STATE, "invalid handle");
    std::lock_guard<std::mutex> guard(idx->mu);
    icd_index *x = idx;
    switch (option) {
    case ICD_OPT_FAMILY_ORDER: x->opt_family_order = (value & 1) != 0; break;
    case ICD_OPT_STREAM_ONE:   x->opt_stream_one = value != 0; break;
    case ICD_OPT_HOST_ONE:     x->opt_host_one = value & 3; break;
    case ICD_OPT_PACING_SHIFT: x->opt_pace_shift = value > 12 ? 12 : value; break;
    case ICD_OPT_PACING_LEAD:  x->opt_pace_lead = value < 1 ? 1 : value; break;
    case ICD_OPT_EXACT_NARROW: x->opt_exact_narrow = value != 0; break;
    case ICD_OPT_WIDE_FROM:    x->opt_wide_from = value < 0 ? 0 : value; break;
    default: return fail(ICD_ERR_INVALID, "option %d: not one of ICD_OPT_*", option);
    }
    return ICD_OK;
}

int icd_index_set_second_pass(icd_index *idx, int32_t enabled) {
    if (!valid(idx)) return fail(ICD_ERR_STATE, "invalid handle");
    std::lock_guard<std::mutex> guard(idx->mu);
    idx->pass2_enabled = enabled != 0;
    idx->adapt_enabled = enabled == 1;
    if (!idx->pass2_enabled || !idx->adapt_enabled) { idx->wide_mode = false; idx->last_narrow_large = false; }
    idx->p2_clean = 0; idx->p2_eval_pending = false;   // (re-armed)
    idx->sparse_need = SPARSE_DISARM_AFTER; idx->sparse_run_seen = 0; idx->sparse_disarmed = false;
    if (idx->nflag) { HIP_TRY(hipSetDevice(idx->device)); HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipMemset(idx->nflag + 4, 0, sizeof(int))); idx->h_nflag[4] = 0; }
    return ICD_OK;
}

int icd_index_set_chunks(icd_index *idx, int32_t chunks) {
    if (!valid(idx)) return fail(ICD_ERR_STATE, "invalid handle");
    std::lock_guard<std::mutex> guard(idx->mu);
    if (chunks < 0 || chunks > COARSE_MAX_P) return fail(ICD_ERR_INVALID, "chunks=%d (0..%d)", chunks, COARSE_MAX_P);
    idx->chunks_override = chunks;
    return ICD_OK;
}

int icd_index_debug_counters(icd_index *idx, unsigned long long *out, int32_t count) {
    if (!valid(idx)) return fail(ICD_ERR_STATE, "invalid handle");
    std::lock_guard<std::mutex> guard(idx->mu);
    if (!out || count <= 0 || count > 8192 * 16) return fail(ICD_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(idx->device));
    HIP_TRY(hipMemcpy(out, idx->dbg, (size_t)count * 8, hipMemcpyDeviceToHost));
    return ICD_OK;
}

int icd_index_set_profiling(icd_index *idx, int32_t enabled) {
    if (!valid(idx)) return fail(ICD_ERR_STATE, "invalid handle");
    std::lock_guard<std::mutex> guard(idx->mu);
    if (enabled < 0) return fail(ICD_ERR_INVALID, "enabled=%d", enabled);
    idx->profiling = enabled != 0;
    idx->prof_every = enabled > 1 ? enabled : 1;
    idx->prof_tick = 0;
    return ICD_OK;
}

int icd_index_profile_summary(icd_index *idx, icd_profile *out_mean, int32_t *out_count) {
    if (!valid(idx)) return fail(ICD_ERR_STATE, "invalid handle");
    std::lock_guard<std::mutex> guard(idx->mu);
    if (!out_mean || !out_count) return fail(ICD_ERR_INVALID, "out is NULL");
    memset(out_mean, 0, sizeof *out_mean);
    HIP_TRY(hipSetDevice(idx->device));
    const long total = idx->prof_count;
    const int n = (int)std::min<long>(total, EV_RING);
    int used = 0;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int j = 0; j < n; ++j) {
        const int slot = (int)((total - 1 - j) % EV_RING);
        hipEvent_t *ev = idx->evring[slot];
        const bool *ok = idx->evring_valid[slot];
        if (!ok[0] || !ok[NUM_EV]) continue;
        HIP_TRY(hipEventSynchronize(ev[NUM_EV]));
        auto span = [&](int a, int b) -> double {
            float ms = 0.f;
            if (ok[a] && ok[b]) hipEventElapsedTime(&ms, ev[a], ev[b]);
            return ms;
        };
        acc[0] += span(0, 1); acc[1] += span(1, 2); acc[2] += span(2, 3); acc[3] += span(3, 4);
        acc[4] += span(4, 5); acc[5] += span(0, NUM_EV);
        ++used;
    }
    if (used) {
        out_mean->ms_prep = (float)(acc[0] / used); out_mean->ms_coarse = (float)(acc[1] / used);
        out_mean->ms_finalize = (float)(acc[2] / used); out_mean->ms_exact = (float)(acc[3] / used);
        out_mean->ms_exact_finalize = (float)(acc[4] / used); out_mean->ms_total = (float)(acc[5] / used);
    }
    *out_count = used;
    idx->prof_count = 0;
    for (int r = 0; r < EV_RING; ++r)
        for (int i = 0; i <= NUM_EV; ++i) idx->evring_valid[r][i] = false;
    return ICD_OK;
}

int icd_index_last_profile(icd_index *idx, icd_profile *out) {
    if (!valid(idx)) return fail(ICD_ERR_STATE, "invalid handle");
    std::lock_guard<std::mutex> guard(idx->mu);
    if (!out) return fail(ICD_ERR_INVALID, "out is NULL");
    memset(out, 0, sizeof *out);
    if (!idx->ev_valid[0] || !idx->ev_valid[NUM_EV]) return fail(ICD_ERR_STATE, "no profiled search recorded");
    HIP_TRY(hipSetDevice(idx->device));
    HIP_TRY(hipEventSynchronize(idx->ev[NUM_EV]));
    auto span = [&](int a, int b) -> float {
        float ms = 0.f;
        if (idx->ev_valid[a] && idx->ev_valid[b]) hipEventElapsedTime(&ms, idx->ev[a], idx->ev[b]);
        return ms;
    };
    out->ms_prep = span(0, 1);
    out->ms_coarse = span(1, 2);
    out->ms_finalize = span(2, 3);
    out->ms_exact = span(3, 4);
    out->ms_exact_finalize = span(4, 5);
    out->ms_total = span(0, NUM_EV);
    return ICD_OK;
}

}  // extern "C"

// the small-input sentence encoder (icd_encoder_*): its own file, this translation unit (fail(), HIP_TRY, packed_attention_kernel)
#include "icd_encoder.hpp"
