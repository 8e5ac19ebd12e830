// icd_group.cpp — multi-GPU search behind the C ABI (include/icd_search.h, icd_group_*): one process per GPU, one
// icd_index per process (a row shard of the corpus, or a replica), an RCCL communicator owned by the group object.
//
// The reference is single-process (SURVEY.md section 5 "Distributed communication backend: none"); this is the surface
// SURVEY.md section 8(b)/(e) asks for: a host that is not PyTorch can run the row-sharded and the query-sharded search
// with nothing but this library and librccl.
//
//   ROW_SHARD    every rank searches ALL queries against its shard (icd_index_search: raw top-k, global ids), looks the
//                levels of its hits up, ONE grouped ncclAllGather moves (score f32 | id i64 | level i32) x k per query,
//                and every rank merges the G lists per query + applies the level reweight (icd_merge_topk).
//   QUERY_SHARD  the corpus is replicated; rank r searches its contiguous slice of the batch (icd_index_search_reweighted:
//                no collective on the data path) and, if asked to, one grouped ncclAllGather hands every rank the whole
//                result.
// Search, collective, merge and reweight are enqueued on ONE stream, nothing synchronises.
//
// librccl is opened with dlopen when the first group is created: the search library itself has no link-time dependency
// on it (a single-GPU deployment needs no RCCL), and inside a PyTorch process the already loaded librccl.so.1 is reused.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/icd_search.h"

extern "C" __attribute__((visibility("hidden"))) int icd_internal_fail(int code, const char *fmt, ...);
// (icd_search.hip) one launch: the gathered padded slices of a query-sharded search -> the contiguous [nq][k] outputs
extern "C" __attribute__((visibility("hidden"))) int icd_internal_unpack_query_slices(const void *r_adj, const void *r_ids, const void *r_raw,
        const void *r_lv, int world, long long nq, int k, long long width, void *out_adj, void *out_raw, void *out_ids, void *out_lv, void *stream);

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;

int load_rccl() {
    if (g_rccl.handle) return ICD_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) return icd_internal_fail(ICD_ERR_UNSUPPORTED, "librccl.so.1 not found (%s): the multi-GPU entry points need RCCL", dlerror());
    Rccl r;
    r.handle = h;
#define ICD_SYM(field, name)                                                                   \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));                             \
    if (!r.field) return icd_internal_fail(ICD_ERR_UNSUPPORTED, "librccl: symbol %s missing", name)
    ICD_SYM(GetUniqueId, "ncclGetUniqueId");
    ICD_SYM(CommInitRank, "ncclCommInitRank");
    ICD_SYM(CommDestroy, "ncclCommDestroy");
    ICD_SYM(AllGather, "ncclAllGather");
    ICD_SYM(GroupStart, "ncclGroupStart");
    ICD_SYM(GroupEnd, "ncclGroupEnd");
    ICD_SYM(GetErrorString, "ncclGetErrorString");
#undef ICD_SYM
    g_rccl = r;
    return ICD_OK;
}

#define NCCL_TRY(expr)                                                                                        \
    do {                                                                                                      \
        ncclResult_t r_ = (expr);                                                                             \
        if (r_ != ncclSuccess)                                                                                \
            return icd_internal_fail(ICD_ERR_HIP, "%s: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)
#define HIPG_TRY(expr)                                                                                        \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess)                                                                                 \
            return icd_internal_fail(e_ == hipErrorOutOfMemory ? ICD_ERR_NOMEM : ICD_ERR_HIP, "%s: %s (%s:%d)", #expr, \
                                     hipGetErrorString(e_), __FILE__, __LINE__);                              \
    } while (0)

// inside icd_group_search: an error between ncclGroupStart and ncclGroupEnd must not leave the RCCL group open (every later
// collective of the process would be queued into it), and the message names the rank: on N ranks the first question is which
#define NCCL_TRY_G(g, open, expr)                                                                             \
    do {                                                                                                      \
        ncclResult_t r_ = (expr);                                                                             \
        if (r_ != ncclSuccess) {                                                                              \
            if (open) g_rccl.GroupEnd();                                                                      \
            return icd_internal_fail(ICD_ERR_HIP, "rank %d of %d: %s: %s (%s:%d)", (g)->rank, (g)->world, #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
        }                                                                                                     \
    } while (0)

// contiguous split of n items over `world` ranks: the first n % world ranks get one more (rag_project_icd10_amd/sharded.py
// shard_bounds - the two must agree)
void shard_bounds(int64_t n, int world, int rank, int64_t *lo, int64_t *hi) {
    const int64_t base = n / world, rem = n % world;
    *lo = rank * base + std::min<int64_t>(rank, rem);
    *hi = *lo + base + (rank < rem ? 1 : 0);
}

}  // namespace

struct icd_group {
    uint32_t magic = 0x1CD16A0Bu;
    icd_index *idx = nullptr;   // borrowed
    int device = 0, rank = 0, world = 1, mode = 0;
    int max_nq = 0, max_k = 0, dim = 0, idx_max_nq = 0;
    bool use_comm = false;      // world > 1, or a one-rank group created WITH an id (the collective path end to end: tests)
    ncclComm_t comm = nullptr;
    // ROW_SHARD: this rank's hits and the gathered ones ([world][max_nq][max_k])
    float *send_s = nullptr; long long *send_i = nullptr; int *send_l = nullptr;
    float *recv_s = nullptr; long long *recv_i = nullptr; int *recv_l = nullptr;
    // QUERY_SHARD with gather: this rank's slice padded to `width` rows and the gathered slices ([world][width][max_k])
    double *qs_adj = nullptr; double *qr_adj = nullptr;
};

namespace {

bool valid(icd_group *g) { return g && g->magic == 0x1CD16A0Bu; }

void free_group(icd_group *g) {
    if (!g) return;
    hipSetDevice(g->device);
    if (g->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(g->comm);
    hipFree(g->send_s); hipFree(g->send_i); hipFree(g->send_l);
    hipFree(g->recv_s); hipFree(g->recv_i); hipFree(g->recv_l);
    hipFree(g->qs_adj); hipFree(g->qr_adj);
    g->magic = 0;
    delete g;
}

}  // namespace

extern "C" {

int icd_group_unique_id(uint8_t *out_id) {
    if (!out_id) return icd_internal_fail(ICD_ERR_INVALID, "out_id is NULL");
    int rc = load_rccl();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == ICD_GROUP_ID_BYTES, "ICD_GROUP_ID_BYTES must equal sizeof(ncclUniqueId)");
    ncclUniqueId id;
    NCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(out_id, &id, sizeof id);
    return ICD_OK;
}

// Everything of a group that can fail on ONE rank alone - argument checks, the buffers, opening librccl - without the
// communicator: a host lets its ranks agree on the outcome (any side channel) BEFORE icd_group_connect, which is collective.
int icd_group_prepare(icd_index *local, int32_t with_comm, int32_t rank, int32_t world, int32_t mode, int32_t max_nq,
                      int32_t max_k, icd_group **out) {
    if (!out) return icd_internal_fail(ICD_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!local) return icd_internal_fail(ICD_ERR_INVALID, "local index is NULL");
    if (world < 1 || rank < 0 || rank >= world) return icd_internal_fail(ICD_ERR_INVALID, "rank %d of %d", rank, world);
    if (mode != ICD_GROUP_ROW_SHARD && mode != ICD_GROUP_QUERY_SHARD) return icd_internal_fail(ICD_ERR_INVALID, "mode=%d", mode);
    icd_stats st;
    int rc = icd_index_stats(local, &st);
    if (rc) return rc;
    if (max_nq <= 0 || max_k <= 0 || max_k > st.max_k) return icd_internal_fail(ICD_ERR_INVALID, "max_nq=%d max_k=%d (index max_k %d)", max_nq, max_k, st.max_k);
    if (mode == ICD_GROUP_ROW_SHARD && (max_nq > st.max_nq || (int64_t)world * max_k > 1024))
        return icd_internal_fail(ICD_ERR_INVALID, "row-sharded: max_nq=%d exceeds the index's %d, or world * max_k = %lld > 1024 (merge kernel)", max_nq,
                                 st.max_nq, (long long)world * max_k);
    icd_group *g = new (std::nothrow) icd_group();
    if (!g) return icd_internal_fail(ICD_ERR_NOMEM, "host allocation failed");
    g->idx = local; g->device = st.device; g->rank = rank; g->world = world; g->mode = mode; g->max_nq = max_nq; g->max_k = max_k; g->dim = st.dim; g->idx_max_nq = st.max_nq;
    g->use_comm = world > 1 || with_comm != 0;
#define GR_TRY(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) {                                                                               \
            free_group(g);                                                                                    \
            return icd_internal_fail(e_ == hipErrorOutOfMemory ? ICD_ERR_NOMEM : ICD_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
        }                                                                                                     \
    } while (0)
    GR_TRY(hipSetDevice(g->device));
    const size_t per = (size_t)max_nq * max_k;
    if (mode == ICD_GROUP_ROW_SHARD) {
        GR_TRY(hipMalloc(reinterpret_cast<void **>(&g->send_s), per * sizeof(float)));
        GR_TRY(hipMalloc(reinterpret_cast<void **>(&g->send_i), per * sizeof(long long)));
        GR_TRY(hipMalloc(reinterpret_cast<void **>(&g->send_l), per * sizeof(int)));
        if (g->use_comm) {
            GR_TRY(hipMalloc(reinterpret_cast<void **>(&g->recv_s), per * world * sizeof(float)));
            GR_TRY(hipMalloc(reinterpret_cast<void **>(&g->recv_i), per * world * sizeof(long long)));
            GR_TRY(hipMalloc(reinterpret_cast<void **>(&g->recv_l), per * world * sizeof(int)));
        }
    } else if (g->use_comm) {
        // one 24-byte record per hit (adj f64 | id i64 | raw f32 | level i32), slices padded to the same length
        const size_t width = ((size_t)max_nq + world - 1) / world;
        GR_TRY(hipMalloc(reinterpret_cast<void **>(&g->qs_adj), width * max_k * 24));
        GR_TRY(hipMalloc(reinterpret_cast<void **>(&g->qr_adj), width * max_k * 24 * world));
    }
#undef GR_TRY
    if (g->use_comm) {
        rc = load_rccl();
        if (rc) { free_group(g); return rc; }
    }
    *out = g;
    return ICD_OK;
}

// COLLECTIVE over the group's ranks (ncclCommInitRank): call it on every rank or on none. A group prepared without a
// communicator (one rank, with_comm = 0) needs no connect. On failure the group stays prepared (destroy it).
int icd_group_connect(icd_group *g, const uint8_t *id) {
    if (!valid(g)) return icd_internal_fail(ICD_ERR_STATE, "invalid group handle");
    if (!g->use_comm) return ICD_OK;
    if (g->comm) return icd_internal_fail(ICD_ERR_STATE, "rank %d of %d: the group is connected already", g->rank, g->world);
    if (!id) return icd_internal_fail(ICD_ERR_INVALID, "rank %d of %d: the unique id of rank 0 is NULL (icd_group_unique_id)", g->rank, g->world);
    HIPG_TRY(hipSetDevice(g->device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclResult_t r = g_rccl.CommInitRank(&g->comm, g->world, uid, g->rank);
    if (r != ncclSuccess) {
        g->comm = nullptr;
        return icd_internal_fail(ICD_ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", g->rank, g->world, g_rccl.GetErrorString(r));
    }
    return ICD_OK;
}

// prepare + connect in one call: for a host whose ranks need no agreement step (a failure of one rank's local part leaves
// the others waiting in ncclCommInitRank: hosts that can fail locally use the two calls above)
int icd_group_create(icd_index *local, const uint8_t *id, int32_t rank, int32_t world, int32_t mode, int32_t max_nq,
                     int32_t max_k, icd_group **out) {
    if (out) *out = nullptr;
    if (world > 1 && !id) return icd_internal_fail(ICD_ERR_INVALID, "a group of %d ranks needs the unique id of rank 0 (icd_group_unique_id)", world);
    icd_group *g = nullptr;
    int rc = icd_group_prepare(local, id != nullptr, rank, world, mode, max_nq, max_k, &g);
    if (rc) return rc;
    if (id) {
        rc = icd_group_connect(g, id);
        if (rc) { free_group(g); return rc; }
    }
    *out = g;
    return ICD_OK;
}

int icd_group_destroy(icd_group *g) {
    if (!valid(g)) return icd_internal_fail(ICD_ERR_STATE, "invalid group handle");
    hipSetDevice(g->device);
    hipDeviceSynchronize();
    free_group(g);
    return ICD_OK;
}

int icd_group_search(icd_group *g, const float *queries, int64_t nq, int32_t k, int32_t gather, double *out_adj,
                     float *out_raw, int64_t *out_ids, int32_t *out_levels, void *stream) {
    if (!valid(g)) return icd_internal_fail(ICD_ERR_STATE, "invalid group handle");
    if (!out_adj || !out_raw || !out_ids || !out_levels) return icd_internal_fail(ICD_ERR_INVALID, "output pointer is NULL");
    if (nq < 0 || nq > g->max_nq || k <= 0 || k > g->max_k) return icd_internal_fail(ICD_ERR_INVALID, "nq=%lld k=%d (group max %d / %d)", (long long)nq, k, g->max_nq, g->max_k);
    if (nq == 0) return ICD_OK;
    if (!queries) return icd_internal_fail(ICD_ERR_INVALID, "queries is NULL");
    if (g->use_comm && !g->comm) return icd_internal_fail(ICD_ERR_STATE, "rank %d of %d: the group was prepared but never connected (icd_group_connect)", g->rank, g->world);
    HIPG_TRY(hipSetDevice(g->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int rc;
    if (g->mode == ICD_GROUP_ROW_SHARD) {
        // local raw top-k (global ids: the index was created with id_base = first row of the shard) + the hits' levels
        rc = icd_index_search(g->idx, queries, nq, k, 1, ICD_MODE_AUTO, g->send_s, reinterpret_cast<int64_t *>(g->send_i), 1, stream);
        if (rc) return rc;
        rc = icd_index_lookup_levels(g->idx, reinterpret_cast<const int64_t *>(g->send_i), nq * k, g->send_l, stream);
        if (rc) return rc;
        const float *ms = g->send_s; const long long *mi = g->send_i; const int *ml = g->send_l;
        if (g->use_comm) {
            const size_t cnt = (size_t)nq * k;
            NCCL_TRY_G(g, false, g_rccl.GroupStart());   // the three arrays travel as ONE fused collective launch
            NCCL_TRY_G(g, true, g_rccl.AllGather(g->send_s, g->recv_s, cnt, ncclFloat32, g->comm, s));
            NCCL_TRY_G(g, true, g_rccl.AllGather(g->send_i, g->recv_i, cnt, ncclInt64, g->comm, s));
            NCCL_TRY_G(g, true, g_rccl.AllGather(g->send_l, g->recv_l, cnt, ncclInt32, g->comm, s));
            NCCL_TRY_G(g, false, g_rccl.GroupEnd());
            ms = g->recv_s; mi = g->recv_i; ml = g->recv_l;   // [world][nq][k]
        }
        return icd_merge_topk(g->device, ms, reinterpret_cast<const int64_t *>(mi), ml, g->world, nq, k, out_adj, out_raw, out_ids, out_levels, stream);
    }
    // query-sharded: this rank's slice
    int64_t lo, hi;
    shard_bounds(nq, g->world, g->rank, &lo, &hi);
    const int64_t m = hi - lo;
    const float *qs = queries + (size_t)lo * g->dim;
    // (a slice larger than the index's max_nq goes through it in pieces)
    auto search_slice = [&](double *adj, float *raw, int64_t *ids, int32_t *lv) -> int {
        for (int64_t off = 0; off < m; off += g->idx_max_nq) {
            const int64_t c = std::min<int64_t>(g->idx_max_nq, m - off);
            const size_t o = (size_t)off * k;
            const int r = icd_index_search_reweighted(g->idx, qs + (size_t)off * g->dim, c, k, 1, ICD_MODE_AUTO, adj + o, raw + o, ids + o, lv + o, 1, stream);
            if (r) return r;
        }
        return ICD_OK;
    };
    if (!g->use_comm || !gather)   // the local slice, written to the first (hi - lo) rows of the outputs
        return search_slice(out_adj, out_raw, out_ids, out_levels);
    const size_t width = ((size_t)nq + g->world - 1) / g->world, per = width * k;
    // four arrays of one padded slice, back to back in the send buffer: adj f64 | ids i64 | raw f32 | levels i32
    char *sb = reinterpret_cast<char *>(g->qs_adj), *rb = reinterpret_cast<char *>(g->qr_adj);
    double *s_adj = reinterpret_cast<double *>(sb);
    long long *s_ids = reinterpret_cast<long long *>(sb + per * 8);
    float *s_raw = reinterpret_cast<float *>(sb + per * 16);
    int *s_lv = reinterpret_cast<int *>(sb + per * 20);
    rc = search_slice(s_adj, s_raw, reinterpret_cast<int64_t *>(s_ids), s_lv);
    if (rc) return rc;
    double *r_adj = reinterpret_cast<double *>(rb);
    long long *r_ids = reinterpret_cast<long long *>(rb + per * g->world * 8);
    float *r_raw = reinterpret_cast<float *>(rb + per * g->world * 16);
    int *r_lv = reinterpret_cast<int *>(rb + per * g->world * 20);
    NCCL_TRY_G(g, false, g_rccl.GroupStart());
    NCCL_TRY_G(g, true, g_rccl.AllGather(s_adj, r_adj, per, ncclFloat64, g->comm, s));
    NCCL_TRY_G(g, true, g_rccl.AllGather(s_ids, r_ids, per, ncclInt64, g->comm, s));
    NCCL_TRY_G(g, true, g_rccl.AllGather(s_raw, r_raw, per, ncclFloat32, g->comm, s));
    NCCL_TRY_G(g, true, g_rccl.AllGather(s_lv, r_lv, per, ncclInt32, g->comm, s));
    NCCL_TRY_G(g, false, g_rccl.GroupEnd());
    // padded slices -> the contiguous [nq][k] outputs: one kernel (it restates shard_bounds; the gloo / torch engine's host-side
    // concatenation in sharded.py is the reference the GPU test compares with)
    if (icd_internal_unpack_query_slices(r_adj, r_ids, r_raw, r_lv, g->world, nq, k, (long long)width, out_adj, out_raw, out_ids, out_levels, stream))
        return icd_internal_fail(ICD_ERR_HIP, "rank %d of %d: the unpack launch of the query-sharded gather failed", g->rank, g->world);
    return ICD_OK;
}

}  // extern "C"
