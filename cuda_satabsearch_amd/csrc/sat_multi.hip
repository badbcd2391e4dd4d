// sat_multi.hip - one search over the GPUs of a node from ONE host thread (include/satabsearch.h,
// sat_multi_*): the database is cut into contiguous shards of equal cost (csrc/host/sat_shard.c),
// every GPU holds only its shard plus the queries, the search is queued on all of them, and ONE
// gather brings the per-shard score arrays (and the int8 solution maps) into device 0's memory, from
// where a single copy takes them to the host, in database file order.
//
// The reference is single-GPU (cudaSaTabsearch.cu:790 "TODO allow multiple GPUs"); SURVEY.md section
// 8e specifies this mode.  Every (query, entry) pair is independent and the random streams are keyed
// by the entry's ordinal in the whole database, so the result is the same for any number of shards.
//
// The gather is RCCL's ncclGather over xGMI (single-process communicators from ncclCommInitAll; the
// library is loaded with dlopen when a multi-GPU context is created, so single-GPU users never pay
// for it).  Shards are padded to the largest one: a fixed-size gather, the rows are put in order on
// the host after the one device-to-host copy.  SAT_MULTI_GATHER=peer selects hipMemcpyPeerAsync
// into device 0 instead (also what is used when librccl cannot be loaded).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types and prototypes only: the entry points are resolved with dlsym

#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "sat_ctx.hpp"
#include "host/sat_shard.h"

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t err__ = (expr);                                                          \
        if (err__ != hipSuccess)                                                            \
            return sat_fail(err__ == hipErrorOutOfMemory ? SAT_ENOMEM : SAT_EDEVICE,        \
                            "%s failed: %s", #expr, hipGetErrorString(err__));              \
    } while (0)

namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGather) Gather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;

    bool load()
    {
        if (handle) return true;
        handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!handle) handle = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!handle) return false;
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(handle, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(handle, "ncclCommDestroy"));
        Gather = reinterpret_cast<decltype(Gather)>(dlsym(handle, "ncclGather"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(handle, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(handle, "ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(handle, "ncclGetErrorString"));
        return CommInitAll && CommDestroy && Gather && GroupStart && GroupEnd && GetErrorString;
    }
};

Rccl g_rccl;      // process-wide: the library is loaded at most once

}  // namespace

struct sat_multi {
    int ndev = 0;
    std::vector<int> devices;
    std::vector<sat_ctx *> ctx;
    std::vector<int32_t> begin;                 // shard g = entries begin[g] .. begin[g+1]-1 of the database
    int n_entries = 0;
    int pad_rows = 0;                           // largest shard: every shard's rows are padded to it in the gather
    bool use_rccl = false;
    bool force_gather = false;                  // SAT_MULTI_GATHER set: gather also with one GPU (tests)
    bool rccl_required = false;                 // SAT_MULTI_GATHER=rccl: an RCCL failure is an error, no peer-copy fallback
    std::vector<ncclComm_t> comm;
    // gathered rows on device 0: [ndev][nq * pad_rows] scores, [ndev][pad_rows * sum(n1)] map bytes
    int32_t *d_all_scores = nullptr;
    size_t all_scores_cap = 0;
    int8_t *d_all_maps = nullptr;
    size_t all_maps_cap = 0;
    // pinned landing zone of the one device-to-host copy
    void *h_stage = nullptr;
    size_t h_stage_cap = 0;
    std::vector<hipEvent_t> done;               // peer-copy path: shard g's rows have arrived on device 0
    unsigned long long d2h_bytes = 0;
};

namespace {

int rccl_fail(ncclResult_t r, const char *what)
{
    return sat_fail(SAT_EDEVICE, "%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
}

template <typename T> int grow_dev(T *&p, size_t &cap, size_t need)
{
    if (need <= cap) return SAT_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    HIP_TRY(hipMalloc(&p, need * sizeof(T)));
    cap = need;
    return SAT_OK;
}

int grow_stage(sat_multi *m, size_t bytes)
{
    if (bytes <= m->h_stage_cap) return SAT_OK;
    if (m->h_stage) (void)hipHostFree(m->h_stage);
    m->h_stage = nullptr;
    m->h_stage_cap = 0;
    HIP_TRY(hipHostMalloc(&m->h_stage, bytes, hipHostMallocDefault));
    m->h_stage_cap = bytes;
    return SAT_OK;
}

// the peer-copy path signals "shard g's rows have arrived on device 0" with one event per sender; they are made
// when that path is first taken (at creation without RCCL, or when a gather falls back to it)
int ensure_peer_events(sat_multi *m)
{
    for (int g = 1; g < m->ndev; g++) {
        if (m->done[(size_t)g]) continue;
        HIP_TRY(hipSetDevice(m->devices[(size_t)g]));
        HIP_TRY(hipEventCreateWithFlags(&m->done[(size_t)g], hipEventDisableTiming));
        (void)hipDeviceEnablePeerAccess(m->devices[0], 0);     // best effort: the copy is staged without it
        (void)hipGetLastError();
    }
    return SAT_OK;
}

// wait for everything queued on every GPU of the set (before an error return: no search may still be writing
// result buffers the caller is about to re-use or free); errors of the waits themselves are dropped
void sync_all(sat_multi *m)
{
    for (int g = 0; g < m->ndev; g++) {
        if (hipSetDevice(m->devices[(size_t)g]) == hipSuccess) (void)hipStreamSynchronize(m->ctx[(size_t)g]->stream);
        (void)hipGetLastError();
    }
}

// bring `count` elements of every device's `src(g)` into block g of `dst` on device 0
template <typename T, typename Src>
int gather_to_device0(sat_multi *m, T *dst, size_t count, ncclDataType_t type, Src src)
{
    sat_ctx *root = m->ctx[0];
    if (m->use_rccl) {
        ncclResult_t r = g_rccl.GroupStart();
        bool ok = r == ncclSuccess;
        for (int g = 0; ok && g < m->ndev; g++) {
            if (hipSetDevice(m->devices[(size_t)g]) != hipSuccess) {       // (never leave the group open)
                (void)g_rccl.GroupEnd();
                return sat_fail(SAT_EDEVICE, "hipSetDevice(%d) failed inside the gather", m->devices[(size_t)g]);
            }
            r = g_rccl.Gather(src(g), dst, count, type, 0, m->comm[(size_t)g], m->ctx[(size_t)g]->stream);
            ok = r == ncclSuccess;
        }
        const ncclResult_t rend = g_rccl.GroupEnd();
        if (ok && rend == ncclSuccess) return SAT_OK;
        // RCCL refused the gather at run time: unless the caller insisted on it, take the peer-copy path from
        // here on (the searches are queued and their rows sit in each GPU's memory; nothing is lost)
        if (m->rccl_required)
            return rccl_fail(ok ? rend : r, ok ? "ncclGroupEnd" : "ncclGather");
        fprintf(stderr, "satabsearch: RCCL gather failed (%s); falling back to peer copies\n",
                g_rccl.GetErrorString ? g_rccl.GetErrorString(ok ? rend : r) : "RCCL error");
        sync_all(m);
        (void)hipGetLastError();
        m->use_rccl = false;
        const int rc = ensure_peer_events(m);
        if (rc != SAT_OK) return rc;
    }
    for (int g = 0; g < m->ndev; g++) {
        sat_ctx *c = m->ctx[(size_t)g];
        HIP_TRY(hipSetDevice(m->devices[(size_t)g]));
        HIP_TRY(hipMemcpyPeerAsync(dst + (size_t)g * count, m->devices[0], src(g), m->devices[(size_t)g], count * sizeof(T), c->stream));
        if (g > 0) {
            HIP_TRY(hipEventRecord(m->done[(size_t)g], c->stream));
            HIP_TRY(hipStreamWaitEvent(root->stream, m->done[(size_t)g], 0));
        }
    }
    return SAT_OK;
}

}  // namespace

extern "C" {

sat_multi *sat_multi_create(int ndev, const int *devices, uint64_t seed)
{
    const int visible = sat_device_count();
    if (ndev <= 0) ndev = visible;
    // an explicit device list may name a device more than once (several shards on one GPU: how the
    // multi-shard path is exercised on a one-GPU box; RCCL refuses duplicates, peer copies do not)
    bool listed_ok = devices != nullptr;
    for (int g = 0; listed_ok && g < ndev; g++) listed_ok = devices[g] >= 0 && devices[g] < visible;
    if (visible <= 0 || (devices ? !listed_ok : ndev > visible)) {
        sat_fail(SAT_ENODEVICE, "%d GPUs asked for, %d visible (this library has no CPU path)", ndev, visible);
        return nullptr;
    }
    sat_multi *m = new (std::nothrow) sat_multi();
    if (!m) {
        sat_fail(SAT_ENOMEM, "out of host memory");
        return nullptr;
    }
    m->ndev = ndev;
    for (int g = 0; g < ndev; g++) m->devices.push_back(devices ? devices[g] : g);
    for (int g = 0; g < ndev; g++) {
        sat_ctx *c = sat_ctx_create(m->devices[(size_t)g], seed);
        if (!c) {
            sat_multi_destroy(m);
            return nullptr;
        }
        m->ctx.push_back(c);
    }
    m->done.assign((size_t)ndev, nullptr);
    const char *how = getenv("SAT_MULTI_GATHER");
    const bool want_peer = how && !strcmp(how, "peer");
    const bool force_rccl = how && !strcmp(how, "rccl");            // also with one GPU (tests)
    m->force_gather = want_peer || force_rccl;
    bool duplicates = false;
    for (int g = 0; g < ndev; g++)
        for (int h = 0; h < g; h++) duplicates = duplicates || m->devices[(size_t)g] == m->devices[(size_t)h];
    if (!want_peer && !duplicates && (ndev > 1 || force_rccl) && g_rccl.load()) {
        m->comm.assign((size_t)ndev, nullptr);
        if (g_rccl.CommInitAll(m->comm.data(), ndev, m->devices.data()) == ncclSuccess) m->use_rccl = true;
        else m->comm.clear();
    }
    if (force_rccl && !m->use_rccl) {
        sat_fail(SAT_EDEVICE, "SAT_MULTI_GATHER=rccl but librccl could not be loaded / initialised");
        sat_multi_destroy(m);
        return nullptr;
    }
    m->rccl_required = force_rccl;
    if (!m->use_rccl && ensure_peer_events(m) != SAT_OK) {
        sat_multi_destroy(m);
        return nullptr;
    }
    return m;
}

void sat_multi_destroy(sat_multi *m)
{
    if (!m) return;
    for (size_t g = 0; g < m->comm.size(); g++)
        if (m->comm[g]) (void)g_rccl.CommDestroy(m->comm[g]);
    if (!m->devices.empty()) (void)hipSetDevice(m->devices[0]);
    if (m->d_all_scores) (void)hipFree(m->d_all_scores);
    if (m->d_all_maps) (void)hipFree(m->d_all_maps);
    if (m->h_stage) (void)hipHostFree(m->h_stage);
    for (size_t g = 0; g < m->done.size(); g++)
        if (m->done[g]) (void)hipEventDestroy(m->done[g]);
    for (sat_ctx *c : m->ctx) sat_ctx_destroy(c);
    delete m;
}

int sat_multi_device_count(const sat_multi *m) { return m ? m->ndev : 0; }

const char *sat_multi_gather_kind(const sat_multi *m)
{
    if (!m) return "";
    if (m->ndev == 1 && !m->force_gather) return "none";
    return m->use_rccl ? "rccl" : "peer";
}

int sat_multi_db_upload_packed(sat_multi *m, int n_entries, const int32_t *orders, const int64_t *cell_off,
                               const uint8_t *tab_tri, const float *dist_tri)
{
    if (!m) return sat_fail(SAT_EINVAL, "null context");
    if (n_entries < m->ndev) return sat_fail(SAT_EINVAL, "%d entries cannot be cut into %d shards", n_entries, m->ndev);
    if (!orders || !cell_off || !tab_tri || !dist_tri) return sat_fail(SAT_EINVAL, "null array");
    // a shard is uploaded as a WINDOW of the packed arrays (from its first entry's first cell): the entries must lie
    // in file order, one after the other without overlap - what every reader here produces
    for (int e = 0; e + 1 < n_entries; e++) {
        const int64_t n = orders[e];
        if (n < 1 || n > SAT_MAXDIM) return sat_fail(SAT_EINVAL, "entry %d: order %lld outside 1..%d", e, (long long)n, SAT_MAXDIM);
        if (cell_off[e] < 0 || cell_off[e + 1] < cell_off[e] + n * (n + 1) / 2)
            return sat_fail(SAT_EINVAL, "entry %d: cell offsets must ascend in file order without overlap for a sharded upload "
                            "(entry %d starts at cell %lld, entry %d at %lld)", e + 1, e, (long long)cell_off[e], e + 1, (long long)cell_off[e + 1]);
    }
    m->begin.assign((size_t)m->ndev + 1, 0);
    if (sat_shard_cuts(n_entries, orders, m->ndev, m->begin.data()) != 0) return sat_fail(SAT_EINVAL, "bad database");
    m->n_entries = n_entries;
    m->pad_rows = 0;
    std::vector<int64_t> ordinal((size_t)n_entries);
    for (int e = 0; e < n_entries; e++) ordinal[(size_t)e] = e;
    // every GPU has its own link to the host: the shards go up concurrently, one host thread per GPU
    std::vector<int> rcs((size_t)m->ndev, SAT_OK);
    std::vector<std::string> errs((size_t)m->ndev);
    auto upload_shard = [&](int g) {
        const int b = m->begin[(size_t)g], n = m->begin[(size_t)g + 1] - b;
        // a shard is a window of the packed arrays: rebase its cell offsets to the window
        std::vector<int64_t> off((size_t)n);
        for (int e = 0; e < n; e++) off[(size_t)e] = cell_off[b + e] - cell_off[b];
        rcs[(size_t)g] = sat_db_upload_packed(m->ctx[(size_t)g], n, orders + b, off.data(), tab_tri + cell_off[b],
                                              dist_tri + cell_off[b], ordinal.data() + b);
        if (rcs[(size_t)g] != SAT_OK) errs[(size_t)g] = sat_last_error();      // the message is per thread
    };
    {
        std::vector<std::thread> pool;
        for (int g = 1; g < m->ndev; g++) pool.emplace_back(upload_shard, g);
        upload_shard(0);
        for (auto &th : pool) th.join();
    }
    for (int g = 0; g < m->ndev; g++) {
        if (rcs[(size_t)g] != SAT_OK) {
            // entry numbers in the message are relative to the shard: say which
            return sat_fail(rcs[(size_t)g], "shard %d (entries from %d): %s", g, m->begin[(size_t)g], errs[(size_t)g].c_str());
        }
        const int n = m->begin[(size_t)g + 1] - m->begin[(size_t)g];
        if (n > m->pad_rows) m->pad_rows = n;
    }
    for (int g = 0; g < m->ndev; g++) m->ctx[(size_t)g]->min_rows = m->pad_rows;     // result buffers hold a padded shard
    return SAT_OK;
}

int sat_multi_shards(const sat_multi *m, int32_t *begin)
{
    if (!m || !begin) return sat_fail(SAT_EINVAL, "null argument");
    if (m->begin.empty()) return sat_fail(SAT_ESTATE, "no database uploaded");
    for (int g = 0; g <= m->ndev; g++) begin[g] = m->begin[(size_t)g];
    return SAT_OK;
}

int sat_multi_queries_set(sat_multi *m, int n_queries, const int32_t *n1s, const uint8_t *qtabs, const float *qdmats,
                          int pitch, const uint8_t *qssetypes, uint32_t first_query_ordinal)
{
    if (!m) return sat_fail(SAT_EINVAL, "null context");
    for (int g = 0; g < m->ndev; g++) {
        int rc = sat_queries_set(m->ctx[(size_t)g], n_queries, n1s, qtabs, qdmats, pitch, qssetypes, first_query_ordinal);
        if (rc != SAT_OK) return rc;
    }
    return SAT_OK;
}

int sat_multi_search(sat_multi *m, int lorder, int lsoln, int maxstart, int32_t *scores, int32_t *ssemaps, double *wall_ms)
{
    if (!m) return sat_fail(SAT_EINVAL, "null context");
    if (!scores) return sat_fail(SAT_EINVAL, "scores buffer is null");
    if (lsoln && !ssemaps) return sat_fail(SAT_EINVAL, "lsoln set but ssemaps buffer is null");
    if (m->begin.empty()) return sat_fail(SAT_ESTATE, "no database uploaded");
    const auto t0 = std::chrono::steady_clock::now();
    // (an error below waits for the searches already queued on the other GPUs before it is returned)
    auto bail = [&](int rc) { const std::string msg = sat_last_error(); sync_all(m); return sat_fail(rc, "%s", msg.c_str()); };
    for (int g = 0; g < m->ndev; g++) {
        int rc = sat_search_async(m->ctx[(size_t)g], lorder, lsoln, maxstart);
        if (rc != SAT_OK) return bail(rc);
    }
    sat_ctx *root = m->ctx[0];
    const size_t nq = root->queries.size(), N = (size_t)m->n_entries, pad = (size_t)m->pad_rows;
    if (m->ndev == 1 && !m->force_gather) {
        int rc = sat_results(root, lsoln, scores, ssemaps);
        if (wall_ms) *wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return rc;
    }
    // ---- one gather of the (padded) per-shard rows to device 0, one copy to the host
    size_t map_bytes_per_row = 0;
    for (const auto &q : root->queries) map_bytes_per_row += (size_t)q.n1;
    const size_t score_count = nq * pad, map_count = pad * map_bytes_per_row;
    int rc;
    if (hipSetDevice(m->devices[0]) != hipSuccess) return bail(sat_fail(SAT_EDEVICE, "hipSetDevice failed"));
    if ((rc = grow_dev(m->d_all_scores, m->all_scores_cap, score_count * (size_t)m->ndev)) != SAT_OK) return bail(rc);
    if (lsoln && (rc = grow_dev(m->d_all_maps, m->all_maps_cap, map_count * (size_t)m->ndev)) != SAT_OK) return bail(rc);
    const size_t stage_bytes = score_count * (size_t)m->ndev * sizeof(int32_t) + (lsoln ? map_count * (size_t)m->ndev : 0);
    if ((rc = grow_stage(m, stage_bytes)) != SAT_OK) return bail(rc);
    if ((rc = gather_to_device0(m, m->d_all_scores, score_count, ncclInt32,
                                [&](int g) { return (const int32_t *)m->ctx[(size_t)g]->d_scores; })) != SAT_OK) return bail(rc);
    if (lsoln && (rc = gather_to_device0(m, m->d_all_maps, map_count, ncclInt8,
                                         [&](int g) { return (const int8_t *)m->ctx[(size_t)g]->d_ssemaps; })) != SAT_OK) return bail(rc);
    int32_t *h_scores = static_cast<int32_t *>(m->h_stage);
    int8_t *h_maps = reinterpret_cast<int8_t *>(h_scores + score_count * (size_t)m->ndev);
    auto to_host = [&]() -> int {
        HIP_TRY(hipSetDevice(m->devices[0]));
        HIP_TRY(hipMemcpyAsync(h_scores, m->d_all_scores, score_count * (size_t)m->ndev * sizeof(int32_t), hipMemcpyDeviceToHost, root->stream));
        if (lsoln) HIP_TRY(hipMemcpyAsync(h_maps, m->d_all_maps, map_count * (size_t)m->ndev, hipMemcpyDeviceToHost, root->stream));
        HIP_TRY(hipStreamSynchronize(root->stream));
        for (int g = 1; g < m->ndev; g++) {                            // the senders' streams are done too
            HIP_TRY(hipSetDevice(m->devices[(size_t)g]));
            HIP_TRY(hipStreamSynchronize(m->ctx[(size_t)g]->stream));
        }
        return SAT_OK;
    };
    if ((rc = to_host()) != SAT_OK) return bail(rc);
    m->d2h_bytes += stage_bytes;
    // rows of shard g: scores [nq][n_g] at block g; maps: query q's [n_g][n1_q] block after those of queries 0..q-1
    for (int g = 0; g < m->ndev; g++) {
        const size_t b = (size_t)m->begin[(size_t)g], n = (size_t)m->begin[(size_t)g + 1] - b;
        const int32_t *src = h_scores + (size_t)g * score_count;
        for (size_t q = 0; q < nq; q++) memcpy(scores + q * N + b, src + q * n, n * sizeof(int32_t));
        if (lsoln) {
            const int8_t *msrc = h_maps + (size_t)g * map_count;
            size_t qoff = 0;
            for (size_t q = 0; q < nq; q++) {
                const size_t n1 = (size_t)root->queries[q].n1;
                int32_t *out = ssemaps + (q * N + b) * SAT_MAXDIM;
                for (size_t e = 0; e < n; e++) {
                    for (size_t i = 0; i < n1; i++) out[e * SAT_MAXDIM + i] = msrc[qoff + e * n1 + i];
                    for (size_t i = n1; i < SAT_MAXDIM; i++) out[e * SAT_MAXDIM + i] = -1;
                }
                qoff += n * n1;
            }
        }
    }
    if (wall_ms) *wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return SAT_OK;
}

int sat_multi_search_topk(sat_multi *m, int lorder, int lsoln, int maxstart, int k, sat_hit *hits, int32_t *ssemaps, double *wall_ms)
{
    if (!m) return sat_fail(SAT_EINVAL, "null context");
    if (!hits || k < 1) return sat_fail(SAT_EINVAL, "bad top-k arguments");
    if (m->begin.empty()) return sat_fail(SAT_ESTATE, "no database uploaded");
    const auto t0 = std::chrono::steady_clock::now();
    auto bail = [&](int rc) { const std::string msg = sat_last_error(); sync_all(m); return sat_fail(rc, "%s", msg.c_str()); };
    for (int g = 0; g < m->ndev; g++) {
        int rc = sat_search_async(m->ctx[(size_t)g], lorder, lsoln, maxstart);
        if (rc != SAT_OK) return bail(rc);
    }
    if (k > m->n_entries) k = m->n_entries;
    const int nq = (int)m->ctx[0]->queries.size();
    // every GPU ranks its own shard (k rows per query leave each GPU), the host merges ndev x k candidates
    std::vector<std::vector<sat_hit>> cand((size_t)m->ndev);
    std::vector<std::vector<int32_t>> cmaps((size_t)m->ndev);
    std::vector<int> got((size_t)m->ndev, 0);
    for (int g = 0; g < m->ndev; g++) {
        cand[(size_t)g].resize((size_t)nq * k);
        if (ssemaps) cmaps[(size_t)g].resize((size_t)nq * k * SAT_MAXDIM);
        const int r = sat_topk_hits(m->ctx[(size_t)g], k, cand[(size_t)g].data(), ssemaps ? cmaps[(size_t)g].data() : nullptr);
        if (r < 0) return bail(r);
        got[(size_t)g] = r;
    }
    for (int q = 0; q < nq; q++) {
        std::vector<int> head((size_t)m->ndev, 0);
        for (int r = 0; r < k; r++) {
            int bg = -1;
            for (int g = 0; g < m->ndev; g++) {
                if (head[(size_t)g] >= got[(size_t)g]) continue;
                // ties in database order: shards are contiguous, so the lower GPU wins a tie
                if (bg < 0 || cand[(size_t)g][(size_t)q * got[(size_t)g] + head[(size_t)g]].score >
                                  cand[(size_t)bg][(size_t)q * got[(size_t)bg] + head[(size_t)bg]].score)
                    bg = g;
            }
            const size_t row = (size_t)q * got[(size_t)bg] + head[(size_t)bg];
            sat_hit h = cand[(size_t)bg][row];
            h.entry += m->begin[(size_t)bg];
            hits[(size_t)q * k + r] = h;
            if (ssemaps) memcpy(ssemaps + ((size_t)q * k + r) * SAT_MAXDIM, cmaps[(size_t)bg].data() + row * SAT_MAXDIM, sizeof(int32_t) * SAT_MAXDIM);
            head[(size_t)bg]++;
        }
    }
    if (wall_ms) *wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return k;
}

unsigned long long sat_multi_stat_d2h_bytes(const sat_multi *m)
{
    if (!m) return 0ull;
    unsigned long long total = m->d2h_bytes;
    for (const sat_ctx *c : m->ctx) total += sat_stat_d2h_bytes(c);
    return total;
}

}  // extern "C"
