// sat_topk.hip - best-k hits of a finished search, selected, ranked and given their statistics
// on the device, so that only k rows per query leave the GPU.
//
// Users of the reference sort the full "name score norm2 z p" listing by raw score and keep the
// head (README_example_usage.txt:100, 256: `sort -k 2,2nr | head`).  Here: one pass packs every
// (score, entry index) of the query batch into 64-bit keys, one segmented radix sort (rocPRIM
// through hipCUB; one segment per query) orders them, and a last kernel turns the first k keys of
// each segment into rows {entry, score, norm2, z, p} (gumbelstats.c:50-94 via csrc/host/sat_gumbel.c)
// and gathers their solution maps when the search ran with LSOLN.  Ties keep database order.
//
// The statistics are bit-identical to the host's: norm2 = 2 * score / (n1 + n2) is one IEEE double
// division on either side, and z and p, which the reference computes from norm2 TRUNCATED TO AN
// INT (gumbelstats.h:26 vs cudaSaTabsearch.cu:446), are looked up in a 256-entry table the host
// fills with its own libm when the context is created - no device exp().
//
// All scratch (keys, sort space, rows) belongs to the context and only ever grows.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <vector>

#include "sat_ctx.hpp"

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t err__ = (expr);                                                          \
        if (err__ != hipSuccess)                                                            \
            return sat_fail(err__ == hipErrorOutOfMemory ? SAT_ENOMEM : SAT_EDEVICE,        \
                            "%s failed: %s", #expr, hipGetErrorString(err__));              \
    } while (0)

namespace {

// key = biased score in the high word, inverted entry index in the low word: a descending
// sort lists higher scores first and, among equal scores, lower entry indices first
__global__ void pack_keys(const int32_t *scores, long long total, int n, unsigned long long *keys)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        const uint32_t e = (uint32_t)(i % n);
        keys[i] = ((unsigned long long)(uint32_t)(scores[i] + 0x40000000) << 32) | (0xFFFFFFFFu - e);
    }
}

struct HitQuery {
    int32_t n1;
    int32_t pad_;
    const int8_t *ssemaps;     // this query's [N][n1] maps, or null
};

// one thread per (query, rank): decode the key, look the statistics up, gather the map
__global__ void finish_hits(const unsigned long long *sorted, int n, int k, int nq, const int32_t *orders,
                            const HitQuery *queries, const double *ztab, const double *ptab,
                            sat_hit *hits, int32_t *maps)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nq * k) return;
    const int q = t / k, r = t - q * k;
    const unsigned long long key = sorted[(size_t)q * n + r];
    const int32_t entry = (int32_t)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFu));
    const int32_t score = (int32_t)(uint32_t)(key >> 32) - 0x40000000;
    const int n1 = queries[q].n1, n2 = orders[entry];
    const double norm2 = 2.0 * score / ((double)(n1 + n2));            // sat_norm2
    int x = (int)norm2;                                                // the reference's double -> int
    x = x < -128 ? -128 : (x > 127 ? 127 : x);                         // |norm2| <= 110 for every legal score
    sat_hit h;
    h.entry = entry;
    h.score = score;
    h.norm2 = norm2;
    h.zscore = ztab[x + 128];
    h.pvalue = ptab[x + 128];
    hits[t] = h;
    if (maps) {
        int32_t *out = maps + (size_t)t * SAT_MAXDIM;
        const int8_t *src = queries[q].ssemaps ? queries[q].ssemaps + (size_t)entry * n1 : nullptr;
        for (int i = 0; i < SAT_MAXDIM; i++) out[i] = (src && i < n1) ? (int32_t)src[i] : -1;
    }
}

template <typename T> int grow(T *&p, size_t &cap, size_t need)
{
    if (need <= cap) return SAT_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    HIP_TRY(hipMalloc(&p, need * sizeof(T)));
    cap = need;
    return SAT_OK;
}

// rank queries [q0, q0 + nq) of the last search; rows land at row `out_row` of ctx->d_hits (and
// ctx->d_hit_maps), which hold `rows_total` rows
int select_hits(sat_ctx *ctx, int q0, int nq, int k, bool want_maps, size_t out_row, size_t rows_total)
{
    const int n = ctx->n_entries;
    const size_t total = (size_t)nq * n;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc;
    if ((rc = grow(ctx->d_keys, ctx->keys_cap, total)) != SAT_OK) return rc;
    if ((rc = grow(ctx->d_sorted, ctx->sorted_cap, total)) != SAT_OK) return rc;
    if (out_row == 0) {                                   // first chunk: size the row buffers for the whole batch
        if ((rc = grow(ctx->d_hits, ctx->hits_cap, rows_total)) != SAT_OK) return rc;
        if (want_maps && (rc = grow(ctx->d_hit_maps, ctx->hit_maps_cap, rows_total * SAT_MAXDIM)) != SAT_OK) return rc;
    }
    if ((rc = grow(ctx->d_seg, ctx->seg_cap, (size_t)nq + 1)) != SAT_OK) return rc;
    if ((rc = grow(ctx->d_hitq, ctx->hitq_cap, (size_t)nq * sizeof(HitQuery))) != SAT_OK) return rc;

    std::vector<int> seg((size_t)nq + 1);
    std::vector<HitQuery> hq((size_t)nq);
    for (int q = 0; q <= nq; q++) seg[(size_t)q] = q * n;            // nq * n < 2^31: the callers cut the batch
    for (int q = 0; q < nq; q++) {
        const auto &info = ctx->queries[(size_t)(q0 + q)];
        hq[(size_t)q].n1 = info.n1;
        hq[(size_t)q].pad_ = 0;
        hq[(size_t)q].ssemaps = want_maps ? ctx->d_ssemaps + info.ssemap_off : nullptr;
    }
    HIP_TRY(hipMemcpyAsync(ctx->d_seg, seg.data(), seg.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->d_hitq, hq.data(), hq.size() * sizeof(HitQuery), hipMemcpyHostToDevice, ctx->stream));

    const int32_t *scores = ctx->d_scores + (size_t)q0 * n;
    hipLaunchKernelGGL(pack_keys, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, scores, (long long)total, n, ctx->d_keys);
    HIP_TRY(hipGetLastError());
    size_t temp_bytes = 0;
    if (nq == 1) {
        HIP_TRY(hipcub::DeviceRadixSort::SortKeysDescending(nullptr, temp_bytes, ctx->d_keys, ctx->d_sorted, (int)total, 0, 64, ctx->stream));
    } else {
        HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortKeysDescending(nullptr, temp_bytes, ctx->d_keys, ctx->d_sorted, (int)total, nq,
                                                                    ctx->d_seg, ctx->d_seg + 1, 0, 64, ctx->stream));
    }
    if ((rc = grow(ctx->d_sort_temp, ctx->sort_temp_cap, temp_bytes ? temp_bytes : 1)) != SAT_OK) return rc;
    if (nq == 1) {
        HIP_TRY(hipcub::DeviceRadixSort::SortKeysDescending(ctx->d_sort_temp, temp_bytes, ctx->d_keys, ctx->d_sorted, (int)total, 0, 64, ctx->stream));
    } else {
        HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortKeysDescending(ctx->d_sort_temp, temp_bytes, ctx->d_keys, ctx->d_sorted, (int)total, nq,
                                                                    ctx->d_seg, ctx->d_seg + 1, 0, 64, ctx->stream));
    }
    hipLaunchKernelGGL(finish_hits, dim3((unsigned)((nq * k + 127) / 128)), dim3(128), 0, ctx->stream, ctx->d_sorted, n, k, nq,
                       ctx->d_orders, reinterpret_cast<const HitQuery *>(ctx->d_hitq), ctx->d_gumbel_z, ctx->d_gumbel_p,
                       ctx->d_hits + out_row, want_maps ? ctx->d_hit_maps + out_row * SAT_MAXDIM : nullptr);
    HIP_TRY(hipGetLastError());
    // the host vectors die at return
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SAT_OK;
}

int check_searched(sat_ctx *ctx)
{
    if (!ctx) return sat_fail(SAT_EINVAL, "null context");
    if (ctx->n_entries <= 0 || ctx->queries.empty() || !ctx->d_scores || ctx->searched_nq != ctx->queries.size())
        return sat_fail(SAT_ESTATE, "no search has run since the last database upload / query change");
    return SAT_OK;
}

}  // namespace

extern "C" int sat_topk(sat_ctx *ctx, int query, int k, int32_t *entry_index, int32_t *scores_out)
{
    int rc = check_searched(ctx);
    if (rc != SAT_OK) return rc;
    if (!entry_index || !scores_out || k < 1) return sat_fail(SAT_EINVAL, "bad top-k arguments");
    if (query < 0 || query >= (int)ctx->queries.size()) return sat_fail(SAT_EINVAL, "query %d out of range", query);
    if (k > ctx->n_entries) k = ctx->n_entries;
    if ((rc = select_hits(ctx, query, 1, k, false, 0, (size_t)k)) != SAT_OK) return rc;
    std::vector<sat_hit> rows((size_t)k);
    HIP_TRY(hipMemcpy(rows.data(), ctx->d_hits, rows.size() * sizeof(sat_hit), hipMemcpyDeviceToHost));
    ctx->d2h_bytes += rows.size() * sizeof(sat_hit);
    for (int i = 0; i < k; i++) {
        entry_index[i] = rows[(size_t)i].entry;
        scores_out[i] = rows[(size_t)i].score;
    }
    return k;
}

extern "C" int sat_topk_hits(sat_ctx *ctx, int k, sat_hit *hits, int32_t *ssemaps)
{
    int rc = check_searched(ctx);
    if (rc != SAT_OK) return rc;
    if (!hits || k < 1) return sat_fail(SAT_EINVAL, "bad top-k arguments");
    if (ssemaps && !ctx->searched_lsoln) return sat_fail(SAT_ESTATE, "the last search ran without lsoln");
    if (k > ctx->n_entries) k = ctx->n_entries;
    const int nq = (int)ctx->queries.size();
    // one segmented sort handles up to 2^31 - 1 keys: long query lists over large databases go in chunks
    const int per_chunk = (int)(0x7FFFFFFFll / ctx->n_entries) < 1 ? 1 : (int)(0x7FFFFFFFll / ctx->n_entries);
    for (int q0 = 0; q0 < nq; q0 += per_chunk) {
        const int nqc = nq - q0 < per_chunk ? nq - q0 : per_chunk;
        if ((rc = select_hits(ctx, q0, nqc, k, ssemaps != nullptr, (size_t)q0 * k, (size_t)nq * k)) != SAT_OK) return rc;
    }
    HIP_TRY(hipMemcpy(hits, ctx->d_hits, (size_t)nq * k * sizeof(sat_hit), hipMemcpyDeviceToHost));
    ctx->d2h_bytes += (size_t)nq * k * sizeof(sat_hit);
    if (ssemaps) {
        HIP_TRY(hipMemcpy(ssemaps, ctx->d_hit_maps, (size_t)nq * k * SAT_MAXDIM * sizeof(int32_t), hipMemcpyDeviceToHost));
        ctx->d2h_bytes += (size_t)nq * k * SAT_MAXDIM * sizeof(int32_t);
    }
    return k;
}
