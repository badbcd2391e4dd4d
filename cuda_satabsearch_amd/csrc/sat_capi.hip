// sat_capi.hip - C ABI (include/satabsearch.h) over the gfx950 SA kernel.
//
// Host side of the drop-in boundary: device memory, the packed database store, the
// query buffer, the Metropolis table, size-class dispatch and launches.  Replaces
// the device glue of nvcc_src_current/cudaSaTabsearch.cu (init_rng :258-264,
// copyQueryToConstantMemory :486-558, alloc/upload :896-984, launch/sync/download
// :1036-1087 and :1128-1270).  No CPU search path exists in this library.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "satabsearch.h"
#ifdef SAT_DIAG
#define SAT_DIAG_HOST 1           // diagnostic builds only: the counters' host side (diag/sat_diag.hpp)
#endif
#include "sat_sa_kernel.hpp"
#include "sat_ctx.hpp"
#include "host/sat_gumbel.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace

int sat_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

namespace {

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t err__ = (expr);                                                      \
        if (err__ != hipSuccess)                                                        \
            return fail(err__ == hipErrorOutOfMemory ? SAT_ENOMEM : SAT_EDEVICE,        \
                        "%s failed: %s", #expr, hipGetErrorString(err__));              \
    } while (0)

// db entries are launched in classes of similar order so that every launch sizes its
// LDS for the largest member of the class only
const int kBucketMax[kNumBuckets] = { 16, 32, 48, 64, 80, 96, 111 };
constexpr size_t kLdsLimit = 160 * 1024;

template <typename T> void dev_free(T *&p)
{
    if (p) (void)hipFree(p);
    p = nullptr;
}

}  // namespace

namespace {

int build_metropolis_table(sat_ctx *ctx)
{
    // P[iter][nd] = expf((float)(-nd) / temp_iter), temp_0 = 10, temp *= 0.95f per step
    // (saparams.h:34-37, K.cu:1030, 1166, 1189), computed with the host libm.  A draw
    // u is never below 2^-32 (rocrand_uniform.h:65-68), so entries <= 2^-32 can never
    // accept and each row is cut after its last entry above that bound.
    const int max_nd = 4 * (SAT_MAXDIM - 1);     // |delta| <= 4 per other query SSE
    const float umin = 2.3283064e-10f;
    std::vector<float> tab;
    std::vector<int32_t> rows(2 * SAT_MAXITER);
    volatile float temp = 10.0f;
    for (int it = 0; it < SAT_MAXITER; it++) {
        int last = 0;
        std::vector<float> row(max_nd + 1);
        for (int nd = 0; nd <= max_nd; nd++) {
            volatile float x = (float)(-nd) / temp;
            row[nd] = expf(x);
            if (row[nd] > umin) last = nd;
        }
        rows[2 * it] = (int32_t)tab.size();
        rows[2 * it + 1] = last;
        // stored times 2^32 (exact): the kernel compares with 2^32 * u.  The row is indexed by
        // 1 - delta clamped to [0, last + 2]: a leading 2^33 for every delta > 0 (expf(x > 0) > 1 >= u)
        // and a trailing 0.0 for "can never be accepted"
        tab.push_back(8589934592.0f);
        for (int nd = 0; nd <= last; nd++) tab.push_back(ldexpf(row[nd], 32));
        tab.push_back(0.0f);
        temp = temp * 0.95f;
    }
    HIP_TRY(hipMalloc(&ctx->d_ptab, tab.size() * sizeof(float)));
    HIP_TRY(hipMalloc(&ctx->d_prow, rows.size() * sizeof(int32_t)));
    HIP_TRY(hipMemcpy(ctx->d_ptab, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_prow, rows.data(), rows.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    return SAT_OK;
}

int build_gumbel_tables(sat_ctx *ctx)
{
    // the reference computes z from the norm2 score TRUNCATED TO AN INT (gumbelstats.h:26 vs H.cu:446),
    // so z and p take one value per integer: tabulated here with the host's libm for x = -128 .. 127
    // (|norm2| <= 110), the device's best-k rows look them up (sat_topk.hip)
    double z[256], p[256];
    for (int x = -128; x < 128; x++) {
        z[x + 128] = sat_z_gumbel_trunc((double)x);
        p[x + 128] = sat_pv_gumbel(z[x + 128]);
    }
    HIP_TRY(hipMalloc(&ctx->d_gumbel_z, sizeof z));
    HIP_TRY(hipMalloc(&ctx->d_gumbel_p, sizeof p));
    HIP_TRY(hipMemcpy(ctx->d_gumbel_z, z, sizeof z, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_gumbel_p, p, sizeof p, hipMemcpyHostToDevice));
    return SAT_OK;
}

void free_db(sat_ctx *ctx)
{
    dev_free(ctx->d_orders);
    dev_free(ctx->d_cell_off);
    dev_free(ctx->d_tab);
    dev_free(ctx->d_dist);
    dev_free(ctx->d_ordinal);
    dev_free(ctx->d_lists);
    dev_free(ctx->d_scores);
    dev_free(ctx->d_ssemaps);
    ctx->scores_cap = 0;
    ctx->ssemaps_cap = 0;
    ctx->desc_dirty = true;
    ctx->n_entries = 0;
    ctx->min_rows = 0;
    ctx->searched_nq = 0;
    ctx->h_orders.clear();
}

typedef void (*kernel_fn)(const SatKernelArgs);

// db-side set width and cell layout (satk::cell_layout of the launch's largest entry): one-word sets go with the
// 8-byte cells, two-word sets with either split layout (entries of up to 48 SSEs: full matrix, above: triangle),
// four-word sets with the triangle
template <int N1P, bool QLDS, int OPT, int WPL> kernel_fn pick_m2w(int m2w, int cells)
{
    if (m2w == 1) return sat_sa_kernel<N1P, 1, QLDS, OPT, WPL, SAT_CELLS_FULL8>;
    if (m2w == 2) return cells == SAT_CELLS_FULL5 ? sat_sa_kernel<N1P, 2, QLDS, OPT, WPL, SAT_CELLS_FULL5>
                                                   : sat_sa_kernel<N1P, 2, QLDS, OPT, WPL, SAT_CELLS_TRI5>;
    return sat_sa_kernel<N1P, 4, QLDS, OPT, WPL, SAT_CELLS_TRI5>;
}

// opt >= 0: an instantiation with the options as compile-time facts (bit 0 LORDER, bit 1 LSOLN; one
// lane per chain, compaction tables exactly when LORDER); with LORDER also `wpl`, the words per
// lane of the compacted rounds when every query of the launch has the same, else 0 (see the
// kernel's OPT and WPL parameters).  These exist for the default placement of the query cells only (LDS for the
// 16 class, L1/L2 for the others) and for the wpl values a class can have
// (satk::compaction_shape); anything else runs the general instantiation.
template <int N1P, int OPT> kernel_fn pick_wpl(int m2w, int cells, int wpl)
{
    constexpr bool kQ = N1P < 32;
    if constexpr ((OPT & 1) == 0) return pick_m2w<N1P, kQ, OPT, 0>(m2w, cells);     // no compaction: wpl unused
    else {
        if (wpl == 4) return pick_m2w<N1P, kQ, OPT, 4>(m2w, cells);
        if constexpr (N1P <= 64)
            if (wpl == 3) return pick_m2w<N1P, kQ, OPT, 3>(m2w, cells);
        if constexpr (N1P == 16) {
            if (wpl == 2) return pick_m2w<N1P, kQ, OPT, 2>(m2w, cells);
            if (wpl == 1) return pick_m2w<N1P, kQ, OPT, 1>(m2w, cells);
        }
        return pick_m2w<N1P, kQ, OPT, 0>(m2w, cells);       // queries of different shapes: wpl read per query
    }
}

// several lanes per chain with the options as compile-time facts: only for the largest entries (M2W = 4),
// words per lane read per query
template <int N1P, bool QLDS> kernel_fn pick_lpc(int opt)
{
    switch (opt) {
    case 4: return sat_sa_kernel<N1P, 4, QLDS, 4, 0, SAT_CELLS_TRI5>;
    case 5: return sat_sa_kernel<N1P, 4, QLDS, 5, 0, SAT_CELLS_TRI5>;
    case 6: return sat_sa_kernel<N1P, 4, QLDS, 6, 0, SAT_CELLS_TRI5>;
    case 7: return sat_sa_kernel<N1P, 4, QLDS, 7, 0, SAT_CELLS_TRI5>;
    case 8: return sat_sa_kernel<N1P, 4, QLDS, 8, 0, SAT_CELLS_TRI5>;
    case 9: return sat_sa_kernel<N1P, 4, QLDS, 9, 0, SAT_CELLS_TRI5>;
    case 10: return sat_sa_kernel<N1P, 4, QLDS, 10, 0, SAT_CELLS_TRI5>;
    default: return sat_sa_kernel<N1P, 4, QLDS, 11, 0, SAT_CELLS_TRI5>;
    }
}

template <int N1P> kernel_fn pick_n1p(int m2w, int cells, bool qlds, int opt, int wpl)
{
    constexpr bool kQ = N1P < 32;
    kernel_fn fn = nullptr;
    if (opt >= 4 && qlds == kQ && m2w == 4) return pick_lpc<N1P, kQ>(opt);
    if (opt >= 0 && opt < 4 && qlds == kQ) {
        switch (opt) {
        case 0: fn = pick_wpl<N1P, 0>(m2w, cells, wpl); break;
        case 1: fn = pick_wpl<N1P, 1>(m2w, cells, wpl); break;
        case 2: fn = pick_wpl<N1P, 2>(m2w, cells, wpl); break;
        default: fn = pick_wpl<N1P, 3>(m2w, cells, wpl); break;
        }
    }
    if (fn) return fn;
    return qlds ? pick_m2w<N1P, true, -1, 0>(m2w, cells) : pick_m2w<N1P, false, -1, 0>(m2w, cells);
}

kernel_fn pick_kernel(int n1p, int m2w, int cells, bool qlds, int opt, int wpl)
{
    switch (n1p) {
    case 16: return pick_n1p<16>(m2w, cells, qlds, opt, wpl);
    case 32: return pick_n1p<32>(m2w, cells, qlds, opt, wpl);
    case 64: return pick_n1p<64>(m2w, cells, qlds, opt, wpl);
    default: return pick_n1p<112>(m2w, cells, qlds, opt, wpl);
    }
}

const int kClassN1P[4] = { 16, 32, 64, 112 };

// (re)build the device query descriptors: pointers into the query blob and into the result
// buffers, grouped by size class
int refresh_descriptors(sat_ctx *ctx, bool lsoln, hipStream_t stream)
{
    const size_t nq = ctx->queries.size();
    const size_t rows = (size_t)(ctx->n_entries > ctx->min_rows ? ctx->n_entries : ctx->min_rows);    // capacity only
    const size_t need_scores = nq * rows;
    if (need_scores > ctx->scores_cap) {
        dev_free(ctx->d_scores);
        HIP_TRY(hipMalloc(&ctx->d_scores, need_scores * sizeof(int32_t)));
        ctx->scores_cap = need_scores;
        ctx->desc_dirty = true;
    }
    if (lsoln) {
        size_t need = 0, n1sum = 0;
        for (auto &q : ctx->queries) {
            q.ssemap_off = need;
            need += (size_t)ctx->n_entries * q.n1;
            n1sum += (size_t)q.n1;
        }
        if (rows * n1sum > need) need = rows * n1sum;
        if (need > ctx->ssemaps_cap) {
            dev_free(ctx->d_ssemaps);
            HIP_TRY(hipMalloc(&ctx->d_ssemaps, need));
            ctx->ssemaps_cap = need;
            ctx->desc_dirty = true;
        }
        if (!ctx->desc_lsoln) ctx->desc_dirty = true;
    }
    if (!ctx->desc_dirty) return SAT_OK;

    std::vector<SatQuery> desc;
    desc.reserve(nq);
    for (int c = 0; c < 4; c++) {
        ctx->class_begin[c] = (int)desc.size();
        ctx->class_n1max[c] = 0;
        ctx->class_wpl[c] = -1;                       // -1: no query yet, 0: mixed
        for (size_t qi = 0; qi < nq; qi++) {
            const auto &q = ctx->queries[qi];
            if (q.n1p != kClassN1P[c]) continue;
            const size_t groups = (size_t)q.n1p / 4 * q.n1p;
            SatQuery d;
            d.qdist = reinterpret_cast<const float4 *>(ctx->d_qblob + q.blob_off);
            d.qcode = reinterpret_cast<const uint32_t *>(ctx->d_qblob + q.blob_off + groups * 16);
            d.qtypes = ctx->d_qblob + q.blob_off + groups * 20;
            d.qpair = reinterpret_cast<const uint2 *>(ctx->d_qblob + q.blob_off + ((groups * 20 + (size_t)q.n1p + 15) & ~(size_t)15));
            d.n1 = q.n1;
            d.pad_ = 0;
            d.seed_q = ctx->seed + ((uint64_t)q.ordinal << 32);
            d.scores = ctx->d_scores + qi * (size_t)ctx->n_entries;
            d.ssemaps = lsoln ? ctx->d_ssemaps + q.ssemap_off : nullptr;
            desc.push_back(d);
            if (q.n1 > ctx->class_n1max[c]) ctx->class_n1max[c] = q.n1;
            int lpi, wpl;
            satk::compaction_shape((q.n1 + 3) >> 2, lpi, wpl);
            ctx->class_wpl[c] = ctx->class_wpl[c] < 0 ? wpl : (ctx->class_wpl[c] == wpl ? wpl : 0);
        }
        if (ctx->class_wpl[c] < 0) ctx->class_wpl[c] = 0;
    }
    ctx->class_begin[4] = (int)desc.size();
    // ordered after earlier launches on the stream; the host vector dies at return, so wait
    HIP_TRY(hipMemcpyAsync(ctx->d_qdesc, desc.data(), desc.size() * sizeof(SatQuery), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    ctx->desc_dirty = false;
    ctx->desc_lsoln = lsoln;
    return SAT_OK;
}

// Entries per workgroup.  A CU hands out its LDS in 128 granules of 1280 bytes (measured,
// scripts/exp/lds_probe.hip: 128-thread workgroups drop from 12 to 11 to 10 per CU at 12 800 and 14 080
// bytes, 384-thread ones from 4 to 3 at 40 960), so a workgroup of one entry wastes up to a granule plus
// what is left over at the end of the CU.  k entries side by side round up once: the bench entry's
// 13 320 bytes fit 11 times alone (11 granules each) and 6 x 2 times in pairs (21 granules a pair).
// Picks the smallest k with the most resident entries, the register file's wave limit included.  Only
// workgroups of a multiple of 4 waves and at most 512 threads are considered: measured on the bench,
// 6-wave workgroups do not spread evenly over the 4 SIMDs (8.5 M scorings/s against 10.6 M), and 12-wave
// ones lose to their own start-up and drain phases what the extra residency gains (10.4 M).
int resident_by_lds(size_t bytes) { return (int)(128 / ((bytes + 1279) / 1280)); }

int pick_epw(kernel_fn fn, int threads, size_t lds_stride)
{
    int best = 1, best_entries = 0;
    for (int k = 1; k * threads <= 512 && (size_t)k * lds_stride <= kLdsLimit; k++) {
        if (k > 1 && (k * threads / 64) % 4 != 0) continue;
        int by_regs = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&by_regs, reinterpret_cast<const void *>(fn), k * threads, 0) != hipSuccess) {
            (void)hipGetLastError();
            return 1;
        }
        const int by_lds = resident_by_lds((size_t)k * lds_stride);
        const int entries = (by_regs < by_lds ? by_regs : by_lds) * k;
        if (entries > best_entries) { best_entries = entries; best = k; }
    }
    return best;
}

// The entries a set of launches covers: indices into the resident shard grouped by order bucket.  A search
// covers the whole shard (the context's lists); the overlapped upload (sat_db_upload_search) searches the
// shard piece by piece, each piece with lists of its own.
struct ListView {
    const int32_t *d_list;       // device array the `begin` offsets index
    const int *begin;            // [kNumBuckets + 1]
    const int *n2max;            // [kNumBuckets] largest order per bucket, 0 = empty
    int n;                       // entries covered = begin[kNumBuckets] - begin[0]
};

int launch_search(sat_ctx *ctx, int lorder, int lsoln, int maxstart, hipStream_t stream, const ListView *piece = nullptr)
{
    if (!ctx) return fail(SAT_EINVAL, "null context");
    if (ctx->n_entries <= 0) return fail(SAT_ESTATE, "no database uploaded");
    const ListView whole = { ctx->d_lists, ctx->bucket_begin, ctx->bucket_n2max, ctx->n_entries };
    const ListView &view = piece ? *piece : whole;
    if (ctx->queries.empty()) return fail(SAT_ESTATE, "no query set");
    if (maxstart < 1) return fail(SAT_EINVAL, "maxstart must be >= 1 (got %d)", maxstart);
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = refresh_descriptors(ctx, lsoln != 0, stream);
    if (rc != SAT_OK) return rc;

    SatKernelArgs a;
    a.orders = ctx->d_orders;
    a.cell_off = ctx->d_cell_off;
    a.tab_tri = ctx->d_tab;
    a.dist_tri = ctx->d_dist;
    a.ordinal = ctx->d_ordinal;
    a.lorder = lorder ? 1 : 0;
    a.lsoln = lsoln ? 1 : 0;
    a.maxstart = maxstart;
    a.ptab = ctx->d_ptab;
    a.prow = ctx->d_prow;
    a.bmap_slabs = nullptr;
    a.bmap_slab_words = 0;
#ifdef SAT_DIAG
    HIP_TRY(satdiag::begin(stream, a.diag));
#endif

    struct Planned {
        kernel_fn fn; SatKernelArgs args; int count, nqc, threads, n2max, max_entries, epw; size_t lds, slab_words;
        int n1p, m2w, qlds, opt, wpl, cells;     // the instantiation's template arguments (sat_last_launch_info)
    };
    std::vector<Planned> plan;
    for (int c = 0; c < 4; c++) {
        const int nqc = ctx->class_begin[c + 1] - ctx->class_begin[c];
        if (nqc == 0) continue;
        const int n1p = kClassN1P[c], n1max = ctx->class_n1max[c];
        a.queries = ctx->d_qdesc + ctx->class_begin[c];
        // A small problem cannot fill the GPU: its run time is the latency of one workgroup per
        // launch, so all order buckets go into ONE launch sized for the largest entry instead of
        // one launch per bucket queued behind each other.
        const bool one_launch = (long long)view.n * nqc <= 4096;
        int overall_n2max = 0;
        for (int b = 0; b < kNumBuckets; b++)
            if (view.n2max[b] > overall_n2max) overall_n2max = view.n2max[b];
        for (int b = 0; b < kNumBuckets; b++) {
            int count = view.begin[b + 1] - view.begin[b];
            int n2max = view.n2max[b];
            if (one_launch) {
                if (b > 0) break;
                count = view.n;
                n2max = overall_n2max;
            }
            if (count == 0) continue;
            const int m2w = n2max <= 32 ? 1 : (n2max <= 64 ? 2 : 4);
            const int cells = satk::cell_layout(n2max);           // (lds_bytes sizes the workgroup for the same layout)

            // chains: one per restart up to 256; shrink until the workgroup fits the LDS.
            // query cells: through L1/L2 for 32-SSE-class queries and up (frees 8+ KB of LDS per
            // workgroup: more resident waves), in LDS for the small class
            int chains = (maxstart + 63) / 64 * 64;
            if (chains > 256) chains = 256;
            if (ctx->tune.chains >= 64 && ctx->tune.chains < chains) chains = ctx->tune.chains / 64 * 64;
            // work compaction needs sparse maps: with LORDER = F almost every step proposes a real
            // new image, the static loops win and the tables would only cost LDS
            bool compact = lorder != 0;
            if (ctx->tune.compact >= 0) compact = ctx->tune.compact != 0;
            bool qlds = n1p < 32;
            if (ctx->tune.qlds >= 0) qlds = ctx->tune.qlds != 0 || n1p < 32;
            size_t lds = 0;
            for (;;) {
                lds = satk::lds_bytes(n1max, n1p, n2max, chains, chains, lsoln != 0, qlds, compact);
                if (lds <= kLdsLimit) break;
                if (chains > 64) { chains -= 64; continue; }
                if (qlds) {                                    // query cells stay in L1/L2 instead
                    qlds = false;
                    chains = (maxstart + 63) / 64 * 64;
                    if (chains > 256) chains = 256;
                    continue;
                }
                return fail(SAT_EINVAL, "workgroup does not fit in LDS (n1=%d n2=%d)", n1max, n2max);
            }
            // lanes per chain: when LDS leaves fewer than 2 waves per SIMD, let 2 or 4 adjacent lanes
            // share a chain (same cells in LDS, 2-4x the waves; they split the pair loops).  Measured:
            // the smallest sharing that reaches 8 waves per CU wins (one lane per chain also runs the
            // option-specialised kernels); beyond that, sharing only adds redundant bookkeeping.
            int lpc_shift = 0;
            for (int l = 0; l <= 2; l++) {
                if ((chains << l) > 1024 || (l > 0 && n1max <= (8 << (l - 1)))) break;
                const size_t lds_l = satk::lds_bytes(n1max, n1p, n2max, chains, chains << l, lsoln != 0, qlds, compact);
                if (lds_l > kLdsLimit) break;
                lpc_shift = l;
                // (target: 8 resident waves per CU; 12 for the 101-SSE query class, whose steps are the longest
                // dependent chains - measured with the triangle cells: configs[4] 2.31 -> 2.45 M scorings/s, the
                // 101-SSE probe 2.48 -> 2.65 M, while 96-SSE entries under a 32-SSE query lose 5 % at 12)
                const int want_waves = ctx->tune.lpc_waves > 0 ? ctx->tune.lpc_waves : (n1p == 112 ? 12 : 8);
                if (resident_by_lds(lds_l) * ((chains << l) / 64) >= want_waves) break;
            }
            if (ctx->tune.lpc >= 0 && ctx->tune.lpc <= 2 && (chains << ctx->tune.lpc) <= 1024) lpc_shift = ctx->tune.lpc;
            // the per-wave tables grow with the lanes: re-size, backing off if that no longer fits
            for (;; lpc_shift--) {
                lds = satk::lds_bytes(n1max, n1p, n2max, chains, chains << lpc_shift, lsoln != 0, qlds, compact);
                if (lds <= kLdsLimit || lpc_shift == 0) break;
            }
            const int threads = chains << lpc_shift;
            // experiment knob: extra (unused) LDS bytes per workgroup, to lower the occupancy
            if (ctx->tune.lds_pad && lds + ctx->tune.lds_pad <= kLdsLimit) lds += ctx->tune.lds_pad;
            a.lpc_shift = lpc_shift;
            a.compact = compact ? 1 : 0;
            // option-specialised instantiation when the layout is the default one for these options
            const bool special = (lpc_shift == 0 || m2w == 4) && compact == (lorder != 0) && !ctx->tune.general;
            const int opt = special ? (lorder ? 1 : 0) | (lsoln ? 2 : 0) | (lpc_shift << 2) : -1;
            kernel_fn fn = pick_kernel(n1p, m2w, cells, qlds, opt, ctx->class_wpl[c]);
            if (!fn) return fail(SAT_EDEVICE, "no kernel variant for n1p=%d m2w=%d", n1p, m2w);
            if (ctx->lds_attr_done.insert(reinterpret_cast<const void *>(fn)).second)
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fn),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
            a.entry_list = view.d_list + (one_launch ? view.begin[0] : view.begin[b]);
            // entries per workgroup (see pick_epw); small launches keep one, for the most workgroups
            const size_t lds_stride = (lds + 15) & ~(size_t)15;
            int epw = 1;
            if ((long long)count * nqc >= 8192) {
                const auto key = std::make_tuple(reinterpret_cast<const void *>(fn), threads, lds_stride);
                auto it = ctx->epw_choice.find(key);
                if (it == ctx->epw_choice.end()) it = ctx->epw_choice.emplace(key, pick_epw(fn, threads, lds_stride)).first;
                epw = it->second;
            }
            if (ctx->tune.epw >= 1 && (size_t)ctx->tune.epw * lds_stride <= kLdsLimit && ctx->tune.epw * threads <= 1024)
                epw = ctx->tune.epw;
            a.epw = epw;
            a.tpe = threads;
            a.lds_stride = (uint32_t)lds_stride;
            Planned pl;
            pl.epw = epw;
            pl.fn = fn;
            pl.args = a;
            pl.count = count;
            pl.nqc = nqc;
            pl.threads = threads;
            pl.lds = lds;
            pl.n2max = n2max;
            pl.slab_words = lsoln ? (size_t)((n1max + 3) / 4) * chains : 0;
            pl.n1p = n1p;
            pl.m2w = m2w;
            pl.cells = cells;
            pl.qlds = qlds ? 1 : 0;
            pl.opt = opt;
            // the words-per-lane argument as pick_kernel resolves it (0 = read per query)
            pl.wpl = (opt >= 0 && opt < 4 && (opt & 1) && qlds == (n1p < 32)) ? ctx->class_wpl[c] : 0;
            if (pl.wpl && !((pl.wpl == 4) || (pl.wpl == 3 && n1p <= 64) || (n1p == 16))) pl.wpl = 0;
            if (opt < 0 || qlds != (n1p < 32)) pl.opt = -1;
            plan.push_back(pl);
        }
    }

    // The launches of one search (order buckets x query classes) are independent.  Queued on ONE stream
    // each would wait for the last workgroups of the one before it (a tail of half-empty CUs per
    // launch); forked over side streams they run concurrently and the next bucket's workgroups fill
    // the tail.  Largest entries first: their workgroups run longest.  One launch needs no fork.
    std::stable_sort(plan.begin(), plan.end(), [](const Planned &x, const Planned &y) { return x.n2max > y.n2max; });
    const bool fork = plan.size() > 1 && ctx->tune.streams != 0 && ctx->side_stream[0] != nullptr;
    const int nlanes = fork ? (int)(plan.size() < (size_t)kNumBuckets ? plan.size() : (size_t)kNumBuckets) : 1;
    if (fork) HIP_TRY(hipEventRecord(ctx->ev_fork, stream));
    // LSOLN: every workgroup of a launch owns a slab of best maps in global memory; launches are cut so
    // that the slabs of all concurrent launches stay under 1 GiB together (a lane of launches reuses
    // its region launch after launch).  grid.y is limited to 65535: very long query lists are split too.
    const size_t lane_budget_words = ((size_t)1 << 30) / 4 / (size_t)nlanes;
    if (lsoln) {
        size_t need_total = 0;
        for (size_t i = 0; i < plan.size(); i++) {
            Planned &pl = plan[i];
            size_t fit = lane_budget_words / pl.slab_words;           // workgroups per launch
            if (fit < 1) fit = 1;
            const int qn_cap = pl.nqc < 65535 ? pl.nqc : 65535;
            pl.max_entries = (int)(fit / (size_t)qn_cap);
            if (pl.max_entries < 1) pl.max_entries = 1;
            if (pl.max_entries > pl.count) pl.max_entries = pl.count;
            // (the spare slots of a launch's last workgroup have slabs too)
            const size_t need = pl.slab_words * (size_t)(pl.max_entries + pl.epw - 1) * (size_t)qn_cap;
            if (need > need_total) need_total = need;
        }
        need_total *= (size_t)nlanes;                                  // one region per lane of launches
        if (need_total > ctx->bmap_slabs_cap) {
            HIP_TRY(hipStreamSynchronize(stream));
            dev_free(ctx->d_bmap_slabs);
            HIP_TRY(hipMalloc(&ctx->d_bmap_slabs, need_total * sizeof(uint32_t)));
            ctx->bmap_slabs_cap = need_total;
        }
    }
    const size_t lane_region_words = lsoln ? ctx->bmap_slabs_cap / (size_t)nlanes : 0;
    for (size_t i = 0; i < plan.size(); i++) {
        const Planned &pl = plan[i];
        const int lane = fork ? (int)(i % (size_t)nlanes) : 0;
        hipStream_t s = fork ? ctx->side_stream[lane] : stream;
        if (fork && i < (size_t)nlanes) HIP_TRY(hipStreamWaitEvent(s, ctx->ev_fork, 0));
        const int max_entries = lsoln ? pl.max_entries : pl.count;
        for (int q0 = 0; q0 < pl.nqc; q0 += 65535) {
            const int qn = pl.nqc - q0 < 65535 ? pl.nqc - q0 : 65535;
            for (int e0 = 0; e0 < pl.count; e0 += max_entries) {
                const int en = pl.count - e0 < max_entries ? pl.count - e0 : max_entries;
                SatKernelArgs part = pl.args;
                part.queries = pl.args.queries + q0;
                part.entry_list = pl.args.entry_list + e0;
                if (lsoln) {
                    part.bmap_slabs = ctx->d_bmap_slabs + (size_t)lane * lane_region_words;
                    part.bmap_slab_words = (uint32_t)pl.slab_words;
                }
                part.n_list = en;
                const size_t lds_launch = pl.epw > 1 ? (size_t)pl.epw * pl.args.lds_stride : pl.lds;
                hipLaunchKernelGGL(pl.fn, dim3((en + pl.epw - 1) / pl.epw, qn), dim3(pl.threads * pl.epw), lds_launch, s, part);
                HIP_TRY(hipGetLastError());
            }
        }
    }
    if (fork)
        for (int lane = 0; lane < nlanes; lane++) {
            HIP_TRY(hipEventRecord(ctx->ev_join[lane], ctx->side_stream[lane]));
            HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_join[lane], 0));
        }
    ctx->searched_nq = ctx->queries.size();
    ctx->searched_lsoln = lsoln != 0;
    ctx->last_launch_info.clear();
    for (size_t i = 0; i < plan.size(); i++) {
        char buf[160];
        snprintf(buf, sizeof buf, "%ssat_sa_kernel<%d, %d, %s, %d, %d, %d> grid %d x %d block %d x %d lds %zu", i ? "; " : "", plan[i].n1p,
                 plan[i].m2w, plan[i].qlds ? "true" : "false", plan[i].opt, plan[i].wpl, plan[i].cells,
                 (plan[i].count + plan[i].epw - 1) / plan[i].epw, plan[i].nqc, plan[i].epw, plan[i].threads, plan[i].lds);
        ctx->last_launch_info += buf;
    }
#ifdef SAT_DIAG
    HIP_TRY(satdiag::end(stream));
#endif
    return SAT_OK;
}

// Upload validation: one wave per db entry reads the entry's packed triangle where the search will
// read it and flags cells outside the kernel's domain; the lowest flagged entry index survives.
// (entries e_begin .. e_end - 1: the overlapped upload checks the shard piece by piece)
__global__ void __launch_bounds__(256) validate_cells(int e_begin, int e_end, const int32_t *orders, const int64_t *cell_off,
                                                      const uint8_t *tab, const float *dist, int32_t *first_bad)
{
    const int e = e_begin + blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= e_end) return;
    const int n = orders[e];
    const int64_t base = cell_off[e];
    const int cells = n * (n + 1) / 2;
    bool bad = false;
    for (int c = lane; c < cells; c += 64) {
        // row i of cell c: the largest i with i (i + 1) / 2 <= c; diagonal cells hold the SSE type
        int i = (int)((sqrtf(8.0f * (float)c + 1.0f) - 1.0f) * 0.5f);
        while ((i + 1) * (i + 2) / 2 <= c) i++;
        while (i * (i + 1) / 2 > c) i--;
        const bool diagonal = c == i * (i + 1) / 2 + i;
        const uint8_t t = tab[base + c];
        if (diagonal) {
            bad |= t > 3;
        } else {
            const float ad = fabsf(dist[base + c]);
            bad |= (t & 0x88u) != 0 || (ad >= 1.0e29f && ad <= 3.4028234e38f);      // finite and out of range
        }
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) atomicMin(first_bad, e);
}

}  // namespace

extern "C" {

const char *sat_last_error(void) { return g_err; }

int sat_abi_version(void) { return SAT_ABI_VERSION; }

int sat_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

sat_ctx *sat_ctx_create(int device, uint64_t seed)
{
    int n = sat_device_count();
    if (n <= 0) {
        fail(SAT_ENODEVICE, "no HIP device available (this library has no CPU path)");
        return nullptr;
    }
    if (device < 0 || device >= n) {
        fail(SAT_ENODEVICE, "device %d out of range (0..%d)", device, n - 1);
        return nullptr;
    }
    sat_ctx *ctx = new (std::nothrow) sat_ctx();
    if (!ctx) {
        fail(SAT_ENOMEM, "out of host memory");
        return nullptr;
    }
    ctx->device = device;
    ctx->seed = seed;
    auto init = [&]() -> int {
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
        ctx->stream = ctx->own_stream;
        HIP_TRY(hipEventCreate(&ctx->ev0));
        HIP_TRY(hipEventCreate(&ctx->ev1));
        // launch-heuristic overrides: read once here, never on the search path
        auto env_int = [](const char *name, int dflt) { const char *v = getenv(name); return v && *v ? atoi(v) : dflt; };
        ctx->tune.compact = env_int("SAT_EXP_COMPACT", -1);
        ctx->tune.qlds = env_int("SAT_EXP_QLDS", -1);
        ctx->tune.lpc = env_int("SAT_EXP_LPC", -1);
        ctx->tune.general = env_int("SAT_EXP_GENERAL", 0);
        ctx->tune.streams = env_int("SAT_EXP_STREAMS", -1);
        ctx->tune.upload_threads = env_int("SAT_EXP_UPLOAD_THREADS", 0);
        ctx->tune.upload_timing = env_int("SAT_EXP_UPLOAD_TIMING", 0);
        ctx->tune.upload_pieces = env_int("SAT_EXP_UPLOAD_PIECES", 0);
        ctx->tune.epw = env_int("SAT_EXP_EPW", 0);
        ctx->tune.lpc_waves = env_int("SAT_EXP_LPC_WAVES", 0);
        ctx->tune.chains = env_int("SAT_EXP_CHAINS", 0);
        const int pad = env_int("SAT_EXP_LDS_PAD", 0);
        ctx->tune.lds_pad = pad > 0 ? (size_t)pad : 0;
        if (ctx->tune.streams != 0) {
            for (int b = 0; b < kNumBuckets; b++) {
                HIP_TRY(hipStreamCreateWithFlags(&ctx->side_stream[b], hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&ctx->ev_join[b], hipEventDisableTiming));
            }
            HIP_TRY(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        }
        const int rc_tab = build_metropolis_table(ctx);
        if (rc_tab != SAT_OK) return rc_tab;
        // load the library's code object now (an empty launch of its smallest kernel): the ~5 ms the
        // first launch of a process pays for it belong to context creation, not to the first upload
        hipLaunchKernelGGL(validate_cells, dim3(1), dim3(256), 0, ctx->stream, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        return build_gumbel_tables(ctx);
    };
    if (init() != SAT_OK) {
        sat_ctx_destroy(ctx);
        return nullptr;
    }
    return ctx;
}

void sat_ctx_destroy(sat_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    free_db(ctx);
    dev_free(ctx->d_qblob);
    dev_free(ctx->d_qdesc);
    dev_free(ctx->d_bmap_slabs);
    dev_free(ctx->d_ptab);
    dev_free(ctx->d_prow);
    dev_free(ctx->d_keys);
    dev_free(ctx->d_sorted);
    dev_free(ctx->d_sort_temp);
    dev_free(ctx->d_hitq);
    dev_free(ctx->d_seg);
    dev_free(ctx->d_hits);
    dev_free(ctx->d_hit_maps);
    dev_free(ctx->d_gumbel_z);
    dev_free(ctx->d_gumbel_p);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    for (int b = 0; b < kNumBuckets; b++) {
        if (ctx->ev_join[b]) (void)hipEventDestroy(ctx->ev_join[b]);
        if (ctx->side_stream[b]) (void)hipStreamDestroy(ctx->side_stream[b]);
    }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

// What sat_db_upload_search asks of the upload: the first search of the current query batch, queued piece
// by piece behind the copies.
struct FirstSearch { int lorder, lsoln, maxstart; };

// Counting sort of entries e_begin .. e_end - 1 by order into `out` (appended at position `pos`): bucket after
// bucket, inside a bucket the LARGEST entries first (file order among equals) - workgroups are dispatched in
// list order and a larger entry runs longer, so a launch ends on its cheapest workgroups instead of its dearest
// (real databases are sorted ascending).  begin[kNumBuckets + 1] / n2max[kNumBuckets] describe the result.
// (two passes over the entries: seven filtered passes and a stable sort per bucket took 3 ms of an 11 ms
// upload of the bench shard)
static void bucket_lists(const int32_t *orders, int e_begin, int e_end, int32_t *out, int pos, int *begin, int *n2max)
{
    int count[SAT_MAXDIM + 1] = { 0 }, start[SAT_MAXDIM + 1] = { 0 };
    for (int e = e_begin; e < e_end; e++) count[orders[e]]++;
    for (int b = 0; b < kNumBuckets; b++) {
        begin[b] = pos;
        n2max[b] = 0;
        const int lo = b == 0 ? 0 : kBucketMax[b - 1];
        for (int n = kBucketMax[b] < SAT_MAXDIM ? kBucketMax[b] : SAT_MAXDIM; n > lo; n--) {
            start[n] = pos;
            pos += count[n];
            if (count[n] && n2max[b] == 0) n2max[b] = n;
        }
    }
    begin[kNumBuckets] = pos;
    for (int e = e_begin; e < e_end; e++) out[(size_t)start[orders[e]]++] = e;
}

static int upload_impl(sat_ctx *ctx, int n_entries, const int32_t *orders,
                       const int64_t *cell_off, const uint8_t *tab_tri,
                       const float *dist_tri, const int64_t *db_ordinal, const FirstSearch *first)
{
    if (!ctx) return fail(SAT_EINVAL, "null context");
    if (n_entries <= 0 || !orders || !cell_off || !tab_tri || !dist_tri)
        return fail(SAT_EINVAL, "empty database or null array");
    if (first) {
        if (ctx->queries.empty()) return fail(SAT_ESTATE, "no query set");
        if (first->maxstart < 1) return fail(SAT_EINVAL, "maxstart must be >= 1 (got %d)", first->maxstart);
    }
    // header pass on the host (orders, offsets, ordinals: a few bytes per entry).  The CELLS - every
    // code byte and distance, 331 MB for the bench shard - are checked on the GPU after the copy, at
    // HBM speed (validate_cells): a host scan of them cost as much as the copy itself.
    int64_t cells_end = 0;
    bool ascending = true;                     // entry e + 1 starts at or after the end of entry e
    for (int e = 0; e < n_entries; e++) {
        const int n = orders[e];
        if (n < 1 || n > SAT_MAXDIM)
            return fail(SAT_EINVAL, "entry %d: order %d outside 1..%d", e, n, SAT_MAXDIM);
        if (cell_off[e] < 0) return fail(SAT_EINVAL, "entry %d: negative cell offset", e);
        if (cell_off[e] < cells_end) ascending = false;
        int64_t end = cell_off[e] + (int64_t)n * (n + 1) / 2;
        if (end > cells_end) cells_end = end;
        if (db_ordinal && (db_ordinal[e] < 0 || db_ordinal[e] > 0xFFFFFFFFll))
            return fail(SAT_EINVAL, "entry %d: db ordinal out of range", e);
    }
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    free_db(ctx);
    const bool timing = ctx->tune.upload_timing != 0;
    auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_mark = now_ms();
    auto lap = [&](const char *what) {
        if (timing) { const double t = now_ms(); fprintf(stderr, "upload: %-18s %7.3f ms\n", what, t - t_mark); t_mark = t; }
    };

    const size_t dist_bytes = (size_t)cells_end * sizeof(float), tab_bytes = (size_t)cells_end;
    // Pieces: with a first search to overlap, the shard goes up in `npieces` runs of whole entries of about
    // equal cell count, and every piece is checked and searched as soon as it has landed - the GPU works on
    // piece c while the host threads copy piece c + 1 (the copies are synchronous calls out of the caller's
    // pageable memory; the kernels run on the context's non-blocking stream).  Needs entries laid out in
    // ascending order (a piece is then one contiguous cell range); small shards go up in one piece.
    int npieces = 1;
    if (first && ascending) {
        // at least 24 MB of distances per piece (each host thread's slice of it is then still a copy of a
        // useful size), at most 8: measured on the 331 MB bench shard, 19.4 ms for upload-then-search,
        // 16.3 / 14.6 / 14.2 / 15.2 / 16.8 ms overlapped in 2 / 4 / 8 / 12 / 16 pieces
        const size_t by_size = dist_bytes / ((size_t)24 << 20);
        npieces = ctx->tune.upload_pieces > 0 ? ctx->tune.upload_pieces : (int)(by_size < 8 ? by_size : 8);
        if (npieces > n_entries) npieces = n_entries;
        if (npieces < 1) npieces = 1;
    }
    std::vector<int> piece_e((size_t)npieces + 1, n_entries);        // piece c = entries piece_e[c] .. piece_e[c+1]-1
    piece_e[0] = 0;
    for (int c = 1, e = 0; c < npieces; c++) {
        const int64_t target = cells_end * c / npieces;
        while (e < n_entries && cell_off[e] < target) e++;
        piece_e[(size_t)c] = e > piece_e[(size_t)c - 1] ? e : piece_e[(size_t)c - 1];
    }
    auto piece_cell = [&](int c) -> int64_t { return c >= npieces || piece_e[(size_t)c] >= n_entries ? cells_end : (c == 0 ? 0 : cell_off[piece_e[(size_t)c]]); };

    // bucket lists of the whole shard (every later search) and, behind them, of each piece
    std::vector<int32_t> lists((size_t)n_entries * (npieces > 1 ? 2 : 1));
    bucket_lists(orders, 0, n_entries, lists.data(), 0, ctx->bucket_begin, ctx->bucket_n2max);
    std::vector<int> piece_begin((size_t)npieces * (kNumBuckets + 1)), piece_n2max((size_t)npieces * kNumBuckets);
    if (npieces > 1) {
        int pos = n_entries;
        for (int c = 0; c < npieces; c++) {
            bucket_lists(orders, piece_e[(size_t)c], piece_e[(size_t)c + 1], lists.data(), pos,
                         &piece_begin[(size_t)c * (kNumBuckets + 1)], &piece_n2max[(size_t)c * kNumBuckets]);
            pos += piece_e[(size_t)c + 1] - piece_e[(size_t)c];
        }
    }

    std::vector<uint32_t> ord(n_entries);
    for (int e = 0; e < n_entries; e++) ord[e] = db_ordinal ? (uint32_t)db_ordinal[e] : (uint32_t)e;

    lap("host lists");
    int32_t *d_bad = nullptr;
    const int32_t none = 0x7FFFFFFF;
    // any failure below leaves the context without a database
    auto body = [&]() -> int {
        HIP_TRY(hipMalloc(&ctx->d_orders, (size_t)n_entries * sizeof(int32_t)));
        HIP_TRY(hipMalloc(&ctx->d_cell_off, (size_t)n_entries * sizeof(int64_t)));
        HIP_TRY(hipMalloc(&ctx->d_ordinal, (size_t)n_entries * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&ctx->d_lists, lists.size() * sizeof(int32_t)));
        HIP_TRY(hipMalloc(&ctx->d_tab, (size_t)cells_end));
        HIP_TRY(hipMalloc(&ctx->d_dist, (size_t)cells_end * sizeof(float)));
        HIP_TRY(hipMalloc(&ctx->d_scores, (size_t)n_entries * sizeof(int32_t)));
        HIP_TRY(hipMalloc(&d_bad, sizeof(int32_t)));
        lap("hipMalloc");
        // the headers first: the piece-wise checks and searches read them
        HIP_TRY(hipMemcpy(ctx->d_orders, orders, (size_t)n_entries * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ctx->d_cell_off, cell_off, (size_t)n_entries * sizeof(int64_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ctx->d_ordinal, ord.data(), (size_t)n_entries * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ctx->d_lists, lists.data(), lists.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemset(ctx->d_scores, 0, (size_t)n_entries * sizeof(int32_t)));
        HIP_TRY(hipMemcpy(d_bad, &none, sizeof none, hipMemcpyHostToDevice));
        lap("header copies");
        ctx->scores_cap = (size_t)n_entries;     // what refresh_descriptors compares with: no re-allocation for one query
        ctx->n_entries = n_entries;

        // The two big arrays go up in slices from a few host threads (each slice a synchronous copy out
        // of the caller's pageable memory: the runtime stages it through its pinned buffers, and several
        // copies in flight keep the link busy while one thread waits for its staging buffer).  The threads
        // walk the pieces together and count themselves off per piece; thread 0 queues the check of a
        // complete piece and (sat_db_upload_search) its search, and goes on copying.
        unsigned hw = std::thread::hardware_concurrency();
        int nthreads = (int)(hw ? (hw < 4 ? hw : 4) : 1);
        if (ctx->tune.upload_threads > 0) nthreads = ctx->tune.upload_threads;
        if (dist_bytes < ((size_t)32 << 20)) nthreads = 1;
        std::vector<hipError_t> err((size_t)nthreads, hipSuccess);
        std::vector<std::atomic<int>> landed((size_t)npieces);
        for (auto &x : landed) x.store(0);
        auto copy_piece = [&](int t, int c) {
            const size_t c0 = (size_t)piece_cell(c), c1 = (size_t)piece_cell(c + 1);
            auto part = [&](const void *src, void *dst, size_t unit) {
                const size_t bytes = (c1 - c0) * unit, base = c0 * unit;
                const size_t lo = (bytes * (size_t)t / (size_t)nthreads) & ~(size_t)255;
                const size_t hi = t + 1 == nthreads ? bytes : (bytes * (size_t)(t + 1) / (size_t)nthreads) & ~(size_t)255;
                if (hi > lo && err[(size_t)t] == hipSuccess)
                    err[(size_t)t] = hipMemcpy((char *)dst + base + lo, (const char *)src + base + lo, hi - lo, hipMemcpyHostToDevice);
            };
            part(dist_tri, ctx->d_dist, sizeof(float));
            part(tab_tri, ctx->d_tab, 1);
            landed[(size_t)c].fetch_add(1, std::memory_order_release);
        };
        // (the runtime takes the copies of all threads through one queue: a thread running ahead into piece
        // c + 1 would delay the last slice of piece c, and with it the piece's search, so nobody starts a
        // piece before the one before it is complete)
        auto piece_complete = [&](int c) {
            while (landed[(size_t)c].load(std::memory_order_acquire) < nthreads) std::this_thread::yield();
        };
        auto helper = [&](int t) {
            (void)hipSetDevice(ctx->device);
            for (int c = 0; c < npieces; c++) {
                copy_piece(t, c);
                if (c + 1 < npieces) piece_complete(c);
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < nthreads; t++) pool.emplace_back(helper, t);
        int rc = SAT_OK;
        for (int c = 0; c < npieces; c++) {
            copy_piece(0, c);
            piece_complete(c);
            if (rc != SAT_OK) continue;                      // (the helpers still finish their copies)
            // ---- check every cell where it now lives: one wave per entry; the kernel's pair arithmetic needs
            // tableau nibbles 0..7 (the reader produces 0..4), SSE types 0..3 and |distance| < 1e29 or non-finite.
            // A search queued behind the check of a bad piece is memory-safe (orders and offsets were checked
            // above; bad cells only give wrong sums) and its results are thrown away below.
            const int e0 = piece_e[(size_t)c], e1 = piece_e[(size_t)c + 1];
            if (e1 <= e0) continue;
            hipLaunchKernelGGL(validate_cells, dim3((unsigned)((e1 - e0 + 3) / 4)), dim3(256), 0, ctx->stream,
                               e0, e1, ctx->d_orders, ctx->d_cell_off, ctx->d_tab, ctx->d_dist, d_bad);
            if (hipGetLastError() != hipSuccess) { rc = fail(SAT_EDEVICE, "launch of the cell check failed"); continue; }
            if (first) {
                if (npieces > 1) {
                    const ListView piece = { ctx->d_lists, &piece_begin[(size_t)c * (kNumBuckets + 1)],
                                             &piece_n2max[(size_t)c * kNumBuckets], e1 - e0 };
                    rc = launch_search(ctx, first->lorder, first->lsoln, first->maxstart, ctx->stream, &piece);
                } else {
                    rc = launch_search(ctx, first->lorder, first->lsoln, first->maxstart, ctx->stream);
                }
            }
        }
        for (auto &th : pool) th.join();
        if (rc != SAT_OK) return rc;
        for (int t = 0; t < nthreads; t++) HIP_TRY(err[(size_t)t]);
        lap(first ? "cell copies, checks and the search queued" : "cell copies");
        int32_t bad = none;
        HIP_TRY(hipStreamSynchronize(ctx->stream));          // a non-blocking stream: the copy below does not wait for it
        HIP_TRY(hipMemcpy(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost));
        lap(first ? "search + validate on GPU" : "validate on GPU");
        if (bad != none) {
            // the earliest flagged entry is looked at again on the host, cell by cell, for the message
            const int e = bad, n = orders[e];
            for (int i = 0; i < n; i++) {
                const int64_t rowbase = cell_off[e] + (int64_t)i * (i + 1) / 2;
                uint8_t ty = tab_tri[rowbase + i];
                if (ty > 3) return fail(SAT_EINVAL, "entry %d: SSE %d has type code %u (0..3 expected)", e, i, ty);
                for (int j = 0; j < i; j++) {
                    if (tab_tri[rowbase + j] & 0x88)
                        return fail(SAT_EINVAL, "entry %d: tableau code 0x%02x at (%d,%d) has a nibble above 7", e, tab_tri[rowbase + j], i, j);
                    float d = dist_tri[rowbase + j];
                    if (std::isfinite(d) && std::fabs(d) >= 1.0e29f)
                        return fail(SAT_EINVAL, "entry %d: distance %g at (%d,%d) out of range", e, d, i, j);
                }
            }
            return fail(SAT_EINVAL, "entry %d: invalid cell", e);     // not reached: the scan and the re-check agree
        }
        return SAT_OK;
    };
    const int rc = body();
    if (d_bad) (void)hipFree(d_bad);
    if (rc != SAT_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        free_db(ctx);
        return rc;
    }
    ctx->h_orders.assign(orders, orders + n_entries);
    return SAT_OK;
}

int sat_db_upload_packed(sat_ctx *ctx, int n_entries, const int32_t *orders,
                         const int64_t *cell_off, const uint8_t *tab_tri,
                         const float *dist_tri, const int64_t *db_ordinal)
{
    return upload_impl(ctx, n_entries, orders, cell_off, tab_tri, dist_tri, db_ordinal, nullptr);
}

int sat_db_upload_search(sat_ctx *ctx, int n_entries, const int32_t *orders,
                         const int64_t *cell_off, const uint8_t *tab_tri,
                         const float *dist_tri, const int64_t *db_ordinal,
                         int lorder, int lsoln, int maxstart)
{
    const FirstSearch first = { lorder, lsoln, maxstart };
    return upload_impl(ctx, n_entries, orders, cell_off, tab_tri, dist_tri, db_ordinal, &first);
}

int sat_db_upload_dense(sat_ctx *ctx, int n_entries, const int32_t *orders,
                        const uint8_t *tabs, const float *dmats, int pitch,
                        const int64_t *db_ordinal)
{
    if (!ctx) return fail(SAT_EINVAL, "null context");
    if (n_entries <= 0 || !orders || !tabs || !dmats || pitch < 1)
        return fail(SAT_EINVAL, "empty database or null array");
    std::vector<int64_t> off(n_entries);
    int64_t cells = 0;
    for (int e = 0; e < n_entries; e++) {
        if (orders[e] < 1 || orders[e] > SAT_MAXDIM || orders[e] > pitch)
            return fail(SAT_EINVAL, "entry %d: order %d outside 1..min(%d, pitch %d)", e, orders[e], SAT_MAXDIM, pitch);
        off[e] = cells;
        cells += (int64_t)orders[e] * (orders[e] + 1) / 2;
    }
    std::vector<uint8_t> tt((size_t)cells);
    std::vector<float> dd((size_t)cells);
    for (int e = 0; e < n_entries; e++) {
        const uint8_t *t = tabs + (size_t)e * pitch * pitch;
        const float *d = dmats + (size_t)e * pitch * pitch;
        int64_t c = off[e];
        for (int i = 0; i < orders[e]; i++)
            for (int j = 0; j <= i; j++, c++) {
                tt[(size_t)c] = t[(size_t)i * pitch + j];
                dd[(size_t)c] = d[(size_t)i * pitch + j];
            }
    }
    return sat_db_upload_packed(ctx, n_entries, orders, off.data(), tt.data(), dd.data(), db_ordinal);
}

int sat_db_size(const sat_ctx *ctx) { return ctx ? ctx->n_entries : 0; }

int sat_queries_set(sat_ctx *ctx, int n_queries, const int32_t *n1s, const uint8_t *qtabs,
                    const float *qdmats, int pitch, const uint8_t *qssetypes, uint32_t first_query_ordinal)
{
    if (!ctx) return fail(SAT_EINVAL, "null context");
    if (n_queries < 1 || !n1s || !qtabs || !qdmats || !qssetypes || pitch < 1)
        return fail(SAT_EINVAL, "bad query batch (n_queries=%d pitch=%d)", n_queries, pitch);
    std::vector<sat_ctx::QueryInfo> infos((size_t)n_queries);
    size_t blob_bytes = 0;
    for (int qi = 0; qi < n_queries; qi++) {
        const int n1 = n1s[qi];
        if (n1 < 1 || n1 > SAT_MAXDIM || n1 > pitch)
            return fail(SAT_EINVAL, "query %d: order %d outside 1..min(%d, pitch %d)", qi, n1, SAT_MAXDIM, pitch);
        auto &q = infos[(size_t)qi];
        q.n1 = n1;
        q.n1p = n1 <= 16 ? 16 : (n1 <= 32 ? 32 : (n1 <= 64 ? 64 : 112));
        q.ordinal = first_query_ordinal + (uint32_t)qi;
        q.blob_off = blob_bytes;
        q.ssemap_off = 0;
        const size_t groups = (size_t)q.n1p / 4 * q.n1p;
        // grouped cells (16 + 4 bytes per group and column), SSE types, then the dense pair cells of the full score
        blob_bytes += ((groups * 20 + (size_t)q.n1p + 15) & ~(size_t)15) + (size_t)q.n1p * q.n1p * 8;
    }
    // grouped, transposed query: group kw, column i holds dmat1[i][4kw..4kw+3] and the four code
    // bytes tab1[i][4kw..4kw+3]; diagonal, padding and non-finite distances get the sentinel
    // so they never score (the reference excludes k == i, K.cu:524, and NaN never passes <= 4)
    std::vector<uint8_t> blob(blob_bytes, 0);
    for (int qi = 0; qi < n_queries; qi++) {
        const auto &q = infos[(size_t)qi];
        const int n1 = q.n1, n1p = q.n1p, groups = n1p / 4;
        const uint8_t *qtab = qtabs + (size_t)qi * pitch * pitch;
        const float *qdmat = qdmats + (size_t)qi * pitch * pitch;
        const uint8_t *types = qssetypes + (size_t)qi * pitch;
        float4 *qdist = reinterpret_cast<float4 *>(blob.data() + q.blob_off);
        uint32_t *qcode = reinterpret_cast<uint32_t *>(blob.data() + q.blob_off + (size_t)groups * n1p * 16);
        uint8_t *qtypes = blob.data() + q.blob_off + (size_t)groups * n1p * 20;
        // dense [i][k] cells {distance, code byte}: the same values as the grouped arrays, for the pair-by-pair
        // full score of an initial map
        uint32_t *qpair = reinterpret_cast<uint32_t *>(blob.data() + q.blob_off + (((size_t)groups * n1p * 20 + (size_t)n1p + 15) & ~(size_t)15));
        for (int i = 0; i < n1p; i++)
            for (int k = 0; k < n1p; k++) {
                float d = SAT_K_QSENT;
                uint32_t code = 0;
                if (k < n1 && i < n1 && k != i) {
                    const float v = qdmat[(size_t)i * pitch + k];
                    if (std::isfinite(v)) d = v;                  // (range and nibbles are checked below)
                    code = qtab[(size_t)i * pitch + k];
                }
                memcpy(&qpair[((size_t)i * n1p + k) * 2], &d, sizeof d);
                qpair[((size_t)i * n1p + k) * 2 + 1] = code;
            }
        for (int i = 0; i < n1; i++) {
            if (types[i] > 3)
                return fail(SAT_EINVAL, "query %d: SSE %d has type code %u (0..3 expected)", qi, i, types[i]);
            qtypes[i] = types[i];
        }
        for (int kw = 0; kw < groups; kw++)
            for (int i = 0; i < n1p; i++) {
                float d[4];
                uint32_t codes = 0;
                for (int sidx = 0; sidx < 4; sidx++) {
                    const int k = 4 * kw + sidx;
                    d[sidx] = SAT_K_QSENT;
                    if (k < n1 && i < n1 && k != i) {
                        const float v = qdmat[(size_t)i * pitch + k];
                        const uint32_t code = qtab[(size_t)i * pitch + k];
                        if (code & 0x88)
                            return fail(SAT_EINVAL, "query %d: tableau code 0x%02x at (%d,%d) has a nibble above 7", qi, code, i, k);
                        if (std::isfinite(v)) {
                            if (std::fabs(v) >= 1.0e29f)
                                return fail(SAT_EINVAL, "query %d: distance %g at (%d,%d) out of range", qi, v, i, k);
                            d[sidx] = v;
                        }
                        codes |= code << (8 * sidx);
                    }
                }
                qdist[(size_t)kw * n1p + i] = float4{ d[0], d[1], d[2], d[3] };
                qcode[(size_t)kw * n1p + i] = codes;
            }
    }
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    dev_free(ctx->d_qblob);
    dev_free(ctx->d_qdesc);
    HIP_TRY(hipMalloc(&ctx->d_qblob, blob_bytes));
    HIP_TRY(hipMalloc(&ctx->d_qdesc, (size_t)n_queries * sizeof(SatQuery)));
    HIP_TRY(hipMemcpy(ctx->d_qblob, blob.data(), blob_bytes, hipMemcpyHostToDevice));
    ctx->queries.swap(infos);
    ctx->desc_dirty = true;
    ctx->searched_nq = 0;                     // the result buffers no longer belong to the current batch
    return SAT_OK;
}

int sat_query_set(sat_ctx *ctx, int n1, const uint8_t *qtab, const float *qdmat,
                  int pitch, const uint8_t *qssetypes, uint32_t query_ordinal)
{
    if (!ctx) return fail(SAT_EINVAL, "null context");
    if (n1 < 1 || n1 > SAT_MAXDIM || !qtab || !qdmat || !qssetypes || pitch < n1)
        return fail(SAT_EINVAL, "bad query (n1=%d pitch=%d)", n1, pitch);
    // a batch of one; the type vector is only read up to n1, so its stride does not matter
    const int32_t n1s[1] = { n1 };
    return sat_queries_set(ctx, 1, n1s, qtab, qdmat, pitch, qssetypes, query_ordinal);
}

int sat_query_count(const sat_ctx *ctx) { return ctx ? (int)ctx->queries.size() : 0; }

int sat_use_stream(sat_ctx *ctx, void *hip_stream)
{
    if (!ctx) return fail(SAT_EINVAL, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->stream = static_cast<hipStream_t>(hip_stream);
    return SAT_OK;
}

int sat_use_own_stream(sat_ctx *ctx)
{
    if (!ctx) return fail(SAT_EINVAL, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->stream = ctx->own_stream;
    return SAT_OK;
}

int sat_search_async(sat_ctx *ctx, int lorder, int lsoln, int maxstart)
{
    if (!ctx) return fail(SAT_EINVAL, "null context");
    return launch_search(ctx, lorder, lsoln, maxstart, ctx->stream);
}

void *sat_device_scores(sat_ctx *ctx) { return ctx ? ctx->d_scores : nullptr; }
void *sat_device_ssemaps(sat_ctx *ctx) { return ctx ? ctx->d_ssemaps : nullptr; }
int sat_query_order(const sat_ctx *ctx) { return (ctx && !ctx->queries.empty()) ? ctx->queries[0].n1 : 0; }

unsigned long long sat_stat_d2h_bytes(const sat_ctx *ctx) { return ctx ? ctx->d2h_bytes : 0ull; }

const char *sat_last_launch_info(const sat_ctx *ctx) { return ctx ? ctx->last_launch_info.c_str() : ""; }

void sat_debug_lds_layout(int m2w, int n1, int n1p, int n2, int chains, int threads, int q_in_lds, int compact,
                          uint32_t out[11])
{
    // m2w: low byte = words of a db-side set; bits 8-9 = 1 + cell layout (SAT_CELLS_*), 0 = the layout launches of
    // such entries get (satk::cell_layout)
    const int cells = (m2w >> 8) ? ((m2w >> 8) & 3) - 1 : satk::cell_layout(n2);
    m2w &= 0xFF;
    const satk::LdsLayout L = satk::lds_layout(m2w, cells, n2, satk::map_words((n1 + 3) >> 2), n1p, chains, threads,
                                               q_in_lds != 0, compact != 0);
    const uint32_t v[11] = { L.code, L.qdist, L.qcode, L.smap, L.tmask, L.qtypes, L.leader, L.red, L.red_stride, L.items, L.total };
    for (int i = 0; i < 11; i++) out[i] = v[i];
}

int sat_sync(sat_ctx *ctx)
{
    if (!ctx) return fail(SAT_EINVAL, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SAT_OK;
}

int sat_results(sat_ctx *ctx, int lsoln, int32_t *scores, int32_t *ssemaps)
{
    if (!ctx) return fail(SAT_EINVAL, "null context");
    if (!scores) return fail(SAT_EINVAL, "scores buffer is null");
    if (lsoln && !ssemaps) return fail(SAT_EINVAL, "lsoln set but ssemaps buffer is null");
    if (ctx->n_entries <= 0) return fail(SAT_ESTATE, "no database uploaded");
    if (ctx->queries.empty() || !ctx->d_scores || ctx->searched_nq != ctx->queries.size())
        return fail(SAT_ESTATE, "no search has run since the last database upload / query change");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const size_t nq = ctx->queries.size(), N = (size_t)ctx->n_entries;
    HIP_TRY(hipMemcpy(scores, ctx->d_scores, nq * N * sizeof(int32_t), hipMemcpyDeviceToHost));
    ctx->d2h_bytes += nq * N * sizeof(int32_t);
    if (lsoln) {
        if (!ctx->d_ssemaps || !ctx->searched_lsoln) return fail(SAT_ESTATE, "the last search ran without lsoln");
        std::vector<int8_t> packed;
        for (size_t qi = 0; qi < nq; qi++) {
            const auto &q = ctx->queries[qi];
            packed.resize(N * q.n1);
            HIP_TRY(hipMemcpy(packed.data(), ctx->d_ssemaps + q.ssemap_off, packed.size(), hipMemcpyDeviceToHost));
            ctx->d2h_bytes += packed.size();
            int32_t *out = ssemaps + qi * N * SAT_MAXDIM;
            for (size_t e = 0; e < N; e++)
                for (int i = 0; i < q.n1; i++)
                    out[e * SAT_MAXDIM + i] = packed[e * q.n1 + i];
        }
    }
    return SAT_OK;
}

int sat_search(sat_ctx *ctx, int lorder, int lsoln, int maxstart,
               int32_t *scores, int32_t *ssemaps, double *kernel_ms)
{
    if (!ctx) return fail(SAT_EINVAL, "null context");
    if (!scores) return fail(SAT_EINVAL, "scores buffer is null");
    if (lsoln && !ssemaps) return fail(SAT_EINVAL, "lsoln set but ssemaps buffer is null");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    int rc = launch_search(ctx, lorder, lsoln, maxstart, ctx->stream);
    if (rc != SAT_OK) return rc;
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (kernel_ms) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        *kernel_ms = ms;
    }
    return sat_results(ctx, lsoln, scores, ssemaps);
}

int sat_search_timed(sat_ctx *ctx, int lorder, int lsoln, int maxstart, int repeats,
                     double *total_ms, double *kernel_ms)
{
    if (!ctx) return fail(SAT_EINVAL, "null context");
    if (repeats < 1) return fail(SAT_EINVAL, "repeats must be >= 1");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    for (int r = 0; r < repeats; r++) {
        int rc = launch_search(ctx, lorder, lsoln, maxstart, ctx->stream);
        if (rc != SAT_OK) return rc;
    }
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    if (total_ms) *total_ms = ms;
    if (kernel_ms) *kernel_ms = ms;
    return SAT_OK;
}

}  // extern "C"
