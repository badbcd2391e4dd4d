// sat_ctx.hpp - private definition of the C ABI's context (shared by sat_capi.hip and
// sat_topk.hip; not part of the public interface).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <string>
#include <tuple>
#include <unordered_set>
#include <vector>

#include "satabsearch.h"
#include "sat_sa_kernel.hpp"

constexpr int kNumBuckets = 7;

struct sat_ctx {
    int device = 0;
    uint64_t seed = SAT_DEFAULT_SEED;
    hipStream_t own_stream = nullptr;   // created with the context
    hipStream_t stream = nullptr;       // where work is queued (own_stream unless sat_use_stream)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    // database shard
    int n_entries = 0;
    int32_t *d_orders = nullptr;
    int64_t *d_cell_off = nullptr;
    uint8_t *d_tab = nullptr;
    float *d_dist = nullptr;
    uint32_t *d_ordinal = nullptr;
    int32_t *d_lists = nullptr;             // entry indices grouped by bucket
    int bucket_begin[kNumBuckets + 1] = { 0 };
    int bucket_n2max[kNumBuckets] = { 0 };
    std::vector<int32_t> h_orders;

    // queries (a batch; one query is a batch of 1), input order
    struct QueryInfo { int n1, n1p; uint32_t ordinal; size_t blob_off; size_t ssemap_off; };
    std::vector<QueryInfo> queries;
    uint8_t *d_qblob = nullptr;             // per query: qdist | qcode | qtypes
    SatQuery *d_qdesc = nullptr;            // descriptors grouped by size class
    int class_begin[5] = { 0, 0, 0, 0, 0 };  // classes: n1p = 16, 32, 64, 112
    int class_n1max[4] = { 0, 0, 0, 0 };
    int class_wpl[4] = { 0, 0, 0, 0 };       // map words per lane shared by the class's queries, 0 = mixed
    bool desc_dirty = true;
    bool desc_lsoln = false;

    // Metropolis table
    float *d_ptab = nullptr;
    int32_t *d_prow = nullptr;

    // launch-heuristic overrides (SAT_EXP_* in satabsearch.h), read ONCE when the context is created
    struct Tuning { int compact = -1, qlds = -1, lpc = -1, general = 0, streams = -1, upload_threads = 0, upload_timing = 0, upload_pieces = 0, epw = 0, lpc_waves = 0, chains = 0; size_t lds_pad = 0; } tune;
    // kernel instantiations whose dynamic-LDS limit has been raised on this device
    std::unordered_set<const void *> lds_attr_done;
    // entries per workgroup chosen for (instantiation, threads per entry, LDS bytes per entry): asked once
    std::map<std::tuple<const void *, int, size_t>, int> epw_choice;
    // side streams: the order buckets of one search run concurrently (each launch has a tail of
    // half-empty CUs; the next bucket's workgroups fill it), forked from / joined to `stream`
    hipStream_t side_stream[kNumBuckets] = { nullptr };
    hipEvent_t ev_fork = nullptr, ev_join[kNumBuckets] = { nullptr };

    // what the result buffers hold: set by a search, cleared by an upload or a new query batch
    size_t searched_nq = 0;                  // 0 = no search since the last upload / query change
    bool searched_lsoln = false;

    // multi-GPU gather (sat_multi.hip): the result buffers are sized for at least this many rows per
    // query, so that a fixed-size gather may read a shard padded to the largest shard
    int min_rows = 0;

    // results: scores [nq][N]; ssemaps: query q's [N][n1_q] block at queries[q].ssemap_off
    int32_t *d_scores = nullptr;
    size_t scores_cap = 0;
    int8_t *d_ssemaps = nullptr;
    size_t ssemaps_cap = 0;
    uint32_t *d_bmap_slabs = nullptr;        // LSOLN scratch: one best-map slab per workgroup of a launch
    size_t bmap_slabs_cap = 0;               // in 32-bit words

    // best-k selection (sat_topk.hip): context-owned scratch that only grows; capacities in elements
    unsigned long long *d_keys = nullptr, *d_sorted = nullptr;
    size_t keys_cap = 0, sorted_cap = 0;
    unsigned char *d_sort_temp = nullptr, *d_hitq = nullptr;
    size_t sort_temp_cap = 0, hitq_cap = 0;
    int *d_seg = nullptr;
    size_t seg_cap = 0;
    sat_hit *d_hits = nullptr;
    size_t hits_cap = 0;
    int32_t *d_hit_maps = nullptr;
    size_t hit_maps_cap = 0;
    // z and p of every truncated norm2 score -128 .. 127, computed by the HOST's libm (sat_gumbel.c)
    double *d_gumbel_z = nullptr, *d_gumbel_p = nullptr;

    // bytes copied device -> host by this context's result calls (sat_stat_d2h_bytes)
    unsigned long long d2h_bytes = 0;
    // kernel instantiations and launch geometry of the last search (sat_last_launch_info)
    std::string last_launch_info;
};


// sets sat_last_error() text and returns `code`
int sat_fail(int code, const char *fmt, ...);
