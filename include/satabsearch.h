/*
 * satabsearch.h - C ABI of the MI355X tableau-search library (libsatabsearch.so).
 *
 * This is the drop-in boundary for ONE path of stivalaa/cuda_satabsearch: the
 * simulated-annealing scoring of one query against every database structure.
 * Plain C: pointers and sizes only, no HIP / torch / C++ types.  Each entry
 * point names the reference interface it stands in for (paths relative to the
 * reference tree, H.cu = nvcc_src_current/cudaSaTabsearch.cu,
 * K.cu = nvcc_src_current/cudaSaTabsearch_kernel.cu).
 *
 * Ownership: the caller owns every host buffer it passes; the context owns all
 * device memory.  Threading: one context per host thread; no globals.
 * Errors: every int-returning call yields 0 on success and a negative SAT_E*
 * code otherwise; sat_last_error() has the text.  Nothing in the library calls
 * exit() or abort().  (Tuning / test overrides of the launch heuristics: include/satabsearch_debug.h.)
 * There is no CPU fallback: without a usable HIP device sat_ctx_create() fails with
 * SAT_ENODEVICE.
 */
#ifndef SATABSEARCH_H
#define SATABSEARCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAT_ABI_VERSION   1
#define SAT_MAXDIM        111    /* saparams.h:15 (MAXDIM): largest structure order        */
#define SAT_MAXITER       100    /* saparams.h:33 (MAXITER): SA steps per restart           */
#define SAT_DEFAULT_SEED  1234   /* H.cu:263, :871                                           */

#define SAT_OK          0
#define SAT_EINVAL     -1   /* bad argument (order out of range, bad type code, ...)        */
#define SAT_ENODEVICE  -2   /* no HIP device / device index out of range                    */
#define SAT_ENOMEM     -3   /* host or device allocation failed                             */
#define SAT_EDEVICE    -4   /* a HIP call or the kernel launch failed (H.cu:1067-1073)      */
#define SAT_ESTATE     -5   /* call order violated (search before upload / query)           */

typedef struct sat_ctx sat_ctx;

/* Message of the most recent failure on this thread ("" if none). */
const char *sat_last_error(void);

/* Library ABI version (SAT_ABI_VERSION it was built with). */
int sat_abi_version(void);

/* Number of visible HIP devices; replaces cudaGetDeviceCount, H.cu:805-812. */
int sat_device_count(void);

/*
 * Create a context on HIP device `device` (the reference picks one device and
 * calls cudaSetDevice, H.cu:813-865, then allocates and seeds one RNG state per
 * thread with init_rng, H.cu:258-264, 896-922).  Here the random streams are
 * counter based, so `seed` is all the state there is: chain (query ordinal, db
 * ordinal, restart) draws from its own Philox4x32-10 stream.
 * Returns NULL on failure.
 */
sat_ctx *sat_ctx_create(int device, uint64_t seed);

/* Release every device and host resource of the context (H.cu:1117-1126, 1326-1337). */
void sat_ctx_destroy(sat_ctx *ctx);

/*
 * Upload a database shard given as packed lower triangles (the layout the
 * library's own reader produces, sat_parse.h):
 *   n_entries          structures in this shard
 *   orders[e]          number of SSEs of entry e, 1..SAT_MAXDIM
 *   cell_off[e]        index of entry e's first cell in tab_tri / dist_tri
 *   tab_tri, dist_tri  cell (i,j), j <= i, at cell_off[e] + i*(i+1)/2 + j;
 *                      diagonal of tab_tri = SSE type code 0..3, off-diagonal =
 *                      two-nibble tableau code (parsetableaux.c:13-33)
 *   db_ordinal[e]      position of entry e in the whole database's file order
 *                      (keys the random streams, so results do not depend on
 *                      how the database is sharded over GPUs); NULL = e
 * Replaces the cudaMalloc3D + cudaMemcpy3D of the dense 96x96 / 111x111 slots,
 * H.cu:924-967 and 1135-1177; both size classes go into the one packed store.
 */
int sat_db_upload_packed(sat_ctx *ctx, int n_entries, const int32_t *orders,
                         const int64_t *cell_off, const uint8_t *tab_tri,
                         const float *dist_tri, const int64_t *db_ordinal);

/*
 * Upload a shard AND run the first search of the current query (or query batch) over it, overlapped:
 * the shard goes up in a few pieces of whole entries, and each piece is checked and searched on the GPU
 * while the host copies the next one - a single query over a freshly read database then costs about
 * max(copy, search) instead of their sum (the reference copies the whole database, H.cu:924-967, then
 * launches, H.cu:1036).  Arguments as sat_db_upload_packed + sat_search_async; the query must be set
 * before.  On return the shard is resident and validated as after sat_db_upload_packed, and the search
 * has completed: collect with sat_results / sat_topk_hits / sat_device_scores.  Results are those of
 * sat_db_upload_packed followed by sat_search, bit for bit.  Entries laid out in ascending cell order
 * (as every reader here produces them) are needed for the overlap; otherwise, and for small shards,
 * the two steps simply run one after the other.  The copies are synchronous calls (the caller's buffers
 * may be pageable), so the overlap also needs the search on a stream they do not wait for: the context's
 * own (non-blocking) stream, or a non-blocking stream given to sat_use_stream; on the device's default
 * stream the pieces are copied and searched in turn - same results, no gain.
 */
int sat_db_upload_search(sat_ctx *ctx, int n_entries, const int32_t *orders,
                         const int64_t *cell_off, const uint8_t *tab_tri,
                         const float *dist_tri, const int64_t *db_ordinal,
                         int lorder, int lsoln, int maxstart);

/*
 * Same, from the reference's dense host layout: entry e occupies
 * pitch*pitch cells at tabs + e*pitch*pitch (row-major, symmetric), exactly the
 * arrays read_database() returns (parsetableaux.c:317-506; pitch 96 or 111).
 * Only the lower triangle is read.
 */
int sat_db_upload_dense(sat_ctx *ctx, int n_entries, const int32_t *orders,
                        const uint8_t *tabs, const float *dmats, int pitch,
                        const int64_t *db_ordinal);

/* Entries currently resident. */
int sat_db_size(const sat_ctx *ctx);

/*
 * Set the query: dense n1 x n1 code and distance matrices with row pitch
 * `pitch` (111 in the reference), SSE types in qssetypes[0..n1).  Replaces the
 * four cudaMemcpy to the c_qn / c_qtab / c_qdmat / c_qssetypes device symbols,
 * copyQueryToConstantMemory H.cu:486-558 and K.cu:118-121.  `query_ordinal` is
 * the query's index in the run (second key of the random streams).
 */
int sat_query_set(sat_ctx *ctx, int n1, const uint8_t *qtab, const float *qdmat,
                  int pitch, const uint8_t *qssetypes, uint32_t query_ordinal);

/*
 * Set a BATCH of queries that one sat_search scores together (grid = entries x queries
 * inside the launches): query q occupies pitch*pitch cells at qtabs + q*pitch*pitch and
 * qdmats + q*pitch*pitch, its SSE types pitch bytes at qssetypes + q*pitch, its order is
 * n1s[q]; its stream key is first_query_ordinal + q.  The reference loops over queries
 * with a sync, four cudaMemcpy and a launch each (H.cu:987-1115); query lists (-q) of
 * hundreds of SIDs are its main workload, and a small database alone cannot fill the GPU.
 * After this call every result buffer has one row per query, in the order given here.
 */
int sat_queries_set(sat_ctx *ctx, int n_queries, const int32_t *n1s, const uint8_t *qtabs,
                    const float *qdmats, int pitch, const uint8_t *qssetypes,
                    uint32_t first_query_ordinal);

/* Queries currently set (1 after sat_query_set). */
int sat_query_count(const sat_ctx *ctx);

/*
 * Run the search for the current query (or query batch) over the resident shard and wait.
 * With a batch of nq queries: scores is [nq][n_entries], ssemaps [nq][n_entries*SAT_MAXDIM].
 * Replaces the sa_tabsearch_gpu / sa_tabsearch_gpu_noshared launches, their
 * cudaDeviceSynchronize and the result cudaMemcpy, H.cu:1036-1087, 1219-1253
 * (kernel contract K.cu:756-802):
 *   lorder     keep sequence order of matched SSEs (LORDER)
 *   lsoln      also return the best SSE map (LSOLN)
 *   maxstart   restarts per db entry (-r, default 128)
 *   scores     [n_entries]            best score per entry, shard order
 *   ssemaps    [n_entries * SAT_MAXDIM] or NULL; entry e's map at e*SAT_MAXDIM,
 *              ssemaps[e*111 + i] = db SSE matched to query SSE i, or -1; only
 *              written when lsoln != 0 (same layout as K.cu:797-800, 1232)
 *   kernel_ms  (may be NULL) device time of the search kernels, launch -> sync,
 *              the window the reference times (H.cu:1036-1077)
 */
int sat_search(sat_ctx *ctx, int lorder, int lsoln, int maxstart,
               int32_t *scores, int32_t *ssemaps, double *kernel_ms);

/*
 * Queue all further work of this context on the caller's stream (`hip_stream` is a
 * hipStream_t passed as void*; NULL selects the device's default stream).  A context
 * starts on a private non-blocking stream; sat_use_own_stream() goes back to it.
 * Lets a caller order the search with its own copies / collectives (bench.py hands
 * over torch's current stream so that the RCCL gather follows the kernel).
 */
int sat_use_stream(sat_ctx *ctx, void *hip_stream);
int sat_use_own_stream(sat_ctx *ctx);

/*
 * Device-resident variant for callers that keep results on the GPU (bench.py,
 * torch.distributed gather over RCCL): launches on the context's current stream,
 * does not synchronise and does not copy.  Results land in the context's device
 * buffers:
 *   sat_device_scores()   int32 [n_queries][n_entries]
 *   sat_device_ssemaps()  int8, query q's [n_entries][n1_q] block after those of queries
 *                         0..q-1; -1 = unmatched; valid after a search with lsoln != 0
 * Pointer lifetime: ask for the pointers AFTER the search has been queued; they stay valid
 * (and keep that search's results) until the next database upload, the next query / query
 * batch change or a search with more queries or lsoln newly set - any of these may
 * re-allocate the buffers.
 * sat_query_order() is the order of query 0.
 */
int sat_search_async(sat_ctx *ctx, int lorder, int lsoln, int maxstart);
void *sat_device_scores(sat_ctx *ctx);
void *sat_device_ssemaps(sat_ctx *ctx);
int sat_query_order(const sat_ctx *ctx);

/* Wait for everything queued on the context's current stream. */
int sat_sync(sat_ctx *ctx);

/*
 * Wait for the queued search and copy its results to the host, same buffers and
 * layout as sat_search (ssemaps may be NULL when lsoln == 0).  SAT_ESTATE when no search
 * has run since the last upload / query change, or lsoln is asked of a search without it.  With sat_search_async
 * this lets one host thread keep several devices busy (one context per GPU, the
 * database sharded contiguously): launch on all, then collect from each - the
 * multi-GPU mode the reference left as a TODO (H.cu:790).
 */
int sat_results(sat_ctx *ctx, int lsoln, int32_t *scores, int32_t *ssemaps);

/*
 * Best-k hits of query `query` (0 for a single query) of the last search, selected and
 * sorted on the device: entry_index[i] / scores_out[i] for i < k, by descending score, ties
 * in database order - what `sort -k 2,2nr | head` does to the reference's output
 * (README_example_usage.txt:100).  Returns the number of hits written (min(k, n_entries))
 * or a negative SAT_E* code.  Waits for the queued search.
 */
int sat_topk(sat_ctx *ctx, int query, int k, int32_t *entry_index, int32_t *scores_out);

/*
 * One row of the reference's output, "name rawscore norm2score z-score p-value"
 * (cudaSaTabsearch.cu:445-453), for a best-k hit: the entry's index in the shard instead of its name.
 */
typedef struct sat_hit {
    int32_t entry;      /* index in the resident shard                                            */
    int32_t score;      /* raw score                                                              */
    double  norm2;      /* 2 * score / (n1 + n2)                         gumbelstats.c:91-94      */
    double  zscore;     /* Gumbel z of the norm2 score truncated to an int   gumbelstats.c:50-58  */
    double  pvalue;     /* 1 - exp(-exp(-(pi / sqrt 6 * z + gamma)))      gumbelstats.c:69-72     */
} sat_hit;

/*
 * Best-k rows of EVERY query of the last search, ranked and given their statistics on the device
 * (one segmented sort for the whole batch; the statistics are bit-identical to the host's
 * csrc/host/sat_gumbel.c): hits[q * k + r] is rank r of query q, by descending score, ties in
 * database order.  ssemaps (may be NULL): [n_queries * k * SAT_MAXDIM] solution maps of those rows,
 * laid out like sat_search's, after a search with lsoln.  Only these k rows per query are copied
 * to the host - what a user who pipes the reference's output through `sort -k 2,2nr | head`
 * wants (README_example_usage.txt:100, 256).  Returns min(k, n_entries) or a negative SAT_E* code.
 */
int sat_topk_hits(sat_ctx *ctx, int k, sat_hit *hits, int32_t *ssemaps);

/* Bytes this context's result calls (sat_results, sat_search, sat_topk, sat_topk_hits) have copied
 * from the device to the host since it was created (diagnostics: the best-k path moves O(k) rows). */
unsigned long long sat_stat_d2h_bytes(const sat_ctx *ctx);

/* Diagnostics: the kernel instantiations (template arguments as rocprofv3 prints them), grids, block
 * sizes and LDS bytes of the launches of this context's last search, "; "-separated. */
const char *sat_last_launch_info(const sat_ctx *ctx);

/*
 * Diagnostics: the LDS carve of one entry slot of the SA kernel (csrc/sat_sa_kernel.hpp, lds_layout -
 * the one function both the kernel and the launch sizing use), byte offsets out[0..10] = code bytes,
 * query distances, query codes, chain maps, type masks, query types, LSOLN leader key, the waves'
 * arg-max keys, the byte stride between those keys (256: inside the item tables), item tables, total.
 * m2w = words of a db-side bit set in the launch's size class: 1 (entries up to 32 SSEs), 2 (64) or 4; its bits
 * 8-9 may name the cell layout, 1 + {0: full matrix of 8-byte cells, 1: full matrix in two arrays, 2: lower
 * triangle in two arrays} (0: the layout launches of such entries get).
 * Lets tests assert alignment and monotonicity without a GPU.
 */
void sat_debug_lds_layout(int m2w, int n1, int n1p, int n2, int chains, int threads, int q_in_lds, int compact,
                          uint32_t out[11]);

/*
 * ---- one search over several GPUs of a node, driven from one host thread -------------------------
 * The reference is single-GPU (cudaSaTabsearch.cu:790 "TODO allow multiple GPUs").  A sat_multi
 * holds one context per GPU; the database is cut into contiguous shards of equal COST
 * (csrc/host/sat_shard.h: real databases are size sorted and a 96-SSE entry costs four 32-SSE
 * ones), each GPU holds its shard and the queries, a search is queued on all of them and ONE
 * gather (RCCL ncclGather over xGMI; SAT_MULTI_GATHER=peer: hipMemcpyPeerAsync; an RCCL gather that fails
 * at run time falls back to the peer copies for the rest of the context's life unless
 * SAT_MULTI_GATHER=rccl insists) brings the shard
 * rows to device 0, from where one copy takes them to the host in database file order.  Results
 * are identical for any number of GPUs (streams are keyed by the entry's ordinal in the database).
 *
 * sat_multi_create      ndev GPUs (<= 0: all visible; devices == NULL: 0 .. ndev-1; a list may name a GPU more
 *                       than once - several shards on one GPU, gathered by peer copies)
 * sat_multi_db_upload_packed   as sat_db_upload_packed for the WHOLE database (ordinals = file order), with ONE more
 *                       requirement: cell_off must ascend in file order without overlap (entry e + 1 starts at or
 *                       after the end of entry e - what every reader here produces), because a shard is uploaded
 *                       as a window of the packed arrays; anything else is SAT_EINVAL
 * sat_multi_shards      begin[ndev + 1]: shard g holds entries begin[g] .. begin[g+1]-1
 * sat_multi_queries_set as sat_queries_set, on every GPU
 * sat_multi_search      as sat_search: scores [nq][n_entries] (and ssemaps) in database order;
 *                       wall_ms = launch on all GPUs .. rows on the host
 * sat_multi_search_topk the best k rows per query, each GPU ranking its own shard (sat_topk_hits) and
 *                       the host merging ndev x k candidates; hits[q * k + r].entry is the index
 *                       in the whole database; ssemaps as in sat_topk_hits
 * sat_multi_gather_kind "rccl", "peer" or "none" (one GPU)
 */
typedef struct sat_multi sat_multi;
sat_multi *sat_multi_create(int ndev, const int *devices, uint64_t seed);
void sat_multi_destroy(sat_multi *m);
int sat_multi_device_count(const sat_multi *m);
const char *sat_multi_gather_kind(const sat_multi *m);
int sat_multi_db_upload_packed(sat_multi *m, int n_entries, const int32_t *orders, const int64_t *cell_off,
                               const uint8_t *tab_tri, const float *dist_tri);
int sat_multi_shards(const sat_multi *m, int32_t *begin);
int sat_multi_queries_set(sat_multi *m, int n_queries, const int32_t *n1s, const uint8_t *qtabs,
                          const float *qdmats, int pitch, const uint8_t *qssetypes, uint32_t first_query_ordinal);
int sat_multi_search(sat_multi *m, int lorder, int lsoln, int maxstart, int32_t *scores, int32_t *ssemaps,
                     double *wall_ms);
int sat_multi_search_topk(sat_multi *m, int lorder, int lsoln, int maxstart, int k, sat_hit *hits,
                          int32_t *ssemaps, double *wall_ms);
unsigned long long sat_multi_stat_d2h_bytes(const sat_multi *m);

/*
 * Time `repeats` back-to-back searches with HIP events on the launch stream
 * (inputs resident, no copies inside the window).  Returns total milliseconds
 * in *total_ms and the dominant SA kernel's summed device time in *kernel_ms.
 */
int sat_search_timed(sat_ctx *ctx, int lorder, int lsoln, int maxstart, int repeats,
                     double *total_ms, double *kernel_ms);

#ifdef __cplusplus
}
#endif
#endif /* SATABSEARCH_H */
