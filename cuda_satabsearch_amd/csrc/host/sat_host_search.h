/*
 * sat_host_search.h - the "-c" host mode of the command line: the same search on one
 * CPU thread, drawing from ONE sequential drand48-compatible stream for the whole run
 * exactly as the reference's host path does (nvcc_src_current/cudaSaTabsearch.cu:871
 * srand48(1234); kernel.cu built without -DCUDA, :804-1236).  Output is byte-identical
 * to the reference's `cudaSaTabsearch -c` (tests/test_cli.py against tests/golden).
 *
 * This is a user-selected mode of the CLI, as in the reference.  It is never used as a
 * fallback: the GPU entry points of libsatabsearch.so fail when no device is usable.
 */
#ifndef SAT_HOST_SEARCH_H
#define SAT_HOST_SEARCH_H
#include <stdint.h>
#include "sat_parse.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct sat_host_stream { uint64_t x; } sat_host_stream;   /* 48-bit LCG state */

void sat_host_stream_seed(sat_host_stream *st, long seedval);     /* = srand48(seedval) */

/*
 * Search structures entries[0..n) of `db` with the query structure `qs` of `queries`.
 * scores[k] / ssemaps[k*SAT_MAXDIM + i] belong to entries[k]; ssemaps may be NULL and
 * is written only when lsoln != 0.  Returns 0, or -1 on allocation failure.
 */
int sat_host_search(const sat_struct_set *db, const int *entries, int n,
                    const sat_struct_set *queries, int qs,
                    int lorder, int lsoln, int maxstart, sat_host_stream *stream,
                    int32_t *scores, int32_t *ssemaps);

#ifdef __cplusplus
}
#endif
#endif
