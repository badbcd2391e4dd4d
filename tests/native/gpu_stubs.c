/* gpu_stubs.c - link-time stand-ins for the device library's entry points, so that the HOST C of the command line
 * (readers, writer, binary image, shard cuts, statistics, the -c host search, printing) can be built on its own with
 * -fsanitize=address,undefined (tests/test_host_sanitizers.py; GPU AddressSanitizer is not available on this pool).
 * Test infrastructure: every stub fails the way the real library does without a GPU, so only -c runs get anywhere. */
#include <stddef.h>
#include "satabsearch.h"

const char *sat_last_error(void) { return "sanitizer build: no device library linked"; }
int sat_device_count(void) { return 0; }
sat_multi *sat_multi_create(int ndev, const int *devices, uint64_t seed) { (void)ndev; (void)devices; (void)seed; return NULL; }
void sat_multi_destroy(sat_multi *m) { (void)m; }
int sat_multi_device_count(const sat_multi *m) { (void)m; return 0; }
const char *sat_multi_gather_kind(const sat_multi *m) { (void)m; return "none"; }
int sat_multi_db_upload_packed(sat_multi *m, int n, const int32_t *o, const int64_t *c, const uint8_t *t, const float *d)
{ (void)m; (void)n; (void)o; (void)c; (void)t; (void)d; return SAT_ENODEVICE; }
int sat_multi_shards(const sat_multi *m, int32_t *begin) { (void)m; (void)begin; return SAT_ENODEVICE; }
int sat_multi_queries_set(sat_multi *m, int nq, const int32_t *n1s, const uint8_t *qt, const float *qd, int pitch,
                          const uint8_t *ty, uint32_t first)
{ (void)m; (void)nq; (void)n1s; (void)qt; (void)qd; (void)pitch; (void)ty; (void)first; return SAT_ENODEVICE; }
int sat_multi_search(sat_multi *m, int lorder, int lsoln, int maxstart, int32_t *scores, int32_t *ssemaps, double *ms)
{ (void)m; (void)lorder; (void)lsoln; (void)maxstart; (void)scores; (void)ssemaps; (void)ms; return SAT_ENODEVICE; }
int sat_multi_search_topk(sat_multi *m, int lorder, int lsoln, int maxstart, int k, sat_hit *hits, int32_t *ssemaps, double *ms)
{ (void)m; (void)lorder; (void)lsoln; (void)maxstart; (void)k; (void)hits; (void)ssemaps; (void)ms; return SAT_ENODEVICE; }
unsigned long long sat_multi_stat_d2h_bytes(const sat_multi *m) { (void)m; return 0ull; }
