/*
 * sat_shard.h - cutting a database into contiguous shards of equal COST (plain C, host side).
 *
 * The reference is single-GPU (cudaSaTabsearch.cu:790 "TODO allow multiple GPUs").  Here the
 * database is sharded contiguously over the GPUs of a node in file order.
 * Real databases are size sorted (scripts/convdb2.py -s), and scoring an entry of 96 SSEs costs
 * four times an entry of 32, so equal entry COUNTS would leave the last GPU with several times the
 * work of the first: the cut points are placed by cumulative cost instead.
 */
#ifndef SAT_SHARD_H
#define SAT_SHARD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Relative cost of scoring one database entry of `order` SSEs (1.0 at 32 SSEs): kernel time per
 * scoring measured on one MI355X with a 32-SSE query at r = 128 on fixed-order databases
 * (scripts/cost_sweep.py, profiles/r02_cost_by_order.txt), linear between the measured orders.
 * Other query sizes shift the curve by less than +-25 % at the ends (8-SSE query: 0.63 .. 7.7,
 * 101-SSE query: 0.60 .. 3.5 against 0.55 .. 5.7 here).
 */
double sat_entry_cost(int order);

/*
 * Cut n_entries entries (orders[e], file order) into nshards contiguous shards of near-equal
 * total cost: begin[g] .. begin[g+1]-1 is shard g, begin[0] = 0, begin[nshards] = n_entries.
 * Every shard is non-empty when n_entries >= nshards.  Returns 0, or -1 on bad arguments.
 */
int sat_shard_cuts(int n_entries, const int32_t *orders, int nshards, int32_t *begin);

#ifdef __cplusplus
}
#endif
#endif
