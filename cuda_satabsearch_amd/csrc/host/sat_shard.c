/* sat_shard.c - see sat_shard.h */
#include <stdlib.h>
#include "sat_shard.h"

/* ns per scoring, 32-SSE query, r = 128, one MI355X (scripts/cost_sweep.py in round 3, after the triangle cell
 * layout for entries above 48 SSEs: profiles/r03_cost_by_order.txt; an entry of 111 SSEs costs 4.2 entries of 32,
 * where the full cell matrix made it 5.4; re-measured after the 64-bit order windows) */
static const struct { int order; double ns; } k_cost[] = {
    { 4, 47.8 }, { 8, 55.4 }, { 12, 63.6 }, { 16, 71.8 }, { 20, 77.5 }, { 24, 81.6 }, { 28, 86.6 }, { 32, 91.6 },
    { 40, 120.1 }, { 48, 146.6 }, { 56, 175.0 }, { 64, 191.6 }, { 72, 241.2 }, { 80, 264.0 }, { 88, 292.1 }, { 96, 321.6 },
    { 104, 370.1 }, { 111, 383.2 },
};
#define K_COST_N ((int)(sizeof(k_cost) / sizeof(k_cost[0])))
#define K_COST_UNIT 91.6

double sat_entry_cost(int order)
{
    if (order <= k_cost[0].order) return k_cost[0].ns / K_COST_UNIT;
    for (int i = 1; i < K_COST_N; i++)
        if (order <= k_cost[i].order) {
            const double t = (double)(order - k_cost[i - 1].order) / (double)(k_cost[i].order - k_cost[i - 1].order);
            return (k_cost[i - 1].ns + t * (k_cost[i].ns - k_cost[i - 1].ns)) / K_COST_UNIT;
        }
    return k_cost[K_COST_N - 1].ns / K_COST_UNIT;
}

int sat_shard_cuts(int n_entries, const int32_t *orders, int nshards, int32_t *begin)
{
    if (n_entries < 0 || nshards < 1 || !begin || (n_entries > 0 && !orders)) return -1;
    double total = 0.0;
    for (int e = 0; e < n_entries; e++) total += sat_entry_cost(orders[e]);
    begin[0] = 0;
    double cum = 0.0;
    int e = 0;
    for (int g = 1; g < nshards; g++) {
        /* shard g starts at the first entry at which the running cost has reached g / nshards of
         * the total, taking an entry when more than half of it lies below the target */
        const double target = total * (double)g / (double)nshards;
        while (e < n_entries) {
            const double c = sat_entry_cost(orders[e]);
            if (cum + 0.5 * c > target) break;
            cum += c;
            e++;
        }
        /* keep every shard non-empty when there are enough entries */
        const int min_begin = begin[g - 1] + 1, max_begin = n_entries - (nshards - g);
        int b = e;
        if (n_entries >= nshards) {
            if (b < min_begin) b = min_begin;
            if (b > max_begin) b = max_begin;
        }
        while (e < b) cum += sat_entry_cost(orders[e++]);
        begin[g] = b;
    }
    begin[nshards] = n_entries;
    return 0;
}
