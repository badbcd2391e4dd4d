/*
 * sat_host_search.c - CLI "-c" host mode.  See sat_host_search.h.
 *
 * Same formulation as the GPU kernel (csrc/sat_sa_kernel.hpp): db entry and query as
 * {distance, code} cells with a NaN "null SSE" row/column for unmatched SSEs, free
 * db SSEs and matched query SSEs as 128-bit sets.  What it computes follows the
 * reference host path: thinit kernel.cu:588-648, tmscord :396-440, neighbour window
 * :1053-1086, randtypeind :677-714, deltasd :502-535, best tracking :1136-1155,
 * Metropolis :1161-1187, cooling :1189; constants saparams.h:26-43, EPS kernel.cu:67.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "sat_host_search.h"

#define EPS_DRAW 1.1e-7
#define MAXITER 100

typedef struct { float d; uint8_t code; } cell_t;
typedef struct { uint64_t w[2]; } set128;

/* ---- drand48-compatible stream (glibc: X <- 0x5DEECE66D * X + 0xB mod 2^48) ---- */
void sat_host_stream_seed(sat_host_stream *st, long seedval)
{
    st->x = ((((uint64_t)(uint32_t)seedval) << 16) | 0x330Eu) & 0xFFFFFFFFFFFFull;
}
static inline float next_uniform(sat_host_stream *st)
{
    st->x = (0x5DEECE66Dull * st->x + 0xBull) & 0xFFFFFFFFFFFFull;
    return (float)ldexp((double)st->x, -48);
}

/* ---- 128-bit sets ---- */
static inline set128 set_below(int pos)          /* bits [0,pos) */
{
    set128 s;
    s.w[0] = pos <= 0 ? 0 : (pos >= 64 ? ~0ull : ((1ull << pos) - 1));
    s.w[1] = pos <= 64 ? 0 : (pos >= 128 ? ~0ull : ((1ull << (pos - 64)) - 1));
    return s;
}
static inline void set_add(set128 *s, int p) { s->w[p >> 6] |= 1ull << (p & 63); }
static inline void set_del(set128 *s, int p) { s->w[p >> 6] &= ~(1ull << (p & 63)); }
static inline int set_lowest(set128 s)
{
    if (s.w[0]) return __builtin_ctzll(s.w[0]);
    if (s.w[1]) return 64 + __builtin_ctzll(s.w[1]);
    return -1;
}
static inline int set_highest(set128 s)
{
    if (s.w[1]) return 127 - __builtin_clzll(s.w[1]);
    if (s.w[0]) return 63 - __builtin_clzll(s.w[0]);
    return -1;
}
static inline int set_count(set128 s) { return __builtin_popcountll(s.w[0]) + __builtin_popcountll(s.w[1]); }
static inline int set_nth(set128 s, int r)       /* r-th set bit, ascending */
{
    for (int w = 0; w < 2; w++) {
        uint64_t v = s.w[w];
        int c = __builtin_popcountll(v);
        if (r < c) {
            while (r--) v &= v - 1;
            return 64 * w + __builtin_ctzll(v);
        }
        r -= c;
    }
    return -1;
}

static inline int code_score(uint8_t x, uint8_t y)      /* tscord */
{
    int hi = ((x ^ y) & 0xF0) == 0, lo = ((x ^ y) & 0x0F) == 0;
    return hi + lo == 0 ? -2 : hi + lo;
}
static inline int term(cell_t q, cell_t d)
{
    return fabsf(q.d - d.d) <= 4.0f ? code_score(q.code, d.code) : 0;
}

int sat_host_search(const sat_struct_set *db, const int *entries, int n,
                    const sat_struct_set *queries, int qs,
                    int lorder, int lsoln, int maxstart, sat_host_stream *stream,
                    int32_t *scores, int32_t *ssemaps)
{
    const int n1 = queries->order[qs];
    const int dp = SAT_MAXDIM + 1;                       /* cell pitch incl. the null SSE */
    cell_t *Q = (cell_t *)malloc(sizeof(cell_t) * (size_t)n1 * n1 + 1);
    cell_t *D = (cell_t *)malloc(sizeof(cell_t) * (size_t)dp * dp);
    if (!Q || !D) { free(Q); free(D); return -1; }

    uint8_t qtype[SAT_MAXDIM];
    {
        const uint8_t *t = queries->tab + queries->cell_off[qs];
        const float *d = queries->dist + queries->cell_off[qs];
        for (int i = 0; i < n1; i++)
            for (int j = 0; j <= i; j++) {
                int64_t c = (int64_t)i * (i + 1) / 2 + j;
                cell_t cell = { i == j ? NAN : d[c], t[c] };   /* NaN diagonal drops k == i */
                Q[i * n1 + j] = Q[j * n1 + i] = cell;
                if (i == j) qtype[i] = t[c];
            }
    }

    for (int e = 0; e < n; e++) {
        const int s = entries[e];
        const int n2 = db->order[s], NULLJ = n2;
        set128 typeset[4] = { { { 0, 0 } }, { { 0, 0 } }, { { 0, 0 } }, { { 0, 0 } } };
        {
            const uint8_t *t = db->tab + db->cell_off[s];
            const float *d = db->dist + db->cell_off[s];
            for (int i = 0; i <= n2; i++)
                for (int j = 0; j <= i; j++) {
                    cell_t cell = { NAN, 0 };
                    if (i < n2) {
                        int64_t c = (int64_t)i * (i + 1) / 2 + j;
                        cell.d = d[c];
                        cell.code = t[c];
                        if (i == j) set_add(&typeset[t[c] & 3], i);
                    }
                    D[i * dp + j] = D[j * dp + i] = cell;
                }
        }

        uint8_t map[SAT_MAXDIM], best[SAT_MAXDIM];
        memset(best, NULLJ, sizeof(best));
        int maxscore = -99999;

        for (int restart = 0; restart < maxstart; restart++) {
            set128 mapped = { { 0, 0 } }, occ = { { 0, 0 } };
            memset(map, NULLJ, (size_t)n1);
            for (int i = 0, j = 0; i < n1; i++) {                 /* random initial map */
                if (!(next_uniform(stream) < 0.5))
                    continue;
                set128 c = typeset[qtype[i] & 3], b = set_below(j);
                c.w[0] &= ~b.w[0]; c.w[1] &= ~b.w[1];
                int jj = set_lowest(c);
                if (jj < 0) break;                                /* no more draws */
                map[i] = (uint8_t)jj;
                set_add(&mapped, i);
                set_add(&occ, jj);
                j = jj + 1;
            }
            int score = 0;
            for (int i = 0; i < n1; i++) {
                if (map[i] == NULLJ) continue;
                const cell_t *row = D + map[i] * dp;
                for (int k = i + 1; k < n1; k++)
                    score += term(Q[i * n1 + k], row[map[k]]);
            }
            if (score > maxscore) { maxscore = score; memcpy(best, map, (size_t)n1); }

            float temp = 10.0f;
            for (int iter = 0; iter < MAXITER; iter++) {
                const int i = (int)((next_uniform(stream) - EPS_DRAW) * n1);
                const int oldj = map[i];
                int lo = 0, hi = n2;
                if (lorder) {
                    set128 upto = set_below(i + 1), a = mapped, b = mapped;
                    a.w[0] &= upto.w[0]; a.w[1] &= upto.w[1];
                    b.w[0] &= ~upto.w[0]; b.w[1] &= ~upto.w[1];
                    int p = set_highest(a), q = set_lowest(b);
                    lo = p < 0 ? n2 : map[p];
                    hi = i == n1 - 1 ? n2 : (q < 0 ? -1 : map[q]);
                }
                set128 cand = typeset[qtype[i] & 3], bl = set_below(lo), bh = set_below(hi);
                cand.w[0] &= ~occ.w[0] & bh.w[0] & ~bl.w[0];
                cand.w[1] &= ~occ.w[1] & bh.w[1] & ~bl.w[1];
                const int cnt = set_count(cand);
                int newj = NULLJ;
                if (cnt == 1) newj = set_lowest(cand);
                else if (cnt > 1) newj = set_nth(cand, (int)((next_uniform(stream) - EPS_DRAW) * cnt));

                int delta = 0;
                const cell_t *qrow = Q + i * n1, *orow = D + oldj * dp, *nrow = D + newj * dp;
                for (int k = 0; k < n1; k++) {
                    if (map[k] == NULLJ) continue;
                    delta += term(qrow[k], nrow[map[k]]) - term(qrow[k], orow[map[k]]);
                }
                const int newscore = score + delta;
                if (newscore > maxscore) {
                    maxscore = newscore;
                    if (lsoln) { memcpy(best, map, (size_t)n1); best[i] = (uint8_t)newj; }
                }
                if (expf((float)delta / temp) > next_uniform(stream)) {
                    score = newscore;
                    map[i] = (uint8_t)newj;
                    if (oldj != NULLJ) set_del(&occ, oldj);
                    if (newj != NULLJ) { set_add(&occ, newj); set_add(&mapped, i); }
                    else set_del(&mapped, i);
                }
                temp *= 0.95f;
            }
        }
        scores[e] = maxscore;
        if (lsoln && ssemaps)
            for (int i = 0; i < n1; i++)
                ssemaps[(size_t)e * SAT_MAXDIM + i] = best[i] == NULLJ ? -1 : best[i];
    }
    free(Q);
    free(D);
    return 0;
}
