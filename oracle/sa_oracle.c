/*
 * sa_oracle.c - CPU ORACLE (test infrastructure, see sa_oracle.h).
 *
 * Restates, function by function, the reference host search:
 *   sa_oracle_pair_score   <- tscord      cudaSaTabsearch_kernel.cu:306-332
 *   sa_oracle_full_score   <- tmscord     cudaSaTabsearch_kernel.cu:396-440
 *   move_delta             <- deltasd     cudaSaTabsearch_kernel.cu:502-535
 *   random_initial_map     <- thinit      cudaSaTabsearch_kernel.cu:588-648
 *   pick_free_same_type    <- randtypeind cudaSaTabsearch_kernel.cu:677-714
 *   sa_oracle_search       <- sa_tabsearch_host, cudaSaTabsearch_kernel.cu:924-1233
 *                             with blockDim = gridDim = 1 (:864-868)
 * Constants: saparams.h:26-43 and EPS cudaSaTabsearch_kernel.cu:67.
 *
 * Build with -ffp-contract=off: every float/double operation below is meant
 * to be the single IEEE operation the reference's -O3 (no fast-math) host
 * build performs (Makefile:50-58, 93-94).
 */
#include <math.h>
#include <stdio.h>
#include <string.h>
#include "sa_oracle.h"

int sa_oracle_trace = 0;

#define SA_EPS            1.1e-7       /* keeps (u - EPS) * n below n when u == 1 */
#define SA_INIT_MATCHPROB 0.5
static const float k_max_sse_dist_diff = 4.0f;   /* MXSSED */
static const float k_temp0 = 10.0f;              /* TEMP0  */
static const float k_alpha = 0.95f;              /* ALPHA  */

/* ------------------------------------------------------------------ streams */

#define LCG_MASK ((UINT64_C(1) << 48) - 1)

uint64_t sa_oracle_srand48(long seedval)
{
    /* glibc srand48: high 32 bits of X from the seed, low 16 bits 0x330E */
    return ((((uint64_t)(uint32_t)seedval) << 16) | UINT64_C(0x330E)) & LCG_MASK;
}

static double lcg_drand48(uint64_t *x)
{
    *x = (UINT64_C(0x5DEECE66D) * (*x) + UINT64_C(0xB)) & LCG_MASK;
    return ldexp((double)*x, -48);
}

void sa_oracle_philox4x32_10(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4])
{
    /* Philox4x32-10 (Salmon et al. 2011); multipliers and Weyl key increments as
     * rocrand_philox4x32_10.h ROCRAND_PHILOX_M4x32_0/1, ROCRAND_PHILOX_W32_0/1 */
    uint32_t c0 = counter[0], c1 = counter[1], c2 = counter[2], c3 = counter[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; round++) {
        uint64_t p0 = (uint64_t)UINT32_C(0xD2511F53) * c0;
        uint64_t p1 = (uint64_t)UINT32_C(0xCD9E8D57) * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += UINT32_C(0x9E3779B9);
        k1 += UINT32_C(0xBB67AE85);
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

float sa_oracle_u32_to_uniform(uint32_t v)
{
    const float two_pow_m32 = 2.3283064e-10f; /* ROCRAND_2POW32_INV == 2^-32 */
    float f = (float)v;
    f = f * two_pow_m32;
    return two_pow_m32 + f;
}

/* draw source for one restart chain */
typedef struct chain_draws {
    sa_oracle_rng *rng;
    uint32_t key[2];
    uint32_t counter_z, counter_w;
    uint32_t block[4];
    int      block_id;
} chain_draws;

static void chain_begin(chain_draws *c, sa_oracle_rng *rng, uint32_t db_ordinal, uint32_t restart)
{
    c->rng = rng;
    if (rng->mode != SA_RNG_DRAND48) {
        uint64_t seed_q = rng->seed + ((uint64_t)rng->query_ordinal << 32);
        c->key[0] = (uint32_t)seed_q;
        c->key[1] = (uint32_t)(seed_q >> 32);
        c->counter_z = db_ordinal;
        c->counter_w = restart;
        c->block_id = -1;
    }
}

static uint32_t chain_word(chain_draws *c, int block, int word)
{
    if (block != c->block_id) {
        uint32_t counter[4] = { (uint32_t)block, 0u, c->counter_z, c->counter_w };
        sa_oracle_philox4x32_10(counter, c->key, c->block);
        c->block_id = block;
    }
    return c->block[word];
}

/* uniform float; (block, word) addresses the draw in PHILOX mode and is
 * ignored by the sequential stream */
static float chain_draw(chain_draws *c, int block, int word)
{
    if (c->rng->mode == SA_RNG_DRAND48)
        return (float)lcg_drand48(&c->rng->lcg);
    return sa_oracle_u32_to_uniform(chain_word(c, block, word));
}

float sa_oracle_u16_to_uniform(uint32_t v16)
{
    return (float)(v16 + 1u) * 1.52587890625e-05f;   /* (v + 1) * 2^-16 in (0, 1], exact */
}

/* uniform float from one 16-bit half (half 1 = high) of a Philox word: the two index draws of
 * an SA step share a word (sa_oracle.h) */
static float chain_draw16(chain_draws *c, int block, int word, int half)
{
    if (c->rng->mode == SA_RNG_DRAND48)
        return (float)lcg_drand48(&c->rng->lcg);
    uint32_t w = chain_word(c, block, word);
    if (c->rng->mode == SA_RNG_PHILOX32)          /* comparison mode: a whole word per index draw (sa_oracle.h) */
        return sa_oracle_u32_to_uniform(w);
    return sa_oracle_u16_to_uniform(half ? (w >> 16) : (w & 0xFFFFu));
}

/* ------------------------------------------------------------------ scoring */

int sa_oracle_pair_score(uint8_t x, uint8_t y)
{
    int same_hi = ((x ^ y) & 0xF0) == 0;
    int same_lo = ((x ^ y) & 0x0F) == 0;
    if (same_hi && same_lo) return 2;
    if (same_hi || same_lo) return 1;
    return -2;
}

static int dist_compatible(float d1, float d2)
{
    return fabsf(d1 - d2) <= k_max_sse_dist_diff;
}

int sa_oracle_full_score(const sa_oracle_query *q, const uint8_t *tab2, const float *dmat2,
                         int pitch2, const int *ssemap)
{
    int total = 0;
    for (int i = 0; i < q->n; i++) {
        int j = ssemap[i];
        if (j < 0) continue;
        for (int k = i + 1; k < q->n; k++) {
            int l = ssemap[k];
            if (l < 0) continue;
            if (dist_compatible(q->dmat[(size_t)i * q->pitch + k], dmat2[(size_t)j * pitch2 + l]))
                total += sa_oracle_pair_score(q->tab[(size_t)i * q->pitch + k],
                                              tab2[(size_t)j * pitch2 + l]);
        }
    }
    return total;
}

/* score change when query SSE `i` moves from db SSE old_j to new_j (-1 = none) */
static int move_delta(const sa_oracle_query *q, const uint8_t *tab2, const float *dmat2,
                      int pitch2, const int *ssemap, int i, int old_j, int new_j)
{
    int delta = 0;
    for (int k = 0; k < q->n; k++) {
        int l = ssemap[k];
        if (l < 0 || k == i) continue;
        float d1 = q->dmat[(size_t)i * q->pitch + k];
        uint8_t t1 = q->tab[(size_t)i * q->pitch + k];
        if (old_j >= 0 && l != old_j && dist_compatible(d1, dmat2[(size_t)old_j * pitch2 + l]))
            delta -= sa_oracle_pair_score(t1, tab2[(size_t)old_j * pitch2 + l]);
        if (new_j >= 0 && l != new_j && dist_compatible(d1, dmat2[(size_t)new_j * pitch2 + l]))
            delta += sa_oracle_pair_score(t1, tab2[(size_t)new_j * pitch2 + l]);
    }
    return delta;
}

/* ------------------------------------------------------------------ moves */

/* Random order-preserving, type-respecting initial map.  Each query SSE is
 * considered with probability 1/2; the first failed type search ends the
 * whole construction without further draws (kernel.cu:633-638). */
static void random_initial_map(const sa_oracle_query *q, const uint8_t *types2, int n2,
                               int *ssemap, int *revmap, chain_draws *draws)
{
    for (int i = 0; i < q->n; i++) ssemap[i] = -1;
    for (int j = 0; j < n2; j++) revmap[j] = -1;
    int j = 0;
    for (int i = 0; i < q->n; i++) {
        float u = chain_draw(draws, i >> 2, i & 3);
        if (!(u < SA_INIT_MATCHPROB))
            continue;
        while (j < n2 && types2[j] != q->ssetypes[i])
            j++;
        if (j >= n2)
            return;
        ssemap[i] = j;
        revmap[j] = i;
        j++;
    }
}

/* uniformly chosen free db SSE of type `type` in [lo, hi); -1 when none.
 * No draw unless there are at least two candidates (kernel.cu:701-712). */
static int pick_free_same_type(const uint8_t *types2, const int *revmap, int lo, int hi,
                               uint8_t type, chain_draws *draws, int block, int word)
{
    int count = 0, only = -1;
    for (int j = lo; j < hi; j++)
        if (types2[j] == type && revmap[j] < 0) {
            count++;
            only = j;
        }
    if (count == 0) return -1;
    if (count == 1) return only;
    float u = chain_draw16(draws, block, word, 0);
    unsigned pick = (unsigned)(int)((u - SA_EPS) * count);
    for (int j = lo; j < hi; j++)
        if (types2[j] == type && revmap[j] < 0) {
            if (pick == 0) return j;
            pick--;
        }
    return -1; /* not reached */
}

/* image of the nearest mapped query SSE at or before i, else n2 (kernel.cu:1055-1063) */
static int lower_bound_image(const int *ssemap, int i, int n2)
{
    for (int k = i; k >= 0; k--)
        if (ssemap[k] >= 0)
            return ssemap[k];
    return n2;
}

/* image of the nearest mapped query SSE after i; the last SSE gets n2; when no
 * later SSE is mapped the reference yields -1, an empty range (kernel.cu:1064-1077) */
static int upper_bound_image(const int *ssemap, int i, int n1, int n2)
{
    if (i == n1 - 1)
        return n2;
    for (int k = i + 1; k < n1; k++)
        if (ssemap[k] != -1)
            return ssemap[k];
    return -1;
}

/* ------------------------------------------------------------------ search */

void sa_oracle_search(const sa_oracle_query *q, int dbsize, const int *orders,
                      const int64_t *db_ordinal,
                      const uint8_t *tabs, const float *dmats, int pitch,
                      int lorder, int lsoln, int maxstart,
                      sa_oracle_rng *rng, int *outscore, int *outssemap)
{
    const int n1 = q->n;
    int ssemap[SA_MAXDIM], bestmap[SA_MAXDIM], revmap[SA_MAXDIM];
    uint8_t types2[SA_MAXDIM];
    chain_draws draws;

    for (int d = 0; d < dbsize; d++) {
        const int n2 = orders[d];
        const uint8_t *tab2 = tabs + (size_t)d * pitch * pitch;
        const float *dmat2 = dmats + (size_t)d * pitch * pitch;
        const uint32_t ordinal = (uint32_t)(db_ordinal ? db_ordinal[d] : d);

        for (int j = 0; j < n2; j++)
            types2[j] = tab2[(size_t)j * pitch + j];
        for (int i = 0; i < n1; i++)
            bestmap[i] = -1;

        int maxscore = -99999;
        for (int restart = 0; restart < maxstart; restart++) {
            chain_begin(&draws, rng, ordinal, (uint32_t)restart);
            random_initial_map(q, types2, n2, ssemap, revmap, &draws);
            int score = sa_oracle_full_score(q, tab2, dmat2, pitch, ssemap);
            if (score > maxscore) {
                maxscore = score;
                memcpy(bestmap, ssemap, (size_t)n1 * sizeof(int));
            }

            float temp = k_temp0;
            for (int iter = 0; iter < SA_MAXITER; iter++) {
                /* PHILOX: one block per two steps, two words per step (sa_oracle.h) */
                /* PHILOX32 (comparison mode): one block per step, a whole word per draw */
                const int wide = rng->mode == SA_RNG_PHILOX32;
                const int block = SA_PHILOX_STEP_BLOCK0 + (wide ? iter : (iter >> 1));
                const int word_a = wide ? 0 : 2 * (iter & 1), word_c = wide ? 1 : word_a, word_b = wide ? 2 : word_a + 1;
                float u = chain_draw16(&draws, block, word_a, 1);
                int ssei = (int)((u - SA_EPS) * n1);

                int startj = 0, endj = n2;
                if (lorder) {
                    startj = lower_bound_image(ssemap, ssei, n2);
                    endj = upper_bound_image(ssemap, ssei, n1, n2);
                }
                int newj = pick_free_same_type(types2, revmap, startj, endj,
                                               q->ssetypes[ssei], &draws, block, word_c);
                if (sa_oracle_trace) {
                    /* same line format as the reference DEBUG build (kernel.cu:1092-1096) */
                    printf("%d %d %d %d %d %d %d\n", 0, restart, iter, ssei, startj, endj, newj);
                    printf("%d ssemap: ", 0);
                    for (int t = 0; t < n1; t++)
                        printf("%d ", ssemap[t]);
                    printf("\n");
                }
                int oldj = ssemap[ssei];
                int delta = move_delta(q, tab2, dmat2, pitch, ssemap, ssei, oldj, newj);
                int newscore = score + delta;

                /* best-so-far is taken from the PROPOSED state, before and
                 * regardless of acceptance (kernel.cu:1136-1155) */
                if (newscore > maxscore) {
                    maxscore = newscore;
                    if (lsoln) {
                        memcpy(bestmap, ssemap, (size_t)n1 * sizeof(int));
                        bestmap[ssei] = newj >= 0 ? newj : -1;
                    }
                }

                /* a Metropolis draw is consumed on every step (kernel.cu:1161-1166) */
                u = chain_draw(&draws, block, word_b);
                if (expf((float)delta / temp) > u) {
                    score = newscore;
                    if (oldj >= 0)
                        revmap[oldj] = -1;
                    if (newj >= 0)
                        revmap[newj] = ssei;
                    ssemap[ssei] = newj >= 0 ? newj : -1;
                }
                temp *= k_alpha;
            }
        }
        outscore[d] = maxscore;
        if (lsoln && outssemap)
            for (int i = 0; i < n1; i++)
                outssemap[(size_t)d * SA_MAXDIM + i] = bestmap[i];
    }
}
