/*
 * sa_oracle.h - CPU ORACLE for the simulated-annealing tableau search.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may build, link, load or run anything in
 * oracle/.  The shipped GPU path (cuda_satabsearch_amd/) never does.
 *
 * What it is: a plain-C restatement of the reference's host ("-c") search,
 * nvcc_src_current/cudaSaTabsearch_kernel.cu:804-1236 built without -DCUDA
 * (sa_tabsearch_host) together with its helpers tscord (:306-332), tmscord
 * (:396-440), deltasd (:502-535), thinit (:588-648), randtypeind (:677-714).
 *
 * Parity status: PINNED.  In SA_RNG_DRAND48 mode the oracle CLI built on this
 * file reproduces, byte for byte, the stdout of the reference's own sources
 * compiled here (oracle/_ref, see oracle/Makefile) on every example input the
 * reference ships, and the recorded 2013 run
 * old/nvcc_src_cuda5/cpu_cudaSaTabsearch.o1462445 (tests/test_oracle_golden.py).
 *
 * Two random streams behind the same algorithm:
 *   SA_RNG_DRAND48  one sequential glibc-drand48-compatible LCG stream for the
 *                   whole run (seeded like srand48(1234), cudaSaTabsearch.cu:871)
 *                   carried across restarts, db entries and queries: the
 *                   reference "-c" semantics, not reproducible in parallel.
 *   SA_RNG_PHILOX   counter-based Philox4x32-10 laid out exactly as rocRAND's
 *                   rocrand_init(seed, subsequence, offset) does
 *                   (rocrand_philox4x32_10.h), addressed by
 *                   (seed, query ordinal, db ordinal, restart, draw slot): every
 *                   restart chain has its own stream, so the result does not
 *                   depend on how chains are scheduled.  This is the mode the
 *                   GPU kernel is compared against, bit for bit.
 */
#ifndef SA_ORACLE_H
#define SA_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SA_MAXDIM   111      /* saparams.h:15 */
#define SA_MAXITER  100      /* saparams.h:33 */

#define SA_RNG_DRAND48 0
#define SA_RNG_PHILOX  1
/* Comparison mode, not a parity target: as SA_RNG_PHILOX but the two index draws of an SA step take a whole
 * 32-bit word each at curand_uniform's resolution (one block per step: word 0 = which query SSE, word 1 = which
 * candidate, word 2 = Metropolis).  The kernel's 16-bit index draws give bins of 590 or 591 values per index for
 * n = 111 (at most 0.17 % apart) where 32-bit draws are uniform to 2^-25; tests/test_oracle_golden.py checks that
 * the two modes differ by no more than two seeds of the reference's own stream do. */
#define SA_RNG_PHILOX32 2

/* Philox draw-slot layout shared with the GPU kernel (DESIGN.md "random stream"):
 *   block b of chain (query, db entry, restart) = Philox4x32-10 with
 *     key     = (lo32(seed_q), hi32(seed_q)),  seed_q = seed + ((uint64)query << 32)
 *     counter = (b, 0, db_ordinal, restart)
 *   initial-map draw i (0 <= i < n1)  -> block i/4, word i%4, uniform = 2^-32 + float(v) * 2^-32
 *   SA step t (0 <= t < 100)          -> block 32 + t/2, words a = 2*(t%2), b = a + 1:
 *                                        word a, high 16 bits = which query SSE,
 *                                        word a, low 16 bits  = which candidate (only consumed
 *                                        when >= 2 candidates), each as uniform = (v16 + 1) * 2^-16;
 *                                        word b = Metropolis, uniform = 2^-32 + float(v) * 2^-32
 *   (one block per two steps: the index draws pick among at most 111 items, 16 bits each are
 *   plenty; the block count is what the GPU kernel pays for)
 */
#define SA_PHILOX_STEP_BLOCK0 32

typedef struct sa_oracle_rng {
    int      mode;           /* SA_RNG_DRAND48 or SA_RNG_PHILOX                      */
    uint64_t lcg;            /* DRAND48: 48-bit state, updated by every call         */
    uint64_t seed;           /* PHILOX                                               */
    uint32_t query_ordinal;  /* PHILOX                                               */
} sa_oracle_rng;

typedef struct sa_oracle_query {
    int            n;        /* order n1                                             */
    int            pitch;    /* row pitch (cells) of tab and dmat                    */
    const uint8_t *tab;      /* dense symmetric code matrix                          */
    const float   *dmat;     /* dense symmetric distance matrix                      */
    const uint8_t *ssetypes; /* [n] SSE type of each query SSE                       */
} sa_oracle_query;

/* state after srand48(seedval) */
uint64_t sa_oracle_srand48(long seedval);

/* one Philox4x32-10 block: out[4] = philox(counter[4], key[2]) */
void sa_oracle_philox4x32_10(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]);

/* uint32 -> float in (0,1] exactly as rocrand_uniform.h:65-68 */
float sa_oracle_u32_to_uniform(uint32_t v);

/* 16-bit draw -> float in (0,1]: (v16 + 1) * 2^-16 */
float sa_oracle_u16_to_uniform(uint32_t v16);

/*
 * Search `dbsize` db structures with one query.
 *   orders[d]            order n2 of entry d
 *   db_ordinal[d]        PHILOX only: the entry's ordinal in db file order
 *                        (NULL means d)
 *   tabs, dmats          dense, entry d at d*pitch*pitch, row pitch `pitch`
 *   outscore[d]          best score over all restarts
 *   outssemap[d*111+i]   (may be NULL) best map, -1 = unmatched; written only
 *                        when lsoln != 0, as in the reference (kernel.cu:1223-1233)
 */
void sa_oracle_search(const sa_oracle_query *q, int dbsize, const int *orders,
                      const int64_t *db_ordinal,
                      const uint8_t *tabs, const float *dmats, int pitch,
                      int lorder, int lsoln, int maxstart,
                      sa_oracle_rng *rng, int *outscore, int *outssemap);

/* when non-zero, every SA step prints the proposal and the current map on stdout in
 * the format of the reference's DEBUG build (step-level golden fixture) */
extern int sa_oracle_trace;

/* building blocks, exported for unit tests */
int sa_oracle_pair_score(uint8_t x, uint8_t y);
int sa_oracle_full_score(const sa_oracle_query *q, const uint8_t *tab2, const float *dmat2,
                         int pitch2, const int *ssemap);

#ifdef __cplusplus
}
#endif
#endif /* SA_ORACLE_H */
