/*
 * sat_gumbel.h - score normalisation and Gumbel tail statistics for the
 * "name rawscore norm2score z-score p-value" output row.
 *
 * Follows nvcc_src_current/gumbelstats.c:50-58 (z), :69-72 (p), :91-94 (norm2)
 * and the location/scale constants gumbelstats.h:22-23.
 */
#ifndef SAT_GUMBEL_H
#define SAT_GUMBEL_H
#ifdef __cplusplus
extern "C" {
#endif

#define SAT_GUMBEL_A 0.3780327676087335
#define SAT_GUMBEL_B 0.3582596175507505

/* norm2 = 2*score / (n1 + n2), in double */
double sat_norm2(int score, int n1, int n2);

/*
 * z of a Gumbel(a,b) variate.  The reference prototype takes an `int x`
 * (gumbelstats.h:26) and is called with the double norm2 score
 * (cudaSaTabsearch.cu:446), so the score is truncated toward zero first:
 * z only takes the values of x = ..., -1, 0, 1, 2, ...  Kept, because the
 * printed z and p columns depend on it.
 */
double sat_z_gumbel_trunc(double norm2score);

/* p = 1 - exp(-exp(-(pi/sqrt(6) * z + euler_gamma))) */
double sat_pv_gumbel(double z);

#ifdef __cplusplus
}
#endif
#endif
