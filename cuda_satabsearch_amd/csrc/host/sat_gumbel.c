/* sat_gumbel.c - see sat_gumbel.h */
#include <math.h>
#include "sat_gumbel.h"

static const double k_euler_gamma = 0.5772156649015328606;

static double pi_over_sqrt6(void)
{
    return M_PI / sqrt(6.0);
}

double sat_norm2(int score, int n1, int n2)
{
    return 2.0 * score / ((double)(n1 + n2));
}

double sat_z_gumbel_trunc(double norm2score)
{
    int x = (int)norm2score; /* the reference's implicit double -> int */
    double mu = SAT_GUMBEL_A + SAT_GUMBEL_B * k_euler_gamma;
    double sigma = pi_over_sqrt6() * SAT_GUMBEL_B;
    return (x - mu) / sigma;
}

double sat_pv_gumbel(double z)
{
    return 1 - exp(-exp(-(pi_over_sqrt6() * z + k_euler_gamma)));
}
