#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../bayhunter_amd/csrc/bh_libm.h"
static uint64_t rng = 88172645463325252ull;
static double urand(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (rng >> 11) * (1.0 / 9007199254740992.0); }
int main(void)
{
    const double *tab = (const double *)bhp_sincos_tab_bits;
    long bad_s = 0, bad_c = 0, bad_e = 0, bad_bl = 0, n = 0;
    const double ranges[][2] = {{0, 1e-8}, {0, 0.126}, {0.1, 0.9}, {0.8, 2.5}, {2.4, 10}, {0, 40}, {0, 1000}, {1e3, 1e5}, {1e5, 1.05e8}};
    for (int r = 0; r < 9; ++r)
        for (long i = 0; i < 20000000; ++i) {
            double x = ranges[r][0] + urand() * (ranges[r][1] - ranges[r][0]);
            if (i & 1) x = -x;
            double s0, c0, s1, c1;
            sincos(x, &s0, &c0);
            if (!bhp_sincos(x, &s1, &c1, tab)) { printf("out of range %g\n", x); return 1; }
            double s2, c2;
            bhp_sincos_bl(x, &s2, &c2, tab);
            if (memcmp(&s1, &s2, 8) || memcmp(&c1, &c2, 8)) { if (bad_bl < 5) printf("branch-light mismatch x=%.17g %a %a | %a %a\n", x, s1, s2, c1, c2); ++bad_bl; }
            if (memcmp(&s0, &s1, 8)) { if (bad_s < 5) printf("sin mismatch x=%.17g %a %a\n", x, s0, s1); ++bad_s; }
            if (memcmp(&c0, &c1, 8)) { if (bad_c < 5) printf("cos mismatch x=%.17g %a %a\n", x, c0, c1); ++bad_c; }
            ++n;
        }
    printf("sincos: %ld inputs, sin mismatches %ld, cos mismatches %ld, branch-light vs branchy %ld\n", n, bad_s, bad_c, bad_bl);
    n = 0;
    const double er[][2] = {{-1e-10, 0}, {-1, 0}, {-40, 0}, {-130, 0}, {-500, 500}, {0, 1}};
    for (int r = 0; r < 6; ++r)
        for (long i = 0; i < 20000000; ++i) {
            double x = er[r][0] + urand() * (er[r][1] - er[r][0]);
            if (!bhp_exp_in_domain(x)) continue;
            double e0 = exp(x), e1 = bhp_exp_core(x, bhp_exp_tab);
            if (memcmp(&e0, &e1, 8)) { if (bad_e < 5) printf("exp mismatch x=%.17g %a %a\n", x, e0, e1); ++bad_e; }
            ++n;
        }
    printf("exp: %ld inputs, mismatches %ld\n", n, bad_e);
    return (bad_s || bad_c || bad_e || bad_bl) ? 2 : 0;
}
