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
    long bad_l = 0, bad_p = 0, skipped = 0;
    n = 0;
    const double lr[][2] = {{0.9375, 1.0648}, {1.0, 1.07}, {0.5, 2.0}, {1e-300, 1e-290}, {0.0, 10.0}, {1.0, 1e6}, {1e6, 1e300}};
    for (int r = 0; r < 7; ++r)
        for (long i = 0; i < 20000000; ++i) {
            double x = lr[r][0] + urand() * (lr[r][1] - lr[r][0]);
            if (r == 6) x = exp(urand() * 690.0);
            double l0 = log(x), l1;
            if (!bhp_log(x, &l1, bhp_log_data)) { ++skipped; continue; }
            if (memcmp(&l0, &l1, 8)) { if (bad_l < 5) printf("log mismatch x=%.17g %a %a\n", x, l0, l1); ++bad_l; }
            ++n;
        }
    printf("log: %ld inputs (%ld outside the table paths), mismatches %ld\n", n, skipped, bad_l);
    n = 0; skipped = 0;
    for (int r = 0; r < 4; ++r)
        for (long i = 0; i < 20000000; ++i) {
            float x = (float)(r == 0 ? 0.9 + 0.2 * urand() : r == 1 ? 0.5 + 1.5 * urand() : r == 2 ? 1e-3 + 50.0 * urand() : exp(-80.0 + 160.0 * urand()));
            float y = (r < 2 && (i & 1)) ? -2.275f : (float)(-12.0 + 24.0 * urand());
            float p0 = powf(x, y), p1;
            if (!bhp_powf(x, y, &p1, bhp_powf_log2_data, bhp_exp2f_data)) { ++skipped; continue; }
            if (memcmp(&p0, &p1, 4)) { if (bad_p < 5) printf("powf mismatch x=%.9g y=%.9g %a %a\n", x, y, p0, p1); ++bad_p; }
            ++n;
        }
    printf("powf: %ld inputs (%ld outside the main path), mismatches %ld\n", n, skipped, bad_p);
    return (bad_s || bad_c || bad_e || bad_bl || bad_l || bad_p) ? 2 : 0;
}
