// bayhunter_amd/csrc/rf_kernel.hip -- receiver-function synthesis on gfx950.
//
// Replaces the reference's rfmini path  synrf_cwrap (rfmini/wrap.cpp:58-80) -> synrf
// (rfmini/synrf.cpp:16-55) -> calcresp/calcresp_core (rfmini/greens.cpp:400-756) -> compute_rf
// (:343-398) -> iftr (:136-158) -> ccfork (rfmini/fork.cpp:11-60) for a batch of models.
//
// Two kernels per batch (all FP64 VALU, no MFMA -- complex 2x2 recursions, not contractions):
//   rf_coef_kernel      lane = model.  Earth-flattening (rfmini/model.cpp:221-252), the
//                       frequency-independent interface reflection/transmission matrices
//                       (greens.cpp:19-112), the free-surface displacement matrix (:307-322) and
//                       the Z/R -> P/SV rotation constants (:324-341), written to a per-model
//                       coefficient record in HBM (L2-resident: 3.4 KB/model at 10 layers) -- and the
//                       model's Nyquist bin (j = nsamp/2), the one frequency that does not fit the
//                       nsamp/2 = 4 x 256 bins the next kernel's lanes share out.
//   rf_synth_kernel     workgroup = model.  The nsamp/2 frequencies of a model are independent
//                       (greens.cpp:528): lane t runs the Kennett/Mueller top-down reflectivity
//                       recursion (:196-224) for bins t, t+256, ...; the coefficient record is
//                       wave-uniform (scalar loads).  Deconvolution + Gauss filter + time shift
//                       (:343-398) are fused and the sample goes, with its Hermitian mirror, straight
//                       to its bit-reversed place in LDS: the spectrum never exists in HBM.  Then the
//                       inverse FFT of length nsamp in LDS (32 KB at nsamp = 2048 + 8 KB twiddles:
//                       four workgroups per CU);
//                       only the first nkeep real samples are written (rfmini_modrf.py:142).
#include "bh_device.h"
#include "bh_tuning.h"
#include <cmath>
#include <cstdlib>

namespace {
// w / a beyond which the Gauss low-pass exp(-(w/a)^2 / 4) is below RF_CUT = 1e-17: 2 sqrt(17 ln 10)
constexpr double RF_CUT_WA = 12.5132;

struct cd {
    double re, im;
};
__device__ __forceinline__ cd C(double r, double i = 0.0) { return cd{r, i}; }
__device__ __forceinline__ cd operator+(cd a, cd b) { return cd{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cd operator-(cd a, cd b) { return cd{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cd operator-(cd a) { return cd{-a.re, -a.im}; }
__device__ __forceinline__ cd operator*(cd a, cd b)
{
    return cd{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
__device__ __forceinline__ cd operator*(double s, cd a) { return cd{s * a.re, s * a.im}; }
__device__ __forceinline__ cd operator*(cd a, double s) { return cd{s * a.re, s * a.im}; }
__device__ __forceinline__ cd operator+(double s, cd a) { return cd{s + a.re, a.im}; }
__device__ __forceinline__ cd conj(cd a) { return cd{a.re, -a.im}; }
__device__ __forceinline__ cd crecip(cd b)
{
    const double den = b.re * b.re + b.im * b.im;
    return cd{b.re / den, -b.im / den};
}
__device__ __forceinline__ cd operator/(cd a, cd b) { return a * crecip(b); }
__device__ __forceinline__ cd csqrt_d(cd z)
{
    if (z.re == 0.0 && z.im == 0.0) return cd{0.0, z.im};
    const double r = hypot(z.re, z.im);
    if (z.re >= 0.0) {
        const double t = sqrt(0.5 * (r + z.re));
        return cd{t, z.im / (2.0 * t)};
    }
    const double t = sqrt(0.5 * (r - z.re));
    return cd{fabs(z.im) / (2.0 * t), copysign(t, z.im)};
}
// ---- elementary functions of the per-frequency recursion ------------------------------------------
// The receiver function is held to 1e-4 of its peak (north_star), the tests to 1e-9; the library's correctly
// rounded division / sqrt / hypot and its sincos / exp with their special-case paths are two thirds of the
// instructions of a layer step.  These are the same functions to ~2 ulp for the arguments that occur here
// (finite, normal range): hardware seed + two Newton steps, Cody-Waite reduction + minimax polynomials
// (coefficients: fdlibm's __kernel_sin / __kernel_cos).
// (v_rcp_f64 / v_rsq_f64 are good to ~2^-23 or better; one Newton step squares that: ~1e-14 relative, five orders below
// the 1e-9 the tests assert; RF_NEWTON2 restores the second step)
__device__ __forceinline__ double rcp_nr(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
#ifdef RF_NEWTON2
    r = fma(r, fma(-x, r, 1.0), r);
#endif
    return r;
}
// 1/sqrt(x), x > 0
__device__ __forceinline__ double rsq_nr(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = fma(y, fma(-hx * y, y, 0.5), y);
#ifdef RF_NEWTON2
    y = fma(y, fma(-hx * y, y, 0.5), y);
#endif
    return y;
}
// sin and cos of a moderate argument (|x| < 2^20: |k| * 2^-107 of reduction error)
__device__ __forceinline__ void sincos_cw(double x, double *sn, double *cs)
{
    const double k = __builtin_rint(x * 0x1.45f306dc9c883p-1);
    double r = fma(-k, 0x1.921fb54442d18p+0, x);
    r = fma(-k, 0x1.1a62633145c07p-54, r);
    const int q = (int)k;
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double s = fma(r * z, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double c = fma(z * z, pc, fma(-0.5, z, 1.0));
    const double so = (q & 1) ? c : s, co = (q & 1) ? s : c;
    *sn = (q & 2) ? -so : so;
    *cs = ((q + 1) & 2) ? -co : co;
}
// e^x, Taylor of degree 13 on |r| <= ln2/2 (remainder 4e-18), scaled by ldexp (under/overflow as ldexp's)
__device__ __forceinline__ double exp_cw(double x)
{
    // |x| beyond the exponent range: the reduction below would lose all bits of r; clamped, the result is ldexp's
    // 0 / inf as for the library's exp (x < -745 -> 0, x > 710 -> inf).  NaN passes through (both tests are false).
    x = (x < -800.0) ? -800.0 : x;
    x = (x > 800.0) ? 800.0 : x;
    const double k = __builtin_rint(x * 0x1.71547652b82fep+0);
    double r = fma(-k, 0x1.62e42fefa39efp-1, x);
    r = fma(-k, 0x1.abc9e3b39803fp-56, r);
    double p = fma(r, 1.0 / 6227020800.0, 1.0 / 479001600.0);
    p = fma(r, p, 1.0 / 39916800.0);
    p = fma(r, p, 1.0 / 3628800.0);
    p = fma(r, p, 1.0 / 362880.0);
    p = fma(r, p, 1.0 / 40320.0);
    p = fma(r, p, 1.0 / 5040.0);
    p = fma(r, p, 1.0 / 720.0);
    p = fma(r, p, 1.0 / 120.0);
    p = fma(r, p, 1.0 / 24.0);
    p = fma(r, p, 1.0 / 6.0);
    p = fma(r, p, 0.5);
    p = fma(r, p, 1.0);
    p = fma(r, p, 1.0);
    const double kk = fmin(fmax(k, -2000.0), 2000.0);
    return ldexp(p, (int)kk);
}
__device__ __forceinline__ cd crecip_f(cd b)
{
    const double i = rcp_nr(b.re * b.re + b.im * b.im);
    return cd{b.re * i, -b.im * i};
}
// principal square root, branch-free; 0 for z = 0
__device__ __forceinline__ cd csqrt_f(cd z)
{
    const double n2 = z.re * z.re + z.im * z.im;
    const double r = n2 * rsq_nr(n2);                 // |z|
    const double y = 0.5 * (r + fabs(z.re));
    const double it = rsq_nr(y), t = y * it;          // t = sqrt(y), it = 1/t
    const double u = 0.5 * z.im * it;                 // im / (2t)
    cd o = (z.re >= 0.0) ? cd{t, u} : cd{fabs(u), copysign(t, z.im)};
    if (!(n2 > 0.0)) o = cd{n2, n2};                  // 0 -> 0, NaN -> NaN
    return o;
}
// e^z
__device__ __forceinline__ cd cexp_f(cd z)
{
    double s, c;
    sincos_cw(z.im, &s, &c);
    const double m = exp_cw(z.re);
    return cd{m * c, m * s};
}

struct cm2 {
    cd c11, c12, c21, c22;
};
__device__ __forceinline__ cm2 operator*(const cm2 &x, const cm2 &y)
{
    return cm2{x.c11 * y.c11 + x.c12 * y.c21, x.c11 * y.c12 + x.c12 * y.c22,
               x.c21 * y.c11 + x.c22 * y.c21, x.c21 * y.c12 + x.c22 * y.c22};
}
__device__ __forceinline__ cm2 operator+(const cm2 &x, const cm2 &y)
{
    return cm2{x.c11 + y.c11, x.c12 + y.c12, x.c21 + y.c21, x.c22 + y.c22};
}

// ---- per-model coefficient record (doubles) ---------------------------------------------------
//  [0] nlay  [1] p (s/km)  [2] do_decomp  [3] bad (NaN propagation of t0 / decomp)
//  (the last two doubles of the record are spare)
//  [4..7] m11 m12 m21 m22 (rotation)   [8..15] 2*h matrix (4 complex)
//  [16..23] free-surface ru (4 complex)
//  [24 + 8*l ...]           layer l = 0..Lmax-1 : 1/vp^2, 1/vs^2, h (flattened), 1/(pi qp), 1/(2 qp), 1/(pi qs), 1/(2 qs), -
//  [24 + 8*Lmax + 32*i ...] interface below layer i (i = 0..Lmax-2): rd, td, ru, tu (4 complex each)
//  [24 + 40*Lmax]           1 when every interface matrix and the free-surface ru are REAL (no post-critical wave at any
//                           interface: vertical slownesses of the elastic, real velocities all real) -- the synthesis kernel
//                           then multiplies by real matrices (rf_one_frequency<true>)
constexpr int REC_HEAD = 24;
__host__ __device__ inline size_t rec_doubles(int Lmax) { return REC_HEAD + 8 * (size_t)Lmax + 32 * (size_t)Lmax + 2; }

__device__ __forceinline__ void store_cm2(double *p, const cm2 &m)
{
    p[0] = m.c11.re; p[1] = m.c11.im; p[2] = m.c12.re; p[3] = m.c12.im;
    p[4] = m.c21.re; p[5] = m.c21.im; p[6] = m.c22.re; p[7] = m.c22.im;
}
__device__ __forceinline__ cm2 load_cm2(const double *p)
{
    return cm2{cd{p[0], p[1]}, cd{p[2], p[3]}, cd{p[4], p[5]}, cd{p[6], p[7]}};
}
// a real 2x2 matrix: the real parts of a stored complex one whose imaginary parts are zero
struct rm2 {
    double c11, c12, c21, c22;
};
__device__ __forceinline__ rm2 load_rm2(const double *p) { return rm2{p[0], p[2], p[4], p[6]}; }
__device__ __forceinline__ cm2 operator*(const rm2 &x, const cm2 &y)
{
    return cm2{cd{x.c11 * y.c11.re + x.c12 * y.c21.re, x.c11 * y.c11.im + x.c12 * y.c21.im},
               cd{x.c11 * y.c12.re + x.c12 * y.c22.re, x.c11 * y.c12.im + x.c12 * y.c22.im},
               cd{x.c21 * y.c11.re + x.c22 * y.c21.re, x.c21 * y.c11.im + x.c22 * y.c21.im},
               cd{x.c21 * y.c12.re + x.c22 * y.c22.re, x.c21 * y.c12.im + x.c22 * y.c22.im}};
}
__device__ __forceinline__ cm2 operator*(const cm2 &x, const rm2 &y)
{
    return cm2{cd{x.c11.re * y.c11 + x.c12.re * y.c21, x.c11.im * y.c11 + x.c12.im * y.c21},
               cd{x.c11.re * y.c12 + x.c12.re * y.c22, x.c11.im * y.c12 + x.c12.im * y.c22},
               cd{x.c21.re * y.c11 + x.c22.re * y.c21, x.c21.im * y.c11 + x.c22.im * y.c21},
               cd{x.c21.re * y.c12 + x.c22.re * y.c22, x.c21.im * y.c12 + x.c22.im * y.c22}};
}
__device__ __forceinline__ cm2 operator+(const rm2 &x, const cm2 &y)
{
    return cm2{cd{x.c11 + y.c11.re, y.c11.im}, cd{x.c12 + y.c12.re, y.c12.im}, cd{x.c21 + y.c21.re, y.c21.im},
               cd{x.c22 + y.c22.re, y.c22.im}};
}
// sum of |imaginary parts|: 0 exactly when the matrix is real
__device__ __forceinline__ double imag_mass(const cm2 &m) { return fabs(m.c11.im) + fabs(m.c12.im) + fabs(m.c21.im) + fabs(m.c22.im); }
// number of entries that are NaN or +-inf (a bit test: under -ffp-contract=fast "x - x" of an expression is not
// reliably zero -- one copy may be contracted into an fma, the other not)
__device__ __forceinline__ double nf1(double x) { return __builtin_isfinite(x) ? 0.0 : 1.0; }
__device__ __forceinline__ double nonfinite(const cm2 &m)
{
    return nf1(m.c11.re) + nf1(m.c11.im) + nf1(m.c12.re) + nf1(m.c12.im) + nf1(m.c21.re) + nf1(m.c21.im) + nf1(m.c22.re) +
           nf1(m.c22.im);
}

// greens.cpp:19-85 (P/SV part)
__device__ void interface_coeffs(double u, double vp1, double vs1, double rho1, double vp2,
                                 double vs2, double rho2, cm2 &rd, cm2 &td, cm2 &ru, cm2 &tu)
{
    const double mue1 = rho1 * vs1 * vs1, mue2 = rho2 * vs2 * vs2;
    const double c = 2. * (mue1 - mue2), u2 = u * u, cu2 = c * u2;
    const cd a1 = conj(csqrt_d(C(1. / (vp1 * vp1) - u2)));
    const cd a2 = conj(csqrt_d(C(1. / (vp2 * vp2) - u2)));
    const cd b1 = conj(csqrt_d(C(1. / (vs1 * vs1) - u2)));
    const cd b2 = conj(csqrt_d(C(1. / (vs2 * vs2) - u2)));
    const double t1 = cu2 - rho1 + rho2;
    const double t2 = cu2 - rho1;
    const double t3 = cu2 + rho2;
    const cd t4 = t3 * a1 - t2 * a2;
    const cd a1a2b1b2 = (c * c * u2) * a1 * a2 * b1 * b2;
    {
        const cd d1 = (t1 * t1 * u2) + (t2 * t2) * a2 * b2 + (rho1 * rho2) * a2 * b1;
        const cd d2 = a1a2b1b2 + (t3 * t3) * a1 * b1 + (rho1 * rho2) * a1 * b2;
        const cd t5 = crecip(d1 + d2);
        const cd t7 = (2. * rho1) * t5;
        const cd mix = (t1 * t3) + (c * t2) * a2 * b2;
        rd.c11 = (d2 - d1) * t5;                                              // rpp
        rd.c21 = (-2. * u) * a1 * t5 * mix;                                   // rps
        td.c11 = a1 * t7 * (t3 * b1 - t2 * b2);                               // tpp
        td.c21 = -(a1 * t7) * u * (t1 + c * (a2 * b1));                       // tps
        rd.c22 = (d2 - d1 - (2. * rho1 * rho2) * (a1 * b2 - a2 * b1)) * t5;   // rss
        rd.c12 = (2. * u) * b1 * t5 * mix;                                    // rsp
        td.c22 = b1 * t7 * t4;                                                // tss
        td.c12 = b1 * t7 * u * (t1 + c * (a1 * b2));                          // tsp
    }
    {
        const cd d1 = (t1 * t1 * u2) + (t3 * t3) * a1 * b1 + (rho1 * rho2) * a1 * b2;
        const cd d2 = a1a2b1b2 + (t2 * t2) * a2 * b2 + (rho1 * rho2) * a2 * b1;
        const cd t5 = crecip(d1 + d2);
        const cd t7 = (2. * rho2) * t5;
        const cd mix = (t1 * t2) + (c * t3) * a1 * b1;
        ru.c11 = (d2 - d1) * t5;
        ru.c21 = (2. * u) * a2 * t5 * mix;
        tu.c11 = a2 * t7 * (t3 * b1 - t2 * b2);
        tu.c21 = -(a2 * t7) * u * (t1 + c * (a1 * b2));
        ru.c22 = (d2 - d1 - (2. * rho1 * rho2) * (a2 * b1 - a1 * b2)) * t5;
        ru.c12 = (-2. * u) * b2 * t5 * mix;
        tu.c22 = b2 * t7 * t4;
        tu.c12 = b2 * t7 * u * (t1 + c * (a2 * b1));
    }
}

// One frequency of one model: greens.cpp:528-585 + :343-398.  `rec` may be wave-uniform.
// REALC: the record's interface matrices are real (its flag says so): products with them take half the operations of a
// complex 2x2 product -- 76 of the ~580 vector instructions of a layer step.  Same values (the general form multiplies
// the same numbers by imaginary parts that are exactly zero).
template <bool REALC>
__device__ __forceinline__ cd rf_one_frequency(const double *__restrict__ rec, int Lmax, int j,
                                               double dw, double qg, double gauss, double tshift,
                                               int waveno)
{
    const int nlay = (int)rec[0];
    const double p2 = rec[1] * rec[1];
    const double w = dw * j;
    const double wref = 2. * M_PI * 1.0;
    const double lgw = j ? log(w / wref) : 0.0;
    const double nanv = __longlong_as_double(0x7ff8000000000000ll);
    if (rec[3] != 0.0) return cd{nanv, nanv};

    cm2 nb, q, g;
    nb = q = g = cm2{C(0), C(0), C(0), C(0)};
    const double *lay = rec + REC_HEAD;
    const double *ifc = rec + REC_HEAD + 8 * Lmax;
    for (int i = 1; i < nlay; ++i) {
        const double *L = lay + 8 * (i - 1);
        // complex velocities with causal Q, v (a + i b) with a = 1 + ln(w/wref)/(pi Q), b = 1/(2Q) (greens.cpp:539-543);
        // 1/v^2 - p^2 with 1/(a + ib)^2 = (a - ib)^2 / (a^2 + b^2)^2; vertical slownesses; phases e^{-i w d q}
        const double d = L[2];
        const double ap = fma(lgw, L[3], 1.0), bp = L[4], as = fma(lgw, L[5], 1.0), bs = L[6];
        const double np = ap * ap + bp * bp, ns = as * as + bs * bs;
        const double ip = L[0] * rcp_nr(np * np), is = L[1] * rcp_nr(ns * ns);
        const cd plc = csqrt_f(cd{ip * (ap * ap - bp * bp) - p2, -2.0 * ip * ap * bp});
        const cd slc = csqrt_f(cd{is * (as * as - bs * bs) - p2, -2.0 * is * as * bs});
        const double wd = w * d;
        const cd e11 = cexp_f(cd{wd * plc.im, -wd * plc.re});
        const cd e22 = cexp_f(cd{wd * slc.im, -wd * slc.re});
        // Mueller (1985) top-down recursion, greens.cpp:196-224
        cm2 nt;
        if (i == 1)
            nt = load_cm2(rec + 16);
        else {
            const double *ic = ifc + 32 * (i - 2); // interface above layer i
            if (REALC) nt = load_rm2(ic + 16) + (load_rm2(ic + 8) * nb) * q;
            else nt = load_cm2(ic + 16) + (load_cm2(ic + 8) * nb) * q; // ru[i] + td[i]*nb[i-1]*q
        }
        const cd e12 = e11 * e22;
        nb = cm2{nt.c11 * (e11 * e11), nt.c12 * e12, nt.c21 * e12, nt.c22 * (e22 * e22)};
        const double *icn = ifc + 32 * (i - 1); // interface below layer i
        cm2 rn;
        if (REALC) rn = load_rm2(icn) * nb;
        else rn = load_cm2(icn) * nb;
        const cm2 m = cm2{C(1.) - rn.c11, -rn.c12, -rn.c21, C(1.) - rn.c22};
        const cd idet = crecip_f(m.c11 * m.c22 - m.c12 * m.c21);
        const cm2 minv = cm2{idet * m.c22, -(idet * m.c12), -(idet * m.c21), idet * m.c11};
        if (REALC) q = minv * load_rm2(icn + 24);
        else q = minv * load_cm2(icn + 24);
        if (i == 1)
            g = cm2{e11 * q.c11, e11 * q.c12, e22 * q.c21, e22 * q.c22};
        else {
            const cm2 ge = cm2{g.c11 * e11, g.c12 * e22, g.c21 * e11, g.c22 * e22};
            g = ge * q;
        }
    }
    const cm2 hm = load_cm2(rec + 8); // already 2*h
    cd cr, cz;
    if (waveno == 0) {
        cr = hm.c11 * g.c11 + hm.c12 * g.c21;
        cz = hm.c21 * g.c11 + hm.c22 * g.c21;
    } else {
        cr = hm.c11 * g.c12 + hm.c12 * g.c22;
        cz = hm.c21 * g.c12 + hm.c22 * g.c22;
    }
    // (the common factor exp(i w t0) of greens.cpp:583-585 cancels in cr*conj(cz)/|cz|^2)
    if (rec[2] != 0.0) {
        const cd cx = cz * rec[4] + cr * rec[5];
        const cd cy = cz * rec[6] + cr * rec[7];
        cz = cx;
        cr = cy;
    }
    if (waveno == 1) {
        const cd t = cz;
        cz = cr;
        cr = t;
    }
    const double idenom = rcp_nr(cz.re * cz.re + cz.im * cz.im);
    const cd num = cr * conj(cz);
    const cd v = cd{num.re * idenom, num.im * idenom};
    double wa = w / gauss;
    wa = (wa > 50.0) ? 50.0 : wa;
    const cd cq = qg * cexp_f(cd{-0.25 * (wa * wa), -w * tshift});
    return v * cq;
}

__global__ __launch_bounds__(256) void rf_coef_kernel(RfKernelArgs A)
{
    const int ib = blockIdx.x * blockDim.x + threadIdx.x;
    if (ib >= A.B) return;
    const int Lmax = A.Lmax;
    double *rec = A.coef + (size_t)ib * rec_doubles(Lmax);
    const int nlay = A.nlay[ib];
    const ptrdiff_t base = (ptrdiff_t)ib * A.sb;
    const double R = 6371.0;
    const double p = A.p_s_per_deg * 0.00899; // wrap.cpp:55
    const double p2 = p * p;
    double bad = 0.0;
    double nf = 0.0; // stays 0 while every coefficient of the record is finite
    double im = 0.0; // stays 0 while every interface matrix is real

    // top-layer quantities before flattening (q = 1 for the top layer anyway)
    const double vp0 = A.vp[base], vs0 = A.vs[base];
    // rfmini_modrf.py:125-130 and wrap.cpp:13,73-74
    const double kap = vp0 / vs0;
    const double poisson = (2 - kap * kap) / (2 - 2 * (kap * kap));
    const double nsv = (A.nsv > 0.0) ? A.nsv : vs0;
    const double vptop = nsv * sqrt((1. - poisson) / (.5 - poisson));
    const double vstop = nsv;

    // flatten layer by layer (model.cpp:221-252); z = depth of the layer top = running sum of h
    double ztop = 0.0, t0 = 0.0;
    double pvp = 0, pvs = 0, prh = 0; // previous (upper) layer, flattened
    for (int l = 0; l < nlay; ++l) {
        const ptrdiff_t o = base + (ptrdiff_t)l * A.sl;
        // thickness the way synrf.cpp:28-32 forms it from the depths z = cumsum(h)
        // (rfmini_modrf.py:119-123): z[l+1] - z[l]; the half-space gets -1
        const double znext = ztop + A.h[o];
        const bool half = (l == nlay - 1);
        double hh = half ? -1.0 : (znext - ztop);
        double vp = A.vp[o], vs = A.vs[o], rh = A.rho[o];
        const double qp = A.qp ? A.qp[o] : 500.0, qs = A.qs ? A.qs[o] : 225.0;
        const double zb = ztop + hh;
        double r = R - ztop;
        double q = R / r;
        const double zf = R * log(q);
        vp *= q;
        vs *= q;
        rh /= q;
        const bool lower_halfspace = !(hh > 0.0) && !(vp < 1.0 && rh < 0.1);
        if (!lower_halfspace) {
            r = R - zb;
            q = R / r;
            hh = R * log(q) - zf;
        }
        double *lay = rec + REC_HEAD + 8 * l;
        lay[0] = 1.0 / (vp * vp); lay[1] = 1.0 / (vs * vs); lay[2] = hh;
        lay[3] = 1.0 / (M_PI * qp); lay[4] = 1.0 / (2.0 * qp); lay[5] = 1.0 / (M_PI * qs); lay[6] = 1.0 / (2.0 * qs);
        for (int k = 0; k < 7; ++k) nf += nf1(lay[k]);
        // direct-wave delay (greens.cpp:510-526); only its NaN-ness can reach the RF
        const double vv = (A.waveno == 0) ? vp : vs;
        t0 += hh * sqrt(1. / (vv * vv) - p2);
        if (l == 0) {
            // free surface, greens.cpp:87-112 (plain sqrt) and displacement matrix :307-322
            const cd a = csqrt_d(C(1. / (vp * vp) - p2));
            const cd b = csqrt_d(C(1. / (vs * vs) - p2));
            const double t1 = 2. * vs * vs;
            const double t2 = t1 * p2 - 1.;
            const cd d1 = C(t2 * t2);
            const cd d2 = (t1 * t1 * p2) * a * b;
            const cd d = d1 + d2;
            const cd t3 = C(2. * t1 * p * t2) / d;
            cm2 ru;
            ru.c11 = (d2 - d1) / d;
            ru.c12 = -(b * t3);
            ru.c21 = a * t3;
            ru.c22 = ru.c11;
            store_cm2(rec + 16, ru);
            const double vp2 = vp * vp, vs2 = vs * vs, x = 1. - 2. * vs2 * p2;
            const cd a1 = conj(a), b1 = conj(b);
            const cd qq = crecip(C(x * x) + (4. * vs2 * vs2 * p2) * a1 * b1);
            cm2 hm;
            hm.c11 = qq * a1 * b1 * (2. * vs2 * p);
            hm.c12 = qq * b1 * (1. - 2. * vs2 * p2);
            hm.c21 = qq * a1 * (1. - 2. * vs2 * p2);
            hm.c22 = -(qq * a1 * b1 * (2. * vs2 * p));
            hm.c11 = 2.0 * hm.c11; hm.c12 = 2.0 * hm.c12; hm.c21 = 2.0 * hm.c21; hm.c22 = 2.0 * hm.c22;
            store_cm2(rec + 8, hm);
            nf += nonfinite(ru) + nonfinite(hm);
            im += imag_mass(ru);
            (void)vp2;
        } else {
            cm2 rd, td, ru, tu;
            interface_coeffs(p, pvp, pvs, prh, vp, vs, rh, rd, td, ru, tu);
            double *ic = rec + REC_HEAD + 8 * Lmax + 32 * (l - 1);
            store_cm2(ic, rd);
            store_cm2(ic + 8, td);
            store_cm2(ic + 16, ru);
            store_cm2(ic + 24, tu);
            nf += nonfinite(rd) + nonfinite(td) + nonfinite(ru) + nonfinite(tu);
            im += imag_mass(rd) + imag_mass(td) + imag_mass(ru) + imag_mass(tu);
        }
        pvp = vp; pvs = vs; prh = rh;
        ztop = znext;
    }
    if (t0 != t0) bad = 1.0;
    if (nlay < 2) bad = 1.0; // the reference reads an uninitialised matrix here
    // rotation Z/R -> P/SV with REAL vertical slownesses (greens.cpp:324-341)
    double do_decomp = 0.0, m11 = 0, m12 = 0, m21 = 0, m22 = 0;
    if (vstop > 0.01 && fabs(p) > 0.0001) {
        do_decomp = 1.0;
        const double aa = sqrt(1. / (vptop * vptop) - p * p), bb = sqrt(1. / (vstop * vstop) - p * p);
        m11 = -(2 * vstop * vstop * p * p - 1.) / (vptop * aa);
        m12 = 2. * p * vstop * vstop / vptop;
        m21 = -2. * p * vstop;
        m22 = (1. - 2. * vstop * vstop * p * p) / (vstop * bb);
        nf += nf1(m11) + nf1(m12) + nf1(m21) + nf1(m22);
    }
    // A non-finite coefficient makes every bin of the reference's spectrum non-finite, hence the whole trace (the
    // inverse FFT sums all bins); with the spectral cut-off the bins above it are not formed here, so the record
    // carries the flag instead of relying on the propagation
    if (nf != 0.0) bad = 1.0;
    rec[0] = (double)nlay; rec[1] = p; rec[2] = do_decomp; rec[3] = bad;
    rec[4] = m11; rec[5] = m12; rec[6] = m21; rec[7] = m22;
    rec[REC_HEAD + 40 * (size_t)Lmax] = (im == 0.0) ? 1.0 : 0.0;
}

// The same record with LP lanes per model, lane l = layer l (LP = 16 or 32 >= Lmax): the serial kernel above is a
// chain of ~10 interface computations per lane with 64 wavefronts on the whole chip (0.10 ms at B = 4096, a sixth of
// the receiver function's time); here every layer's flattening and the interface above it are one lane's work.
// Same operations per layer: the depth of a layer's top is the reference's running sum, formed in its order; the
// direct-wave delay is summed by the model's first lane in layer order.  (The Nyquist bin is the synthesis kernel's.)
template <int LP>
__device__ __forceinline__ void rf_coef_layers_body(const RfKernelArgs &A)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int ibr = t / LP, l = t % LP, lane = threadIdx.x & 63, lbase = lane - l;
    const bool vm = ibr < A.B;
    const int ib = vm ? ibr : 0;
    const int Lmax = A.Lmax;
    double *rec = A.coef + (size_t)ib * rec_doubles(Lmax);
    const int nlay = A.nlay[ib];
    const ptrdiff_t base = (ptrdiff_t)ib * A.sb;
    const double R = 6371.0;
    const double p = A.p_s_per_deg * 0.00899; // wrap.cpp:55
    const double p2 = p * p;
    const bool on = vm && l < nlay;
    const int lc = (l < nlay) ? l : (nlay > 0 ? nlay - 1 : 0);

    // this layer, flattened (model.cpp:221-252); ztop = the running sum of the thicknesses above, in the loop's order
    double ztop = 0.0;
    for (int i = 0; i < lc; ++i) ztop = ztop + A.h[base + (ptrdiff_t)i * A.sl];
    const ptrdiff_t o = base + (ptrdiff_t)lc * A.sl;
    const double znext = ztop + A.h[o];
    const bool half = (lc == nlay - 1);
    double hh = half ? -1.0 : (znext - ztop);
    double vp = A.vp[o], vs = A.vs[o], rh = A.rho[o];
    const double qp = A.qp ? A.qp[o] : 500.0, qs = A.qs ? A.qs[o] : 225.0;
    {
        const double zb = ztop + hh;
        double r = R - ztop;
        double q = R / r;
        const double zf = R * log(q);
        vp *= q;
        vs *= q;
        rh /= q;
        const bool lower_halfspace = !(hh > 0.0) && !(vp < 1.0 && rh < 0.1);
        if (!lower_halfspace) {
            r = R - zb;
            q = R / r;
            hh = R * log(q) - zf;
        }
    }
    double nf = 0.0; // stays 0 while every coefficient this lane writes is finite
    double im = 0.0; // stays 0 while the interface matrices this lane writes are real
    if (on) {
        double *lay = rec + REC_HEAD + 8 * l;
        lay[0] = 1.0 / (vp * vp); lay[1] = 1.0 / (vs * vs); lay[2] = hh;
        lay[3] = 1.0 / (M_PI * qp); lay[4] = 1.0 / (2.0 * qp); lay[5] = 1.0 / (M_PI * qs); lay[6] = 1.0 / (2.0 * qs);
        for (int k = 0; k < 7; ++k) nf += nf1(lay[k]);
    }
    // direct-wave delay (greens.cpp:510-526); only its NaN-ness can reach the RF
    const double vv = (A.waveno == 0) ? vp : vs;
    const double term = hh * sqrt(1. / (vv * vv) - p2);
    // the layer above, from the neighbouring lane
    const double pvp = __shfl(vp, lane - 1), pvs = __shfl(vs, lane - 1), prh = __shfl(rh, lane - 1);
    if (on && l > 0) {
        cm2 rd, td, ru, tu;
        interface_coeffs(p, pvp, pvs, prh, vp, vs, rh, rd, td, ru, tu);
        double *ic = rec + REC_HEAD + 8 * Lmax + 32 * (l - 1);
        store_cm2(ic, rd);
        store_cm2(ic + 8, td);
        store_cm2(ic + 16, ru);
        store_cm2(ic + 24, tu);
        nf += nonfinite(rd) + nonfinite(td) + nonfinite(ru) + nonfinite(tu);
        im = imag_mass(rd) + imag_mass(td) + imag_mass(ru) + imag_mass(tu);
    }
    double t0 = 0.0, nfall = 0.0, imall = 0.0;
    for (int i = 0; i < LP; ++i) {
        const double ti = __shfl(term, lbase + i), ni = __shfl(nf, lbase + i), ii = __shfl(im, lbase + i);
        if (i < nlay) {
            t0 += ti;
            nfall += ni;
            imall += ii;
        }
    }
    if (vm && l == 0) {
        // free surface, greens.cpp:87-112 (plain sqrt) and displacement matrix :307-322
        const cd a = csqrt_d(C(1. / (vp * vp) - p2));
        const cd b = csqrt_d(C(1. / (vs * vs) - p2));
        const double t1 = 2. * vs * vs;
        const double t2 = t1 * p2 - 1.;
        const cd d1 = C(t2 * t2);
        const cd d2 = (t1 * t1 * p2) * a * b;
        const cd d = d1 + d2;
        const cd t3 = C(2. * t1 * p * t2) / d;
        cm2 ru;
        ru.c11 = (d2 - d1) / d;
        ru.c12 = -(b * t3);
        ru.c21 = a * t3;
        ru.c22 = ru.c11;
        store_cm2(rec + 16, ru);
        const double vs2 = vs * vs, x = 1. - 2. * vs2 * p2;
        const cd a1 = conj(a), b1 = conj(b);
        const cd qq = crecip(C(x * x) + (4. * vs2 * vs2 * p2) * a1 * b1);
        cm2 hm;
        hm.c11 = qq * a1 * b1 * (2. * vs2 * p);
        hm.c12 = qq * b1 * (1. - 2. * vs2 * p2);
        hm.c21 = qq * a1 * (1. - 2. * vs2 * p2);
        hm.c22 = -(qq * a1 * b1 * (2. * vs2 * p));
        hm.c11 = 2.0 * hm.c11; hm.c12 = 2.0 * hm.c12; hm.c21 = 2.0 * hm.c21; hm.c22 = 2.0 * hm.c22;
        store_cm2(rec + 8, hm);
        // top-layer quantities before flattening (q = 1 for the top layer anyway): rfmini_modrf.py:125-130, wrap.cpp:13,73-74
        const double vp0 = A.vp[base], vs0 = A.vs[base];
        const double kap = vp0 / vs0;
        const double poisson = (2 - kap * kap) / (2 - 2 * (kap * kap));
        const double nsv = (A.nsv > 0.0) ? A.nsv : vs0;
        const double vptop = nsv * sqrt((1. - poisson) / (.5 - poisson));
        const double vstop = nsv;
        double bad = 0.0;
        if (t0 != t0) bad = 1.0;
        if (nlay < 2) bad = 1.0; // the reference reads an uninitialised matrix here
        // rotation Z/R -> P/SV with REAL vertical slownesses (greens.cpp:324-341)
        double do_decomp = 0.0, m11 = 0, m12 = 0, m21 = 0, m22 = 0;
        if (vstop > 0.01 && fabs(p) > 0.0001) {
            do_decomp = 1.0;
            const double aa = sqrt(1. / (vptop * vptop) - p * p), bb = sqrt(1. / (vstop * vstop) - p * p);
            m11 = -(2 * vstop * vstop * p * p - 1.) / (vptop * aa);
            m12 = 2. * p * vstop * vstop / vptop;
            m21 = -2. * p * vstop;
            m22 = (1. - 2. * vstop * vstop * p * p) / (vstop * bb);
            nfall += nf1(m11) + nf1(m12) + nf1(m21) + nf1(m22);
        }
        // non-finite coefficients anywhere in the record: the whole trace is non-finite in the reference (see rf_coef_kernel)
        nfall += nonfinite(ru) + nonfinite(hm);
        if (nfall != 0.0) bad = 1.0;
        rec[0] = (double)nlay; rec[1] = p; rec[2] = do_decomp; rec[3] = bad;
        rec[4] = m11; rec[5] = m12; rec[6] = m21; rec[7] = m22;
        rec[REC_HEAD + 40 * (size_t)Lmax] = (imall + imag_mass(ru) == 0.0 && nfall == 0.0) ? 1.0 : 0.0;
    }
}

template <int LP>
__global__ __launch_bounds__(256) void rf_coef_layers_kernel(RfKernelArgs A)
{
    rf_coef_layers_body<LP>(A);
}
// The same with at most 96 registers (a few spilled): in the fused call the coefficient kernel is dispatched behind
// the start gate, when two dispersion wavefronts of 208 registers sit on every SIMD -- the 124-register build could
// only start where such wavefronts have ended (3.4 ms later), right before the synthesis kernel that waits for it; this
// one becomes resident beside them and is long done when the first synthesis workgroup finds room.
template <int LP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void rf_coef_layers_kernel_small(RfKernelArgs A)
{
    rf_coef_layers_body<LP>(A);
}

// Spectrum of one model into LDS, then the inverse real FFT there:
// iftr (greens.cpp:136-158) + ccfork(+1) (fork.cpp:11-60): f[n] = (1/N) Re sum_k X[k] e^{+2 pi i k n / N} with the
// Hermitian extension X[N-k] = conj(X[k]) of the bins k = 0..N/2 the frequency loop produces.
// A real trace of N samples is a COMPLEX transform of length M = N/2 (round 3; round 2 ran the full length-N
// complex transform on the extended spectrum: twice the LDS, twice the butterflies):
//     Z[k] = (X[k] + conj(X[M-k])) + i w^k (X[k] - conj(X[M-k])),  w = e^{+2 pi i / N},  k = 0..M-1
//     z[m] = sum_k Z[k] e^{+2 pi i k m / M}        ->  f[2m] = Re z[m] / N,  f[2m+1] = Im z[m] / N
// (the even samples come from X[k] + X[k+M], the odd ones from (X[k] - X[k+M]) w^k, and X[k+M] = conj(X[M-k]);
// only the real parts of X[0] and X[N/2] reach a real output).  Pairs (k, M-k) share their work:
// with E = X[k] + conj(X[M-k]), T = i w^k (X[k] - conj(X[M-k])):  Z[k] = E + T,  Z[M-k] = conj(E - T).
// The transform is decimation in frequency (natural order in, bit-reversed out), so that the bins go to LDS in
// natural order and only the nkeep output samples are read through the bit reversal.
// Twiddles: w^k = TW1[k >> 6] TW0[k & 63] (two small tables instead of N/4 entries: one more complex product per
// butterfly, 1-3 KB of LDS instead of 8-64 KB).  LDS per workgroup: 8 N + 16 + 1024 + N/8 bytes --
// 17.7 KB at nsamp = 2048 (round 2: 40 KB), 134 KB at nsamp = 16384, the largest trace one workgroup holds.
// GLOBALZ: the half-length spectrum lives in the model's slice of A.zwork (HBM / L2) instead of LDS -- traces longer than a
// workgroup's LDS holds (nsamp > 16384).  Same bins, same butterflies, same order; a workgroup's barrier orders its own
// global accesses (all its wavefronts share the CU's cache).
template <bool GLOBALZ = false>
__device__ __forceinline__ void rf_synth_body(const RfKernelArgs &A, int logm, int jcut)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int N = A.nsamp, M = N / 2;
    double2 *z = GLOBALZ ? reinterpret_cast<double2 *>(A.zwork) + (size_t)blockIdx.x * (size_t)M : reinterpret_cast<double2 *>(smem); // [M]
    double2 *nyq = GLOBALZ ? reinterpret_cast<double2 *>(smem) : z + M;                                                               // [1]   Re X[N/2]
    double2 *tw0 = nyq + 1;                                // [64]  w^r
    double2 *tw1 = tw0 + 64;                               // [max(1, N/128)]  w^(64 q)
    const int ib = blockIdx.x;
    const int nthr = blockDim.x; // 256 (four wavefronts) or 128
    // Which wavefront takes the bins of the last, partly filled pass rotates with the workgroup (553 bins are 3 + 2 + 2 + 2
    // passes of 64 lanes: without the rotation the same hardware wave slot of every workgroup carries the third pass)
    const int rot = (blockIdx.x * 64) & (nthr - 1);
    const int tid = threadIdx.x;
    const int tbin = (A.no_rot ? tid : ((tid + rot) & (nthr - 1)));
    const double dw = 2.0 * M_PI * A.fsamp / A.nsamp;
    const double qg = sqrt(M_PI) * A.fsamp / A.gauss;
    const size_t recsz = rec_doubles(A.Lmax);
    const double *rec = A.coef + (size_t)ib * recsz;
    const int n1 = (N >= 128) ? N / 128 : 1;
    for (int k = tid; k < 64 + n1; k += nthr) {
        const int idx = (k < 64) ? k : (k - 64) * 64;
        double sn, cs;
        sincos_cw((2.0 * M_PI / (double)N) * (double)idx, &sn, &cs);
        tw0[k] = make_double2(cs, sn); // (tw1 follows tw0)
    }
    const bool realc = rec[REC_HEAD + 40 * (size_t)A.Lmax] != 0.0 && !A.no_realc; // (uniform over the workgroup)
    for (int j = tbin; j < M; j += nthr) {
        // bins from jcut on: the Gauss low-pass has them below RF_CUT of the pass band (see bh_launch_rf)
        cd s = cd{0.0, 0.0};
        if (j < jcut) {
            if (realc) s = rf_one_frequency<true>(rec, A.Lmax, j, dw, qg, A.gauss, A.tshift, A.waveno);
            else s = rf_one_frequency<false>(rec, A.Lmax, j, dw, qg, A.gauss, A.tshift, A.waveno);
        }
        if (j == 0) s.im = __builtin_isfinite(s.im) ? 0.0 : s.im; // Re X[0] only (a non-finite bin stays non-finite)
        z[j] = make_double2(s.re, s.im);
    }
    if (tid == 0) { // the Nyquist bin: one more frequency for one thread, and only when the filter keeps it
        const cd s = (M < jcut) ? rf_one_frequency<false>(rec, A.Lmax, M, dw, qg, A.gauss, A.tshift, A.waveno) : cd{0.0, 0.0};
        nyq[0] = make_double2(s.re, __builtin_isfinite(s.im) ? 0.0 : s.im);
    }
    __syncthreads();
    auto twiddle = [&](int k) {
        const double2 a = tw1[k >> 6], b = tw0[k & 63];
        return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
    };
    for (int k = tid; k <= M / 2; k += nthr) {
        const double2 xa = z[k];
        const double2 xb = (k == 0) ? nyq[0] : z[M - k];
        const double2 w = twiddle(k);
        const double ex = xa.x + xb.x, ey = xa.y - xb.y;    // E = X[k] + conj(X[M-k])
        const double dx = xa.x - xb.x, dy = xa.y + xb.y;    // D = X[k] - conj(X[M-k])
        const double px = w.x * dx - w.y * dy, py = w.x * dy + w.y * dx; // w^k D
        const double tx = -py, ty = px;                     // T = i w^k D
        z[k] = make_double2(ex + tx, ey + ty);
        if (k > 0 && 2 * k < M) z[M - k] = make_double2(ex - tx, -(ey - ty));
    }
    __syncthreads();
    for (int s = logm - 1; s >= 0; --s) {
        const int l = 1 << s;          // half-size of the butterflies of this stage
        for (int bfly = tid; bfly < M / 2; bfly += nthr) {
            const int m = bfly & (l - 1);
            const int i = ((bfly >> s) << (s + 1)) + m;
            const double2 w = twiddle(m << (logm - s));     // e^{2 pi i m / (2l)} = w^(m M / l)
            const double2 u = z[i], v = z[i + l];
            const double dx = u.x - v.x, dy = u.y - v.y;
            z[i] = make_double2(u.x + v.x, u.y + v.y);
            z[i + l] = make_double2(w.x * dx - w.y * dy, w.x * dy + w.y * dx);
        }
        __syncthreads();
    }
    const double scale = 1.0 / (double)N;
    const int shift = 32 - logm;
    if (A.sums != nullptr && nthr == 256) {
        // Fused likelihood: the kept samples never leave the CU.  The sums are formed exactly as like_kernel forms them
        // from a stored trace -- thread t takes samples t, t + 256, ... in order, a butterfly per wavefront, the four
        // wavefronts' values added in order -- with every product and sum rounded on its own (BH_NOFUSE pins a value to a
        // register between two operations: this file is compiled with contraction on, HIP's __dmul_rn / __dadd_rn are plain
        // operators, and `#pragma clang fp contract(off)` did not keep the backend from fusing here): the same bits.
        const int n = A.nkeep;
#define BH_RF_SAMPLE(i) (scale * (((i) & 1) ? z[(int)(__brev((unsigned)((i) >> 1)) >> shift)].y : z[(int)(__brev((unsigned)((i) >> 1)) >> shift)].x))
#define BH_NOFUSE(x) asm volatile("" : "+v"(x))
        double s0 = 0.0, s1 = 0.0;
        for (int i = tid; i < n; i += 256) {
            double yi = BH_RF_SAMPLE(i);
            BH_NOFUSE(yi);
            double d = yi - A.yobs[i];
            BH_NOFUSE(d);
            double dd = d * d;
            BH_NOFUSE(dd);
            s0 = s0 + dd;
            if (i + 1 < n) {
                double yj = BH_RF_SAMPLE(i + 1);
                BH_NOFUSE(yj);
                double dj = yj - A.yobs[i + 1];
                BH_NOFUSE(dj);
                double pj = d * dj;
                BH_NOFUSE(pj);
                s1 = s1 + pj;
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            s0 = s0 + __shfl_xor(s0, off);
            s1 = s1 + __shfl_xor(s1, off);
        }
        if (n > 64) { // (like_kernel: block_sum; up to 64 samples only the first wavefront's value counts)
            double *red = reinterpret_cast<double *>(tw0); // (the twiddle tables are dead by now: 64 complex numbers)
            __syncthreads();
            if ((tid & 63) == 0) {
                red[tid >> 6] = s0;
                red[4 + (tid >> 6)] = s1;
            }
            __syncthreads();
            s0 = red[0] + red[1] + red[2] + red[3];
            s1 = red[4] + red[5] + red[6] + red[7];
        }
        if (tid == 0) {
            double *o = A.sums + (size_t)ib * 4;
            double y0 = BH_RF_SAMPLE(0), yn = BH_RF_SAMPLE(n - 1);
            BH_NOFUSE(y0);
            BH_NOFUSE(yn);
            o[0] = s0;
            o[1] = s1;
            o[2] = y0 - A.yobs[0];
            o[3] = yn - A.yobs[n - 1];
        }
#undef BH_RF_SAMPLE
#undef BH_NOFUSE
        return;
    }
    double *out = A.rf + (size_t)ib * A.ldr;
    for (int m = tid; 2 * m < A.nkeep; m += nthr) {
        const double2 v = z[(int)(__brev((unsigned)m) >> shift)];
        out[2 * m] = scale * v.x;
        if (2 * m + 1 < A.nkeep) out[2 * m + 1] = scale * v.y;
    }
}
// Two register budgets of the same text: 4 wavefronts per SIMD (128 VGPRs, a few spilled) and 3 (168, none);
// bh_launch_rf picks (BH_RF_WAVES overrides, for measurements).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void rf_synth_kernel(RfKernelArgs A, int logm, int jcut)
{
    rf_synth_body<false>(A, logm, jcut);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void rf_synth_kernel_w3(RfKernelArgs A, int logm, int jcut)
{
    rf_synth_body<false>(A, logm, jcut);
}
// Traces longer than a workgroup's LDS holds: the spectrum in the HBM workspace (GLOBALZ)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void rf_synth_kernel_long(RfKernelArgs A, int logm, int jcut)
{
    rf_synth_body<true>(A, logm, jcut);
}

} // namespace

size_t bh_rf_coef_doubles(int Lmax) { return rec_doubles(Lmax); }

size_t bh_rf_lds_bytes(int nsamp)
{
    return (size_t)(nsamp / 2) * 16 + 16 + 64 * 16 + (size_t)(nsamp >= 128 ? nsamp / 128 : 1) * 16;
}

int bh_launch_rf(const RfKernelArgs &a_in, hipStream_t stream)
{
    RfKernelArgs a = a_in;
    const BhTuning &tun = bh_tuning(); // (experiment switches, bh_tuning.h)
    a.no_realc = tun.rf_no_realc != 0 ? 1 : 0;
    a.no_rot = tun.rf_no_rot != 0 ? 1 : 0;
    const int nthr = (tun.rf_threads == 128) ? 128 : 256;
    const int half = a.nsamp / 2;
    int logm = 0;
    while ((1 << logm) < half) ++logm;
    size_t lds = bh_rf_lds_bytes(a.nsamp); // half-length complex spectrum + Nyquist bin + two twiddle tables
    const bool longtrace = lds > BH_RF_MAX_LDS;
    if (longtrace) {
        if (a.zwork == nullptr || a.nsamp > BH_RF_MAX_NSAMP) return -1;
        lds -= (size_t)(a.nsamp / 2) * 16; // (the spectrum is in the workspace)
        a.lds_min = 0;
    }
    if (lds < (size_t)a.lds_min) lds = (size_t)a.lds_min;
    if (lds > 64 * 1024) { // beyond the default dynamic-LDS limit: a workgroup may take the CU's whole 160 KB
        static std::atomic<unsigned long long> allowed{0};
        const void *k[2] = {reinterpret_cast<const void *>(rf_synth_kernel), reinterpret_cast<const void *>(rf_synth_kernel_w3)};
        if (!bh_allow_big_lds(&allowed, k, 2, (int)BH_RF_MAX_LDS)) return -1;
    }
    if (a.Lmax <= 16 && a.coef_small)
        hipLaunchKernelGGL((rf_coef_layers_kernel_small<16>), dim3((a.B + 15) / 16), dim3(256), 0, stream, a);
    else if (a.Lmax <= 32 && a.coef_small)
        hipLaunchKernelGGL((rf_coef_layers_kernel_small<32>), dim3((a.B + 7) / 8), dim3(256), 0, stream, a);
    else if (a.Lmax <= 16)
        hipLaunchKernelGGL((rf_coef_layers_kernel<16>), dim3((a.B + 15) / 16), dim3(256), 0, stream, a);
    else if (a.Lmax <= 32)
        hipLaunchKernelGGL((rf_coef_layers_kernel<32>), dim3((a.B + 7) / 8), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(rf_coef_kernel, dim3((a.B + 255) / 256), dim3(256), 0, stream, a);
    // Spectral cut-off.  Every bin carries the Gauss low-pass exp(-w^2 / (4 a^2)) (greens.cpp:343-398); where that
    // factor is below RF_CUT = 1e-17 the bin is below 1e-17 of the pass band (|R/Z| is of order one): a tenth of the
    // rounding unit (2^-53 = 1.1e-16) of the sums the transform forms, i.e. it is lost in the reference's own additions.
    // Such bins are set to zero instead of being computed (the reference computes them and multiplies by ~0).  With
    // a = 2.5, 20 Hz, nsamp 2048 that is every bin above 5.0 Hz: 510 of 1025 are computed, 8 passes of 64 lanes per model
    // (round 2 cut at 1e-30: 678 bins, 11 passes).  tests/test_gpu_rf.py compares with the uncut transform (1e-13).
    const bool no_cut = tun.rf_no_cut != 0; // (tests flip it with bh_engine_set_tuning)
    const double dw = 2.0 * M_PI * a.fsamp / a.nsamp;
    const double jc = std::floor(RF_CUT_WA * a.gauss / dw) + 1.0;
    const int jcut = (no_cut || !(jc < (double)half)) ? half + 1 : (int)jc;
    if (longtrace)
        hipLaunchKernelGGL(rf_synth_kernel_long, dim3(a.B), dim3(256), lds, stream, a, logm, jcut);
    else if (tun.rf_waves == 3)
        hipLaunchKernelGGL(rf_synth_kernel_w3, dim3(a.B), dim3(nthr), lds, stream, a, logm, jcut);
    else
        hipLaunchKernelGGL(rf_synth_kernel, dim3(a.B), dim3(nthr), lds, stream, a, logm, jcut);
    return 0;
}
