// bayhunter_amd/csrc/swd_kernel.hip -- Rayleigh/Love phase & group dispersion on gfx950.
//
// Replaces the reference's surfdisp96 (src/extensions/surfdisp96.f:55-360 and the routines it
// calls) for a batch of models.  Mapping: ONE WAVEFRONT LANE = ONE CANDIDATE MODEL, one
// 64-lane wavefront per workgroup, grid = ceil(B/64).  No MFMA: the work is a scalar FP64
// recurrence (5-vector x 5x5 compound matrix per layer for Rayleigh, 2-vector for Love) inside
// a data-dependent root search.
//
// Design points
//   * EVALUATION-SYNCHRONOUS STATE MACHINE.  The reference's control flow is
//       for period: bracket-step until sign change; refine (bisection / inverse Neville)
//     and the number of secular-function evaluations differs per model and per period.  A
//     literal SIMT translation would make every lane wait for the slowest lane in every inner
//     loop.  Here each lane keeps an explicit search state (period index, which root, bracket,
//     Neville table, continuation tag) and the wavefront's loop body is exactly ONE secular
//     evaluation for all lanes followed by a short per-lane state transition.  Lanes drift
//     apart in period index freely; the wave ends when its slowest lane has used up its own
//     total, not the sum of per-period maxima.
//   * The model (thickness, vp, vs, rho), rounded to binary32 as the f2py boundary of the
//     reference does, is staged once through LDS in [array][layer][lane]
//     order (bank-conflict free: lane = bank) from coalesced global loads of the layer-major
//     (vp, vs, rho, h) arrays; the per-lane Neville tables x[11], y[11] and the period table
//     live in LDS as well.
//   * Rounding points, the search sequence (start value, 0.005 km/s stepping, direction logic,
//     Neville/bisection decisions, the 1e-6 stop test, which point is returned) and the
//     binary32 arithmetic of the start value and of the group-velocity formula follow the
//     reference exactly; only the device sin/cos/exp differ from the host
//     libm in the last ulp.
#include "bh_device.h"

namespace {

constexpr int NEV_MAX = 11; // Neville table entries: order grows to m <= 10 (surfdisp96.f:655)

__device__ __forceinline__ bool signs_differ(double x, double y)
{
    return ((__double_as_longlong(x) ^ __double_as_longlong(y)) < 0);
}

// LDS views -----------------------------------------------------------------------------------
struct ModelLds {
    const float *d, *a, *b, *rho; // each [Lmax][64], this lane's column pre-offset
    __device__ __forceinline__ double D(int m) const { return (double)d[m * BH_WAVE]; }
    __device__ __forceinline__ double A(int m) const { return (double)a[m * BH_WAVE]; }
    __device__ __forceinline__ double Bv(int m) const { return (double)b[m * BH_WAVE]; }
    __device__ __forceinline__ double R(int m) const { return (double)rho[m * BH_WAVE]; }
};

// ---- Love: SH Thomson-Haskell (surfdisp96.f:710-769) ----------------------------------------
__device__ double love_secular(double wvno, double omega, const ModelLds &md, int mmax, int llw,
                               int mtop)
{
    double beta1 = md.Bv(mmax - 1);
    double rho1 = md.R(mmax - 1);
    double xkb = omega / beta1;
    double wvnop = wvno + xkb;
    double wvnom = fabs(wvno - xkb);
    double rb = sqrt(wvnop * wvnom);
    double e1 = rho1 * rb;
    double e2 = 1.0 / (beta1 * beta1);
    for (int m = mtop - 2; m >= 0; --m) {
        if (m <= mmax - 2 && m >= llw - 1) {
            beta1 = md.Bv(m);
            rho1 = md.R(m);
            const double dm = md.D(m);
            const double xmu = rho1 * beta1 * beta1;
            xkb = omega / beta1;
            wvnop = wvno + xkb;
            wvnom = fabs(wvno - xkb);
            rb = sqrt(wvnop * wvnom);
            const double q = dm * rb;
            double cosq, y, z;
            if (wvno < xkb) {
                double sinq;
                sincos(q, &sinq, &cosq);
                y = sinq / rb;
                z = -rb * sinq;
            } else if (wvno == xkb) {
                cosq = 1.0;
                y = dm;
                z = 0.0;
            } else {
                double fac = 0.0;
                if (q < 16.0) fac = exp(-2.0 * q);
                cosq = (1.0 + fac) * 0.5;
                const double sinq = (1.0 - fac) * 0.5;
                y = sinq / rb;
                z = rb * sinq;
            }
            const double e10 = e1 * cosq + e2 * xmu * z;
            const double e20 = e1 * y / xmu + e2 * cosq;
            double xnor = fabs(e10);
            const double ynor = fabs(e20);
            if (ynor > xnor) xnor = ynor;
            if (xnor < 1.0e-40) xnor = 1.0;
            e1 = e10 / xnor;
            e2 = e20 / xnor;
        }
    }
    return e1;
}

// ---- Rayleigh: eigenfunction products (surfdisp96.f:874-991, `var`) --------------------------
struct LayerTerms {
    double a0, cpcq, cpy, cpz, cqw, cqx, xy, xz, wy, wz, w, cosp;
};

__device__ __forceinline__ void layer_products(double p, double q, double ra, double rb,
                                               double wvno, double xka, double xkb, double dpth,
                                               LayerTerms &o)
{
    double cosp, cosq, w, x, y, z;
    double pex = 0.0, sex = 0.0;
    if (wvno < xka) {
        double sinp;
        sincos(p, &sinp, &cosp);
        w = sinp / ra;
        x = -ra * sinp;
    } else if (wvno == xka) {
        cosp = 1.0;
        w = dpth;
        x = 0.0;
    } else {
        pex = p;
        double fac = 0.0;
        if (p < 16.0) fac = exp(-2.0 * p);
        cosp = (1.0 + fac) * 0.5;
        const double sinp = (1.0 - fac) * 0.5;
        w = sinp / ra;
        x = ra * sinp;
    }
    if (wvno < xkb) {
        double sinq;
        sincos(q, &sinq, &cosq);
        y = sinq / rb;
        z = -rb * sinq;
    } else if (wvno == xkb) {
        cosq = 1.0;
        y = dpth;
        z = 0.0;
    } else {
        sex = q;
        double fac = 0.0;
        if (q < 16.0) fac = exp(-2.0 * q);
        cosq = (1.0 + fac) * 0.5;
        const double sinq = (1.0 - fac) * 0.5;
        y = sinq / rb;
        z = rb * sinq;
    }
    const double exa = pex + sex;
    double a0 = 0.0;
    if (exa < 60.0) a0 = exp(-exa);
    o.a0 = a0;
    o.cpcq = cosp * cosq;
    o.cpy = cosp * y;
    o.cpz = cosp * z;
    o.cqw = cosq * w;
    o.cqx = cosq * x;
    o.xy = x * y;
    o.xz = x * z;
    o.wy = w * y;
    o.wz = w * z;
    o.w = w;
    o.cosp = cosp;
}

// One layer of the Dunkin recursion: e <- normalise(e * CA(layer)).  CA is the 5x5 compound
// matrix of surfdisp96.f:1024-1068 (`dnka`); its entries are formed with the reference's
// operation order and consumed column by column so that only one column is live at a time.
__device__ __forceinline__ void rayleigh_layer(double e[5], double wvno2, double gam, double gammk,
                                               double rho, const LayerTerms &v)
{
    const double two = 2.0;
    const double gamm1 = gam - 1.0;
    const double twgm1 = gam + gamm1;
    const double gmgmk = gam * gammk;
    const double gmgm1 = gam * gamm1;
    const double gm1sq = gamm1 * gamm1;
    const double rho2 = rho * rho;
    const double a0pq = v.a0 - v.cpcq;
    const double ca11 = v.cpcq - two * gmgm1 * a0pq - gmgmk * v.xz - wvno2 * gm1sq * v.wy;
    const double ca12 = (wvno2 * v.cpy - v.cqx) / rho;
    const double ca13 = -(twgm1 * a0pq + gammk * v.xz + wvno2 * gamm1 * v.wy) / rho;
    const double ca14 = (v.cpz - wvno2 * v.cqw) / rho;
    const double ca15 = -(two * wvno2 * a0pq + v.xz + wvno2 * wvno2 * v.wy) / rho2;
    const double ca21 = (gmgmk * v.cpz - gm1sq * v.cqw) * rho;
    const double ca22 = v.cpcq;
    const double ca23 = gammk * v.cpz - gamm1 * v.cqw;
    const double ca24 = -v.wz;
    const double ca25 = ca14;
    const double ca41 = (gm1sq * v.cpy - gmgmk * v.cqx) * rho;
    const double ca42 = -v.xy;
    const double ca43 = gamm1 * v.cpy - gammk * v.cqx;
    const double ca44 = ca22;
    const double ca45 = ca12;
    const double ca51 =
        -(two * gmgmk * gm1sq * a0pq + gmgmk * gmgmk * v.xz + gm1sq * gm1sq * v.wy) * rho2;
    const double ca52 = ca41;
    const double ca53 =
        -(gammk * gamm1 * twgm1 * a0pq + gam * gammk * gammk * v.xz + gamm1 * gm1sq * v.wy) * rho;
    const double ca54 = ca21;
    const double ca55 = ca11;
    const double t = -two * wvno2;
    const double ca31 = t * ca53;
    const double ca32 = t * ca43;
    const double ca33 = v.a0 + two * (v.cpcq - ca11);
    const double ca34 = t * ca23;
    const double ca35 = t * ca13;
    // ee(i) = sum_j e(j)*ca(j,i), accumulated from 0.0 in j order (surfdisp96.f:836-842)
    double ee0 = 0.0, ee1 = 0.0, ee2 = 0.0, ee3 = 0.0, ee4 = 0.0;
    ee0 = ee0 + e[0] * ca11; ee0 = ee0 + e[1] * ca21; ee0 = ee0 + e[2] * ca31; ee0 = ee0 + e[3] * ca41; ee0 = ee0 + e[4] * ca51;
    ee1 = ee1 + e[0] * ca12; ee1 = ee1 + e[1] * ca22; ee1 = ee1 + e[2] * ca32; ee1 = ee1 + e[3] * ca42; ee1 = ee1 + e[4] * ca52;
    ee2 = ee2 + e[0] * ca13; ee2 = ee2 + e[1] * ca23; ee2 = ee2 + e[2] * ca33; ee2 = ee2 + e[3] * ca43; ee2 = ee2 + e[4] * ca53;
    ee3 = ee3 + e[0] * ca14; ee3 = ee3 + e[1] * ca24; ee3 = ee3 + e[2] * ca34; ee3 = ee3 + e[3] * ca44; ee3 = ee3 + e[4] * ca54;
    ee4 = ee4 + e[0] * ca15; ee4 = ee4 + e[1] * ca25; ee4 = ee4 + e[2] * ca35; ee4 = ee4 + e[3] * ca45; ee4 = ee4 + e[4] * ca55;
    // normc (surfdisp96.f:995-1020): max-norm rescale; its log() result is never used.
    double t1 = 0.0;
    if (fabs(ee0) > t1) t1 = fabs(ee0);
    if (fabs(ee1) > t1) t1 = fabs(ee1);
    if (fabs(ee2) > t1) t1 = fabs(ee2);
    if (fabs(ee3) > t1) t1 = fabs(ee3);
    if (fabs(ee4) > t1) t1 = fabs(ee4);
    if (t1 < 1.0e-40) t1 = 1.0;
    e[0] = ee0 / t1;
    e[1] = ee1 / t1;
    e[2] = ee2 / t1;
    e[3] = ee3 / t1;
    e[4] = ee4 / t1;
}

// ---- Rayleigh: Dunkin compound-matrix secular function (surfdisp96.f:773-871) -----------------
__device__ double rayleigh_secular(double wvno, double omga, const ModelLds &md, int mmax, int llw,
                                   int mtop)
{
    double e[5];
    LayerTerms v;
    double omega = omga;
    if (omega < 1.0e-4) omega = 1.0e-4;
    const double wvno2 = wvno * wvno;
    {
        const double ah = md.A(mmax - 1), bh = md.Bv(mmax - 1);
        const double xka = omega / ah;
        const double xkb = omega / bh;
        double wvnop = wvno + xka;
        double wvnom = fabs(wvno - xka);
        const double ra = sqrt(wvnop * wvnom);
        wvnop = wvno + xkb;
        wvnom = fabs(wvno - xkb);
        const double rb = sqrt(wvnop * wvnom);
        const double t = bh / omega;
        const double gammk = 2.0 * t * t;
        const double gam = gammk * wvno2;
        const double gamm1 = gam - 1.0;
        const double rho1 = md.R(mmax - 1);
        e[0] = rho1 * rho1 * (gamm1 * gamm1 - gam * gammk * ra * rb);
        e[1] = -rho1 * ra;
        e[2] = rho1 * (gamm1 - gammk * ra * rb);
        e[3] = rho1 * rb;
        e[4] = wvno2 - ra * rb;
    }
    for (int m = mtop - 2; m >= 0; --m) {
        if (m <= mmax - 2 && m >= llw - 1) {
            const double am = md.A(m), bm = md.Bv(m);
            const double xka = omega / am;
            const double xkb = omega / bm;
            const double t = bm / omega;
            const double gammk = 2.0 * t * t;
            const double gam = gammk * wvno2;
            double wvnop = wvno + xka;
            double wvnom = fabs(wvno - xka);
            const double ra = sqrt(wvnop * wvnom);
            wvnop = wvno + xkb;
            wvnom = fabs(wvno - xkb);
            const double rb = sqrt(wvnop * wvnom);
            const double dpth = md.D(m);
            const double rho1 = md.R(m);
            const double p = ra * dpth;
            const double q = rb * dpth;
            layer_products(p, q, ra, rb, wvno, xka, xkb, dpth, v);
            rayleigh_layer(e, wvno2, gam, gammk, rho1, v);
        }
    }
    double result = e[0];
    if (llw != 1) { // water layer on top (surfdisp96.f:850-866); unreachable from BayHunter
        const double xka = omega / md.A(0);
        const double wvnop = wvno + xka;
        const double wvnom = fabs(wvno - xka);
        const double ra = sqrt(wvnop * wvnom);
        const double dpth = md.D(0);
        const double rho1 = md.R(0);
        const double p = ra * dpth;
        const double znul = 1.0e-5;
        layer_products(p, znul, ra, znul, wvno, xka, znul, dpth, v);
        const double w0 = -rho1 * v.w;
        result = v.cosp * e[0] + w0 * e[1];
    }
    return result;
}

// ---- half-space Rayleigh velocity, binary32 throughout (surfdisp96.f:367-388) -----------------
__device__ float gtsolh_f32(float a, float b)
{
    float c = 0.95f * b;
    for (int i = 0; i < 5; ++i) {
        const float gamma = b / a;
        const float kappa = c / b;
        const float k2 = kappa * kappa;
        const float gk = gamma * kappa;
        const float gk2 = gk * gk;
        const float fac1 = sqrtf(1.0f - gk2);
        const float fac2 = sqrtf(1.0f - k2);
        const float tk = 2.0f - k2;
        const float fr = tk * tk - 4.0f * fac1 * fac2;
        float frp = -4.0f * (2.0f - k2) * kappa + 4.0f * fac2 * gamma * gamma * kappa / fac1 +
                    4.0f * fac1 * kappa / fac2;
        frp = frp / b;
        c = c - fr / frp;
    }
    return c;
}

// continuation tags: what the pending secular evaluation is for
enum : int {
    ST_FIRST = 0, // del1 at the start value c1                      (surfdisp96.f:421-423)
    ST_STEP = 1,  // del2 at c2 = c1 +- dc                           (:447-449)
    ST_NEV0 = 2,  // first midpoint inside nevill                    (:582-583)
    ST_NEVL = 3,  // midpoint / Neville estimate, then top of loop   (:586-...)
    ST_NEVF = 4   // forced midpoint after the estimate left the bracket (:594-598)
};

template <int IFUNC>
__global__ __launch_bounds__(BH_WAVE) void swd_kernel(SwdKernelArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const int ib = blockIdx.x * BH_WAVE + lane;
    const bool valid = ib < A.B;
    const int Lmax = A.Lmax;
    const int K = A.K;

    float *mdl = reinterpret_cast<float *>(smem);                       // [4][Lmax][64]
    double *xs = reinterpret_cast<double *>(smem + (size_t)4 * Lmax * BH_WAVE * sizeof(float));
    double *ys = xs + NEV_MAX * BH_WAVE;                                 // [11][64] each
    double *per = ys + NEV_MAX * BH_WAVE;                                // [K]

    for (int k = lane; k < K; k += BH_WAVE) per[k] = A.periods[k];

    // ---- stage the model through LDS, rounding to binary32 like the f2py boundary -----------
    const int mmax = valid ? A.nlay[ib] : 2;
    int mtop = mmax; // wave-wide maximum layer count = loop bound of the secular functions
    for (int off = 32; off > 0; off >>= 1) mtop = max(mtop, __shfl_xor(mtop, off));
    {
        const ptrdiff_t base = (ptrdiff_t)ib * A.sb;
        for (int l = 0; l < Lmax; ++l) {
            float fd = 0.f, fa = 1.f, fb = 1.f, fr = 1.f;
            if (valid && l < mmax) {
                const ptrdiff_t o = base + (ptrdiff_t)l * A.sl;
                fd = (float)A.h[o];
                fa = (float)A.vp[o];
                fb = (float)A.vs[o];
                fr = (float)A.rho[o];
            }
            mdl[(0 * Lmax + l) * BH_WAVE + lane] = fd;
            mdl[(1 * Lmax + l) * BH_WAVE + lane] = fa;
            mdl[(2 * Lmax + l) * BH_WAVE + lane] = fb;
            mdl[(3 * Lmax + l) * BH_WAVE + lane] = fr;
        }
    }
    __syncthreads();
    ModelLds md;
    md.d = mdl + 0 * Lmax * BH_WAVE + lane;
    md.a = mdl + 1 * Lmax * BH_WAVE + lane;
    md.b = mdl + 2 * Lmax * BH_WAVE + lane;
    md.rho = mdl + 3 * Lmax * BH_WAVE + lane;
    double *xl = xs + lane; // element j at xl[j*64]
    double *yl = ys + lane;

    // ---- driver set-up (surfdisp96.f:124-217) ------------------------------------------------
    const float b0 = md.b[0];
    const int llw = (b0 <= 0.0f) ? 2 : 1;
    float betmx = -1.e20f, betmn = 1.e20f;
    int jmn = 0, jsol = 1;
    for (int i = 0; i < mmax; ++i) {
        const float bi = md.b[i * BH_WAVE], ai = md.a[i * BH_WAVE];
        if (bi > 0.01f && bi < betmn) {
            betmn = bi;
            jmn = i;
            jsol = 1;
        } else if (bi <= 0.01f && ai < betmn) {
            betmn = ai;
            jmn = i;
            jsol = 0;
        }
        if (bi > betmx) betmx = bi;
    }
    const float h32 = 0.005f;
    const double one = 1.0e-2;
    const double onea = (double)1.5f;
    const double dc = fabs((double)0.005f);
    const double twopi = 2.0 * 3.141592653589793;
    const double pct = (double)0.01f; // `0.01*ss1` with a default-real literal (:623-626)
    float cc1 = (jsol == 0) ? betmn : gtsolh_f32(md.a[jmn * BH_WAVE], md.b[jmn * BH_WAVE]);
    cc1 = 0.95f * cc1;
    cc1 = 0.90f * cc1;
    const double cc = (double)cc1;
    const double cm = cc;
    const double betmxd = (double)betmx;
    const bool group = A.igr > 0;

    // ---- per-lane search state ----------------------------------------------------------------
    int k = 0;        // period index
    int root = 0;     // 0: root at t (or t/(1+h)), 1: second root at t/(1-h) for group velocity
    int st = ST_FIRST;
    int ifirst = 1;
    bool active = valid && K > 0;
    int errflag = 0;
    double c1 = cc, c2 = 0.0, clow = cc, del1 = 0.0, del2 = 0.0, del1st = 0.0;
    double c3 = 0.0, del3 = 0.0, ck = 0.0;
    int idir = 1, nev = 1, mnev = 1, nctrl = 1;
    float t1a = 0.f, t1b = 0.f;
    double t1 = 1.0, omega = 1.0;
    unsigned int myevals = 0;

    auto set_period = [&](int kk) {
        double tt = per[kk];
        if (group) {
            t1a = (float)(tt / (double)(1.0f + h32));
            t1b = (float)(tt / (double)(1.0f - h32));
            tt = (double)t1a;
        } else {
            t1a = (float)tt;
        }
        t1 = tt;
        omega = twopi / t1;
    };
    if (active) set_period(0);
    double ceval = c1; // phase velocity the pending evaluation is for

    while (__ballot(active) != 0ull) {
        double del = 0.0;
        if (active) {
            const double wvno = omega / ceval;
            if (IFUNC == 1)
                del = love_secular(wvno, omega, md, mmax, llw, mtop);
            else
                del = rayleigh_secular(wvno, omega, md, mmax, llw, mtop);
            ++myevals;
        }
        if (!active) continue;

        // ---- state transition.  `todo`: 0 nothing, 1 prepare next bracket step, 2 root search
        //      failed (iret = -1), 3 refinement finished with c3, 4 nevill top-of-loop,
        //      5 nevill post-bracket section ---------------------------------------------------
        int todo = 0;
        switch (st) {
        case ST_FIRST:
            del1 = del;
            if (ifirst == 1) del1st = del1;
            idir = (ifirst != 1 && signs_differ(del1st, del1)) ? -1 : +1;
            todo = 1;
            break;
        case ST_STEP:
            del2 = del;
            if (signs_differ(del1, del2)) { // bracketed: enter nevill with (c1,c2,del1,del2)
                c3 = 0.5 * (c1 + c2);
                ceval = c3;
                st = ST_NEV0;
            } else {
                c1 = c2;
                del1 = del2;
                if (c1 < cm || c1 >= betmxd + dc) todo = 2;
                else todo = 1;
            }
            break;
        case ST_NEV0:
            del3 = del;
            nev = 1;
            nctrl = 1;
            mnev = 1;
            todo = 4;
            break;
        case ST_NEVL:
            del3 = del;
            todo = 4;
            break;
        case ST_NEVF:
            del3 = del;
            todo = 5;
            break;
        }
        if (todo == 4) { // label 100 of nevill
            nctrl = nctrl + 1;
            if (nctrl >= 100) {
                todo = 3;
            } else if (c3 < fmin(c1, c2) || c3 > fmax(c1, c2)) {
                nev = 0;
                c3 = 0.5 * (c1 + c2);
                ceval = c3;
                st = ST_NEVF;
                todo = 0;
            } else {
                todo = 5;
            }
        }
        if (todo == 5) {
            const double s13 = del1 - del3;
            const double s32 = del3 - del2;
            if (signs_differ(del3, del1)) {
                c2 = c3;
                del2 = del3;
            } else {
                c1 = c3;
                del1 = del3;
            }
            if (fabs(c1 - c2) <= 1.0e-6 * c1) {
                todo = 3;
            } else {
                if (signs_differ(s13, s32)) nev = 0;
                const double ss1 = fabs(del1), s1 = pct * ss1;
                const double ss2 = fabs(del2), s2 = pct * ss2;
                bool halve = (s1 > ss2 || s2 > ss1 || nev == 0);
                if (!halve) {
                    if (nev == 2) {
                        xl[mnev * BH_WAVE] = c3;
                        yl[mnev * BH_WAVE] = del3;
                    } else {
                        xl[0] = c1;
                        yl[0] = del1;
                        xl[BH_WAVE] = c2;
                        yl[BH_WAVE] = del2;
                        mnev = 1;
                    }
                    const double ym = yl[mnev * BH_WAVE];
                    for (int kk = 1; kk <= mnev; ++kk) {
                        const int j = mnev - kk;
                        const double yj = yl[j * BH_WAVE];
                        const double denom = ym - yj;
                        if (fabs(denom) < 1.0e-10 * fabs(ym)) {
                            halve = true;
                            break;
                        }
                        xl[j * BH_WAVE] = (-yj * xl[(j + 1) * BH_WAVE] + ym * xl[j * BH_WAVE]) / denom;
                    }
                    if (!halve) {
                        c3 = xl[0];
                        nev = 2;
                        mnev = mnev + 1;
                        if (mnev > 10) mnev = 10;
                    }
                }
                if (halve) {
                    c3 = 0.5 * (c1 + c2);
                    nev = 1;
                    mnev = 1;
                }
                ceval = c3;
                st = ST_NEVL;
                todo = 0;
            }
        }
        if (todo == 3) { // getsol after nevill (:468-471)
            c1 = c3;
            todo = (c1 > betmxd) ? 2 : 6;
        }
        if (todo == 2 || todo == 6) { // a root search ended: 6 = found c1, 2 = failed
            bool period_done = false;
            double c1b = 0.0; // the "c1" the driver uses after the (optional) second search
            if (root == 0) {
                if (todo == 2) { // no root in the fundamental mode: err, zero-fill, stop (:313-354)
                    errflag = 1;
                    for (int i = k; i < K; ++i) A.vel[(size_t)ib * A.ldv + i] = 0.0;
                    active = false;
                } else {
                    ck = c1;
                    if (group) { // second root at the slightly longer period (:282-287)
                        root = 1;
                        t1 = (double)t1b;
                        omega = twopi / t1;
                        ifirst = 0;
                        clow = 0.0 + one * dc; // cb(k) is still 0 for the fundamental mode
                        c1 = c1 - onea * dc;
                        st = ST_FIRST;
                        ceval = c1;
                    } else {
                        period_done = true;
                    }
                }
            } else {
                c1b = (todo == 2) ? ck : c1; // second root failed: reuse the first (:291-293)
                period_done = true;
            }
            if (period_done) {
                const float cc0 = (float)ck;
                double out;
                if (!group) {
                    out = (double)cc0;
                } else { // all binary32 (:305)
                    const float cc1s = (float)c1b;
                    const float gvel =
                        (1.0f / t1a - 1.0f / t1b) / (1.0f / (t1a * cc0) - 1.0f / (t1b * cc1s));
                    out = (double)gvel;
                }
                A.vel[(size_t)ib * A.ldv + k] = out;
                k = k + 1;
                if (k >= K) {
                    active = false;
                } else { // initial guess for the next period (:268-272)
                    set_period(k);
                    root = 0;
                    ifirst = 0;
                    c1 = ck - onea * dc;
                    clow = cm;
                    st = ST_FIRST;
                    ceval = c1;
                }
            }
            todo = 0;
        }
        if (todo == 1) { // label 1000 of getsol: next bracket step (:437-446)
            c2 = (idir > 0) ? c1 + dc : c1 - dc;
            if (c2 <= clow) {
                idir = +1;
                c1 = clow;
                c2 = c1 + dc;
                // dc > 0, so the retried c2 = clow + dc is above clow: no further loop
            }
            ceval = c2;
            st = ST_STEP;
        }
    }
    if (valid) A.err[ib] = errflag;
    if (A.neval != nullptr) {
        unsigned long long tot = myevals;
        for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
        if (lane == 0) atomicAdd(A.neval, tot);
    }
}

} // namespace

size_t bh_swd_lds_bytes(int Lmax, int K)
{
    return (size_t)4 * Lmax * BH_WAVE * sizeof(float) + (size_t)2 * NEV_MAX * BH_WAVE * sizeof(double) +
           (size_t)K * sizeof(double);
}

void bh_launch_swd(const SwdKernelArgs &a, int iwave, hipStream_t stream)
{
    const int grid = (a.B + BH_WAVE - 1) / BH_WAVE;
    const size_t lds = bh_swd_lds_bytes(a.Lmax, a.K);
    if (iwave == 1)
        hipLaunchKernelGGL(swd_kernel<1>, dim3(grid), dim3(BH_WAVE), lds, stream, a);
    else
        hipLaunchKernelGGL(swd_kernel<2>, dim3(grid), dim3(BH_WAVE), lds, stream, a);
}
