// bayhunter_amd/csrc/swd_lean.hip -- fundamental-mode phase velocities with the FAST ARITHMETIC, one lane = one TRIAL.
//
// The kernel of the engine's default settings (BH_SEARCH_FAST + BH_ARITH_FAST) for batches of up to a few ten thousand
// models: replaces surfdisp96's driver, getsol and nevill (surfdisp96.f:172-357, :390-482, :557-686) for targets that are
// fundamental-mode phase velocities.  Everything else (reference sequence, group velocities, higher modes, the re-run of
// guarded models) stays with swd_group_kernel / swd_kernel.
//
//   * J = 4 ... 64 neighbouring lanes are one model; a lane evaluates the secular function (swd_fa.h) at ONE trial velocity,
//     all layers serial in the lane -- no layer-parallel phases, no exchange inside an evaluation, every lane busy.  One
//     such evaluation costs a wavefront 14 k cycles (Rayleigh, 10 layers; tools/ubench/faeval.hip), whatever the number of
//     trials it carries, so a round of 16 trials costs what one costs.
//   * THE SCAN IS THE REFERENCE'S (getsol, :437-460): the same start value c(k-1) - 1.5 dc, the same grid (repeated additions of
//     dc), the same direction rule, floor (clow), bounds (cm, betmx + dc) and first sign change -- but a round evaluates the
//     next J grid points at once and the first event among them (sign change, bound, floor) is found with one ballot.
//   * Inside the bracket the root is located by a round of trials clustered around the inverse-quadratic estimate through the
//     bracket's ends and the grid point before them (1e-7 |x| * 4^i both sides; beside the next period's window the cluster is
//     2 or 8 lanes, see the kernel), by J-section where a cluster does not close in; the root returned is the inverse-quadratic
//     point of the final bracket (<= 1.3e-6 |c| wide, typically 8e-7).  The reference (nevill) stops at a bracket of 1e-6 c1 and
//     returns one of its ends: velocities agree to ~1e-6 relative (north_star: 1e-5).
//   * THE GUARD of the short refinement (SearchT, swd_common.h) with its rules and probes -- a root within two steps of a
//     half-space velocity with a sign change right outside the bracket, a scan step over a half-space velocity that showed no sign
//     change -- plus, since round 6 (models drawn from a sampler's prior): a cell that holds betmx or a half-space velocity has its
//     sign changes COUNTED (32 lanes per model and more; one: the refinement goes on, several: the guard; fewer lanes: the guard at
//     once), a J-section counts the sign changes among its points (several roots in a cell: the guard), a cell's first estimate
//     that does not bracket the root between two of its trials is followed by a J-section of the whole cell, a start value next to
//     a sign change of the wrong polarity is guarded -- and the rule of the fast arithmetic: a scan or probe value that is not a
//     number or below fa::SIGN_FLOOR fires it.  A guarded model is listed and run again by the engine with the reference's sequence
//     in the reference's arithmetic (launch_swd_rerun), so failure flags and zero rows are the reference's.
//   * A model with a water layer (unreachable from BayHunter) is guarded at once.
#include "../../include/bh_engine_debug.h"
#include "bh_device.h"
#include "bh_tuning.h"
#include <algorithm>
#define BH_HD __device__ __forceinline__
#define BH_TAB static __device__ const
#include "bh_libm.h"

namespace {
#include "swd_common.h"

// The model in LDS as binary64 (binary32-valued, like the f2py boundary) with the per-layer reciprocals: [7][rows][S]
struct LeanModel {
    const double *p; // column pre-offset
    int S, LS;       // models per wavefront, rows * S
    __device__ __forceinline__ double D(int m) const { return p[m * S]; }
    __device__ __forceinline__ double A(int m) const { return p[LS + m * S]; }
    __device__ __forceinline__ double Bv(int m) const { return p[2 * LS + m * S]; }
    __device__ __forceinline__ double R(int m) const { return p[3 * LS + m * S]; }
    __device__ __forceinline__ double IA(int m) const { return p[4 * LS + m * S]; }
    __device__ __forceinline__ double IB(int m) const { return p[5 * LS + m * S]; }
    __device__ __forceinline__ double IR(int m) const { return p[6 * LS + m * S]; }
};

// dltar4 (surfdisp96.f:773-871) with the fast arithmetic, no water layer; iom = 1 / omega
__device__ __forceinline__ double lean_rayleigh(double wvno, double omega, double iom, const LeanModel &md, int mmax, int mtop)
{
    const double wvno2 = wvno * wvno;
    double e[5];
    {
        const double t = md.Bv(mmax - 1) * iom;
        fa::rayleigh_halfspace(e, wvno, wvno2, omega * md.IA(mmax - 1), omega * md.IB(mmax - 1), 2.0 * t * t, md.R(mmax - 1));
    }
    for (int m = mtop - 2; m >= 0; --m) {
        if (m <= mmax - 2) {
            const double t = md.Bv(m) * iom;
            const double gammk = 2.0 * t * t;
            LayerTerms v;
            fa::layer_products(wvno, omega * md.IA(m), omega * md.IB(m), md.D(m), v);
            Ca19 c;
            fa::ca19(c, wvno2, gammk * wvno2, gammk, md.R(m), md.IR(m), v);
            double ee[5];
            fa::apply5(e, c.c, ee);
            // normc (:995-1020) after every second layer and after the last: a positive scale factor of the vector cancels in
            // the final max-norm scaling, and two layers cannot leave the binary64 range (inputs are bounded: SearchT::init)
            if ((m & 1) == 0) {
                fa::normalize5(ee[0], ee[1], ee[2], ee[3], ee[4], e);
            } else {
#pragma unroll
                for (int i = 0; i < 5; ++i) e[i] = ee[i];
            }
        }
    }
    return e[0];
}
// dltar1 (surfdisp96.f:710-769)
__device__ __forceinline__ double lean_love(double wvno, double omega, const LeanModel &md, int mmax, int mtop)
{
    double e1, e2;
    {
        const double ib = md.IB(mmax - 1);
        const double xkb = omega * ib;
        const double r2 = (wvno + xkb) * fabs(wvno - xkb);
        double rb, t_;
        fa::sqrt_rsqrt(r2 > 1.0e-290 ? r2 : 1.0, rb, t_);
        rb = r2 > 1.0e-290 ? rb : 0.0;
        e1 = md.R(mmax - 1) * rb;
        e2 = ib * ib;
    }
    for (int m = mtop - 2; m >= 0; --m) {
        if (m <= mmax - 2) {
            const double ib = md.IB(m), bm = md.Bv(m), rho1 = md.R(m);
            double cosq, y, z, q;
            fa::love_terms(wvno, omega * ib, md.D(m), cosq, y, z, q);
            fa::love_step(e1, e2, cosq, y * (ib * ib * md.IR(m)), z * (rho1 * bm * bm));
        }
    }
    return e1;
}

enum : int {
    PH_START = 0,      // a period's first round: the start value (trial 0) and the first J - 1 upward steps
    PH_SCAN = 1,       // J further steps of the scan
    PH_REF1 = 2,       // J-section of the bracket
    PH_REFC = 3,       // trials clustered around the estimate
    PH_PROBE_STEP = 4, // the guard's probes of a scan step over a half-space velocity (ST_GS1 / ST_GS2 of SearchT)
    PH_PROBE_ACC = 5,  // the guard's probes outside an accepted bracket (ST_GH / ST_GL)
    PH_SPECIAL = 6,    // a bracket that contains betmx or a half-space velocity: the sign changes BELOW that velocity (see the kernel)
    PH_PROBE_START = 7, // the guard's probes next to a start value that lies next to a root
    PH_SPECIAL_B = 8,  // ... and those above it
    PH_SPECIAL_C = 9   // ... and, where the cell holds ONE, a look inside the guard's distance of that velocity
};

__device__ __forceinline__ bool sign_neg(double x) { return __double_as_longlong(x) < 0; }
// LDS ordering inside ONE wavefront: its LDS instructions execute in order; only the compiler must not reorder
__device__ __forceinline__ void lean_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifndef BH_LEAN_CLUSTER_W
#define BH_LEAN_CLUSTER_W 4.0e-7 // half width (relative) of the two-lane cluster: 1e-7 0.683, 2e-7 0.652, 4e-7 0.646, 6e-7 0.647 ms (c2)
#endif
#ifndef BH_LEAN_RIDE_TOL
#define BH_LEAN_RIDE_TOL 4.0e-7   // the window that rode along is taken over if the root is within this (relative) of the estimate it was anchored on
#endif
#ifndef BH_LEAN_CLUSTER_BIG
#define BH_LEAN_CLUSTER_BIG 8 // the same with 32 and 64 lanes per model
#endif
#ifndef BH_LEAN_CLUSTER
#define BH_LEAN_CLUSTER 2 // lanes of the cluster of trials when the next period's window rides along (16 lanes per model and more)
#endif
constexpr int LEAN_WPB = 4; // wavefronts per workgroup: independent (no barrier), one per SIMD of the CU the workgroup lands on

// J: trials per model and round: 4, 8, 16, 32 or 64 lanes = one model.
// CNT: the build with the evaluation counters and the round clocks (launches with A.neval set: bh_engine_set_instrumentation's
// `counting`).  The counters cost 12 registers; without them the <16> build needs 197 (200 allocated), so that an 88-register
// receiver-function wavefront fits beside the two dispersion wavefronts of a SIMD (2 x 200 + 88 <= 512).
template <int J, bool CNT>
__global__ __launch_bounds__(BH_WAVE * LEAN_WPB) void swd_lean_kernel(SwdMultiArgs A, int wave_lds, int flip)
{
    // "this workgroup is resident": what a second stream waits for before it dispatches wavefronts beside these
    if (A.started != nullptr && threadIdx.x == 0) atomicAdd(A.started, 1u);
    const int lane = threadIdx.x & (BH_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / BH_WAVE));
    // Wavefront -> (target, index).  Two targets are interleaved wavefront by wavefront in proportion to their wavefront
    // counts (wg_n0 / wg_n1), so that every CU -- a workgroup's four wavefronts sit on its four SIMDs -- gets its share of both.
    int wid = (int)blockIdx.x * LEAN_WPB + wave, ty = (int)blockIdx.y;
    if (A.wg_n1 > 0) {
        const long long n1 = A.wg_n1, N = (long long)A.wg_n0 + n1;
        if (wid >= (int)N) return;
        const int l0 = (int)(((long long)wid * n1) / N), l1 = (int)((((long long)wid + 1) * n1) / N);
        ty = (l1 > l0) ? 1 : 0;
        wid = (l1 > l0) ? l0 : wid - l0;
        // The second target takes the models in the OPPOSITE order: the wavefronts run longest search first (bh_launch_pair_order), and
        // the wavefront that shares a SIMD with the longest Rayleigh searches should be the one with the shortest Love searches
        // (c2: 0.668 -> 0.656 ms/step).  Scheduling only.
        if ((flip & 256) && ty == 1) {
            if (flip & 2048) { // inside the wavefront's XCD (the models may be ordered per XCD, bh_launch_pair_order): its q-th wavefront of the
                               // target takes the (Q - 1 - q)-th group
                const int x = (wid >> 1) & 7, q = ((wid >> 4) << 1) | (wid & 1), qr = A.wg_n1 / 8 - 1 - q;
                wid = ((qr >> 1) << 4) + 2 * x + (qr & 1);
            } else {
                wid = A.wg_n1 - 1 - wid;
            }
        }
    }
    const SwdTarget T = A.t[ty];
    constexpr int MPW = BH_WAVE / J; // models per wavefront
    if (wid * MPW >= A.B) return;
    const int g = lane / J, r = lane & (J - 1), lbase = lane & ~(J - 1);
    const int sidx = wid * MPW + g;
    const bool valid = sidx < A.B;
    const int32_t *perm = T.perm != nullptr ? T.perm : A.perm;
    const int ib = valid ? (perm ? perm[sidx] : sidx) : 0;
    const int Lmax = A.Lmax, K = T.K, ifunc = T.iwave;
    extern __shared__ __align__(16) unsigned char smem_lean[];
    double *omg = reinterpret_cast<double *>(smem_lean + (size_t)wave * wave_lds); // [K]: 2 pi / period
    double *mdl = omg + ((K + 1) & ~1);                                            // [7][Lmax][MPW]
    const int LS = Lmax * MPW;
    constexpr double dc = (double)0.005f, onea = (double)1.5f, twopi = 2.0 * 3.141592653589793, guard_rel = 3.0e-6;
    for (int k = lane; k < K; k += BH_WAVE) omg[k] = twopi / T.periods[k];
    for (int idx = lane; idx < LS; idx += BH_WAVE) {
        const int l = idx / MPW, mg = idx - l * MPW;
        const int sb = wid * MPW + mg;
        const int b = sb < A.B ? (perm ? perm[sb] : sb) : 0;
        float fd = 0.f, fa_ = 1.f, fb = 1.f, fr = 1.f;
        if (sb < A.B && l < A.nlay[b]) {
            const ptrdiff_t o = (ptrdiff_t)b * T.sb + (ptrdiff_t)l * T.sl;
            fd = (float)T.h[o];
            fa_ = (float)T.vp[o];
            fb = (float)T.vs[o];
            fr = (float)T.rho[o];
        }
        mdl[idx] = (double)fd;
        mdl[LS + idx] = (double)fa_;
        mdl[2 * LS + idx] = (double)fb;
        mdl[3 * LS + idx] = (double)fr;
        mdl[4 * LS + idx] = fa::rcp((double)fa_);
        mdl[5 * LS + idx] = fa::rcp((double)fb);
        mdl[6 * LS + idx] = fa::rcp((double)fr);
    }
    int mmax = valid ? A.nlay[ib] : 2;
    mmax = mmax < 1 ? 1 : (mmax > Lmax ? Lmax : mmax);
    int mtop = mmax;
    for (int off = 32; off > 0; off >>= 1) mtop = max(mtop, __shfl_xor(mtop, off));
    mtop = __builtin_amdgcn_readfirstlane(mtop);
    lean_wave_sync();
    LeanModel md;
    md.p = mdl + g;
    md.S = MPW;
    md.LS = LS;

    // ---- driver set-up (surfdisp96.f:124-217; SearchT::init): extremal velocities, start value, input sanity
    float betmx = -1.e20f, betmn = 1.e20f;
    int jmn = 0, jsol = 1;
    bool sane = true;
    for (int i = 0; i < mmax; ++i) {
        const float bi = (float)md.Bv(i), ai = (float)md.A(i), di = (float)md.D(i), ri = (float)md.R(i);
        sane = sane && (ai > 0.0f) && (ai <= 100.0f) && (bi >= 0.0f) && (bi <= 100.0f) && (ri > 0.0f) && (ri < 1.0e6f) &&
               (i == mmax - 1 || (di >= 0.0f && di < 1.0e7f));
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
    float cc1 = (jsol == 0) ? betmn : gtsolh_f32((float)md.A(jmn), (float)md.Bv(jmn));
    cc1 = 0.95f * cc1;
    cc1 = 0.90f * cc1;
    const double cm = (double)cc1, betmxd = (double)betmx;
    const double vh0 = md.Bv(mmax - 1), vh1 = (ifunc == 2) ? md.A(mmax - 1) : betmxd, vsafe = fmin(vh0, vh1);
    // The velocities at which the secular function is not smooth in c or at which getsol's acceptance changes: the half-space's S
    // (and, Rayleigh, P) velocity -- above it the half-space term takes |k - k_beta|: a root below has a mirror image above -- and
    // betmx (a root above it fails the period).  The one of them inside the open cell (lo, hi), 0 if none, -1 if more than one.
    auto special_in_cell = [&](double lo_, double hi_) -> double {
        double sp = 0.0;
        int n = 0;
        if (lo_ < betmxd && betmxd < hi_) sp = betmxd, ++n;
        if (vh0 != betmxd && lo_ < vh0 && vh0 < hi_) sp = vh0, ++n;
        if (vh1 != betmxd && vh1 != vh0 && lo_ < vh1 && vh1 < hi_) sp = vh1, ++n;
        return n > 1 ? -1.0 : sp;
    };
    double *vel = T.vel + (size_t)ib * T.ldv;
    const bool writer = valid && r == 0;
    bool active = valid && K > 0 && sane, guard = false;
#define LEAN_GUARD(n) do { guard = true; if (CNT) greason = (n); } while (0)
    int greason = 0; // (development aid: which rule fired the guard -- 1 water layer, 2 / 3 small start / scan value, 4 step probes, 5 bracket
                     //  probes, 6 a bracket that contains betmx with sign changes on both sides of it or next to it, 7 root at a bracket end or at betmx)
    int errflag = 0;
    if (valid && !sane) {
        errflag = 1;
        if (writer)
            for (int i = 0; i < K; ++i) vel[i] = 0.0;
    }
    if (active && md.Bv(0) <= 0.0) { // water layer on top: the reference's sequence
        LEAN_GUARD(1);
        active = false;
    }
    // ---- search state, the same in every lane of the model
    int k = 0, ph = PH_START, idir = +1, ifirst = 1, nref = 0;
    double c1 = cm, clow = cm, del1 = 0.0, ck = 0.0, omega = 1.0, iom = 1.0;
    bool s1stneg = false;
    double cp = 0.0, delp = 0.0; // the grid point before c1 and its value (third point of the first estimate)
    bool havep = false;
    double lo = 0.0, hi = 0.0, flo = 0.0, fhi = 0.0, p3 = 0.0, fp3 = 0.0, c3 = 0.0, wprev = 0.0;
    bool have3 = false;
    double cell_lo = 0.0, cell_hi = 0.0, pb = 0.0, delb = 0.0;
    bool pp_neg = false; // PH_SPECIAL -> PH_SPECIAL_B: the sign just above the special velocity
    int sp_nb = 0;       // ... and the sign changes found below it
    bool flo_neg = false, chk = false; // chk: this period's start value has been probed (PH_PROBE_START)
    unsigned evals = 0;
    if (active) {
        omega = omg[0];
        iom = fa::rcp(omega);
    }
    constexpr unsigned long long maskJ = (J >= 64) ? ~0ull : ((1ull << J) - 1ull);
    // THE NEXT PERIOD'S FIRST ROUND RIDES ALONG.  The estimate x that a round of clustered trials is centred on -- the inverse
    // quadratic point through the bracket's ends and the grid point before them -- is within 2e-7 |x| of the root in 99.9 %
    // (Rayleigh) / 99 % (Love) of the periods of the bench's models (tools/cpu_scan_steps.py): the scaled secular function is that
    // smooth over three grid steps, except next to a layer velocity.  So with the next period's window beside it the cluster is
    // only NC lanes and the other NR = J - NC lanes evaluate -- at the NEXT period's frequency -- the first round of the next
    // period's scan on the grid anchored at x - 1.5 dc instead of root - 1.5 dc.  If the root comes out within 4e-7 |x| of x,
    // those values ARE the next period's first round (its grid is anchored 4e-7 relative off the root: the reference's own root
    // is known to 1e-6, and the guard covers grids that differ by 3e-6); otherwise they are dropped.
    //   NC: with 8 or 16 lanes per model 2 (x -+ 4e-7 |x|; a miss costs two rounds: J-section, then the cluster again).  The
    // bench's sign change is 14.5 steps from the start value on average: a window of 14 lanes reaches it in 59 % of the periods
    // (one round for the period), the rest take a second round of 16 steps -- c2 with clusters of 2 / 4 / 8 lanes: 0.66 / 0.70 /
    // 0.76 ms.  With 32 and 64 lanes 8 (x -+ 1e-7 |x| 4^i, i < 4): that is where a sampler's windows run, whose models (up to 20
    // thin layers, any order of velocities) have a layer velocity next to every tenth root -- 8 chains with clusters of
    // 2 / 4 / 8 / 16 lanes: 66.6 / 70.0 / 73.1 / 73.1 thousand iterations/s (B = 2048 of the bench's models: 0.478 / 0.489 / 0.500 / 0.556 ms).
    constexpr int NC = (J >= 32) ? BH_LEAN_CLUSTER_BIG : ((J >= 8) ? BH_LEAN_CLUSTER : J / 2), NR = J - NC;
    constexpr bool can_spec = J >= 4; // (every trial count the launcher offers)
    constexpr unsigned long long maskC = (1ull << NC) - 1ull, maskR = (1ull << NR) - 1ull;
    const double invJ1 = 1.0 / (double)(J + 1);
    unsigned nrounds = 0;
    long long t_eval = 0;
    const long long t_start = ((CNT && A.neval != nullptr)) ? clock64() : 0;
    while (__ballot(active) != 0ull) {
        ++nrounds;
        // ---- this lane's trial velocity (and frequency)
        double cev = c1, cprev = c1; // cprev: the grid point before a scan trial
        double om_l = omega, iom_l = iom;
        bool pt = false;             // this lane's value takes part in the refinement's decision
        bool spec = false;           // the model's lanes beyond the cluster's carry the next period's first round
        double xc = 0.0;             // centre of the cluster
        if (ph <= PH_SCAN) {
            // The grid of getsol's scan (:437-446).  The reference forms it by repeated additions of dc; here point n is
            // base + n dc in one fused operation -- the two differ in the last bits (1e-16 relative: a thousandth of what the
            // fast arithmetic itself moves a sign change by, swd_fa.h).
            const bool start = ph == PH_START;
            if (!start) { // (label 1000 of getsol: the floor in a reversed search)
                if (idir > 0) {
                    if (c1 + dc <= clow) c1 = clow;
                } else if (c1 - dc <= clow) {
                    idir = +1;
                    c1 = clow;
                    havep = false;
                }
            }
            const double base = (start && c1 + dc <= clow) ? clow : c1;
            const int n = start ? r : r + 1; // steps from there
            const double step = (idir > 0 || start) ? dc : -dc;
            cprev = __builtin_fma((double)(n - 1), step, base);
            cev = (n == 0) ? c1 : __builtin_fma((double)n, step, base);
        } else if (ph == PH_REF1) {
            cev = __builtin_fma(hi - lo, (double)(r + 1) * invJ1, lo);
            pt = cev > lo && cev < hi;
        } else if (ph == PH_REFC) {
            double x = 0.0;
            bool ok = false;
            if (have3) { // inverse quadratic interpolation through the three points
                const double d12 = flo - fhi, d1p = flo - fp3, d2p = fhi - fp3;
                const double den = d12 * d1p * d2p;
                if (den != 0.0) {
                    const double id = fa::rcp(den);
                    x = (lo * fhi * fp3 * d2p - hi * flo * fp3 * d1p + p3 * flo * fhi * d12) * id;
                    ok = x > lo && x < hi;
                }
            }
            if (!ok) {
                x = lo - flo * (hi - lo) * fa::rcp(fhi - flo);
                if (!(x > lo && x < hi)) x = 0.5 * (lo + hi);
            }
            spec = can_spec && k + 1 < K;
            xc = x;
            const int nc = spec ? NC : J; // lanes of the cluster
            if (r < nc) {
                const int h = nc / 2;
                const double w = ((spec && NC == 2) ? BH_LEAN_CLUSTER_W : 1.0e-7) * fabs(x);
                cev = (r < h) ? x - __builtin_ldexp(w, 2 * (h - 1 - r)) : x + __builtin_ldexp(w, 2 * (r - h));
                pt = cev > lo && cev < hi;
            } else { // the next period's first round on the grid anchored at the estimate (k + 1 >= 1: clow = cm)
                const int s_ = r - NC;
                const double c1n = x - onea * dc;
                const double basen = (c1n + dc <= cm) ? cm : c1n;
                cprev = __builtin_fma((double)(s_ - 1), dc, basen);
                cev = (s_ == 0) ? c1n : __builtin_fma((double)s_, dc, basen);
                om_l = omg[k + 1];
                iom_l = fa::rcp(om_l);
            }
        } else if (J >= 32 && (ph == PH_SPECIAL || ph == PH_SPECIAL_B)) { // (fewer lanes: such a cell is guarded at once)
            // lanes 0 / 1: the two points s (1 -+ 3e-6) next to the special velocity s of the cell; the others: J - 2 points between
            // there and the cell's lower (PH_SPECIAL) / upper (PH_SPECIAL_B) end, their distances from s in geometric progression
            // (the sign changes of such a cell crowd towards s: the root just below it, its image just above, the next ones at
            // 2 - 10 times the distance)
            const double sp = special_in_cell(cell_lo, cell_hi);
            const double g = guard_rel * sp;
            const bool up = ph == PH_SPECIAL_B;
            const float lg = log2f((float)(((up ? cell_hi - sp : sp - cell_lo)) / g)) * (1.0f / (float)(J - 1));
            const double dist = g * (double)exp2f(lg * (float)(r - 1));
            cev = (r == 0) ? (up ? sp + g : sp - g) : ((r == 1) ? sp + g : (up ? sp + dist : sp - dist));
        } else if (J >= 32 && ph == PH_SPECIAL_C) {
            // J points inside s (1 -+ 3e-6), J / 2 on either side of s, their distances from s in geometric progression from 1e-8 s
            // (ratio 300^(2/J): 1.43 with 32 lanes)
            const double sp = special_in_cell(cell_lo, cell_hi);
            const double dist = (1.0e-8 * sp) * (double)exp2f((8.2288187f / (float)(J / 2)) * (float)(r >> 1));
            cev = (r & 1) ? sp + dist : sp - dist;
        } else if (ph == PH_PROBE_START) {
            cev = (r == 0) ? c1 - guard_rel * c1 : ((r == 1) ? c1 + guard_rel * c1 : c1);
        } else if (ph == PH_PROBE_STEP) {
            const double l_ = fmin(c1, pb), h_ = fmax(c1, pb);
            cev = (r == 0) ? l_ + guard_rel * h_ : h_ - guard_rel * h_;
        } else {
            cev = (r == 0) ? cell_hi + guard_rel * fabs(c3) : cell_lo - guard_rel * fabs(c3);
        }
        if (!active || !(cev > 0.0)) cev = 1.0; // (finished or broken models compute on a harmless value)
        const long long te0 = ((CNT && A.neval != nullptr)) ? clock64() : 0;
        const double wvno = om_l * fa::rcp(cev);
        const double del = (ifunc == 2) ? lean_rayleigh(wvno, om_l, iom_l, md, mmax, mtop) : lean_love(wvno, om_l, md, mmax, mtop);
        if ((CNT && A.neval != nullptr)) t_eval += clock64() - te0;

        // ---- the round's decision.  Signs and "not a number" travel as ballots; one event code per lane and a ballot give the
        // first event of a window of lanes; the VALUES at the events are fetched by two small sets of exchanges (all outside the
        // per-model branches); the velocities there are recomputed from the lane index.
        const bool dneg = sign_neg(del);
        const unsigned long long mneg = (__ballot(dneg) >> lbase) & maskJ;
        const unsigned long long m_small = (__ballot(!(fabs(del) >= fa::SIGN_FLOOR)) >> lbase) & maskJ;
        const unsigned long long m_pt = (__ballot(pt) >> lbase) & maskJ;
        const bool inB = ph == PH_REFC && spec && r >= NC;
        int ev = 0; // 1 floor of a reversed search reached, 2 sign change, 3 step needs the guard's probes, 4 out of bounds
        // (NOT guarded: a root within ~1e-6 c of a grid point with a second root less than a step away -- the cell that holds both
        // shows no sign change, so one grid sees a bracket where the reference's, 1e-6 c beside it, walks past the pair to another
        // mode, or the other way round -- and, in a cell with a half-space velocity, two sign changes closer to each other than the
        // count's points resolve (PH_SPECIAL: a distance ratio of <= 1.23; pairs INSIDE the guard's 3e-6 are looked for: PH_SPECIAL_C): 7 models
        // in 9.2 million drawn from a sampler's prior, three of them with another failure flag further along the other branch
        // (profiles/r06_fuzz_prior_final.txt).  Two forms of a rule for the first were built in round 6 and dropped:
        // "|f| two orders of magnitude below both neighbours'" also fired on a near-tangency of one of the bench's 16 384 models
        // (a re-run launch every fourth step: 0.65 -> 0.83 ms); step probes behind every point that the local slope sends through
        // zero, plus a look at the grid point after an accepted root next to its bracket's end, caught 3 of 7 and cost every round
        // two more neighbour exchanges: c2 0.670 -> 0.702 ms.  Neither can see a channel mode's steep pole-zero pair.)
        if (ph <= PH_SCAN || inB) {
            const bool first = ph == PH_START || inB;        // the window is a period's first round: trial 0 = the start value,
            const int rr = inB ? r - NC : r;                 // the steps upward (consumed only if the start value says so)
            if (!(first && rr == 0)) {
                const bool down = !first && idir < 0;
                const double a_ = fmin(cprev, cev), b_ = fmax(cprev, cev);
                const bool refneg = inB ? ((mneg >> NC) & 1ull) != 0ull : (first ? (mneg & 1ull) != 0ull : sign_neg(del1));
                const double floor_ = inB ? cm : clow;
                if (down && cev <= floor_) ev = 1;
                else if (dneg != refneg) ev = 2;
                else if (b_ >= vsafe && ((a_ <= vh0 && vh0 <= b_) || (a_ <= vh1 && vh1 <= b_))) ev = 3;
                else if (cev < cm || cev >= betmxd + dc) ev = 4;
            }
        } else if (ph <= PH_REFC) {
            ev = (pt && dneg != sign_neg(flo)) ? 2 : 0;
        }
        const unsigned long long m_ev = (__ballot(ev != 0) >> lbase) & maskJ;
        // stage 1: the refinement's window A = all J lanes, or the cluster's NC
        const int nA = (ph == PH_REFC && spec) ? NC : J;
        const unsigned long long m_evA = (ph == PH_REFC && spec) ? (m_ev & maskC) : m_ev;
        const int eA = m_evA ? (int)__builtin_ctzll(m_evA) : nA; // first trial beyond the sign change (nA: none)
        const int r0 = m_pt ? (int)__builtin_ctzll(m_pt) : 0, r1 = m_pt ? 63 - (int)__builtin_clzll(m_pt) : -1; // the trials that take part: one run of lanes
        const int i3 = (eA < nA) ? ((eA + 1 <= r1) ? eA + 1 : eA - 2) : r1 - 1; // a third point next to the new bracket
        const double dA_e = __shfl(del, lbase + (eA < nA ? eA : nA - 1)), dA_1 = __shfl(del, lbase + (eA > 0 ? eA - 1 : 0));
        const double d_3 = __shfl(del, lbase + (i3 >= 0 && i3 < J ? i3 : 0)), d_r1 = __shfl(del, lbase + (r1 >= 0 ? r1 : 0));

        int todo = 0;          // 2 root search failed; 3 root c3 accepted; 4 period done with c3
        bool scan_now = false; // stage 2 consumes a scan window for this model
        int w0 = 0;            // its first lane
        bool sc_first = false; // it is a period's first round
        // A bracket with betmx or a half-space velocity inside (rare: the round's wavefront-uniform branch).  Above the half-space's
        // S velocity the half-space term takes |k - k_beta|: a root just below it has a mirror image just above, and further sign
        // changes may follow up to betmx -- the reference's function has them, and getsol takes whatever nevill ends at provided it
        // is not above betmx (:468-471).  A cell with SEVERAL sign changes is therefore a matter of nevill's sequence (which root;
        // with betmx inside: whether the period fails), a cell with ONE is not.  Two rounds count them: the two points s (1 -+ 3e-6)
        // next to the special velocity s and J - 2 points between s and the cell's lower end, then J - 2 between s and its upper
        // end, their distances from s in geometric progression (ratio <= 1.23 with 32 lanes per model: the sign changes of such
        // a cell crowd towards s).  With fewer than 32 lanes per model the cell is guarded at once.
        //   * one sign change in the whole cell: a third round looks inside s (1 -+ 3e-6) (PH_SPECIAL_C below), then the refinement
        //     goes on in the sign change's section (s = betmx and the sign change above it: the period fails as the reference's
        //     does -- nevill ends within 1e-6 c of a sign change above betmx (1 + 3e-6));
        //   * anything else -- several sign changes, one between the two points next to s, a value that is no number -- the guard.
        // (Before this rule a cell with betmx inside was always guarded -- 99 % of the guarded models of LVZ-rich batches,
        // profiles/r05_lean_guard.txt -- and a cell with a half-space velocity below betmx inside was refined like any other:
        // on models drawn from a sampler's prior, one in 10^4 then came back with another root of the cell than the reference's,
        // up to 1.5e-3 away.)
        bool special_now = false;
        // Third round, for a cell whose one sign change is about to be refined (or to fail the period): J points INSIDE the guard's
        // distance of s.  The two points s (1 -+ 3e-6) showed the same sign, so between them lies no sign change or a PAIR -- a root
        // 4e-7 below a half-space velocity with its image 9e-7 above: with a third sign change 6.5e-6 above, the cell counted one,
        // this search refined it and the reference's nevill ended at the first, 6.9e-6 away (the suite's fuzz, seed 3790146708;
        // docs/HISTORY.md).  Any other sign in there than the two points': the guard.
        if (J >= 32 && __ballot(active && ph == PH_SPECIAL_C) != 0ull) {
            if (active && ph == PH_SPECIAL_C) {
                special_now = true;
                evals += (unsigned)J;
                if (m_small != 0ull || mneg != (pp_neg ? maskJ : 0ull)) {
                    LEAN_GUARD(6);
                } else if (sp_nb != 0) { // (the cell's one sign change lies above betmx: getsol's "c1 > betmx", :470)
                    todo = 2;
                } else {
                    // (the section is a few per cent of the cell wide and holds one sign change: a cluster around the secant
                    // point closes in on it; if not, the next round is a J-section of the section)
                    have3 = false;
                    nref = 0;
                    wprev = hi - lo;
                    ph = PH_REFC;
                }
            }
        }
        if (J >= 32 && __ballot(active && (ph == PH_SPECIAL || ph == PH_SPECIAL_B)) != 0ull) {
            // Signs in the order of the distance from s.  Element 0 = the point next to s (lane 0: s - g below, lane 1: s + g above),
            // elements 1 .. J - 2 = lanes 2 .. J - 1, element J - 1 = the cell's end on that side (sign of cell_lo: flo_neg; the other
            // end's is the opposite).  Bit e of ch: a sign change between elements e and e + 1.
            const bool up = ph == PH_SPECIAL_B;
            const bool s_near = up ? pp_neg : (mneg & 1ull) != 0ull, s_end = up ? !flo_neg : flo_neg;
            const unsigned long long inner = (J > 2) ? ((mneg >> 2) & ((1ull << (J - 2)) - 1ull)) : 0ull;
            const unsigned long long seq = (s_near ? 1ull : 0ull) | (inner << 1) | ((s_end ? 1ull : 0ull) << (J - 1));
            const unsigned long long ch = (seq ^ (seq >> 1)) & ((1ull << (J - 1)) - 1ull);
            const int nch = __builtin_popcountll(ch);
            const int te = ch ? (int)__builtin_ctzll(ch) : 0;
            // the lanes of elements te (nearer to s) and te + 1 (-1: the cell's end)
            const int e_near = (te == 0) ? (up ? 1 : 0) : te + 1, e_far = (te + 1 == J - 1) ? -1 : te + 2;
            const double d_near = __shfl(del, lbase + e_near), d_far = __shfl(del, lbase + (e_far < 0 ? 0 : e_far));
            const double c_near = __shfl(cev, lbase + e_near), c_far = __shfl(cev, lbase + (e_far < 0 ? 0 : e_far));
            if (active && (ph == PH_SPECIAL || ph == PH_SPECIAL_B)) {
                special_now = true;
                evals += (unsigned)J;
                const bool is_betmx = special_in_cell(cell_lo, cell_hi) == betmxd;
                if (m_small != 0ull || nch > 1) {
                    LEAN_GUARD(6);
                } else if (!up) { // below s
                    pp_neg = (mneg & 2ull) != 0ull;
                    if (pp_neg != ((mneg & 1ull) != 0ull)) LEAN_GUARD(6); // (a sign change between the two points next to s)
                    sp_nb = nch;
                    if (nch == 1) { // the section with the sign change: [far, near]
                        hi = c_near;
                        fhi = d_near;
                        if (e_far >= 0) {
                            lo = c_far;
                            flo = d_far;
                        }
                    }
                    ph = PH_SPECIAL_B;
                } else { // above s: the cell's sign changes are counted
                    if (sp_nb + nch != 1) {
                        if (is_betmx && sp_nb == 0 && nch > 0) todo = 2; // (unreachable with nch <= 1; kept for the rule's sake)
                        else LEAN_GUARD(6);
                    } else {
                        sp_nb = (nch == 1 && is_betmx) ? 1 : 0; // (from here on: the cell's one sign change lies above betmx -- the period fails)
                        if (nch == 1) { // [near, far]
                            lo = c_near;
                            flo = d_near;
                            if (e_far >= 0) {
                                hi = c_far;
                                fhi = d_far;
                            }
                        }
                        ph = PH_SPECIAL_C; // (a look inside the guard's distance of s first)
                    }
                }
            }
        }
        if (active) {
            scan_now = ph <= PH_SCAN;
            sc_first = ph == PH_START;
            bool probed = false;
            if (special_now) {
                // (decided above)
            } else if (ph == PH_PROBE_STEP) {
                evals += 2;
                const bool n1 = sign_neg(del1);
                if ((m_small & 3ull) || ((mneg & 1ull) != 0ull) != n1 || ((mneg & 2ull) != 0ull) != n1) {
                    LEAN_GUARD(4);
                } else { // the step is an ordinary one
                    cp = c1;
                    delp = del1;
                    havep = true;
                    c1 = pb;
                    del1 = delb;
                    ph = PH_SCAN;
                    if (c1 < cm || c1 >= betmxd + dc) todo = 2;
                }
            } else if (ph == PH_REF1 || ph == PH_REFC) {
                // the velocity of trial i of this round's window A
                const double olo = lo, ohi = hi, oflo = flo, ofhi = fhi;
                const int nc = nA, h = nc / 2;
                const double w = ((nA == 2) ? BH_LEAN_CLUSTER_W : 1.0e-7) * fabs(xc);
                const bool sect = ph == PH_REF1;
                auto ctrial = [&](int i) -> double {
                    return sect ? __builtin_fma(ohi - olo, (double)(i + 1) * invJ1, olo)
                                : ((i < h) ? xc - __builtin_ldexp(w, 2 * (h - 1 - i)) : xc + __builtin_ldexp(w, 2 * (i - h)));
                };
                evals += (unsigned)__builtin_popcountll(m_pt);
                ++nref;
                if (sect && m_pt != 0ull) {
                    // Several sign changes among the section points: a cell with several roots (modes closer than dc: short periods
                    // over thick slow layers; above a half-space velocity the function's oscillations) -- which of them nevill ends
                    // at is a matter of its sequence: the guard.  (A cluster's round that does not close in is followed by a J-section:
                    // a cell whose first estimate is poor -- a cell with several roots -- comes by here.)
                    const int np = r1 - r0 + 1;
                    const unsigned long long sg = (mneg >> r0) & ((np >= 64) ? ~0ull : ((1ull << np) - 1ull));
                    const int inner = __builtin_popcountll((sg ^ (sg >> 1)) & ((np >= 65) ? ~0ull : ((1ull << (np - 1)) - 1ull)));
                    const int ends = ((((sg & 1ull) != 0ull) != sign_neg(oflo)) ? 1 : 0) + (((((sg >> (np - 1)) & 1ull) != 0ull) != sign_neg(ofhi)) ? 1 : 0);
                    if (inner + ends > 1) LEAN_GUARD(6);
                }
                if (m_pt != 0ull) {
                    if (eA < nA) { // first trial beyond the sign change
                        hi = ctrial(eA);
                        fhi = dA_e;
                        if (eA - 1 >= r0) {
                            lo = ctrial(eA - 1);
                            flo = dA_1;
                        }
                    } else { // all trials on the lower end's side
                        lo = ctrial(r1);
                        flo = d_r1;
                    }
                    if (i3 >= r0 && i3 <= r1) {
                        p3 = ctrial(i3);
                        fp3 = d_3;
                        have3 = true;
                    } else if (hi != ohi) {
                        p3 = ohi;
                        fp3 = ofhi;
                        have3 = true;
                    } else if (lo != olo) {
                        p3 = olo;
                        fp3 = oflo;
                        have3 = true;
                    }
                }
                // The estimate's round did not close in (a poor estimate: an end value that is not the function's, a kink, several
                // roots in the cell): J-section next.  If it was the cell's first round the J-section takes the WHOLE cell again,
                // not the part the estimate's trials left of it: it counts the cell's sign changes (above), and a cell with three
                // roots cut at a poor estimate shows one.
                // (the cell's first round: "closed in" = the sign change lies BETWEEN two of the cluster's trials -- an estimate that is
                // merely in the right fifth of a cell with three roots leaves a bracket around one of them)
                const bool poor = ph == PH_REFC && (hi - lo > 0.25 * wprev || (nref == 1 && !(eA > 0 && eA < nA)));
                if (poor && nref == 1) {
                    lo = olo;
                    hi = ohi;
                    flo = oflo;
                    fhi = ofhi;
                }
                ph = poor ? PH_REF1 : PH_REFC;
                wprev = hi - lo;
                if (hi - lo <= 1.3e-6 * fabs(hi) || nref >= 16 || m_pt == 0ull) {
                    // the root: inverse quadratic interpolation through the bracket's ends and the nearest third point (the secant
                    // point alone is off by a good part of the bracket where the function bends), else the secant point
                    c3 = lo - flo * (hi - lo) * fa::rcp(fhi - flo);
                    if (have3) {
                        const double d12 = flo - fhi, d1p = flo - fp3, d2p = fhi - fp3;
                        const double den = d12 * d1p * d2p;
                        const double xq = (lo * fhi * fp3 * d2p - hi * flo * fp3 * d1p + p3 * flo * fhi * d12) * fa::rcp(den);
                        if (den != 0.0 && xq >= lo && xq <= hi) c3 = xq;
                    }
                    if (!(c3 >= lo && c3 <= hi)) c3 = 0.5 * (lo + hi);
                    todo = 3;
                }
            } else if (ph == PH_PROBE_START) {
                // Signs just below (nA), at (n2) and just above (nB) the start value.  No sign change next to it: nothing hinges on
                // it.  One sign change r, and below it the sign of the first period's start value (the fundamental mode's own
                // polarity): harmless -- a start value below r searches upward and one above it downward (:425-435), both find r
                // in their first cell, as this search will.  The other polarity: a start value below r searches downward and one
                // above it upward, AWAY from r, to different roots -- which one the reference takes hangs on the last digits of
                // its previous root: the guard.
                evals += 3;
                const bool nA = (mneg & 1ull) != 0ull, nB = (mneg & 2ull) != 0ull, n2 = (mneg & 4ull) != 0ull;
                if ((m_small & 7ull) || (nA != nB && nA != s1stneg) || (nA == nB && n2 != nA)) LEAN_GUARD(2);
                ph = PH_START; // (the start round again, its start value checked)
                chk = true;
            } else if (ph == PH_PROBE_ACC) {
                evals += 2;
                if ((m_small & 3ull) || ((mneg & 1ull) != 0ull) == flo_neg || ((mneg & 2ull) != 0ull) != flo_neg) LEAN_GUARD(5);
                todo = 4;
                probed = true; // (the period ends a round after its cluster: nothing rode along)
            }
            if (todo == 3) { // the guard at an accepted bracket
                const double m2 = 2.0 * dc;
                todo = 4;
                if (fabs(c3 - vh0) < m2 || fabs(c3 - vh1) < m2 || fabs(c3 - betmxd) < m2) {
                    const double eps = guard_rel * fabs(c3);
                    if (cell_hi - c3 < eps || c3 - cell_lo < eps || fabs(c3 - betmxd) < eps) {
                        LEAN_GUARD(7);
                    } else {
                        ph = PH_PROBE_ACC;
                        todo = 0;
                    }
                }
            }
            if (guard) { // this run's results are not used (the engine runs the model again)
                active = false;
                todo = 0;
            }
            if (todo == 4) { // getsol after the refinement (:468-471), then the driver's next period (:253-272)
                if (c3 > betmxd) {
                    todo = 2;
                } else {
                    ck = c3;
                    if (writer) vel[k] = (double)(float)ck;
                    k = k + 1;
                    if (k >= K) {
                        active = false;
                    } else {
                        omega = omg[k];
                        iom = fa::rcp(omega);
                        ifirst = 0;
                        c1 = ck - onea * dc;
                        clow = cm;
                        ph = PH_START;
                        if (!probed && spec && fabs(c3 - xc) <= BH_LEAN_RIDE_TOL * fabs(c3)) { // the upper lanes' values are this period's first round
                            c1 = xc - onea * dc; // (the grid they evaluated)
                            scan_now = true;
                            sc_first = true;
                            w0 = NC;
                        }
                    }
                }
            }
            if (todo == 2) { // no root at period k: err, zeros from there on (:313-354)
                errflag = 1;
                if (writer)
                    for (int i = k; i < K; ++i) vel[i] = 0.0;
                active = false;
            }
        }
        // stage 2: the scan window [w0, w0 + wn) of the models that consume one this round
        const int wn = (w0 != 0) ? NR : J;
        const unsigned long long m_evS = (w0 != 0) ? ((m_ev >> NC) & maskR) : m_ev;
        const int e = m_evS ? (int)__builtin_ctzll(m_evS) : wn; // first event among the window's trials (wn: none)
        const int wb = lbase + w0;
        const int ev_e = __shfl(ev, wb + (e < wn ? e : wn - 1));
        const double d_e = __shfl(del, wb + (e < wn ? e : wn - 1)), d_em = __shfl(del, wb + (e > 0 ? e - 1 : 0)),
                     d_em2 = __shfl(del, wb + (e > 1 ? e - 2 : 0)), d_first = __shfl(del, wb), d_second = __shfl(del, wb + 1);
        if (active && scan_now) {
            const bool start = sc_first;
            const unsigned long long msm = (w0 != 0) ? ((m_small >> NC) & maskR) : m_small;
            // the velocity of trial t of the window (what its lane evaluated)
            const double base = (start && c1 + dc <= clow) ? clow : c1;
            const double step = (start || idir > 0) ? dc : -dc;
            const double c1s = c1;
            auto cgrid = [&](int t) -> double { return (start && t == 0) ? c1s : __builtin_fma((double)(start ? t : t + 1), step, base); };
            int off = 0, todo2 = 0; // 1 bracket (c1, del1) - (pb, delb) found; 2 root search failed
            bool consume = true;
            // A root within the guard's distance of the start value: its sign there decides the direction of the search (:425-435),
            // and the reference's start value -- its own previous root - 1.5 dc -- lies up to 3e-6 c beside this one.  |f(c1)|
            // against the change of f over the first step (linear over a step but next to a layer velocity: a root within 3e-6 c
            // of c1 gives a ratio of 3e-6 c / dc = 2e-3; the rule looks closer below 1e-2): the round is not consumed, the next one
            // probes c1 (1 -+ 3e-6) -- all three signs equal: the start round again, else the guard.  (On models drawn from a
            // sampler's prior one search in 10^6 took the other direction than the reference's, and failed where it did not.)
            const bool look_closer = start && !chk && ifirst != 1 && !(msm & 1ull) && !(flip & 4096) && fabs(d_first) < 1.0e-2 * fabs(d_second - d_first);
            if (look_closer) {
                ph = PH_PROBE_START;
                consume = false;
            } else if (start) {
                chk = false;
                ++evals;
                if (msm & 1ull) LEAN_GUARD(2);
                del1 = d_first;
                havep = false;
                if (ifirst == 1) s1stneg = sign_neg(d_first);
                idir = (ifirst != 1 && s1stneg != sign_neg(d_first)) ? -1 : +1;
                off = 1;
                if (idir > 0) {
                    if (c1 + dc <= clow) {
                        c1 = clow;
                        havep = false;
                    }
                } else {
                    consume = false; // reversed search: the upward steps are not the scan's
                }
                ph = PH_SCAN;
            }
            if (consume) {
                // trials off .. e are consumed (a floor event: off .. e - 1)
                const int last = (e < wn) ? ((ev_e == 1) ? e - 1 : e) : wn - 1;
                if (last >= off) {
                    evals += (unsigned)(last - off + 1);
                    const unsigned long long used = ((last >= 63) ? ~0ull : ((1ull << (last + 1)) - 1ull)) & ~((1ull << off) - 1ull);
                    if (msm & used) LEAN_GUARD(3);
                }
                // (c1, del1) and the point before it after the steps that precede the event (e = wn: after all of them)
                const int ns = e - off; // steps taken before the event
                if (ns >= 2) {
                    cp = cgrid(e - 2);
                    delp = d_em2;
                    havep = true;
                } else if (ns == 1) {
                    cp = c1;
                    delp = del1;
                    havep = true;
                }
                if (ns >= 1) {
                    c1 = cgrid(e - 1);
                    del1 = d_em;
                }
                if (e < wn) {
                    if (ev_e == 1) {
                        idir = +1;
                        c1 = clow;
                        havep = false;
                    } else if (ev_e == 2) {
                        pb = cgrid(e);
                        delb = d_e;
                        todo2 = 1;
                    } else if (ev_e == 3) {
                        pb = cgrid(e);
                        delb = d_e;
                        ph = PH_PROBE_STEP;
                    } else {
                        todo2 = 2;
                    }
                }
            }
            if (todo2 == 1) { // a bracket: set up its refinement (SearchT::bracketed)
                cell_lo = fmin(c1, pb);
                cell_hi = fmax(c1, pb);
                flo = (c1 < pb) ? del1 : delb;
                fhi = (c1 < pb) ? delb : del1;
                flo_neg = sign_neg(flo);
                lo = cell_lo;
                hi = cell_hi;
                p3 = cp;
                fp3 = delp;
                have3 = havep;
                nref = 0;
                wprev = hi - lo;
                ph = have3 ? PH_REFC : PH_REF1;
                {
                    // betmx or a half-space velocity inside the cell: the next round counts the cell's sign changes (PH_SPECIAL above).
                    // Two of them in one cell, or one within the guard's distance of a grid point (the reference's cell may be the
                    // neighbouring one): the reference's sequence.
                    const double sp = special_in_cell(cell_lo, cell_hi);
                    if (sp != 0.0) {
                        ph = PH_SPECIAL;
                        if (J < 32 || !(sp > 0.0 && sp - guard_rel * sp > cell_lo && sp + guard_rel * sp < cell_hi)) LEAN_GUARD(6);
                    } else if (cell_lo > vsafe) {
                        // A cell above a half-space velocity: the reference's function oscillates there (leaking modes' images): the
                        // refinement starts with a J-section, which counts the sign changes it sees (below).  (No measurable cost: c4 8.72
                        // against 8.68 x 10^4 chain-iterations/s with / without.)
                        have3 = false;
                        ph = PH_REF1;
                    }
                }
            }
            if (guard) active = false;
            if (todo2 == 2 && active) { // no root at period k: err, zeros from there on (:313-354)
                errflag = 1;
                if (writer)
                    for (int i = k; i < K; ++i) vel[i] = 0.0;
                active = false;
            }
        }
    }
    if (writer) {
        T.err[ib] = errflag;
        if (guard && T.gcount != nullptr) { // to be run again with the reference's sequence
            T.glist[atomicAdd(T.gcount, 1)] = ib;
            atomicAdd(T.gcount + 2 * BH_MAX_TARGETS, 1); // (cumulative, for bh_engine_guard_stats)
        }
    }
    if ((CNT && A.neval != nullptr)) {
        unsigned long long tot = writer ? evals : 0u, lps = tot * (unsigned long long)(valid ? mmax - 1 : 0);
        for (int off = 32; off > 0; off >>= 1) {
            tot += __shfl_xor(tot, off);
            lps += __shfl_xor(lps, off);
        }
        if (lane == 0) {
            atomicAdd(A.neval, tot);
            atomicAdd(A.neval + (ifunc == 2 ? 8 : 9), tot);
            atomicAdd(A.neval + (ifunc == 2 ? 10 : 11), lps);
            // development aid: rounds and cycles of the wavefronts, [1..3] Rayleigh (sum of rounds, sum of cycles, most rounds), [4..6] Love
            const int o = (ifunc == 2) ? 1 : 4;
            atomicAdd(A.neval + o, (unsigned long long)nrounds);
            atomicAdd(A.neval + o + 1, (unsigned long long)(clock64() - t_start));
            atomicMax(A.neval + o + 2, (unsigned long long)nrounds);
            atomicAdd(A.neval + 7, 1ull);
            atomicAdd(A.neval + (ifunc == 2 ? 12 : 13), (unsigned long long)t_eval); // cycles inside the secular evaluations
        }
        if (writer && guard && greason >= 1 && greason <= 7) // guard reasons, 16 bits each: 1 .. 4 in word 14, 5 .. 7 in word 15
            atomicAdd(A.neval + (greason <= 4 ? 14 : 15), 1ull << (16 * ((greason - 1) & 3)));
    }
}
} // namespace

// LDS of one workgroup (LEAN_WPB wavefronts, a private region each)
static size_t lean_wave_lds(int J, int Lmax, int Kmax)
{
    return ((((size_t)((Kmax + 1) & ~1) + (size_t)7 * Lmax * (BH_WAVE / J)) * sizeof(double)) + 15) & ~(size_t)15;
}
size_t bh_swd_lean_lds_bytes(int J, int Lmax, int Kmax) { return LEAN_WPB * lean_wave_lds(J, Lmax, Kmax); }

// Trials per model and round for a call of `nt` targets over B models.  A round of J trials costs a wavefront what one trial
// costs, so few models get many trials (the latency regime: fewer rounds) and many models few (the throughput regime: fewer
// evaluations that the scan does not consume).  Measured on the c2 shape (two targets; ms per step with 4 / 8 / 16 trials):
// 128 pairs (16 / 32 / 64 trials) 0.60 / 0.45 / 0.40, 2048: 0.64 / 0.47 / 0.50, 4096: 0.66 / 0.58 / 0.98;
// 8192 pairs - / 1.02 / 0.81, 12 288: 1.69 / 1.27 / 1.28, 16 384: 1.69 / 1.29 / 1.56, 24 576: 2.14 / 2.07 / 2.19,
// 32 768: 2.15 / 2.56 / -, 65 536: 4.31 / 4.68 / 5.49, 131 072: 7.99 / 8.98 / 10.7.
// A function of the call's shape alone, NOT of the device: a model's result depends on it in the last bits (the refinement's
// trial points do), and a sampler's windows must not (it pins the number: bh_engine_set_swd_trials).
int bh_swd_lean_trials(int B, int nt)
{
    const long pairs = (long)B * nt;
    return pairs <= 1024 ? 64 : (pairs <= 5120 ? 32 : (pairs <= 10240 ? 16 : (pairs <= 28672 ? 8 : 4)));
}

// All targets of `a` (fundamental-mode phase velocities, a.t[t].look = trials per round, gcount / glist set) in one launch.
int bh_launch_swd_lean(const SwdMultiArgs &a0, hipStream_t stream, SwdLaunchInfo *info)
{
    SwdMultiArgs a = a0;
    int kmax = 0, jmin = BH_WAVE;
    long wmax = 1, wsum = 0, nw[BH_MAX_TARGETS] = {0};
    for (int t = 0; t < a.ntargets; ++t) {
        const int J = a.t[t].look;
        if (J < 4 || J > BH_WAVE || (J & (J - 1)) != 0 || a.t[t].igr != 0 || a.t[t].mode > 1) return -1;
        kmax = a.t[t].K > kmax ? a.t[t].K : kmax;
        jmin = J < jmin ? J : jmin;
        const int mpw = BH_WAVE / J;
        nw[t] = (a.B + mpw - 1) / mpw;
        wmax = nw[t] > wmax ? nw[t] : wmax;
        wsum += nw[t];
    }
    const size_t wave_lds = lean_wave_lds(jmin, a.Lmax, kmax);
    const size_t lds = LEAN_WPB * wave_lds;
    if (lds > 64 * 1024) return -1;
    dim3 grid((unsigned)((wmax + LEAN_WPB - 1) / LEAN_WPB), (unsigned)a.ntargets);
    a.wg_n0 = a.wg_n1 = 0;
    if (a.ntargets == 2) { // interleaved (see the kernel)
        a.wg_n0 = (int)nw[0];
        a.wg_n1 = (int)nw[1];
        grid = dim3((unsigned)((wsum + LEAN_WPB - 1) / LEAN_WPB), 1);
    }
    const dim3 block(BH_WAVE * LEAN_WPB);
    if (info != nullptr) {
        info->workgroups = grid.x * grid.y;
        info->waves = wsum;
        info->lds = lds;
        info->fast_arith = 1;
        info->restarts_in_place = 0;
    }
    int flip = bh_tuning().swd_lean_flip;
    if (a.ntargets == 2 && nw[0] == nw[1] && nw[1] % 16 == 0) flip |= 2048; // (the opposite order can stay inside the XCDs)
    const int J = a.t[0].look; // (one trial count per launch: the kernel is compiled per count)
    for (int t = 1; t < a.ntargets; ++t)
        if (a.t[t].look != J) return -1;
    const bool cnt = a.neval != nullptr; // (the build with the counters: bh_engine_set_instrumentation)
#define LEAN_LAUNCH(JJ)                                                                                                             \
    do {                                                                                                                            \
        if (cnt) hipLaunchKernelGGL((swd_lean_kernel<JJ, true>), grid, block, lds, stream, a, (int)wave_lds, flip);                 \
        else hipLaunchKernelGGL((swd_lean_kernel<JJ, false>), grid, block, lds, stream, a, (int)wave_lds, flip);                    \
    } while (0)
    switch (J) {
    case 4: LEAN_LAUNCH(4); break;
    case 8: LEAN_LAUNCH(8); break;
    case 16: LEAN_LAUNCH(16); break;
    case 32: LEAN_LAUNCH(32); break;
    default: LEAN_LAUNCH(64); break;
    }
#undef LEAN_LAUNCH
    return 0;
}
