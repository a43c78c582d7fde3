// bayhunter_amd/csrc/chain_kernel.hip -- device-resident rj-McMC chain step (one lane = one chain).
//
// Replaces, for thousands of chains advanced in lock-step, the host part of the reference's
// per-chain sampler around the forward call (src/SingleChain.py:511-589 `iterate`):
//   chain_propose_kernel   draw the modification and the proposal (:246-328, :394-413), sort the
//                          nuclei (:315-328), check priors and validity (:330-420), convert the
//                          Voronoi nuclei to layers (src/Models.py:26-52) and write the layered model
//                          straight into the layer-major [Lmax][C] arrays bh_evaluate_batch reads;
//   chain_accept_kernel    acceptance probability incl. the birth/death terms (:452-487), accept or
//                          keep, move counters, proposal-width adaptation every 1000 iterations
//                          (:425-450, :584-587).
// The forward models + likelihood between the two are bh_evaluate_batch on device pointers.
//
// Speculative windows (bh_chain_propose_window / bh_chain_accept_window): a chain's iterations are sequential, but
// every draw is a pure function of (chain, iteration, purpose), so the proposals of the NEXT `depth` iterations can be
// written down for both outcomes of every accept/reject decision before any of them is evaluated: a binary tree with
// 2^depth - 1 nodes per chain (heap order: node 0 = this iteration's proposal, node 2j+1 / 2j+2 = the next
// iteration's proposal after node j was rejected / accepted).  ONE bh_evaluate_batch over all nodes of all chains
// then costs what a batch of that size costs (for a few chains: the same ~1.5 ms latency floor as one proposal per
// chain), and the accept kernel walks the realised path through the tree: `depth` iterations per evaluation launch,
// bit for bit the sequential walk (same arithmetic on the same operands in the same order; the evaluation of a
// model does not depend on the batch it is in).  Windows end at the iterations where something outside the
// chain's own state changes: the proposal-width adaptation (every 1000th iteration must be the LAST of its window)
// and, on the host side, snapshots and temperature exchanges.
// Random numbers: Philox4x32-10, counter = (global chain index, iteration, purpose), key = seed -- every draw is
// a pure function of its coordinates, so results do not depend on scheduling.  (The reference uses
// one Mersenne-Twister stream per chain; trajectories therefore agree statistically, not draw by
// draw.  The draw-by-draw replay of the reference is bayhunter_amd/chains.py.)  For testing, the
// draws can be injected from a buffer instead.
#include "../../include/bh_engine.h"
#include "bh_device.h"

namespace {

struct Philox {
    uint32_t k0, k1;
    __device__ void round(uint32_t c[4])
    {
        const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
        const uint64_t p0 = (uint64_t)M0 * c[0], p1 = (uint64_t)M1 * c[2];
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    __device__ void gen(uint64_t seed, uint32_t a, uint32_t b, uint32_t cc, uint32_t out[4])
    {
        k0 = (uint32_t)seed;
        k1 = (uint32_t)(seed >> 32);
        uint32_t c[4] = {a, b, cc, 0u};
        for (int i = 0; i < 10; ++i) round(c);
        out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
    }
};

__device__ __forceinline__ double u01(uint32_t hi, uint32_t lo) // [0, 1), 53 bits
{
    return (double)((((uint64_t)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
}

// the draws one iteration of one chain may need
struct Draws {
    double u_move, u_index, u_z, u_accept, u_noise, normal;
};

// k = position of the iteration inside a speculative window (injected draws: [depth][6][C])
__device__ Draws get_draws(const bh_chain_config &cfg, const bh_chain_state &S, int c, int C, int iiter, int k)
{
    Draws d;
    if (S.inject != nullptr) { // test mode
        const double *in = S.inject + (size_t)k * 6 * C;
        d.u_move = in[0 * (size_t)C + c]; d.u_index = in[1 * (size_t)C + c];
        d.u_z = in[2 * (size_t)C + c]; d.u_accept = in[3 * (size_t)C + c];
        d.u_noise = in[4 * (size_t)C + c]; d.normal = in[5 * (size_t)C + c];
        return d;
    }
    Philox ph;
    // every draw gets its own 64 bits of Philox output: draws of one iteration must be independent of
    // each other (a normal deviate correlated with the choice of the move makes the walk drift)
    // the counter carries the chain's GLOBAL index (cfg.chain_offset + c): a job sharded over ranks with one
    // job-wide seed draws, for every chain, the numbers the unsharded job would draw
    uint32_t r[4], q[4], t[4], n[4];
    const uint32_t gc = (uint32_t)(cfg.chain_offset + (int64_t)c);
    ph.gen(cfg.seed, gc, (uint32_t)iiter, 0u, r);
    ph.gen(cfg.seed, gc, (uint32_t)iiter, 1u, q);
    ph.gen(cfg.seed, gc, (uint32_t)iiter, 2u, t);
    ph.gen(cfg.seed, gc, (uint32_t)iiter, 3u, n);
    d.u_move = u01(r[0], r[1]); d.u_index = u01(r[2], r[3]);
    d.u_z = u01(q[0], q[1]); d.u_accept = u01(q[2], q[3]);
    d.u_noise = u01(t[0], t[1]);
    const double a = 1.0 - u01(n[0], n[1]); // (0, 1]
    const double bq = u01(n[2], n[3]);
    d.normal = sqrt(-2.0 * log(a)) * cospi(2.0 * bq); // Box-Muller
    return d;
}

// the accept step's draw alone (the same bits as get_draws(...).u_accept, without the other five)
__device__ double get_accept_draw(const bh_chain_config &cfg, const bh_chain_state &S, int c, int C, int iiter, int k)
{
    if (S.inject != nullptr) return S.inject[(size_t)k * 6 * C + 3 * (size_t)C + c];
    Philox ph;
    uint32_t q[4];
    ph.gen(cfg.seed, (uint32_t)(cfg.chain_offset + (int64_t)c), (uint32_t)iiter, 1u, q);
    return u01(q[2], q[3]);
}

enum { MV_VS = 0, MV_Z = 1, MV_BIRTH = 2, MV_DEATH = 3, MV_NOISE = 4, MV_VPVS = 5 };
__device__ __forceinline__ int par_index(int mv) { return mv <= 1 ? mv : (mv <= 3 ? 2 : mv - 1); } // PAR_MAP

// A chain's model parameters: n, vp/vs and pointers to the arrays (vs, z: nuclei; noise; h: layer thicknesses), which live
// in the lane's private memory (lane = chain kernel) or in the node's LDS record (window kernel) -- the arithmetic on
// them is the same text either way.
struct Params {
    int n;
    double vpvs;
    double *vs, *z, *noise, *h;
};
// LDS record of one tree node (doubles): [0] n  [1] valid  [2] vp/vs  [3 .. 3+2nt) noise  then vs[ML+1], z[ML+1], h[ML+1]
__host__ __device__ inline int node_rec_doubles(int nt, int ML) { return 3 + 2 * nt + 3 * (ML + 1); }

// The state a tree node's proposal starts from: the proposal of the nearest ancestor that is entered through
// its "accepted" edge (and was valid), else the chain's current state.  node < 0: the chain's current state.
// lds_from: the LDS record of `from_node` when the window kernel keeps the tree there (same values as the global copy)
__device__ void load_base(const bh_chain_state &S, int C, size_t ldp, int nt, int ML, int c, int from_node, Params &P,
                          const double *lds_from = nullptr)
{
    if (from_node >= 0 && lds_from != nullptr) {
        // (source and destination records never overlap: told to the compiler so that the LDS reads of several
        // elements are in flight together -- one wavefront per chain, every dependent LDS round trip is ~100 cycles)
        const double *__restrict__ src = lds_from;
        double *__restrict__ pn = P.noise, *__restrict__ pv = P.vs, *__restrict__ pz = P.z;
        P.n = (int)src[0];
        P.vpvs = src[2];
        for (int i = 0; i < 2 * nt; ++i) pn[i] = src[3 + i];
        const double *__restrict__ fv = src + 3 + 2 * nt, *__restrict__ fz = fv + (ML + 1);
        const int n = P.n;
#pragma unroll 4
        for (int i = 0; i < n; ++i) {
            pv[i] = fv[i];
            pz[i] = fz[i];
        }
        return;
    }
    if (from_node < 0) {
        P.n = S.n[c];
        for (int i = 0; i < P.n; ++i) {
            P.vs[i] = S.vs[(size_t)i * C + c];
            P.z[i] = S.z[(size_t)i * C + c];
        }
        P.vpvs = S.vpvs[c];
        for (int i = 0; i < 2 * nt; ++i) P.noise[i] = S.noise[(size_t)i * C + c];
    } else {
        const size_t col = (size_t)from_node * C + c;
        P.n = S.pn[col];
        for (int i = 0; i < P.n; ++i) {
            P.vs[i] = S.pvs[(size_t)i * ldp + col];
            P.z[i] = S.pz[(size_t)i * ldp + col];
        }
        P.vpvs = S.pvpvs[col];
        for (int i = 0; i < 2 * nt; ++i) P.noise[i] = S.pnoise[col * 2 * nt + i];
    }
}

__device__ __forceinline__ void layer_thicknesses(const Params &P, double *h_)
{
    const double *__restrict__ z = P.z;
    double *__restrict__ h = h_;
    const int n = P.n;
    double prev = 0.0, zi = n >= 1 ? z[0] : 0.0;
#pragma unroll 4
    for (int i = 0; i + 1 < n; ++i) {
        const double zn = z[i + 1];
        const double zd = (zi + zn) / 2.;
        h[i] = zd - prev;
        prev = zd;
        zi = zn;
    }
    if (n >= 1) h[n - 1] = 0.0;
}

// index of the nucleus nearest in depth to zb (first one on ties, like the reference's argmin); reads only
__device__ __forceinline__ int nearest_nucleus(const double *__restrict__ z, int n, double zb)
{
    int near = 0;
    double best = fabs(z[0] - zb);
#pragma unroll 4
    for (int i = 1; i < n; ++i) {
        const double di = fabs(z[i] - zb);
        if (di < best) {
            best = di;
            near = i;
        }
    }
    return near;
}

// One proposal (SingleChain.py:246-420, :511-556; Models.py:26-52): from the state of `from_node`, with the draws of
// iteration `iiter`, into column node*C + c of the proposal arrays (leading dimension ldp).
// `P` brings the storage; `lds_from` as in load_base.
__device__ __forceinline__ void propose_node(const bh_chain_config &cfg, const bh_chain_state &S, int C, size_t ldp, int c, int iiter,
                                             int from_node, int node, Params &P, const double *lds_from, bool *valid_out, const Draws &d)
{
    const int ML = cfg.maxlayers, nt = cfg.nt;
    // ---- which modification (SingleChain.py:512-517, :596-599) ---------------------------------
    int nnoise = 0;
    for (int i = 0; i < 2 * nt; ++i) nnoise += (cfg.noise_lo[i] != cfg.noise_hi[i]);
    const bool vpvs_free = cfg.vpvsmin != cfg.vpvsmax;
    const bool early = (double)iiter < (-(double)cfg.iter_burnin + (double)cfg.iterations * 0.01);
    int moves[6], nm = 0;
    moves[nm++] = MV_VS;
    moves[nm++] = MV_Z;
    if (!early) {
        moves[nm++] = MV_BIRTH;
        moves[nm++] = MV_DEATH;
    }
    if (nnoise > 0) moves[nm++] = MV_NOISE;
    if (vpvs_free) moves[nm++] = MV_VPVS;
    int mi = (int)(d.u_move * nm);
    if (mi >= nm) mi = nm - 1;
    const int mv = moves[mi];

    // ---- proposal -----------------------------------------------------------------------------------
    load_base(S, C, ldp, nt, ML, c, from_node, P, lds_from);
    double *vs = P.vs, *z = P.z, *noise = P.noise;
    int n = P.n;
    double vpvs = P.vpvs;
    double dvs2 = 0.0;
    bool valid = true;
    if (mv == MV_VS) {
        int ind = (int)(d.u_index * n);
        if (ind >= n) ind = n - 1;
        vs[ind] = vs[ind] + d.normal * S.propdist[0 * (size_t)C + c];
    } else if (mv == MV_Z) {
        int ind = (int)(d.u_index * n);
        if (ind >= n) ind = n - 1;
        z[ind] = z[ind] + d.normal * S.propdist[1 * (size_t)C + c];
    } else if (mv == MV_BIRTH) {
        const double zb = cfg.zmin + d.u_z * (cfg.zmax - cfg.zmin);
        const int near = nearest_nucleus(z, n, zb);
        const double vb = vs[near] + d.normal * S.propdist[2 * (size_t)C + c];
        dvs2 = (vb - vs[near]) * (vb - vs[near]);
        if (n >= ML) valid = false; // would exceed the layer prior anyway
        else {
            vs[n] = vb;
            z[n] = zb;
            n += 1;
        }
    } else if (mv == MV_DEATH) {
        int ind = (int)(d.u_index * n);
        if (ind >= n) ind = n - 1;
        const double zb = z[ind], vb = vs[ind];
        for (int i = ind; i + 1 < n; ++i) {
            vs[i] = vs[i + 1];
            z[i] = z[i + 1];
        }
        n -= 1;
        if (n < 1) valid = false;
        else {
            const int near = nearest_nucleus(z, n, zb);
            dvs2 = (vs[near] - vb) * (vs[near] - vb);
        }
    } else if (mv == MV_NOISE) {
        int pick = (int)(d.u_noise * nnoise);
        if (pick >= nnoise) pick = nnoise - 1;
        int idx = 0;
        for (int i = 0, seen = 0; i < 2 * nt; ++i)
            if (cfg.noise_lo[i] != cfg.noise_hi[i]) {
                if (seen == pick) idx = i;
                ++seen;
            }
        noise[idx] = noise[idx] + d.normal * S.propdist[3 * (size_t)C + c];
        for (int i = 0; i < 2 * nt; ++i)
            if (cfg.noise_lo[i] != cfg.noise_hi[i] && (noise[i] < cfg.noise_lo[i] || noise[i] > cfg.noise_hi[i])) valid = false;
    } else {
        vpvs = vpvs + d.normal * S.propdist[4 * (size_t)C + c];
        if (vpvs < cfg.vpvsmin || vpvs > cfg.vpvsmax) valid = false;
    }
    // nuclei sorted by depth (:315-328); insertion sort is stable like the reference's argsort use
    double ztop = n > 0 ? z[0] : 0.0; // largest depth so far = z[i - 1] once the first i nuclei are in order
    for (int i = 1; i < n; ++i) {
        const double zi = z[i];
        if (!(ztop > zi)) { // already in place (the usual case: one nucleus moved at most)
            ztop = zi;
            continue;
        }
        const double vi = vs[i];
        int j = i - 1;
        while (j >= 0 && z[j] > zi) {
            z[j + 1] = z[j];
            vs[j + 1] = vs[j];
            --j;
        }
        z[j + 1] = zi;
        vs[j + 1] = vi;
    }
    P.n = n;
    P.vpvs = vpvs;
    // ---- nuclei -> layers (Models.py:39-52), validity with the CURRENT vp/vs (:330-392) ------------
    double *h = P.h;
    layer_thicknesses(P, h);
    if (valid && (mv <= MV_DEATH)) {
        const int layermodel = n - 1;
        if (!(layermodel >= cfg.layermin && layermodel <= cfg.layermax)) valid = false;
        // (every layer is looked at, no early exit: the tests have no side effects and the reads can be in flight together)
        const double *__restrict__ rh = h, *__restrict__ rv = vs;
        double zc = 0.0, vi = n > 0 ? rv[0] : 0.0;
        bool ok = true;
#pragma unroll 4
        for (int i = 0; i < n; ++i) {
            const double hi = rh[i];
            const double vn = (i + 1 < n) ? rv[i + 1] : 0.0;
            if (i < n - 1 && hi < cfg.thickmin) ok = false;
            if (vi < cfg.vsmin || vi > cfg.vsmax) ok = false;
            zc += hi;
            if (zc < cfg.zmin || zc > cfg.zmax) ok = false;
            if (i + 1 < n) {
                if (cfg.lvz >= 0.0 && !(vn - vi * (1 - cfg.lvz) > 0)) ok = false;
                if (cfg.hvz >= 0.0 && !(vi * (1 + cfg.hvz) - vn > 0)) ok = false;
            }
            vi = vn;
        }
        valid = valid && ok;
    }
    // ---- write the proposal and the layered model (invalid: re-evaluate the model it started from) ---------
    const size_t col = (size_t)node * C + c;
    S.move[col] = mv;
    S.valid[col] = valid ? 1 : 0;
    S.dvs2[col] = dvs2;
    if (!valid) { // keep the evaluate batch well-formed: it sees the unchanged model, result ignored
        load_base(S, C, ldp, nt, ML, c, from_node, P, lds_from);
        layer_thicknesses(P, h);
        n = P.n;
        vpvs = P.vpvs;
    }
    P.n = n;
    P.vpvs = vpvs;
    *valid_out = valid;
    S.pn[col] = n;
    S.pvpvs[col] = vpvs;
    for (int i = 0; i < 2 * nt; ++i) S.pnoise[col * 2 * nt + i] = noise[i]; // [columns][2nt]: evaluate's layout
    // vp: crustal vp/vs down to the first layer with vs >= mantle[0], mantle[1] below (Models.py:26-37)
    {
        // (one pass over the node's arrays; they are only read here and the global arrays only written)
        const double *__restrict__ rv = vs, *__restrict__ rz = z, *__restrict__ rh = h;
        bool deep = false;
#pragma unroll 4
        for (int i = 0; i < n; ++i) {
            const double vi = rv[i], zi = rz[i], hi = rh[i];
            if (cfg.mantle_vs > 0.0 && vi >= cfg.mantle_vs) deep = true;
            S.pvs[(size_t)i * ldp + col] = vi;
            S.pz[(size_t)i * ldp + col] = zi;
            S.lay_h[(size_t)i * ldp + col] = hi;
            S.lay_vs[(size_t)i * ldp + col] = vi;
            const double vpi = vi * (deep ? cfg.mantle_vpvs : vpvs);
            S.lay_vp[(size_t)i * ldp + col] = vpi;
            if (S.lay_rho != nullptr) S.lay_rho[(size_t)i * ldp + col] = vpi * 0.32 + 0.77; // as rho_from_vp_kernel (Targets.py:319)
        }
    }
    S.lay_n[col] = n;
}

// lane = chain, one proposal per chain (depth 1; ldp = C)
__global__ void chain_propose_kernel(bh_chain_config cfg, bh_chain_state S, int C, int iiter)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double vs[BH_CHAIN_MAXLAYERS + 1], z[BH_CHAIN_MAXLAYERS + 1], h[BH_CHAIN_MAXLAYERS + 1], noise[2 * BH_MAX_TARGETS];
    Params P;
    P.vs = vs; P.z = z; P.h = h; P.noise = noise;
    bool valid;
    propose_node(cfg, S, C, (size_t)C, c, iiter, -1, 0, P, nullptr, &valid, get_draws(cfg, S, c, C, iiter, 0));
}

// Speculative window: T = 2^(depth-1) lanes per chain (64 / T chains per single-wavefront workgroup); level k of the
// tree is proposed by the first 2^k lanes of a chain, the levels one after the other.  The tree lives in LDS while it
// is built (one record per node, node_rec_doubles): a node starts from its ancestor's record and works in its own --
// no private arrays (dynamic indexing would put them in scratch memory) and no wait for global stores between the
// levels (round 3, first form: state through global memory and __syncthreads -- 128 us per launch at depth 7, 7 % of a
// c4 launch).  A workgroup is one wavefront: its LDS operations execute in order, the barrier only pins the compiler.
__global__ __launch_bounds__(64) void chain_propose_window_kernel(bh_chain_config cfg, bh_chain_state S, int C, size_t ldp,
                                                                   int iiter, int depth)
{
    extern __shared__ __align__(16) double tree[];
    const int T = 1 << (depth - 1);
    const int per_wg = 64 / T;
    const int N = (1 << depth) - 1;
    const int slot = (int)threadIdx.x / T;
    const int c = blockIdx.x * per_wg + slot;
    const int p = (int)threadIdx.x % T;
    const int nt = cfg.nt, ML = cfg.maxlayers;
    const int RS = node_rec_doubles(nt, ML) | 1; // odd stride: neighbouring nodes start in different banks
    double *mine = tree + (size_t)slot * N * RS;  // this chain's records
    // The draws of iteration iiter + k are the same for every node of level k: lane k of the chain computes them (four Philox
    // blocks, a logarithm, a square root and a cosine: 3 us when every level did it for itself, 21 of the kernel's 70 us at
    // depth 7), the level's lanes fetch them.  (2^(depth-1) >= depth: the chain has a lane for every level.)
    Draws mydraws = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (c < C && p < depth) mydraws = get_draws(cfg, S, c, C, iiter + p, p);
    for (int k = 0; k < depth; ++k) {
        const int srcl = slot * T + k;
        Draws d;
        d.u_move = __shfl(mydraws.u_move, srcl); d.u_index = __shfl(mydraws.u_index, srcl);
        d.u_z = __shfl(mydraws.u_z, srcl); d.u_accept = __shfl(mydraws.u_accept, srcl);
        d.u_noise = __shfl(mydraws.u_noise, srcl); d.normal = __shfl(mydraws.normal, srcl);
        if (c < C && p < (1 << k)) {
            const int node = (1 << k) - 1 + p;
            // nearest ancestor entered through its "accepted" edge whose proposal was valid
            int from = -1;
            for (int j = node; j > 0; j = (j - 1) >> 1) {
                const int parent = (j - 1) >> 1;
                if ((j & 1) == 0 && mine[(size_t)parent * RS + 1] != 0.0) {
                    from = parent;
                    break;
                }
            }
            double *rec = mine + (size_t)node * RS;
            Params P;
            P.noise = rec + 3;
            P.vs = rec + 3 + 2 * nt;
            P.z = P.vs + (ML + 1);
            P.h = P.z + (ML + 1);
            bool valid;
            propose_node(cfg, S, C, ldp, c, iiter + k, from, node, P, from >= 0 ? mine + (size_t)from * RS : nullptr, &valid, d);
            rec[0] = (double)P.n;
            rec[1] = valid ? 1.0 : 0.0;
            rec[2] = P.vpvs;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// lane = chain: walk the realised path through the window's tree (depth 1: the plain accept step)
__global__ void chain_accept_kernel(bh_chain_config cfg, bh_chain_state S, int C, size_t ldp, int iiter, int depth,
                                    const double *logL, const double *misfits)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int nt = cfg.nt;
    int node = 0, last = -1;
    double cur = S.like[c];
    const double beta = S.beta ? S.beta[c] : 1.0;
    for (int k = 0; k < depth; ++k) {
        const size_t col = (size_t)node * C + c;
        bool accepted = false;
        if (S.valid[col]) {
            const double u_accept = get_accept_draw(cfg, S, c, C, iiter + k, k);
            const int mv = S.move[col];
            const int pi = par_index(mv);
            S.proposed[pi * (size_t)C + c] += 1.0;
            const double like = logL[col];
            const double dl = (S.beta ? beta * (like - cur) : like - cur);
            double alpha;
            if (mv == MV_BIRTH || mv == MV_DEATH) { // Bodin et al. (2012), SingleChain.py:468-485
                const double theta = S.propdist[2 * (size_t)C + c];
                const double dv = cfg.vsmax - cfg.vsmin;
                const double Bt = S.dvs2[col] / (2. * (theta * theta));
                if (mv == MV_BIRTH) alpha = log((theta * sqrt(2 * M_PI)) / dv) + Bt + dl;
                else alpha = log(dv / (theta * sqrt(2 * M_PI))) - Bt + dl;
            } else {
                alpha = dl;
            }
            if (log(u_accept) < alpha) {
                accepted = true;
                last = node;
                cur = like;
                S.accepted[pi * (size_t)C + c] += 1.0;
                S.naccepted[c] += 1;
            }
            // proposal-width adaptation (SingleChain.py:425-450): only reached with a valid proposal (:584).
            // The host ends a window at such an iteration (the widths enter the next proposals).
            if ((iiter + k) % 1000 == 0) {
                bool all = true;
                for (int i = 0; i < 5; ++i) all = all && (S.proposed[i * (size_t)C + c] != 0.0);
                if (all)
                    for (int i = 0; i < 5; ++i) {
                        const double rate = S.accepted[i * (size_t)C + c] / S.proposed[i * (size_t)C + c] * 100;
                        double pd = S.propdist[i * (size_t)C + c];
                        if (rate < cfg.acc_lo) {
                            pd = pd * 0.95;
                            if (pd < 0.001) pd = 0.001;
                        } else if (rate > cfg.acc_hi) {
                            pd = pd * 1.05;
                        }
                        S.propdist[i * (size_t)C + c] = pd;
                    }
            }
        }
        node = 2 * node + (accepted ? 2 : 1);
    }
    if (last >= 0) { // the last accepted proposal of the window becomes the chain's state
        const size_t col = (size_t)last * C + c;
        const int n = S.pn[col];
        S.n[c] = n;
        for (int i = 0; i < n; ++i) {
            S.vs[(size_t)i * C + c] = S.pvs[(size_t)i * ldp + col];
            S.z[(size_t)i * C + c] = S.pz[(size_t)i * ldp + col];
        }
        // rows beyond n are kept at zero: the state arrays are then a function of the trajectory alone, not of how
        // it was cut into windows (only the last accepted proposal of a window is written)
        for (int i = n; i < cfg.maxlayers; ++i) {
            S.vs[(size_t)i * C + c] = 0.0;
            S.z[(size_t)i * C + c] = 0.0;
        }
        S.vpvs[c] = S.pvpvs[col];
        for (int i = 0; i < 2 * nt; ++i) S.noise[(size_t)i * C + c] = S.pnoise[col * 2 * nt + i];
        S.like[c] = cur;
        for (int i = 0; i <= nt; ++i) S.misfits[(size_t)i * C + c] = misfits[col * (nt + 1) + i];
    }
}

// A wavefront per chain: walk the realised path through the window's tree (depth > 1).  Same decisions and the same bits as
// `depth` rounds of chain_accept_kernel; what differs is where the operands wait: the tree's per-node values (valid, move,
// logL, birth/death term) are fetched by all lanes at once -- node j in lane j mod 64 -- and the window's accept draws by
// the first `depth` lanes, so that the walk itself is shuffles instead of three dependent global loads and a Philox
// evaluation per level (24 -> 9 us at depth 7); the counters of the five proposal types sit in lanes 0..4 and are written
// back once; the committed model is copied one layer per lane.
__global__ __launch_bounds__(64) void chain_accept_window_kernel(bh_chain_config cfg, bh_chain_state S, int C, size_t ldp, int iiter, int depth,
                                                                  const double *logL, const double *misfits)
{
    const int c = (int)blockIdx.x, lane = (int)threadIdx.x;
    if (c >= C) return;
    const int nt = cfg.nt, N = (1 << depth) - 1;
    int valid0 = 0, valid1 = 0, mv0 = 0, mv1 = 0;
    double like0 = 0.0, like1 = 0.0, dv0 = 0.0, dv1 = 0.0;
    if (lane < N) {
        const size_t col = (size_t)lane * C + c;
        valid0 = S.valid[col]; mv0 = S.move[col]; like0 = logL[col]; dv0 = S.dvs2[col];
    }
    if (lane + 64 < N) {
        const size_t col = (size_t)(lane + 64) * C + c;
        valid1 = S.valid[col]; mv1 = S.move[col]; like1 = logL[col]; dv1 = S.dvs2[col];
    }
    const double udraw = (lane < depth) ? get_accept_draw(cfg, S, c, C, iiter + lane, lane) : 0.5;
    double cur = S.like[c];
    const double beta = S.beta ? S.beta[c] : 1.0;
    const double theta = S.propdist[2 * (size_t)C + c]; // (changes only in the adaptation, after the window's last decision)
    double prop = lane < 5 ? S.proposed[lane * (size_t)C + c] : 0.0, acc = lane < 5 ? S.accepted[lane * (size_t)C + c] : 0.0;
    long long nacc = (long long)S.naccepted[c];
    int node = 0, last = -1;
    for (int k = 0; k < depth; ++k) {
        const int src = node & 63;
        const bool hi = node >= 64;
        const int va = __shfl(valid0, src), vb = __shfl(valid1, src), ma = __shfl(mv0, src), mb = __shfl(mv1, src);
        const double la = __shfl(like0, src), lb = __shfl(like1, src), da = __shfl(dv0, src), db = __shfl(dv1, src);
        const double u_accept = __shfl(udraw, k);
        bool accepted = false;
        if (hi ? vb : va) {
            const int mv = hi ? mb : ma;
            const int pi = par_index(mv);
            if (lane == pi) prop += 1.0;
            const double like = hi ? lb : la;
            const double dl = (S.beta ? beta * (like - cur) : like - cur);
            double alpha;
            if (mv == MV_BIRTH || mv == MV_DEATH) { // Bodin et al. (2012), SingleChain.py:468-485
                const double dv = cfg.vsmax - cfg.vsmin;
                const double Bt = (hi ? db : da) / (2. * (theta * theta));
                if (mv == MV_BIRTH) alpha = log((theta * sqrt(2 * M_PI)) / dv) + Bt + dl;
                else alpha = log(dv / (theta * sqrt(2 * M_PI))) - Bt + dl;
            } else {
                alpha = dl;
            }
            if (log(u_accept) < alpha) {
                accepted = true;
                last = node;
                cur = like;
                if (lane == pi) acc += 1.0;
                nacc += 1;
            }
            // proposal-width adaptation (SingleChain.py:425-450): the last decision of a window only (the host sees to it)
            if ((iiter + k) % 1000 == 0) {
                const bool all = (__ballot(lane < 5 && prop != 0.0) & 0x1full) == 0x1full;
                if (all && lane < 5) {
                    const double rate = acc / prop * 100;
                    double pd = S.propdist[lane * (size_t)C + c];
                    if (rate < cfg.acc_lo) {
                        pd = pd * 0.95;
                        if (pd < 0.001) pd = 0.001;
                    } else if (rate > cfg.acc_hi) {
                        pd = pd * 1.05;
                    }
                    S.propdist[lane * (size_t)C + c] = pd;
                }
            }
        }
        node = 2 * node + (accepted ? 2 : 1);
    }
    if (lane < 5) {
        S.proposed[lane * (size_t)C + c] = prop;
        S.accepted[lane * (size_t)C + c] = acc;
    }
    if (lane == 0) S.naccepted[c] = nacc;
    if (last >= 0) { // the last accepted proposal of the window becomes the chain's state
        const size_t col = (size_t)last * C + c;
        const int n = S.pn[col];
        for (int i = lane; i < cfg.maxlayers; i += 64) { // (rows beyond n are kept at zero, see chain_accept_kernel)
            S.vs[(size_t)i * C + c] = i < n ? S.pvs[(size_t)i * ldp + col] : 0.0;
            S.z[(size_t)i * C + c] = i < n ? S.pz[(size_t)i * ldp + col] : 0.0;
        }
        for (int i = lane; i < 2 * nt; i += 64) S.noise[(size_t)i * C + c] = S.pnoise[col * 2 * nt + i];
        for (int i = lane; i <= nt; i += 64) S.misfits[(size_t)i * C + c] = misfits[col * (nt + 1) + i];
        if (lane == 0) {
            S.n[c] = n;
            S.vpvs[c] = S.pvpvs[col];
            S.like[c] = cur;
        }
    }
}

} // namespace

extern "C" {

int bh_chain_propose_window(void *stream, const bh_chain_config *cfg, const bh_chain_state *state, int C, int iiter,
                            int depth, ptrdiff_t ld)
{
    if (!cfg || !state || C < 0 || cfg->maxlayers > BH_CHAIN_MAXLAYERS || cfg->nt > BH_MAX_TARGETS) return BH_EINVAL;
    if (depth < 1 || depth > BH_CHAIN_MAXDEPTH || ld < (ptrdiff_t)C * ((1 << depth) - 1)) return BH_EINVAL;
    for (int k = 0; k + 1 < depth; ++k)
        if ((iiter + k) % 1000 == 0) return BH_EINVAL; // an adaptation iteration must be the last of its window
    if (C == 0) return BH_OK;
    if (depth == 1 && ld == C) {
        hipLaunchKernelGGL(chain_propose_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, *cfg, *state, C, iiter);
    } else {
        const int per_wg = 64 >> (depth - 1);
        const size_t lds = (size_t)per_wg * ((1 << depth) - 1) * (node_rec_doubles(cfg->nt, cfg->maxlayers) | 1) * sizeof(double);
        if (lds > 160 * 1024) return BH_EUNSUPPORTED;
        if (lds > 64 * 1024) {
            static std::atomic<unsigned long long> big{0};
            const void *k[1] = {reinterpret_cast<const void *>(chain_propose_window_kernel)};
            if (!bh_allow_big_lds(&big, k, 1, 160 * 1024)) return BH_EHIP;
        }
        hipLaunchKernelGGL(chain_propose_window_kernel, dim3((C + per_wg - 1) / per_wg), dim3(64), lds, (hipStream_t)stream, *cfg,
                           *state, C, (size_t)ld, iiter, depth);
    }
    return hipGetLastError() == hipSuccess ? BH_OK : BH_EHIP;
}

int bh_chain_accept_window(void *stream, const bh_chain_config *cfg, const bh_chain_state *state, int C, int iiter, int depth,
                           ptrdiff_t ld, const double *logL, const double *misfits)
{
    if (!cfg || !state || C < 0 || !logL || !misfits) return BH_EINVAL;
    if (depth < 1 || depth > BH_CHAIN_MAXDEPTH || ld < (ptrdiff_t)C * ((1 << depth) - 1)) return BH_EINVAL;
    for (int k = 0; k + 1 < depth; ++k)
        if ((iiter + k) % 1000 == 0) return BH_EINVAL;
    if (C == 0) return BH_OK;
    if (depth > 1) // a wavefront per chain
        hipLaunchKernelGGL(chain_accept_window_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream, *cfg, *state, C, (size_t)ld, iiter, depth,
                           logL, misfits);
    else
        hipLaunchKernelGGL(chain_accept_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, *cfg, *state, C, (size_t)ld,
                           iiter, depth, logL, misfits);
    return hipGetLastError() == hipSuccess ? BH_OK : BH_EHIP;
}

int bh_chain_propose(void *stream, const bh_chain_config *cfg, const bh_chain_state *state, int C, int iiter)
{
    return bh_chain_propose_window(stream, cfg, state, C, iiter, 1, C);
}

int bh_chain_accept(void *stream, const bh_chain_config *cfg, const bh_chain_state *state, int C, int iiter,
                    const double *logL, const double *misfits)
{
    return bh_chain_accept_window(stream, cfg, state, C, iiter, 1, C, logL, misfits);
}

} // extern "C"
