// small_samplers.hpp -- mcmc::hmc / mcmc::mala / mcmc::rwmh for small-dimensional targets, one lane per chain.
//
// Companion of rmhmc_small.hpp: the same one-chain-per-lane layout (compile-time D <= 4, everything in registers) for the other
// samplers, so that the reference's own example programs (examples/eigen/{hmc,mala}_normal.cpp: the d = 2 normal model)
// run on the device.  With D x D matrices per lane the full generality of the reference costs nothing here: any dense
// precond_mat / cov_mat together with any box constraints, including bounded MALA with a dense preconditioner, which the
// MFMA path refuses (INV of eps^2 J M per draw).  Reference loops: src/hmc.cpp:155-205, src/mala.cpp:149-190 +
// include/mcmc/mala.ipp:30-70, src/rwmh.cpp:123-151.  Arithmetic = oracle (orc_hmc / orc_mala / orc_rwmh with reduce_width 1):
// sequential fma chains, dense products including their zero entries.
#pragma once

#include "rmhmc_small.hpp"

namespace mi {

template <int D>
__device__ __forceinline__ void sm_matmul(const double (&A)[D][D], const double (&B)[D][D], double (&Cm)[D][D])
{
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) acc = dfma(A[i][k], B[k][j], acc);
            Cm[i][j] = acc;
        }
}

// BMO_MATOPS_DOT_PROD in the order the target's policy states (oracle: orc_dot): W = 1 one sequential fma chain; W = 4 four
// strided chains q_j over i = j (mod 4), combined (q0 + q2) + (q1 + q3) -- the order of the MFMA-layout kernels, so that a
// small-d run of this engine and of those kernels agree bit for bit (LogisticSmallModel)
template <int D, int W = 1>
__device__ __forceinline__ double sm_dot(const double (&x)[D], const double (&y)[D])
{
    static_assert(W == 1 || W == 4, "reduction orders the oracle states");
    if constexpr (W == 1) {
        double q = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) q = dfma(x[i], y[i], q);
        return q;
    } else {
        double q[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < D; ++i) q[i & 3] = dfma(x[i], y[i], q[i & 3]);
        return (q[0] + q[2]) + (q[1] + q[3]);
    }
}
// reduction order of a target policy: Target::W when it declares one, else 1
template <class T, class = void> struct small_reduce_width { static constexpr int value = 1; };
template <class T> struct small_reduce_width<T, decltype((void)T::W)> { static constexpr int value = T::W; };

// what every small sampler shares: the chain's lane, the box maps, the precond matrix and the draw bookkeeping
template <class Target>
struct SmallChain {
    static constexpr int D = Target::D;
    static constexpr int W = small_reduce_width<Target>::value;
    static_assert(D >= 1 && D <= SMALL_MAX_D, "Target::D out of range");
    const SmallParams& prm;
    const Target& tgt;
    const bool bounded;
    uint64_t c, chain;

    __device__ __forceinline__ SmallChain(const SmallParams& p, const Target& t, uint64_t c_)
        : prm(p), tgt(t), bounded(p.vals_bound != 0), c(c_), chain(p.chain0 + c_) {}

    __device__ __forceinline__ void inv_tr(const double (&v)[D], double (&o)[D]) const
    {
#pragma unroll
        for (int i = 0; i < D; ++i) o[i] = bounded ? box_inv_transform(v[i], prm.btype[i], prm.lb[i], prm.ub[i]) : v[i];
    }
    // box_log_kernel (hmc.cpp:84-95, mala.cpp:84-95, rwmh.cpp:84-95)
    __device__ __forceinline__ double box_log_kernel(const double (&v)[D]) const
    {
        double vi[D], g[D];
        inv_tr(v, vi);
        const double k = tgt.kernel(vi, g, false);
        if (!bounded) return k;
        double lj = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) if (prm.btype[i] != 1) lj += box_log_jacobian_term(v[i], prm.btype[i], prm.lb[i], prm.ub[i]);
        return k + lj;
    }
    // gradient of the target at inv_transform(v); J = inv_jacobian_adjust(v) as the dense matrix the reference forms
    __device__ __forceinline__ void grad_at(const double (&v)[D], double (&grad)[D]) const
    {
        double vi[D];
        inv_tr(v, vi);
        (void)tgt.kernel(vi, grad, true);
    }
    __device__ __forceinline__ void jacobian(const double (&v)[D], double (&J)[D][D]) const
    {
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) J[i][j] = (i == j) ? box_inv_jacobian(v[i], prm.btype[i], prm.lb[i], prm.ub[i]) : 0.0;
    }
    __device__ __forceinline__ void load_state(double (&v)[D]) const
    {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const double t = prm.theta[(size_t)i * prm.C + c];
            v[i] = bounded ? box_transform(t, prm.btype[i], prm.lb[i], prm.ub[i]) : t;
        }
    }
    __device__ __forceinline__ void store_natural(double* base, const double (&v)[D]) const   // + the epilogue inv_transform
    {
#pragma unroll
        for (int i = 0; i < D; ++i) base[(size_t)i * prm.C] = bounded ? box_inv_transform(v[i], prm.btype[i], prm.lb[i], prm.ub[i]) : v[i];
    }
    __device__ __forceinline__ void normals(uint32_t draw, double (&z)[D]) const
    {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double z0, z1;
            // dimension i = 8 b + 4 h + j takes component h of Philox slot 4 b + j (det_math.hpp)
            rng_normal_pair(prm.seed, chain, draw + prm.draw0, (uint32_t)(4 * (i / 8) + (i & 3)), STREAM_NORMAL, z0, z1);
            z[i] = ((i >> 2) & 1) ? z1 : z0;
        }
    }
    __device__ __forceinline__ void precond(double (&M)[D][D]) const
    {
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) M[i][j] = prm.M[i][j];
    }
};

// ------------------------------------------------------------------ mcmc::hmc (src/hmc.cpp:30-227)
template <class Target>
__global__ __launch_bounds__(256) void hmc_small_kernel(const SmallParams prm, const Target tgt)
{
    constexpr int D = Target::D;
    [[maybe_unused]] constexpr int W = small_reduce_width<Target>::value;
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= prm.C) return;
    const SmallChain<Target> ch(prm, tgt, c);
    const double eps = prm.eps;
    double M[D][D], Minv[D][D], L[D][D];
    ch.precond(M);
    sm_inv<D>(M, Minv);                                              // hmc.cpp:58
    sm_chol<D>(M, L);                                                // :59

    // mntm_update_fn (hmc.cpp:99-128)
    auto mntm_update = [&](const double (&pos)[D], double (&p)[D]) {
        double grad[D];
        ch.grad_at(pos, grad);
        if (ch.bounded) {
            double J[D][D], jg[D];
            ch.jacobian(pos, J);
            sm_gemv<D>(J, grad, jg);
#pragma unroll
            for (int i = 0; i < D; ++i) p[i] = p[i] + (eps * jg[i]) / 2.0;      // :122
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i) p[i] = p[i] + (eps * grad[i]) / 2.0;    // :126
        }
    };
    auto kinetic = [&](const double (&p)[D]) -> double {
        double t[D];
        sm_gemv<D>(Minv, p, t);
        return sm_dot<D, W>(p, t) / 2.0;
    };

    double prev[D], cur[D];
    ch.load_state(prev);                                             // :134-136
    double prev_U = -ch.box_log_kernel(prev);                        // :140
    uint64_t n_acc = 0;
    const uint32_t n_total = prm.n_burnin + prm.n_keep;
    const size_t slab = (size_t)prm.d * prm.C;
    for (uint32_t draw = 0; draw < n_total; ++draw) {                // :155
        double z[D], p[D];
        ch.normals(draw, z);
        sm_gemv<D>(L, z, p);                                         // :158
        const double prev_K = kinetic(p);                            // :160
#pragma unroll
        for (int i = 0; i < D; ++i) cur[i] = prev[i];
        for (uint32_t k = 0; k < prm.n_leap_steps; ++k) {            // :164-176
            mntm_update(cur, p);
            double mp[D];
            sm_gemv<D>(Minv, p, mp);
#pragma unroll
            for (int i = 0; i < D; ++i) cur[i] = cur[i] + eps * mp[i];           // :171
            mntm_update(cur, p);
        }
        double prop_U = -ch.box_log_kernel(cur);                     // :178
        if (!is_finite(prop_U)) prop_U = INF;
        const double prop_K = kinetic(p);                            // :184
        const double x = -(prop_U + prop_K) + (prev_U + prev_K);
        const double comp_val = (x < 0.01) ? x : 0.01;               // :188
        const bool accept = rng_uniform(prm.seed, ch.chain, draw + prm.draw0, 0u) < det_exp(comp_val);
        if (accept) {
            prev_U = prop_U;
#pragma unroll
            for (int i = 0; i < D; ++i) prev[i] = cur[i];
        }
        if (draw >= prm.n_burnin) {
            n_acc += accept ? 1u : 0u;
            if (prm.draws) ch.store_natural(prm.draws + (size_t)(draw - prm.n_burnin) * slab + c, prev);
        }
    }
    ch.store_natural(prm.theta + c, prev);
    if (prm.n_accept) prm.n_accept[c] = n_acc;
    if (prm.n_leap) prm.n_leap[c] = (uint64_t)n_total * prm.n_leap_steps;
}

// ------------------------------------------------------------------ mcmc::mala (src/mala.cpp:30-208, mala.ipp:30-70)
template <class Target>
__global__ __launch_bounds__(256) void mala_small_kernel(const SmallParams prm, const Target tgt)
{
    constexpr int D = Target::D;
    [[maybe_unused]] constexpr int W = small_reduce_width<Target>::value;
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= prm.C) return;
    const SmallChain<Target> ch(prm, tgt, c);
    const double eps = prm.eps;
    const double s2 = eps * eps;
    double M[D][D], L[D][D];
    ch.precond(M);
    sm_chol<D>(M, L);                                                // mala.cpp:58

    // factorisation of Sigma: INV and LOG_DET (through CHOL_LOWER), as dmvnorm does (dmvnorm.hpp:39-41)
    auto factorise = [&](const double (&Sigma)[D][D], double (&Sinv)[D][D]) -> double {
        sm_inv<D>(Sigma, Sinv);
        return sm_log_det<D>(Sigma);
    };
    // mala_mean_fn (mala.cpp:97-125); J filled when bounded
    auto mean_of = [&](const double (&v)[D], double (&J)[D][D], double (&out)[D]) {
        double grad[D], t[D];
        ch.grad_at(v, grad);
        if (ch.bounded) {
            double JM[D][D];
            ch.jacobian(v, J);
            sm_matmul<D>(J, M, JM);
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) JM[i][j] = s2 * JM[i][j];
            sm_gemv<D>(JM, grad, t);
#pragma unroll
            for (int i = 0; i < D; ++i) out[i] = v[i] + t[i] / 2.0;             // :121
        } else {
            sm_gemv<D>(M, grad, t);
#pragma unroll
            for (int i = 0; i < D; ++i) out[i] = v[i] + (s2 * t[i]) / 2.0;      // :123
        }
    };
    auto dmvnorm = [&](const double (&x)[D], const double (&mu)[D], const double (&Sinv)[D][D], double log_det) -> double {
        const double cons_term = -0.5 * (double)D * LOG_2PI;         // dmvnorm.hpp:36
        double xc[D], t[D];
#pragma unroll
        for (int i = 0; i < D; ++i) xc[i] = x[i] - mu[i];
        sm_gemv<D>(Sinv, xc, t);
        return cons_term - 0.5 * (log_det + sm_dot<D, W>(xc, t));       // :41
    };

    double Sinv_h[D][D];                                             // unbounded: Sigma = eps^2 M never changes
    double log_det_h = 0.0;
    if (!ch.bounded) {
        double Sigma[D][D];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) Sigma[i][j] = s2 * M[i][j];
        log_det_h = factorise(Sigma, Sinv_h);
    }

    double prev[D], cur[D];
    ch.load_state(prev);                                             // mala.cpp:132-134
    double prev_LP = ch.box_log_kernel(prev);                        // :138
    uint64_t n_acc = 0;
    const uint32_t n_total = prm.n_burnin + prm.n_keep;
    const size_t slab = (size_t)prm.d * prm.C;
    for (uint32_t draw = 0; draw < n_total; ++draw) {                // :149
        double z[D], t[D], prev_mean[D], Jprev[D][D];
        ch.normals(draw, z);
        mean_of(prev, Jprev, prev_mean);           // also the prev_mean of the adjustment below (same inputs, same bits)
        if (ch.bounded) {                                            // :152-157
            double CJ[D][D], T[D][D];
            sm_chol<D>(Jprev, CJ);
            sm_matmul<D>(CJ, L, T);
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) T[i][j] = eps * T[i][j];
            sm_gemv<D>(T, z, t);
#pragma unroll
            for (int i = 0; i < D; ++i) cur[i] = prev_mean[i] + t[i];
        } else {                                                     // :159
            sm_gemv<D>(L, z, t);
#pragma unroll
            for (int i = 0; i < D; ++i) cur[i] = prev_mean[i] + eps * t[i];
        }
        double prop_LP = ch.box_log_kernel(cur);                     // :162
        if (!is_finite(prop_LP)) prop_LP = -INF;                     // :164-166
        // mala_prop_adjustment (mala.ipp:30-70)
        double prop_mean[D], Jprop[D][D], adj;
        mean_of(cur, Jprop, prop_mean);
        if (ch.bounded) {
            double Sigma[D][D], Sinv[D][D];
            sm_matmul<D>(Jprop, M, Sigma);                           // prop_inv_jacob in BOTH terms (:52-53)
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) Sigma[i][j] = s2 * Sigma[i][j];
            const double log_det = factorise(Sigma, Sinv);
            adj = dmvnorm(prev, prop_mean, Sinv, log_det) - dmvnorm(cur, prev_mean, Sinv, log_det);
        } else {
            adj = dmvnorm(prev, prop_mean, Sinv_h, log_det_h) - dmvnorm(cur, prev_mean, Sinv_h, log_det_h);
        }
        const double x = prop_LP - prev_LP + adj;
        const double comp_val = (x < 0.01) ? x : 0.01;               // mala.cpp:170
        const bool accept = rng_uniform(prm.seed, ch.chain, draw + prm.draw0, 0u) < det_exp(comp_val);
        if (accept) {
            prev_LP = prop_LP;
#pragma unroll
            for (int i = 0; i < D; ++i) prev[i] = cur[i];
        }
        if (draw >= prm.n_burnin) {
            n_acc += accept ? 1u : 0u;
            if (prm.draws) ch.store_natural(prm.draws + (size_t)(draw - prm.n_burnin) * slab + c, prev);
        }
    }
    ch.store_natural(prm.theta + c, prev);
    if (prm.n_accept) prm.n_accept[c] = n_acc;
    if (prm.n_leap) prm.n_leap[c] = 0;
}

// ------------------------------------------------------------------ mcmc::rwmh (src/rwmh.cpp:30-175); eps = par_scale, M = cov_mat
template <class Target>
__global__ __launch_bounds__(256) void rwmh_small_kernel(const SmallParams prm, const Target tgt)
{
    constexpr int D = Target::D;
    [[maybe_unused]] constexpr int W = small_reduce_width<Target>::value;
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= prm.C) return;
    const SmallChain<Target> ch(prm, tgt, c);
    double M[D][D], Lc[D][D];
    ch.precond(M);
    sm_chol<D>(M, Lc);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) Lc[i][j] = prm.eps * Lc[i][j];   // cov_mcmc_chol = par_scale * CHOL_LOWER(cov) (:119)

    double prev[D], cur[D];
    ch.load_state(prev);                                             // :105-107
    double prev_LP = ch.box_log_kernel(prev);                        // :113
    uint64_t n_acc = 0;
    const uint32_t n_total = prm.n_burnin + prm.n_keep;
    const size_t slab = (size_t)prm.d * prm.C;
    for (uint32_t draw = 0; draw < n_total; ++draw) {                // :123
        double z[D], t[D];
        ch.normals(draw, z);
        sm_gemv<D>(Lc, z, t);
#pragma unroll
        for (int i = 0; i < D; ++i) cur[i] = prev[i] + t[i];         // :126
        double prop_LP = ch.box_log_kernel(cur);                     // :128
        if (!is_finite(prop_LP)) prop_LP = -INF;                     // :130-132
        const double x = prop_LP - prev_LP;
        const double comp_val = (x < 0.0) ? x : 0.0;                 // std::min(0.0, x): NaN -> 0 (:136)
        const bool accept = rng_uniform(prm.seed, ch.chain, draw + prm.draw0, 0u) < det_exp(comp_val);
        if (accept) {
            prev_LP = prop_LP;
#pragma unroll
            for (int i = 0; i < D; ++i) prev[i] = cur[i];
        }
        if (draw >= prm.n_burnin) {
            n_acc += accept ? 1u : 0u;
            if (prm.draws) ch.store_natural(prm.draws + (size_t)(draw - prm.n_burnin) * slab + c, prev);
        }
    }
    ch.store_natural(prm.theta + c, prev);
    if (prm.n_accept) prm.n_accept[c] = n_acc;
    if (prm.n_leap) prm.n_leap[c] = 0;
}

// ------------------------------------------------------------------ mcmc::nuts (src/nuts.cpp:30-332, nuts.ipp:30-241)
// The recursive nuts_build_tree runs as an explicit call / return machine over a per-lane frame stack (one frame per tree
// level, in private memory), keeping the reference's argument plumbing: every doubling restarts from (prev_draw, mntm_vec)
// (nuts.cpp:241-256), the second-half calls cross their edge outputs (nuts.ipp:195,207), one runif per completed second
// half in post-order.  Lanes of a wave diverge freely (each chain has its own tree); cost is irrelevant at d = 2.
constexpr int NUTS_SMALL_MAX_DEPTH = 12;

template <int D>
struct NutsRes {            // what one nuts_build_tree call hands back
    double new_draw[D], pos[D], neg[D], mpos[D], mneg[D];
    uint32_t n, s, n_alpha;
    double alpha;
};
template <int D>
struct NutsFrame {
    double start_draw[D], start_mntm[D];
    NutsRes<D> res;         // results of the first half, then of the node
    uint32_t depth, phase;
};

template <class Target>
__global__ __launch_bounds__(64) void nuts_small_kernel(const SmallParams prm, const Target tgt)
{
    constexpr int D = Target::D;
    [[maybe_unused]] constexpr int W = small_reduce_width<Target>::value;
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= prm.C) return;
    const SmallChain<Target> ch(prm, tgt, c);
    double M[D][D], Minv[D][D], L[D][D];
    ch.precond(M);
    sm_inv<D>(M, Minv);                                              // nuts.cpp:62-64
    sm_chol<D>(M, L);
    uint64_t n_leap = 0;

    auto mntm_update = [&](const double (&pos)[D], double (&p)[D], double step) {    // nuts.cpp:108-135
        double grad[D];
        ch.grad_at(pos, grad);
        if (ch.bounded) {
            double J[D][D], jg[D];
            ch.jacobian(pos, J);
            sm_gemv<D>(J, grad, jg);
#pragma unroll
            for (int i = 0; i < D; ++i) p[i] = p[i] + (step * jg[i]) / 2.0;
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i) p[i] = p[i] + (step * grad[i]) / 2.0;
        }
    };
    auto leap_frog = [&](double step, double (&draw)[D], double (&p)[D]) {           // one step of nuts.cpp:139-154
        mntm_update(draw, p, step);
        double mp[D];
        sm_gemv<D>(Minv, p, mp);
#pragma unroll
        for (int i = 0; i < D; ++i) draw[i] = draw[i] + step * mp[i];
        mntm_update(draw, p, step);
        ++n_leap;
    };
    auto kinetic = [&](const double (&p)[D]) -> double {
        double t[D];
        sm_gemv<D>(Minv, p, t);
        return sm_dot<D, W>(p, t) / 2.0;
    };
    auto potential = [&](const double (&v)[D]) -> double {           // -box_log_kernel, non-finite -> +inf
        const double u = -ch.box_log_kernel(v);
        return is_finite(u) ? u : INF;
    };
    auto uturn_ok = [&](const double (&pos)[D], const double (&neg)[D], const double (&mpos)[D], const double (&mneg)[D]) -> uint32_t {
        double diff[D];
#pragma unroll
        for (int i = 0; i < D; ++i) diff[i] = pos[i] - neg[i];
        const uint32_t c1 = sm_dot<D, W>(diff, mneg) >= 0.0 ? 1u : 0u;  // nuts.ipp:226 / nuts.cpp:286
        const uint32_t c2 = sm_dot<D, W>(diff, mpos) >= 0.0 ? 1u : 0u;  // :227 / :287
        return c1 * c2;
    };

    const uint32_t n_total = prm.n_burnin + prm.n_keep;
    const size_t slab = (size_t)prm.d * prm.C;
    double prev_draw[D];
    ch.load_state(prev_draw);                                        // nuts.cpp:160-162
    double step_size, mu_val = 0.0, h_val = 0.0, epsilon_bar = prm.eps;
    if (prm.draw0 == 0) {
        // nuts_find_initial_step_size (nuts.ipp:30-93) from the INIT-stream momentum (nuts.cpp:166-172)
        double z[D], p0[D], nd[D], np_[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double z0, z1;
            rng_normal_pair(prm.seed, ch.chain, 0u, (uint32_t)(i & 3), STREAM_INIT, z0, z1);
            z[i] = (i >> 2) ? z1 : z0;
        }
        sm_gemv<D>(L, z, p0);
        step_size = 1.0;
        const double pU = potential(prev_draw), pK = kinetic(p0);
#pragma unroll
        for (int i = 0; i < D; ++i) { nd[i] = prev_draw[i]; np_[i] = p0[i]; }
        leap_frog(step_size, nd, np_);
        double qU = potential(nd), qK = kinetic(np_);
        const double log_half = det_log(0.5), neg_log2 = -det_log(2.0);
        int a_val = 2 * ((-(qU + qK) + (pU + pK)) > log_half ? 1 : 0) - 1;
        bool check_cond = (-(qU + qK) + (pU + pK)) > neg_log2;
        while (check_cond) {
            step_size *= (a_val == 1) ? 2.0 : 0.5;
            leap_frog(step_size, nd, np_);
            qU = potential(nd); qK = kinetic(np_);
            a_val = 2 * ((-(qU + qK) + (pU + pK)) > log_half ? 1 : 0) - 1;
            check_cond = (-(qU + qK) + (pU + pK)) > neg_log2;
        }
        mu_val = det_log(10 * step_size);                            // nuts.cpp:174
    } else {
        step_size = prm.step_out[c];                                 // continuation: the step size comes back in ...
        epsilon_bar = step_size;                                     // (after the window every draw runs at the adapted epsilon_bar)
        if (prm.draw0 <= prm.n_adapt && prm.adapt_state != nullptr) {   // ... and, inside the adaptation window, the dual-averaging state
            h_val = prm.adapt_state[c]; epsilon_bar = prm.adapt_state[prm.C + c]; mu_val = prm.adapt_state[2 * prm.C + c];
        }
    }
    double prev_U = -ch.box_log_kernel(prev_draw);                   // :181
    uint64_t n_acc = 0;

    NutsFrame<D> F[NUTS_SMALL_MAX_DEPTH];
    for (uint32_t draw = 0; draw < n_total; ++draw) {                // :199
        const uint32_t gd = draw + prm.draw0;                        // index in the chain's random stream / adaptation schedule
        uint32_t uslot = 0;
        double z[D], mntm_vec[D];
        ch.normals(draw, z);
        sm_gemv<D>(L, z, mntm_vec);                                  // :202
        const double prev_K = kinetic(mntm_vec);                     // :204
        const double log_rand_val = det_log(rng_uniform(prm.seed, ch.chain, gd, uslot++)) - prev_U - prev_K;   // :206
        double new_draw[D], draw_pos[D], draw_neg[D], mntm_pos[D], mntm_neg[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            new_draw[i] = prev_draw[i]; draw_pos[i] = prev_draw[i]; draw_neg[i] = prev_draw[i];
            mntm_pos[i] = mntm_vec[i]; mntm_neg[i] = mntm_vec[i];
        }
        uint32_t tree_depth = 0, n_val = 1, s_val = 1, n_alpha_val = 0, good_round = 0;
        double alpha_val = 0.0;

        while (s_val == 1 && tree_depth < prm.max_depth) {           // :227
            const double zd = rng_uniform(prm.seed, ch.chain, gd, uslot++);        // :233
            const int dir = (zd <= 0.5) ? -1 : 1;                    // :235
            const double dstep = (double)dir * step_size;

            // ---- nuts_build_tree(dir, ..., start = (prev_draw, mntm_vec), depth = tree_depth)
            NutsRes<D> R;
            uint32_t sp = 0, cur_depth = tree_depth;
            double cur_draw[D], cur_mntm[D];
#pragma unroll
            for (int i = 0; i < D; ++i) { cur_draw[i] = prev_draw[i]; cur_mntm[i] = mntm_vec[i]; }
            bool calling = true;
            for (;;) {
                if (calling) {
                    if (cur_depth == 0) {                            // nuts.ipp:126-160
                        double nd[D], np_[D];
#pragma unroll
                        for (int i = 0; i < D; ++i) { nd[i] = cur_draw[i]; np_[i] = cur_mntm[i]; }
                        leap_frog(dstep, nd, np_);
                        const double pU = potential(nd), pK = kinetic(np_);
                        R.n = (log_rand_val <= -pU - pK) ? 1u : 0u;  // :146
                        R.s = (log_rand_val < 1000.0 - pU - pK) ? 1u : 0u;   // :147
#pragma unroll
                        for (int i = 0; i < D; ++i) { R.new_draw[i] = nd[i]; R.pos[i] = nd[i]; R.neg[i] = nd[i]; R.mpos[i] = np_[i]; R.mneg[i] = np_[i]; }
                        const double dd = -(pU + pK) + (prev_U + prev_K);
                        R.alpha = det_exp((dd < 0.0) ? dd : 0.0);    // :157
                        R.n_alpha = 1;
                        calling = false;
                    } else {                                         // first half: same start, one level down (:166-171)
                        NutsFrame<D>& f = F[sp];
#pragma unroll
                        for (int i = 0; i < D; ++i) { f.start_draw[i] = cur_draw[i]; f.start_mntm[i] = cur_mntm[i]; }
                        f.depth = cur_depth; f.phase = 1;
                        ++sp; --cur_depth;
                    }
                } else {
                    if (sp == 0) break;                              // R is the result of the top call
                    NutsFrame<D>& g = F[sp - 1];
                    if (g.phase == 1) {
                        g.res = R;
                        if (R.s == 1) {                              // second half from the edge on the dir side (:186-208)
                            g.phase = 2;
#pragma unroll
                            for (int i = 0; i < D; ++i) {
                                cur_draw[i] = (dir == -1) ? g.res.neg[i] : g.res.pos[i];
                                cur_mntm[i] = (dir == -1) ? g.res.mneg[i] : g.res.mpos[i];
                            }
                            cur_depth = g.depth - 1;
                            calling = true;
                        } else {
                            R = g.res; --sp;                         // :234-239
                        }
                    } else {
                        // crossed outputs: the callee's pos-side results land in the caller's neg side for dir = -1 (:195),
                        // its neg-side results in the caller's pos side for dir = +1 (:207); the other pair is a dummy
#pragma unroll
                        for (int i = 0; i < D; ++i) {
                            if (dir == -1) { g.res.neg[i] = R.pos[i]; g.res.mneg[i] = R.mpos[i]; }
                            else           { g.res.pos[i] = R.neg[i]; g.res.mpos[i] = R.mneg[i]; }
                        }
                        const double prob_val = (double)R.n / (double)(g.res.n + R.n);                 // :212
                        const double zz = rng_uniform(prm.seed, ch.chain, gd, uslot++);                // :213
                        if (zz < prob_val) {
#pragma unroll
                            for (int i = 0; i < D; ++i) g.res.new_draw[i] = R.new_draw[i];           // :215-217
                        }
                        g.res.n += R.n; g.res.alpha += R.alpha; g.res.n_alpha += R.n_alpha;          // :220-222
                        g.res.s = R.s * uturn_ok(g.res.pos, g.res.neg, g.res.mpos, g.res.mneg);      // :226-229
                        R = g.res; --sp;
                    }
                }
            }
            // ---- back in nuts_impl: the top call's outputs (nuts.cpp:241-256)
#pragma unroll
            for (int i = 0; i < D; ++i) {
                new_draw[i] = R.new_draw[i];
                if (dir == -1) { draw_neg[i] = R.neg[i]; mntm_neg[i] = R.mneg[i]; }
                else           { draw_pos[i] = R.pos[i]; mntm_pos[i] = R.mpos[i]; }
            }
            alpha_val = R.alpha; n_alpha_val = R.n_alpha;
            if (R.s == 1) {                                          // :260
                const double za = rng_uniform(prm.seed, ch.chain, gd, uslot++);    // :261
                if (za < (double)R.n / (double)n_val) {              // :263
                    prev_U = potential(new_draw);                    // :264-270
#pragma unroll
                    for (int i = 0; i < D; ++i) prev_draw[i] = new_draw[i];       // :272
                    good_round = 1;
                }
            }
            n_val += R.n;                                            // :283
            tree_depth += 1;
            s_val = R.s * uturn_ok(draw_pos, draw_neg, mntm_pos, mntm_neg);       // :286-289
        }

        if (gd < prm.n_adapt) {                                      // :294-302
            const double m = (double)(gd + 1);
            h_val += (1 / (m + prm.t0)) * (prm.delta - (alpha_val / (double)n_alpha_val) - h_val);
            step_size = det_exp(mu_val - h_val * __builtin_sqrt(m) / prm.gamma);
            epsilon_bar *= det_exp(det_pow(m, -prm.kappa) * (det_log(step_size) - det_log(epsilon_bar)));
        } else {
            step_size = epsilon_bar;
        }
        if (prm.depth_trace) prm.depth_trace[(size_t)draw * prm.C + c] = tree_depth;
        if (draw >= prm.n_burnin) {                                  // :306-309
            n_acc += good_round;
            if (prm.draws) ch.store_natural(prm.draws + (size_t)(draw - prm.n_burnin) * slab + c, prev_draw);
        }
    }
    ch.store_natural(prm.theta + c, prev_draw);
    if (prm.n_accept) prm.n_accept[c] = n_acc;
    if (prm.n_leap) prm.n_leap[c] = n_leap;
    if (prm.step_out) prm.step_out[c] = step_size;
    if (prm.adapt_state) { prm.adapt_state[c] = h_val; prm.adapt_state[prm.C + c] = epsilon_bar; prm.adapt_state[2 * prm.C + c] = mu_val; }
}

}  // namespace mi
