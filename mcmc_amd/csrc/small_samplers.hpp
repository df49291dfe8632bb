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

template <int D>
__device__ __forceinline__ double sm_dot(const double (&x)[D], const double (&y)[D])
{
    double q = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) q = dfma(x[i], y[i], q);
    return q;
}

// what every small sampler shares: the chain's lane, the box maps, the precond matrix and the draw bookkeeping
template <class Target>
struct SmallChain {
    static constexpr int D = Target::D;
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
            rng_normal_pair(prm.seed, chain, draw + prm.draw0, (uint32_t)(i & 3), STREAM_NORMAL, z0, z1);
            z[i] = (i >> 2) ? z1 : z0;
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
        return sm_dot<D>(p, t) / 2.0;
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
        return cons_term - 0.5 * (log_det + sm_dot<D>(xc, t));       // :41
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

}  // namespace mi
