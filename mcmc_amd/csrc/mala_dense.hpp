// mala_dense.hpp -- many-chain MALA for dense-gradient Gaussian targets (d <= 128) on the fp64
// matrix cores; same wavefront mapping as hmc_dense.hpp (16 chains per wave).
//
// Replaces mcmc::internal::mala_impl's draw loop (/root/reference/src/mala.cpp:149-186) with
// mala_mean_fn (:97-125), mala_prop_adjustment (/root/reference/include/mcmc/mala.ipp:30-70) and
// stats_mcmc::dmvnorm (/root/reference/include/stats/dmvnorm.hpp:28-54) for the identity
// preconditioner, where Sigma = eps^2 I:  INV(Sigma) = diag(1/eps^2), LOG_DET(Sigma) = sum_i 2 log sqrt(eps^2)
// (both precomputed on the host with the same arithmetic the oracle uses).
// The reference evaluates the gradient three times per draw (mu(theta), mu(theta'), mu(theta) again);
// the values are deterministic, so P*theta of the current state is cached and ONE mat-vec per draw
// (at the proposal) reproduces all of them bit for bit.
#pragma once

#include "hmc_dense.hpp"

namespace mi {

struct MalaParams {
    const double* P;
    uint32_t d;
    uint64_t C, chain0;
    double* theta;          // [d][C] in/out
    double* draws;
    uint64_t* n_accept;
    uint64_t seed;
    uint32_t n_burnin, n_keep;
    double eps;             // step_size
    double s2;              // eps*eps (mala.cpp:123, mala.ipp:41)
    double rs;              // 1.0 / s2: diagonal of INV(eps^2 I)
    double cons_term;       // -0.5 * d * log(2 pi)   (dmvnorm.hpp:36)
    double log_det;         // LOG_DET(eps^2 I)
};

template <int NT>
__global__ MI_NO_DS_MERGE __launch_bounds__(256, 1) void mala_gauss_mfma_kernel(const MalaParams prm)
{
    constexpr int NS = 4 * NT;
    extern __shared__ __attribute__((aligned(16))) double lds_P[];
    stage_precision<NT>(prm.P, prm.d, lds_P);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane >> 4;
    const uint64_t cl = ((uint64_t)blockIdx.x * 4 + wave) * 16 + (lane & 15);
    const bool live = cl < prm.C;
    const uint64_t cld = live ? cl : prm.C - 1;
    const uint64_t chain = prm.chain0 + cl;
    const uint32_t d = prm.d;
    const uint64_t C = prm.C;
    const double eps = prm.eps, s2 = prm.s2, rs = prm.rs;
    const double* afrag = lds_P + lane;
    const size_t lane_off = (size_t)j * C + cld;

    double th[NS], w[NS];        // current state and P*theta (grad = -w)
    double tp[NS], wp[NS];       // proposal and P*theta'
    double xc[NS], tt[NS];

#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const uint32_t dim = 4 * s + j;
        th[s] = (dim < d) ? prm.theta[(size_t)(4 * s) * C + lane_off] : 0.0;
    }
    matvec_mfma<NT>(afrag, th, w);
    double prev_LP = -0.5 * dot4<NS>(th, w);            // box_log_kernel(first_draw), mala.cpp:138
    uint64_t n_acc = 0;
    const uint32_t n_total = prm.n_burnin + prm.n_keep;

#pragma unroll 1
    for (uint32_t draw = 0; draw < n_total; ++draw) {
        // proposal: mala_mean_fn(prev) + eps * L z   (mala.cpp:150,159; L = I)
#pragma unroll
        for (int b = 0; b < NS / 2; ++b) {
            double z0, z1;
            rng_normal_pair(prm.seed, chain, draw, (uint32_t)(4 * b + j), STREAM_NORMAL, z0, z1);
            const double za = (8u * b + j < d) ? z0 : 0.0;
            const double zb = (8u * b + 4 + j < d) ? z1 : 0.0;
            const double ma = th[2 * b] - (s2 * w[2 * b]) / 2.0;           // theta + (s2*grad)/2, :123
            const double mb = th[2 * b + 1] - (s2 * w[2 * b + 1]) / 2.0;
            tp[2 * b] = ma + eps * za;
            tp[2 * b + 1] = mb + eps * zb;
            __builtin_amdgcn_sched_barrier(0);
        }
        matvec_mfma<NT>(afrag, tp, wp);
        double prop_LP = -0.5 * dot4<NS>(tp, wp);        // :162
        if (!is_finite(prop_LP)) prop_LP = -INF;         // :164-166
        // mala_prop_adjustment (mala.ipp:59-64): dmvnorm(prev | mu(prop)) - dmvnorm(prop | mu(prev))
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const double mean_prop = tp[s] - (s2 * wp[s]) / 2.0;
            xc[s] = th[s] - mean_prop;                   // dmvnorm.hpp:37
            tt[s] = rs * xc[s];                          // INV(Sigma) * X_cent
        }
        const double quad_a = dot4<NS>(xc, tt);          // :39
        const double da = prm.cons_term - 0.5 * (prm.log_det + quad_a);   // :41
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const double mean_prev = th[s] - (s2 * w[s]) / 2.0;
            xc[s] = tp[s] - mean_prev;
            tt[s] = rs * xc[s];
        }
        const double quad_b = dot4<NS>(xc, tt);
        const double db = prm.cons_term - 0.5 * (prm.log_det + quad_b);
        const double x = prop_LP - prev_LP + (da - db);
        const double comp_val = (x < 0.01) ? x : 0.01;   // std::min(0.01, x), mala.cpp:170
        const double z = rng_uniform(prm.seed, chain, draw, 0u);          // :171
        const bool accept = z < det_exp(comp_val);       // :173
        if (accept) {
#pragma unroll
            for (int s = 0; s < NS; ++s) { th[s] = tp[s]; w[s] = wp[s]; }
            prev_LP = prop_LP;
        }
        if (draw >= prm.n_burnin) {
            n_acc += accept ? 1u : 0u;
            if (prm.draws != nullptr && live) {
                double* out = prm.draws + (size_t)(draw - prm.n_burnin) * d * C;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const uint32_t dim = 4 * s + j;
                    if (dim < d) (out + (size_t)(4 * s) * C)[lane_off] = th[s];
                }
            }
        }
    }
    if (live) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const uint32_t dim = 4 * s + j;
            if (dim < d) prm.theta[(size_t)(4 * s) * C + lane_off] = th[s];
        }
        if (j == 0 && prm.n_accept) prm.n_accept[cl] = n_acc;
    }
}

}  // namespace mi
