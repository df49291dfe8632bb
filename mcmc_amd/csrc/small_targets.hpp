// small_targets.hpp -- target policies of the one-lane-per-chain engine (small_samplers.hpp, rmhmc_small.hpp) besides the
// normal model of rmhmc_small.hpp.  A policy is what the reference's callback is on the host (ref: include/mcmc/hmc.hpp:42-48):
//
//     struct Target {
//         static constexpr int D = ...;                  // dimension, compile time, <= SMALL_MAX_D
//         static constexpr int W = 1;                    // optional: order of the samplers' dot products (1 or 4, see sm_dot)
//         __device__ double kernel(const double (&vals)[D], double (&grad)[D], bool want_grad) const;   // log kernel, fills grad
//         __device__ void tensor(const double (&vals)[D], double (&G)[D][D], double (*dG)[D][D]) const;  // rmhmc only
//     };
//
// passed to the kernels BY VALUE (plain data + device pointers).  include/mi_mcmc_target.hpp turns a user's policy into a library.
#pragma once

#include "det_math.hpp"
#include "rmhmc_small.hpp"

namespace mi {

// Bayesian logistic regression with D <= 8 coefficients (MI_TARGET_LOGISTIC; SURVEY 8(d) C3's model at small d):
//   log K = sum_r [y_r eta_r - log(1 + e^eta_r)] - |beta|^2 / 2,  eta = X beta,  grad = X^T (y - sigmoid(eta)) - beta.
// Reduction orders = what logit_lds_kernel produces for d <= 8 (oracle knobs W = 4, 4 blocks of 16, 2 eta sub-chains: with
// d <= 8 all dimensions sit in the first sub-chain of the first block): eta_r one fma chain over j; the row sum four strided
// chains over r (mod 4), (q0 + q2) + (q1 + q3); |beta|^2 and the samplers' dots in the same four-chain order (W = 4); X^T r one
// fma chain over the rows.  So this engine and the LDS kernel give the same bits on the same problem -- it serves what the LDS
// kernel does not implement: mcmc::nuts, box constraints, precond_mat / cov_mat.
template <int D_>
struct LogisticSmallModel {
    static constexpr int D = D_;
    static constexpr int W = 4;
    const double* X;      // n x D row-major (device)
    const double* y;      // n (device)
    uint32_t n;

    __device__ __forceinline__ double kernel(const double (&v)[D_], double (&g)[D_], bool want_grad) const
    {
        // every lane reads the same rows: scalar loads through the constant address space (rmhmc_small.hpp: NormalModel)
        typedef const double __attribute__((address_space(4)))* cptr_t;
        cptr_t Xc = (cptr_t)(uintptr_t)X;
        cptr_t yc = (cptr_t)(uintptr_t)y;
        double q[4] = {0.0, 0.0, 0.0, 0.0};
        double acc[D_];
#pragma unroll
        for (int j = 0; j < D_; ++j) acc[j] = 0.0;
        for (uint32_t r = 0; r < n; ++r) {
            double eta = 0.0;
#pragma unroll
            for (int j = 0; j < D_; ++j) eta = dfma(Xc[(size_t)r * D_ + j], v[j], eta);
            const double yr = yc[r];
            q[r & 3] = q[r & 3] + (yr * eta - softplus(eta));
            if (want_grad) {
                const double res = yr - sigmoid(eta);
#pragma unroll
                for (int j = 0; j < D_; ++j) acc[j] = dfma(Xc[(size_t)r * D_ + j], res, acc[j]);
            }
        }
        const double ll = (q[0] + q[2]) + (q[1] + q[3]);
        double b[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < D_; ++j) b[j & 3] = dfma(v[j], v[j], b[j & 3]);
        if (want_grad) {
#pragma unroll
            for (int j = 0; j < D_; ++j) g[j] = acc[j] - v[j];
        }
        return ll - 0.5 * ((b[0] + b[2]) + (b[1] + b[3]));
    }
    // mcmc::rmhmc: the Fisher information plus the prior precision, G = X^T diag(lam) X + I, lam_r = s_r (1 - s_r), and
    // dG/dbeta_i = X^T diag(lam_r (1 - 2 s_r) X_ri) X; rows ascending, one fma per row and entry (oracle: orc_target_tensor)
    __device__ __forceinline__ void tensor(const double (&v)[D_], double (&G)[D_][D_], double (*dG)[D_][D_]) const
    {
        typedef const double __attribute__((address_space(4)))* cptr_t;
        cptr_t Xc = (cptr_t)(uintptr_t)X;
#pragma unroll
        for (int r = 0; r < D_; ++r)
#pragma unroll
            for (int c = 0; c < D_; ++c) {
                G[r][c] = 0.0;
                if (dG) {
#pragma unroll
                    for (int i = 0; i < D_; ++i) dG[i][r][c] = 0.0;
                }
            }
        for (uint32_t k = 0; k < n; ++k) {
            double x[D_];
            double eta = 0.0;
#pragma unroll
            for (int j = 0; j < D_; ++j) { x[j] = Xc[(size_t)k * D_ + j]; eta = dfma(x[j], v[j], eta); }
            const double sg = sigmoid(eta);
            const double lam = sg * (1.0 - sg);
            const double dl = lam * (1.0 - 2.0 * sg);
#pragma unroll
            for (int r = 0; r < D_; ++r)
#pragma unroll
                for (int c = 0; c < D_; ++c) {
                    const double xx = x[r] * x[c];
                    G[r][c] = dfma(xx, lam, G[r][c]);
                    if (dG) {
#pragma unroll
                        for (int i = 0; i < D_; ++i) dG[i][r][c] = dfma(xx, dl * x[i], dG[i][r][c]);
                    }
                }
        }
#pragma unroll
        for (int r = 0; r < D_; ++r) G[r][r] = G[r][r] + 1.0;
    }
};

}  // namespace mi
