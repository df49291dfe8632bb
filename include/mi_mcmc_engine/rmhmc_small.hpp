// rmhmc_small.hpp -- many-chain Riemannian-manifold HMC for small-dimensional targets, one lane per chain.
//
// Replaces, for C independent chains, the draw loop of mcmc::internal::rmhmc_impl (/root/reference/src/rmhmc.cpp:30-287;
// SURVEY section 8 row f-4).  RM-HMC needs, per chain and per fixed-point iteration, a d x d inverse of a position-dependent
// metric tensor and d products Ginv * dG_i: O(d^4) work on matrices that differ between chains, so nothing is shared across
// a chain tile and there is no contraction for the matrix cores.  The kernel therefore keeps one chain per lane with the whole
// state (position, momentum, three d x d matrices, two d x d x d derivative cubes) in registers, for compile-time D (the cubes
// make D <= 4 the practical range; the other samplers of small_samplers.hpp take D <= SMALL_MAX_D = 8):
// the reference's own use of RM-HMC is the d = 2 normal model of examples/eigen/rmhmc_normal.cpp.  Bound: fp64 VALU (the
// data sums of the target); HBM sees theta once per call and one slab per kept draw.
//
// Arithmetic = oracle/mcmc_oracle.c:orc_rmhmc, operation for operation: sequential sums, fma only where the oracle writes
// fma (mat-vec / mat-mat rows, dots), Gauss-Jordan inverse with partial pivoting, Cholesky, log-det = sum 2 log L_ii.
// The sign of the momentum increment is the reference's (rmhmc.cpp:116,139,145: p + eps/2 * dH/dtheta) -- see DESIGN.md.
#pragma once

#include "det_math.hpp"
#include "hmc_dense.hpp"      // box_* element-wise maps

namespace mi {

constexpr double LOG_2PI = 1.83787706640934548356;   // /root/reference/include/stats/mcmc_stats.hpp:28-30

constexpr int SMALL_MAX_D = 8;      // dimensions of a one-lane-per-chain target (Target::D <= SMALL_MAX_D; rmhmc: D <= 4 is practical)

struct SmallParams {
    const double* data;     // NORMAL_MODEL: the observations x_1..x_n (device); other targets carry their own data
    uint32_t n_rows;
    uint32_t d;
    uint64_t C, chain0;
    double* theta;          // [d][C] in/out
    double* draws;          // [n_keep][d][C] or nullptr
    uint64_t* n_accept;
    uint64_t* n_leap;
    uint64_t seed;
    uint32_t n_burnin, n_keep, n_leap_steps, n_fp_steps, draw0;
    double eps;
    int vals_bound;
    int btype[SMALL_MAX_D];
    double lb[SMALL_MAX_D], ub[SMALL_MAX_D];
    double M[SMALL_MAX_D][SMALL_MAX_D];   // hmc / mala / nuts: precond_mat; rwmh: cov_mat (identity when the settings carry none); unused by rmhmc
    // nuts (nuts_settings_t, mcmc_structs.hpp:82-101); eps carries epsilon_bar_0
    uint32_t n_adapt, max_depth;
    double delta, gamma, t0, kappa;
    double* step_out;       // [C] or nullptr: in (continuation, draw0 > 0) / out: step size
    double* adapt_state;    // [3][C] or nullptr: nuts dual-averaging state (h, epsilon_bar, mu): out always, in when 0 < draw0 <= n_adapt
    uint32_t* depth_trace;  // [n_burnin + n_keep][C] or nullptr
};

// ---- the d = 2 normal model of the reference's example programs (examples/eigen/rmhmc_normal.cpp:44-106):
//      vals = (mu, sigma), log K = -n (log(2 pi)/2 + log sigma) - sum (x - mu)^2 / (2 sigma^2), Fisher metric.
struct NormalModel {
    static constexpr int D = 2;
    const double* x;
    uint32_t n;

    __device__ __forceinline__ double kernel(const double (&v)[2], double (&g)[2], bool want_grad) const
    {
        const double mu = v[0], sigma = v[1];
        const double nn = (double)n;
        double m1 = 0.0, m2 = 0.0;
        // The observations are read-only for the whole launch and every lane reads the same one: through the constant address
        // space the loads become scalar (s_load_dwordx16 = 8 observations per request into SGPRs, no per-lane address, no
        // vmcnt wait per observation), double-buffered one block ahead.  Summation order unchanged: r ascending.
        typedef const double __attribute__((address_space(4)))* cptr_t;
        cptr_t xc = (cptr_t)(uintptr_t)x;
        constexpr uint32_t B = 8;
        double nxt[B];
        const uint32_t nb = n / B;
        if (nb) {
#pragma unroll
            for (uint32_t k = 0; k < B; ++k) nxt[k] = xc[k];
        }
        for (uint32_t b = 0; b < nb; ++b) {
            double cur[B];
#pragma unroll
            for (uint32_t k = 0; k < B; ++k) cur[k] = nxt[k];
            if (b + 1 < nb) {
#pragma unroll
                for (uint32_t k = 0; k < B; ++k) nxt[k] = xc[(b + 1) * B + k];
            }
#pragma unroll
            for (uint32_t k = 0; k < B; ++k) {
                const double e = cur[k] - mu;
                m1 = m1 + e;
                m2 = dfma(e, e, m2);
            }
        }
        for (uint32_t r = nb * B; r < n; ++r) {
            const double e = xc[r] - mu;
            m1 = m1 + e;
            m2 = dfma(e, e, m2);
        }
        const double s2 = sigma * sigma;
        const double ret = -(nn * (0.5 * LOG_2PI + det_log(sigma))) - m2 / (2.0 * s2);
        if (want_grad) {
            g[0] = m1 / s2;
            g[1] = m2 / (s2 * sigma) - nn / sigma;
        }
        return ret;
    }
    // G and (optionally) dG[i] = dG/dvals_i
    __device__ __forceinline__ void tensor(const double (&v)[2], double (&G)[2][2], double (*dG)[2][2]) const
    {
        const double sigma = v[1];
        const double nn = (double)n;
        const double s2 = sigma * sigma;
        G[0][0] = nn / s2; G[0][1] = 0.0; G[1][0] = 0.0; G[1][1] = (2.0 * nn) / s2;
        if (dG) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) { dG[0][r][c] = 0.0; dG[1][r][c] = (-2.0 * G[r][c]) / sigma; }
        }
    }
};

// ---- small dense helpers, all unrolled (registers), operation order of oracle/mcmc_oracle.c
template <int D>
__device__ __forceinline__ void sm_gemv(const double (&A)[D][D], const double (&x)[D], double (&y)[D])
{
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) acc = dfma(A[i][j], x[j], acc);
        y[i] = acc;
    }
}

// orc_inv: Gauss-Jordan with partial pivoting (first strict maximum)
template <int D>
__device__ __forceinline__ void sm_inv(const double (&A)[D][D], double (&Ai)[D][D])
{
    double a[D][D];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) { a[i][j] = A[i][j]; Ai[i][j] = (i == j) ? 1.0 : 0.0; }
#pragma unroll
    for (int c = 0; c < D; ++c) {
        int piv = c;
        double best = __builtin_fabs(a[c][c]);
#pragma unroll
        for (int r = c + 1; r < D; ++r) {
            const double m = __builtin_fabs(a[r][c]);
            if (m > best) { best = m; piv = r; }
        }
#pragma unroll
        for (int r = c + 1; r < D; ++r) {
            const bool sw = (piv == r);
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double t0 = a[c][j], t1 = a[r][j];
                a[c][j] = sw ? t1 : t0; a[r][j] = sw ? t0 : t1;
                const double u0 = Ai[c][j], u1 = Ai[r][j];
                Ai[c][j] = sw ? u1 : u0; Ai[r][j] = sw ? u0 : u1;
            }
        }
        const double pv = a[c][c];
#pragma unroll
        for (int j = 0; j < D; ++j) { a[c][j] = a[c][j] / pv; Ai[c][j] = Ai[c][j] / pv; }
#pragma unroll
        for (int r = 0; r < D; ++r) {
            if (r == c) continue;
            const double f = a[r][c];
            const bool upd = !(f == 0.0);
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double na = a[r][j] - f * a[c][j];
                const double ni = Ai[r][j] - f * Ai[c][j];
                a[r][j] = upd ? na : a[r][j];
                Ai[r][j] = upd ? ni : Ai[r][j];
            }
        }
    }
}

// orc_chol_lower
template <int D>
__device__ __forceinline__ void sm_chol(const double (&A)[D][D], double (&L)[D][D])
{
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) L[i][j] = 0.0;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        double sum = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) sum = sum - L[j][k] * L[j][k];
        const double ljj = __builtin_sqrt(sum);
        L[j][j] = ljj;
#pragma unroll
        for (int i = j + 1; i < D; ++i) {
            double t = A[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) t = t - L[i][k] * L[j][k];
            L[i][j] = t / ljj;
        }
    }
}

template <int D>
__device__ __forceinline__ double sm_log_det(const double (&G)[D][D])          // LOG_DET via Cholesky: sum 2 log L_ii
{
    double L[D][D];
    sm_chol<D>(G, L);
    double ld = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) ld = ld + 2.0 * det_log(L[i][i]);
    return ld;
}
template <int D>
__device__ __forceinline__ double sm_half_log_det(const double (&G)[D][D]) { return 0.5 * sm_log_det<D>(G); }

template <class Target>
__global__ __launch_bounds__(256) void rmhmc_small_kernel(const SmallParams prm, const Target tgt)
{
    constexpr int D = Target::D;
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= prm.C) return;
    const uint64_t chain = prm.chain0 + c;
    const uint64_t C = prm.C;
    const double eps = prm.eps;
    const bool bounded = prm.vals_bound != 0;

    auto inv_tr = [&](const double (&v)[D], double (&o)[D]) {
#pragma unroll
        for (int i = 0; i < D; ++i) o[i] = bounded ? box_inv_transform(v[i], prm.btype[i], prm.lb[i], prm.ub[i]) : v[i];
    };
    // box_log_kernel (rmhmc.cpp:84-95)
    auto box_log_kernel = [&](const double (&v)[D]) -> double {
        double vi[D], g[D];
        inv_tr(v, vi);
        const double k = tgt.kernel(vi, g, false);
        if (!bounded) return k;
        double lj = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) if (prm.btype[i] != 1) lj += box_log_jacobian_term(v[i], prm.btype[i], prm.lb[i], prm.ub[i]);
        return k + lj;
    };
    // box_tensor_fn (rmhmc.cpp:152-164)
    auto box_tensor = [&](const double (&v)[D], double (&G)[D][D], double (*dG)[D][D]) {
        double vi[D];
        inv_tr(v, vi);
        tgt.tensor(vi, G, dG);
    };
    // mntm_update_fn (rmhmc.cpp:99-150): the increment eps * [J] grad_obj / 2
    auto mntm_incr = [&](const double (&pos)[D], const double (&p)[D], const double (&Gi)[D][D], const double (&dG)[D][D][D],
                         double (&out)[D]) {
        double pi[D], grad[D], gobj[D], b[D];
        inv_tr(pos, pi);
        (void)tgt.kernel(pi, grad, true);
        sm_gemv<D>(Gi, p, b);                                        // Ginv p
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double T[D][D];                                          // Ginv * dG_i
#pragma unroll
            for (int r = 0; r < D; ++r)
#pragma unroll
                for (int s = 0; s < D; ++s) {
                    double acc = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) acc = dfma(Gi[r][k], dG[i][k][s], acc);
                    T[r][s] = acc;
                }
            double tr = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) tr = tr + T[j][j];
            double dp = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                double a = 0.0;                                      // (T^T p)_j
#pragma unroll
                for (int k = 0; k < D; ++k) a = dfma(T[k][j], p[k], a);
                dp = dfma(a, b[j], dp);
            }
            gobj[i] = -grad[i] + 0.5 * (tr - dp);
        }
        if (bounded) {
            // jacob_matrix * grad_obj as the dense product the reference forms (zeros off the diagonal take part)
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const double Jij = (i == j) ? box_inv_jacobian(pos[i], prm.btype[i], prm.lb[i], prm.ub[i]) : 0.0;
                    acc = dfma(Jij, gobj[j], acc);
                }
                out[i] = (eps * acc) / 2.0;
            }
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i) out[i] = (eps * gobj[i]) / 2.0;
        }
    };
    auto kinetic = [&](const double (&p)[D], const double (&Gi)[D][D]) -> double {
        double t[D];
        sm_gemv<D>(Gi, p, t);
        double k = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) k = dfma(p[i], t[i], k);
        return k / 2.0;
    };

    double prev_draw[D], new_draw[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const double v = prm.theta[(size_t)i * C + c];
        prev_draw[i] = bounded ? box_transform(v, prm.btype[i], prm.lb[i], prm.ub[i]) : v;     // rmhmc.cpp:170-172
        new_draw[i] = prev_draw[i];
    }
    double new_tensor[D][D], prev_tensor[D][D], inv_new[D][D], inv_prev[D][D];
    double new_deriv[D][D][D], prev_deriv[D][D][D];
    box_tensor(new_draw, new_tensor, new_deriv);                     // :187
    sm_inv<D>(new_tensor, inv_new);                                  // :190
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            prev_tensor[i][j] = new_tensor[i][j]; inv_prev[i][j] = inv_new[i][j];
#pragma unroll
            for (int k = 0; k < D; ++k) prev_deriv[i][j][k] = new_deriv[i][j][k];
        }
    const double cons_term = 0.5 * (double)D * LOG_2PI;              // :195
    double prev_U = cons_term - box_log_kernel(prev_draw) + sm_half_log_det<D>(new_tensor);   // :197
    uint64_t n_acc = 0;
    const uint32_t n_total = prm.n_burnin + prm.n_keep;
    const size_t slab = (size_t)prm.d * C;

    for (uint32_t draw = 0; draw < n_total; ++draw) {                // :206
        double z[D], p[D], pp[D], incr[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double z0, z1;
            rng_normal_pair(prm.seed, chain, draw + prm.draw0, (uint32_t)(i & 3), STREAM_NORMAL, z0, z1);
            z[i] = (i >> 2) ? z1 : z0;
        }
        {
            double L[D][D];
            sm_chol<D>(prev_tensor, L);
            sm_gemv<D>(L, z, p);                                     // :209
        }
        const double prev_K = kinetic(p, inv_prev);                  // :211
#pragma unroll
        for (int i = 0; i < D; ++i) new_draw[i] = prev_draw[i];      // :213

        for (uint32_t k = 0; k < prm.n_leap_steps; ++k) {            // :215
#pragma unroll
            for (int i = 0; i < D; ++i) pp[i] = p[i];
            for (uint32_t kk = 0; kk < prm.n_fp_steps; ++kk) {       // :220-222
                mntm_incr(new_draw, pp, inv_prev, prev_deriv, incr);
#pragma unroll
                for (int i = 0; i < D; ++i) pp[i] = p[i] + incr[i];
            }
#pragma unroll
            for (int i = 0; i < D; ++i) p[i] = pp[i];                // :224
            double prop[D];
#pragma unroll
            for (int i = 0; i < D; ++i) prop[i] = new_draw[i];       // :228
            for (uint32_t kk = 0; kk < prm.n_fp_steps; ++kk) {       // :231-235
                double Tn[D][D], S[D][D], t[D];
                box_tensor(prop, Tn, nullptr);
                sm_inv<D>(Tn, inv_new);
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int j = 0; j < D; ++j) S[i][j] = inv_prev[i][j] + inv_new[i][j];
                sm_gemv<D>(S, p, t);
#pragma unroll
                for (int i = 0; i < D; ++i) prop[i] = new_draw[i] + (0.5 * eps) * t[i];
            }
#pragma unroll
            for (int i = 0; i < D; ++i) new_draw[i] = prop[i];       // :237
            box_tensor(new_draw, new_tensor, new_deriv);             // :239
            sm_inv<D>(new_tensor, inv_new);                          // :240
            mntm_incr(new_draw, p, inv_new, new_deriv, incr);        // :244
#pragma unroll
            for (int i = 0; i < D; ++i) p[i] = p[i] + incr[i];
        }

        double prop_U = cons_term - box_log_kernel(new_draw) + sm_half_log_det<D>(new_tensor);   // :247
        if (!is_finite(prop_U)) prop_U = INF;                        // :249-251
        const double prop_K = kinetic(p, inv_new);                   // :253
        const double x = -(prop_U + prop_K) + (prev_U + prev_K);
        const double comp_val = (x < 0.01) ? x : 0.01;               // :257
        const double zu = rng_uniform(prm.seed, chain, draw + prm.draw0, 0u);   // :258
        const bool accept = zu < det_exp(comp_val);                  // :260
        if (accept) {
            prev_U = prop_U;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                prev_draw[i] = new_draw[i];
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    prev_tensor[i][j] = new_tensor[i][j]; inv_prev[i][j] = inv_new[i][j];
#pragma unroll
                    for (int k2 = 0; k2 < D; ++k2) prev_deriv[i][j][k2] = new_deriv[i][j][k2];
                }
            }
        }
        if (draw >= prm.n_burnin) {
            n_acc += accept ? 1u : 0u;
            if (prm.draws) {
                double* row = prm.draws + (size_t)(draw - prm.n_burnin) * slab + c;
#pragma unroll
                for (int i = 0; i < D; ++i)                           // :264,:270 + the epilogue inv_transform (:277-284)
                    row[(size_t)i * C] = bounded ? box_inv_transform(prev_draw[i], prm.btype[i], prm.lb[i], prm.ub[i]) : prev_draw[i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < D; ++i)
        prm.theta[(size_t)i * C + c] = bounded ? box_inv_transform(prev_draw[i], prm.btype[i], prm.lb[i], prm.ub[i]) : prev_draw[i];
    if (prm.n_accept) prm.n_accept[c] = n_acc;
    if (prm.n_leap) prm.n_leap[c] = (uint64_t)n_total * prm.n_leap_steps;
}

}  // namespace mi
