// rwmh_dense.hpp -- many-chain random-walk Metropolis-Hastings for dense-gradient Gaussian targets (d <= 128) on the fp64
// matrix cores; same wavefront mapping as hmc_dense.hpp (16 chains per wave).  SURVEY 8 (f-4).
//
// Replaces the draw loop of mcmc::internal::rwmh_impl (/root/reference/src/rwmh.cpp:123-151):
//   new_draw = prev_draw + (par_scale * CHOL_LOWER(cov_mat)) * z,   prop_LP = box_log_kernel(new_draw) (value only),
//   accept iff runif < exp(min(0, prop_LP - prev_LP))               (min(0, NaN) = 0: a NaN difference is accepted, :136)
// One MFMA mat-vec per draw (the Gaussian log kernel is -x.Px/2).  Variants:
//   plain                    cov_mat absent (identity, :58), no bounds: step = par_scale * z
//   GENERAL                  settings.vals_bound (:64-79,105-107,157-166) and / or a DIAGONAL cov_mat: step_i = c_i z_i with
//                            c = par_scale * sqrt(cov_ii) from the host; chain in the transformed space, log_jacobian summed
//                            over dimensions in order, rows stored through inv_transform
//   GENERAL + DENSE_C        a dense cov_mat (d <= 64): par_scale * CHOL_LOWER(cov) from the host as a second set of MFMA
//                            A-fragments, step = Lc z as a mat-vec with the oracle's fma order
#pragma once

#include "hmc_dense.hpp"

namespace mi {

struct RwmhParams {
    const double* P;
    uint32_t d;
    uint64_t C, chain0;
    double* theta;          // [d][C] in/out
    double* draws;
    uint64_t* n_accept;
    uint64_t seed;
    uint32_t n_burnin, n_keep;
    uint32_t draw0;         // index of this call's first draw in the chains' random streams (mi_chains.draw0)
    double par_scale;       // rwmh_settings.par_scale
    int vals_bound;
    const int* btype;       // GENERAL: [d] each
    const double* lb;
    const double* ub;
    const double* c_diag;   // par_scale * sqrt(cov_ii)
    const double* Lc;       // DENSE_C: par_scale * CHOL_LOWER(cov_mat), d*d row-major (device)
};

template <int NT, bool GENERAL = false, bool DENSE_C = false>
__global__ MI_NO_DS_MERGE __launch_bounds__(256, 1) void rwmh_gauss_mfma_kernel(const RwmhParams prm)
{
    static_assert(!DENSE_C || GENERAL, "the dense proposal covariance rides the general variant");
    constexpr int NS = 4 * NT;
    extern __shared__ __attribute__((aligned(16))) double lds_P[];
    double* const lds_lb = lds_P + (size_t)NT * 4 * NT * 64;
    double* const lds_ub = lds_lb + 16 * NT;
    double* const lds_c = lds_ub + 16 * NT;
    int* const lds_bt = reinterpret_cast<int*>(lds_c + 16 * NT);
    double* const lds_Lc = lds_c + 16 * NT + 8 * NT;      // after the int table, DENSE_C only
    if constexpr (GENERAL) {
        for (int i = threadIdx.x; i < 16 * NT; i += blockDim.x) {
            const bool in = (uint32_t)i < prm.d;
            lds_lb[i] = in ? prm.lb[i] : 0.0;
            lds_ub[i] = in ? prm.ub[i] : 0.0;
            lds_bt[i] = in ? prm.btype[i] : 1;
            lds_c[i] = in ? prm.c_diag[i] : 0.0;
        }
    }
    if constexpr (DENSE_C && !dense_m_from_global<NT>()) stage_precision<NT>(prm.Lc, prm.d, lds_Lc);      // d > 64: read from L2 in fragment order
    stage_precision<NT>(prm.P, prm.d, lds_P);           // ends with a barrier

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane >> 4;
    const uint64_t cl = ((uint64_t)blockIdx.x * 4 + wave) * 16 + (lane & 15);
    const bool live = cl < prm.C;
    const uint64_t cld = live ? cl : prm.C - 1;
    const uint64_t chain = prm.chain0 + cl;
    const uint32_t d = prm.d;
    const uint64_t C = prm.C;
    const double* afrag = lds_P + lane;
    [[maybe_unused]] const double* afrag_lc = (DENSE_C && dense_m_from_global<NT>()) ? prm.Lc + lane : lds_Lc + lane;
    const size_t lane_off = (size_t)j * C + cld;
    const bool vb = GENERAL && prm.vals_bound != 0;

    double th[NS], tp[NS], xp[NS], wp[NS];
    // box_log_kernel (rwmh.cpp:64-79) of the state in (tt, with x / P x in xp / wp)
    auto log_kernel = [&](const double (&tt)[NS]) __attribute__((always_inline)) -> double {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int dim = 4 * s + j;
            if constexpr (GENERAL) xp[s] = vb ? (((uint32_t)dim < d) ? box_inv_transform(tt[s], lds_bt[dim], lds_lb[dim], lds_ub[dim]) : 0.0) : tt[s];
            else xp[s] = tt[s];
        }
        matvec_mfma<NT>(afrag, xp, wp);
        const double kval = -0.5 * dot4<NS>(xp, wp);
        if constexpr (GENERAL) {
            if (!vb) return kval;
            double lj = 0.0;                             // log_jacobian.hpp:36-57: scalar loop, i ascending
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int i0 = 4 * s;
                const double term = box_log_jacobian_term(tt[s], lds_bt[i0 + j], lds_lb[i0 + j], lds_ub[i0 + j]);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const double tg = __shfl(term, (lane & 15) + 16 * g);
                    if ((uint32_t)(i0 + g) < d && lds_bt[i0 + g] != 1) lj = lj + tg;
                }
            }
            return kval + lj;
        } else {
            return kval;
        }
    };

#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const uint32_t dim = 4 * s + j;
        const double v = (dim < d) ? prm.theta[(size_t)(4 * s) * C + lane_off] : 0.0;
        if constexpr (GENERAL) th[s] = (vb && dim < d) ? box_transform(v, lds_bt[dim], lds_lb[dim], lds_ub[dim]) : v;   // rwmh.cpp:105-107
        else th[s] = v;
    }
    double prev_LP = log_kernel(th);                    // :113
    uint64_t n_acc = 0;
    const uint32_t n_total = prm.n_burnin + prm.n_keep;

#pragma unroll 1
    for (uint32_t draw = 0; draw < n_total; ++draw) {
        double zz[DENSE_C ? NS : 1];
#pragma unroll
        for (int b = 0; b < NS / 2; ++b) {              // new_draw = prev_draw + cov_mcmc_chol * rand_vec (:124-126)
            double z0, z1;
            rng_normal_pair_at(prm.seed, chain, draw + prm.draw0, (uint32_t)(4 * b), (uint32_t)j, STREAM_NORMAL, z0, z1);
            const double za = (8u * b + j < d) ? z0 : 0.0;
            const double zb = (8u * b + 4 + j < d) ? z1 : 0.0;
            if constexpr (DENSE_C) {
                zz[2 * b] = za; zz[2 * b + 1] = zb;
            } else if constexpr (GENERAL) {
                tp[2 * b] = th[2 * b] + lds_c[8 * b + j] * za;
                tp[2 * b + 1] = th[2 * b + 1] + lds_c[8 * b + 4 + j] * zb;
            } else {
                tp[2 * b] = th[2 * b] + prm.par_scale * za;
                tp[2 * b + 1] = th[2 * b + 1] + prm.par_scale * zb;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (DENSE_C) {
            double t[NS];
            matvec_m2<NT>(afrag_lc, zz, t);
#pragma unroll
            for (int s = 0; s < NS; ++s) tp[s] = th[s] + t[s];
        }
        double prop_LP = log_kernel(tp);                 // :128
        if (!is_finite(prop_LP)) prop_LP = -INF;         // :130-132
        const double x = prop_LP - prev_LP;
        const double comp_val = (x < 0.0) ? x : 0.0;     // std::min(0.0, x), :136
        const double z = rng_uniform(prm.seed, chain, draw + prm.draw0, 0u);   // :137
        const bool accept = z < det_exp(comp_val);       // :139
        if (accept) {
#pragma unroll
            for (int s = 0; s < NS; ++s) th[s] = tp[s];
            prev_LP = prop_LP;
        }
        if (draw >= prm.n_burnin) {
            n_acc += accept ? 1u : 0u;
            if (prm.draws != nullptr && live) {
                double* out = prm.draws + (size_t)(draw - prm.n_burnin) * d * C;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const uint32_t dim = 4 * s + j;
                    double v = th[s];
                    if constexpr (GENERAL) { if (vb && dim < d) v = box_inv_transform(v, lds_bt[dim], lds_lb[dim], lds_ub[dim]); }   // :157-166
                    if (dim < d) (out + (size_t)(4 * s) * C)[lane_off] = v;
                }
            }
        }
    }
    if (live) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const uint32_t dim = 4 * s + j;
            double v = th[s];
            if constexpr (GENERAL) { if (vb && dim < d) v = box_inv_transform(v, lds_bt[dim], lds_lb[dim], lds_ub[dim]); }
            if (dim < d) prm.theta[(size_t)(4 * s) * C + lane_off] = v;
        }
        if (j == 0 && prm.n_accept) prm.n_accept[cl] = n_acc;
    }
}

}  // namespace mi
