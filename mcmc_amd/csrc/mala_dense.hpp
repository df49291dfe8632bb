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
//
// GENERAL variant: settings.vals_bound (mala.cpp:84-95,105-121,132-134,152-157,193-200 with transform_vals.hpp,
// log_jacobian.hpp, inv_jacobian_adjust.hpp) and / or a DIAGONAL precond_mat M.  Every matrix of the reference's formulas is
// then diagonal (J = inv_jacobian_adjust, J*M, CHOL_LOWER(J), Sigma = eps^2 J M, INV(Sigma)), and Gauss-Jordan / Cholesky /
// the fma-chain products of the oracle's BMO shim applied to a diagonal matrix are the element-wise 1/x, sqrt(x) and a*b:
//   unbounded: mu(v) = v + (eps^2 (M g)) / 2,           proposal = mu + eps (sqrt(M) z),        Sigma = eps^2 M (hoisted)
//   bounded:   mu(v) = v + ((eps^2 (J(v) M)) g) / 2,    proposal = mu + ((eps (sqrt(J) sqrt(M))) z),
//              Sigma = eps^2 (J(proposal) M) in BOTH dmvnorm terms (mala.ipp:52-53), LOG_DET summed over dimensions in order,
//              log kernel = K(inv_transform(v)) + log_jacobian(v), draws reported through inv_transform.
// Identity bounds / unit M reproduce the plain kernel's bits.
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
    uint32_t draw0;         // index of this call's first draw in the chains' random streams (mi_chains.draw0)
    double s2;              // eps*eps (mala.cpp:123, mala.ipp:41)
    double rs;              // 1.0 / s2: diagonal of INV(eps^2 I)
    double cons_term;       // -0.5 * d * log(2 pi)   (dmvnorm.hpp:36)
    double log_det;         // LOG_DET(eps^2 M), hoisted (unbounded runs)
    // general variant (GENERAL = true): settings.vals_bound and / or a diagonal precond_mat M
    int vals_bound;
    const int* btype;       // [d] determine_bounds_type (1 none, 2 lower, 3 upper, 4 both)
    const double* lb;
    const double* ub;
    const double* m;        // [d] diagonal of precond_mat
    const double* m_sqrt;   // [d] diagonal of CHOL_LOWER(precond_mat)
    // dense precond_mat (mala_gauss_dense_m_kernel): d*d row-major device matrices from the host; sep_target: P is the diagonal matrix of
    // an ISO / DIAG target, whose gradient the reference's target function takes ELEMENT-WISE (no 0 * inf from the other dimensions)
    uint32_t sep_target;
    const double* Mfull;    // precond_mat
    const double* Lchol;    // CHOL_LOWER(precond_mat)
    const double* Sinv;     // INV(eps^2 precond_mat)
    uint32_t* nf_flag;      // [C + 1] or nullptr: chains that reached the non-finite regime are flagged and left to literal.hpp
};

template <int NT, bool GENERAL = false>
__global__ MI_NO_DS_MERGE __launch_bounds__(256, 1) void mala_gauss_mfma_kernel(const MalaParams prm)
{
    constexpr int NS = 4 * NT;
    extern __shared__ __attribute__((aligned(16))) double lds_P[];
    // GENERAL: per-dimension bounds / preconditioner tables behind the precision fragments
    double* const lds_lb = lds_P + (size_t)NT * 4 * NT * 64;
    double* const lds_ub = lds_lb + 16 * NT;
    double* const lds_m = lds_ub + 16 * NT;
    double* const lds_ms = lds_m + 16 * NT;
    int* const lds_bt = reinterpret_cast<int*>(lds_ms + 16 * NT);
    if constexpr (GENERAL) {
        for (int i = threadIdx.x; i < 16 * NT; i += blockDim.x) {
            const bool in = (uint32_t)i < prm.d;
            lds_lb[i] = in ? prm.lb[i] : 0.0;
            lds_ub[i] = in ? prm.ub[i] : 0.0;
            lds_bt[i] = in ? prm.btype[i] : 1;
            lds_m[i] = in ? prm.m[i] : 1.0;
            lds_ms[i] = in ? prm.m_sqrt[i] : 1.0;
        }
    }
    stage_precision<NT>(prm.P, prm.d, lds_P);           // ends with a barrier

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

    double th[NS], w[NS];        // current state (transformed space when bounded) and P*x(theta) (grad = -w)
    double tp[NS], wp[NS];       // proposal and P*x(theta')
    double xp[GENERAL ? NS : 1]; // GENERAL: x(theta') = inv_transform(theta')
    const bool vb = GENERAL && prm.vals_bound != 0;

    int tab_o = 0;                 // (an opaque zero, renewed at the top of every draw: see there)
    // mala_mean_fn (mala.cpp:97-125) for one dimension; jm_out = eps^2 (J M) when bounded (= Sigma_ii with J = J(v))
    auto mean_of = [&](double v, double pw, int dim, double& jm_out) __attribute__((always_inline)) -> double {
        if constexpr (GENERAL) {
            const double M = lds_m[dim + tab_o];
            if (vb) {
                const double J = box_inv_jacobian(v, lds_bt[dim], lds_lb[dim], lds_ub[dim]);   // :113
                const double JM = s2 * (J * M);                                                // :115-117
                jm_out = JM;
                return v + (JM * (-pw)) / 2.0;                                                 // :121
            }
            jm_out = s2 * M;
            return v + (s2 * (M * (-pw))) / 2.0;                                               // :123
        } else {
            jm_out = s2;
            return v - (s2 * pw) / 2.0;
        }
    };
    // box_log_kernel at the state whose x / P x are (xx, ww) (mala.cpp:84-95): K(x) [+ log_jacobian(theta), i ascending]
    auto log_kernel = [&](const double (&tt_)[NS], const double (&xx)[GENERAL ? NS : 1], const double (&ww)[NS]) __attribute__((always_inline)) -> double {
        if constexpr (GENERAL) {
            const double kval = -0.5 * dot4<NS>(xx, ww);
            if (!vb) return kval;
            double lj = 0.0;                             // log_jacobian.hpp:36-57: scalar loop
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int i0 = 4 * s;
                const double term = box_log_jacobian_term(tt_[s], lds_bt[i0 + j], lds_lb[i0 + j], lds_ub[i0 + j]);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const double tg = __shfl(term, (lane & 15) + 16 * g);
                    if ((uint32_t)(i0 + g) < d && lds_bt[i0 + g] != 1) lj = lj + tg;
                }
            }
            return kval + lj;
        } else {
            return -0.5 * dot4<NS>(tt_, ww);
        }
    };

#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const uint32_t dim = 4 * s + j;
        const double v = (dim < d) ? prm.theta[(size_t)(4 * s) * C + lane_off] : 0.0;
        if constexpr (GENERAL) th[s] = (vb && dim < d) ? box_transform(v, lds_bt[dim], lds_lb[dim], lds_ub[dim]) : v;   // mala.cpp:132-134
        else th[s] = v;
    }
    if constexpr (GENERAL) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int dim = 4 * s + j;
            xp[s] = vb ? (((uint32_t)dim < d) ? box_inv_transform(th[s], lds_bt[dim], lds_lb[dim], lds_ub[dim]) : 0.0) : th[s];
        }
        matvec_mfma<NT>(afrag, xp, w);
    } else {
        matvec_mfma<NT>(afrag, th, w);
    }
    double prev_LP = log_kernel(th, xp, w);             // box_log_kernel(first_draw), mala.cpp:138
    uint64_t n_acc = 0;
    const uint32_t n_total = prm.n_burnin + prm.n_keep;
    // Non-finite regime: `precond_matrix * grad_obj`, INV(Sigma) * (x - mu), and, bounded, `inv_jacob *
    // precond_matrix`, CHOL_LOWER(J), INV / LOG_DET of eps^2 J M are dense operations in the reference; with every matrix diagonal
    // they are the element-wise arithmetic below only while every value is finite.  A non-finite entry of theta, theta', either
    // gradient or either Jacobian makes one of the two proposal densities non-finite: the chain is then flagged and replayed
    // literally (literal.hpp) instead of finished here.
    bool nf_seen = false;

#pragma unroll 1
    for (uint32_t draw = 0; draw < n_total; ++draw) {
        // (the per-dimension tables are read at an index that is opaque per draw: s2 * M, 1 / (s2 * M), sqrt(m) of every slice are
        //  loop invariants otherwise -- 3 x NS values the kernel has no registers for; they were spilled and reloaded from scratch)
        tab_o = 0;
        asm volatile("" : "+v"(tab_o));
        // proposal: mala_mean_fn(prev) + eps * L z   (mala.cpp:150-159)
#pragma unroll
        for (int b = 0; b < NS / 2; ++b) {
            double z0, z1;
            rng_normal_pair_at(prm.seed, chain, draw + prm.draw0, (uint32_t)(4 * b), (uint32_t)j, STREAM_NORMAL, z0, z1);
            const double za = (8u * b + j < d) ? z0 : 0.0;
            const double zb = (8u * b + 4 + j < d) ? z1 : 0.0;
            double jma, jmb;
            const double ma = mean_of(th[2 * b], w[2 * b], 8 * b + j, jma);
            const double mb = mean_of(th[2 * b + 1], w[2 * b + 1], 8 * b + 4 + j, jmb);
            if constexpr (GENERAL) {
                const double sa = lds_ms[8 * b + j + tab_o], sb = lds_ms[8 * b + 4 + j + tab_o];
                if (vb) {                                // CHOL_LOWER(J) * sqrt_precond, scaled by eps (mala.cpp:155-157)
                    const double Ja = box_inv_jacobian(th[2 * b], lds_bt[8 * b + j], lds_lb[8 * b + j], lds_ub[8 * b + j]);
                    const double Jb = box_inv_jacobian(th[2 * b + 1], lds_bt[8 * b + 4 + j], lds_lb[8 * b + 4 + j], lds_ub[8 * b + 4 + j]);
                    tp[2 * b] = ma + (eps * (__builtin_sqrt(Ja) * sa)) * za;
                    tp[2 * b + 1] = mb + (eps * (__builtin_sqrt(Jb) * sb)) * zb;
                } else {
                    tp[2 * b] = ma + eps * (sa * za);    // :159
                    tp[2 * b + 1] = mb + eps * (sb * zb);
                }
            } else {
                tp[2 * b] = ma + eps * za;
                tp[2 * b + 1] = mb + eps * zb;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (GENERAL) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int dim = 4 * s + j;
                xp[s] = vb ? (((uint32_t)dim < d) ? box_inv_transform(tp[s], lds_bt[dim], lds_lb[dim], lds_ub[dim]) : 0.0) : tp[s];
            }
            matvec_mfma<NT>(afrag, xp, wp);
        } else {
            matvec_mfma<NT>(afrag, tp, wp);
        }
        double prop_LP = log_kernel(tp, xp, wp);         // :162
        if (!is_finite(prop_LP)) prop_LP = -INF;         // :164-166
        // mala_prop_adjustment (mala.ipp:45-65): dmvnorm(prev | mu(prop), Sigma) - dmvnorm(prop | mu(prev), Sigma),
        // Sigma = eps^2 [J(prop)] M in both terms; quadratic forms in dot4 order, accumulated on the fly
        double qa = 0.0, qb = 0.0, ld_term[GENERAL ? NS : 1];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int dim = 4 * s + j;
            double sig_p, sig_c;
            const double mean_prop = mean_of(tp[s], wp[s], dim, sig_p);
            const double mean_prev = mean_of(th[s], w[s], dim, sig_c);
            const double sinv = GENERAL ? 1.0 / sig_p : rs;              // INV(Sigma)_ii
            const double xa = th[s] - mean_prop;                          // dmvnorm.hpp:37
            qa = dfma(xa, sinv * xa, qa);                                 // :39
            const double xb = tp[s] - mean_prev;
            qb = dfma(xb, sinv * xb, qb);
            if constexpr (GENERAL) { if (vb) ld_term[s] = 2.0 * det_log(__builtin_sqrt(sig_p)); }   // LOG_DET via CHOL_LOWER, term i (Sigma is the host's constant otherwise)
        }
        qa = qa + __shfl_xor(qa, 32); qa = qa + __shfl_xor(qa, 16);
        qb = qb + __shfl_xor(qb, 32); qb = qb + __shfl_xor(qb, 16);
        double log_det = prm.log_det;
        if constexpr (GENERAL) {
            if (vb) {                                    // Sigma depends on the proposal: sum_i 2 log L_ii, i ascending
                log_det = 0.0;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const double tg = __shfl(ld_term[s], (lane & 15) + 16 * g);
                        if ((uint32_t)(4 * s + g) < d) log_det = log_det + tg;
                    }
                }
            }
        }
        const double da = prm.cons_term - 0.5 * (log_det + qa);          // :41
        const double db = prm.cons_term - 0.5 * (log_det + qb);
        nf_seen = nf_seen || !is_finite(da) || !is_finite(db);
        const double x = prop_LP - prev_LP + (da - db);
        const double comp_val = (x < 0.01) ? x : 0.01;   // std::min(0.01, x), mala.cpp:170
        const double z = rng_uniform(prm.seed, chain, draw + prm.draw0, 0u);          // :171
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
                    double v = th[s];
                    if constexpr (GENERAL) { if (vb && dim < d) v = box_inv_transform(v, lds_bt[dim], lds_lb[dim], lds_ub[dim]); }   // :193-200
                    if (dim < d) (out + (size_t)(4 * s) * C)[lane_off] = v;
                }
            }
        }
    }
    const bool replay = nf_seen && prm.nf_flag != nullptr;
    if (live && replay && j == 0) { prm.nf_flag[cl] = 1u; prm.nf_flag[C] = 1u; }
    if (live && !replay) {
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

// MALA with a DENSE precond_mat M, unbounded (mala.cpp:123,159; mala.ipp:60-64): four fragment sets share the LDS (d <= 64; beyond, three of them stay in L2)
// (P, M, L = CHOL_LOWER(M), INV(eps^2 M); the last two and LOG_DET(eps^2 M) come from the host in the oracle's operation order).
//   mu(v) = v + (eps^2 (M g)) / 2,   proposal = mu + eps (L z),   dmvnorm terms with INV(Sigma) as mat-vecs:
// five mat-vecs per draw (P x', M g', L z, Sinv xa, Sinv xb); M g of the current state is carried.  A bounded run would need
// INV(eps^2 J(theta') M) per draw and is refused on the host.
template <int NT>
__global__ MI_NO_DS_MERGE __launch_bounds__(256, 1) void mala_gauss_dense_m_kernel(const MalaParams prm)
{
    constexpr int NS = 4 * NT;
    constexpr int MAT = NT * NS * 64;
    extern __shared__ __attribute__((aligned(16))) double lds_P[];
    constexpr bool MG = dense_m_from_global<NT>();     // d > 64: M, L, INV(Sigma) are read from L2 in fragment order (hmc_dense.hpp)
    if constexpr (!MG) {
        stage_precision<NT>(prm.Mfull, prm.d, lds_P + MAT);
        stage_precision<NT>(prm.Lchol, prm.d, lds_P + 2 * MAT);
        stage_precision<NT>(prm.Sinv, prm.d, lds_P + 3 * MAT);
    }
    stage_precision<NT>(prm.P, prm.d, lds_P);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane >> 4;
    const uint64_t cl = ((uint64_t)blockIdx.x * 4 + wave) * 16 + (lane & 15);
    const bool live = cl < prm.C;
    const uint64_t cld = live ? cl : prm.C - 1;
    const uint64_t chain = prm.chain0 + cl;
    const uint32_t d = prm.d;
    const uint64_t C = prm.C;
    const double eps = prm.eps, s2 = prm.s2;
    const double* afrag = lds_P + lane;
    const double* afrag_m = MG ? prm.Mfull + lane : lds_P + MAT + lane;
    const double* afrag_l = MG ? prm.Lchol + lane : lds_P + 2 * MAT + lane;
    const double* afrag_si = MG ? prm.Sinv + lane : lds_P + 3 * MAT + lane;
    const size_t lane_off = (size_t)j * C + cld;

    double th[NS], w[NS], mg[NS];      // current state, P theta, M grad (grad = -w)
    double tp[NS], wp[NS], mgp[NS];    // the same at the proposal
    double a[NS], b[NS];
    // The padding dimensions (d is not a multiple of 16) have zero rows and columns in every staged matrix.  While the state is finite they stay
    // 0; once a real dimension is +-inf a padded ROW of a product is 0 * inf = NaN, and the next dense product spreads it over every real
    // dimension through the (zero) padded column -- the reference has no such dimensions (found by the round-5 fuzz: d = 31, a chain started at
    // -inf, draws NaN where the oracle has +-inf).  So every product's padded entries are put back to 0.
    auto clear_pad = [&](double (&x)[NS]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < NS; ++s) x[s] = ((uint32_t)(4 * s + j) < d) ? x[s] : 0.0;
    };
    // P x: the dense target's mat-vec; an ISO / DIAG target multiplies element-wise (oracle: orc_target_kernel; the user's target function of the
    // reference does): the bits of the mat-vec over the expanded diagonal while everything is finite, and +-inf stays in its own dimension
    auto p_times = [&](const double (&x)[NS], double (&out)[NS]) __attribute__((always_inline)) {
        if (prm.sep_target) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const uint32_t dim = 4 * s + j;
                out[s] = (dim < d) ? prm.P[(size_t)dim * (d + 1)] * x[s] : 0.0;
            }
        } else {
            matvec_mfma<NT>(afrag, x, out);
            clear_pad(out);
        }
    };
    auto m_times_grad = [&](const double (&ww)[NS], double (&out)[NS]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < NS; ++s) a[s] = -ww[s];
        matvec_m2<NT>(afrag_m, a, out);                // precond_matrix * grad_obj (mala.cpp:123)
        clear_pad(out);
    };
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const uint32_t dim = 4 * s + j;
        th[s] = (dim < d) ? prm.theta[(size_t)(4 * s) * C + lane_off] : 0.0;
    }
    p_times(th, w);
    m_times_grad(w, mg);
    double prev_LP = -0.5 * dot4<NS>(th, w);            // mala.cpp:138
    uint64_t n_acc = 0;
    const uint32_t n_total = prm.n_burnin + prm.n_keep;

#pragma unroll 1
    for (uint32_t draw = 0; draw < n_total; ++draw) {
#pragma unroll
        for (int bb = 0; bb < NS / 2; ++bb) {           // rand_vec (:150)
            double z0, z1;
            rng_normal_pair_at(prm.seed, chain, draw + prm.draw0, (uint32_t)(4 * bb), (uint32_t)j, STREAM_NORMAL, z0, z1);
            a[2 * bb] = (8u * bb + j < d) ? z0 : 0.0;
            a[2 * bb + 1] = (8u * bb + 4 + j < d) ? z1 : 0.0;
            __builtin_amdgcn_sched_barrier(0);
        }
        matvec_m2<NT>(afrag_l, a, b);                  // sqrt_precond_matrix * rand_vec (:159)
        clear_pad(b);
#pragma unroll
        for (int s = 0; s < NS; ++s) tp[s] = (th[s] + (s2 * mg[s]) / 2.0) + eps * b[s];   // :123, :159
        clear_pad(tp);                                 // (s2 = inf: inf * 0 in the padding)
        p_times(tp, wp);
        double prop_LP = -0.5 * dot4<NS>(tp, wp);        // :162
        if (!is_finite(prop_LP)) prop_LP = -INF;         // :164-166
        m_times_grad(wp, mgp);
        // mala_prop_adjustment (mala.ipp:60-64): dmvnorm(prev | mu(prop), Sigma) - dmvnorm(prop | mu(prev), Sigma)
#pragma unroll
        for (int s = 0; s < NS; ++s) a[s] = th[s] - (tp[s] + (s2 * mgp[s]) / 2.0);       // X - mu (dmvnorm.hpp:37)
        clear_pad(a);
        matvec_m2<NT>(afrag_si, a, b);
        clear_pad(b);
        const double quad_a = dot4<NS>(a, b);            // :39
#pragma unroll
        for (int s = 0; s < NS; ++s) a[s] = tp[s] - (th[s] + (s2 * mg[s]) / 2.0);
        clear_pad(a);
        matvec_m2<NT>(afrag_si, a, b);
        clear_pad(b);
        const double quad_b = dot4<NS>(a, b);
        const double da = prm.cons_term - 0.5 * (prm.log_det + quad_a);   // :41
        const double db = prm.cons_term - 0.5 * (prm.log_det + quad_b);
        const double x = prop_LP - prev_LP + (da - db);
        const double comp_val = (x < 0.01) ? x : 0.01;   // mala.cpp:170
        const double z = rng_uniform(prm.seed, chain, draw + prm.draw0, 0u);   // :171
        const bool accept = z < det_exp(comp_val);       // :173
        if (accept) {
#pragma unroll
            for (int s = 0; s < NS; ++s) { th[s] = tp[s]; w[s] = wp[s]; mg[s] = mgp[s]; }
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
