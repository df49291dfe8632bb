// mala_logistic.hpp -- many-chain MALA for the Bayesian logistic-regression target
//   log K(beta) = sum_r [ y_r eta_r - log(1 + e^eta_r) ] - 1/2 |beta|^2,   eta = X beta,
//   grad        = X^T (y - sigmoid(eta)) - beta
// (BASELINE config 3: d = 512, N = 1024 rows, 262 144 chains) on the fp64 matrix cores.
//
// Replaces the draw loop of mcmc::internal::mala_impl (/root/reference/src/mala.cpp:149-186) with
// mala_mean_fn (:97-125), mala_prop_adjustment (/root/reference/include/mcmc/mala.ipp:30-70) and
// stats_mcmc::dmvnorm (/root/reference/include/stats/dmvnorm.hpp:28-54), identity preconditioner.
// The reference makes 3 gradient + 1 value callbacks per draw (mu(theta), K(theta'), mu(theta'),
// mu(theta) again); they are deterministic, so ONE fused value+gradient evaluation at the proposal per
// draw, with the current state's gradient cached, reproduces every one of them bit for bit.
//
// Mapping: a workgroup of 4 waves owns 16 chains; wave q owns the dimension block
// [q*dq, (q+1)*dq), dq = 16*NTQ (d = 512 -> 128 dims per wave), in the same MFMA B/D register layout
// as hmc_dense.hpp, so beta / grad / proposal of the block are 3 x 2*NSQ VGPRs.  X is streamed from
// L2 in two fragment-packed copies (built once per call): XE feeds eta = X beta (A = X rows),
// XG feeds X^T r (A = X columns).  Per block of 16 rows:
//   partial eta tile (NSQ MFMAs, this wave's dims) -> LDS -> barrier -> wave q finishes row group q
//   (sum of the four partials, softplus / sigmoid ONCE per row and chain) -> LDS -> barrier ->
//   every wave accumulates its X^T r tiles (4*NTQ MFMAs); the D layout of the eta tile is the B
//   layout of the residual slices, so no shuffles.  Next block's fragments are fetched while the other
//   phase's MFMAs run.
// Reduction orders (the oracle states the same): eta_r = ((e0+e1)+e2)+e3 with e_q the fma chain over
// wave q's dims; X^T r rows ascending as one fma chain; the row sum of the log-likelihood 4-strided
// + butterfly; dot products over dimensions ((S0+S1)+S2)+S3 with S_q the 4-strided dot of block q.
#pragma once

#include "hmc_dense.hpp"

#ifndef MI_LOGIT_EXTRA_NOPS
#define MI_LOGIT_EXTRA_NOPS 0
#endif
#ifndef MI_LOGIT_PREFETCH
#define MI_LOGIT_PREFETCH 1
#endif

namespace mi {

struct MalaLogitParams {
    const double* XE;       // [NB][4][NSQ][64]  eta fragments
    const double* XG;       // [NB][4][NTQ][4][64]  gradient fragments
    const double* ypad;     // [16*NB] labels, zero padded
    uint32_t d, n_rows, NB;
    uint64_t C, chain0;
    double* theta;          // [d][C] in/out
    double* draws;
    uint64_t* n_accept;
    uint64_t seed;
    uint32_t n_burnin, n_keep;
    double eps, s2, rs, cons_term, log_det;
};

// pack X (row-major n_rows x d) into the two fragment orders; zero padding outside
template <int NTQ>
__global__ void pack_logistic_kernel(const double* __restrict__ X, const double* __restrict__ y, uint32_t d,
                                     uint32_t n_rows, uint32_t NB, double* XE, double* XG, double* ypad)
{
    constexpr int NSQ = 4 * NTQ, DQ = 16 * NTQ;
    const uint32_t b = blockIdx.x;
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    for (int s = 0; s < NSQ; ++s) {
        const uint32_t row = 16 * b + (lane & 15), col = q * DQ + 4 * s + (lane >> 4);
        XE[(((size_t)b * 4 + q) * NSQ + s) * 64 + lane] = (row < n_rows && col < d) ? X[(size_t)row * d + col] : 0.0;
    }
    for (int t = 0; t < NTQ; ++t)
        for (int sp = 0; sp < 4; ++sp) {
            const uint32_t row = 16 * b + 4 * sp + (lane >> 4), col = q * DQ + 16 * t + (lane & 15);
            XG[((((size_t)b * 4 + q) * NTQ + t) * 4 + sp) * 64 + lane] = (row < n_rows && col < d) ? X[(size_t)row * d + col] : 0.0;
        }
    if (threadIdx.x < 16) {
        const uint32_t row = 16 * b + threadIdx.x;
        ypad[row] = row < n_rows ? y[row] : 0.0;
    }
}

template <int NTQ>
__global__ __launch_bounds__(256, 1) void mala_logistic_kernel(const MalaLogitParams prm)
{
    constexpr int NSQ = 4 * NTQ, DQ = 16 * NTQ;
    __shared__ double lds_part[2][4][4][64];     // partial eta tiles  [buf][wave][reg][lane]
    __shared__ double lds_rt[2][4][2][64];       // residual / log-lik term of row group q  [buf][q][0/1][lane]
    __shared__ double lds_dot[4][4][64];         // block dot exchange [which][wave][lane]

    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int j4 = lane >> 4;
    const uint64_t cl = (uint64_t)blockIdx.x * 16 + (lane & 15);
    const bool live = cl < prm.C;
    const uint64_t cld = live ? cl : prm.C - 1;
    const uint64_t chain = prm.chain0 + cl;
    const uint32_t d = prm.d;
    const uint64_t C = prm.C;
    const double eps = prm.eps, s2 = prm.s2, rs = prm.rs;
    const uint32_t NB = prm.NB;

    double be[NSQ], gr[NSQ];       // current state and its gradient (this wave's dims)
    double bp[NSQ], gp[NSQ];       // proposal and its gradient
    double ae[NSQ], ag[4 * NTQ];   // X fragments in flight

    auto dim_of = [&](int s) -> uint32_t { return (uint32_t)(q * DQ + 4 * s + j4); };
    auto load_xe = [&](uint32_t b) __attribute__((always_inline)) {
        const double* src = prm.XE + (((size_t)b * 4 + q) * NSQ) * 64 + lane;
#pragma unroll
        for (int s = 0; s < NSQ; ++s) ae[s] = src[(size_t)s * 64];
    };
    auto load_xg = [&](uint32_t b) __attribute__((always_inline)) {
        const double* src = prm.XG + (((size_t)b * 4 + q) * NTQ * 4) * 64 + lane;
#pragma unroll
        for (int k = 0; k < 4 * NTQ; ++k) ag[k] = src[(size_t)k * 64];
    };
    // ((S0 + S1) + S2) + S3 of up to 3 per-wave partial dots (each already butterflied inside the wave)
    auto exchange3 = [&](double& a, double& b2, double& c) __attribute__((always_inline)) {
        __syncthreads();
        lds_dot[0][q][lane] = a; lds_dot[1][q][lane] = b2; lds_dot[2][q][lane] = c;
        __syncthreads();
        a = ((lds_dot[0][0][lane] + lds_dot[0][1][lane]) + lds_dot[0][2][lane]) + lds_dot[0][3][lane];
        b2 = ((lds_dot[1][0][lane] + lds_dot[1][1][lane]) + lds_dot[1][2][lane]) + lds_dot[1][3][lane];
        c = ((lds_dot[2][0][lane] + lds_dot[2][1][lane]) + lds_dot[2][2][lane]) + lds_dot[2][3][lane];
    };

    // value and gradient at x: returns log K(x), fills gout (gradient on this wave's dims)
    auto evaluate = [&](const double (&x)[NSQ], double (&gout)[NSQ]) __attribute__((always_inline)) -> double {
        double4_t gacc[NTQ];
#pragma unroll
        for (int t = 0; t < NTQ; ++t) gacc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
        double llq = 0.0;
        load_xe(0);
        load_xg(0);
#pragma unroll 1
        for (uint32_t b = 0; b < NB; ++b) {
            const int buf = (int)(b & 1u);
            if (!MI_LOGIT_PREFETCH && b > 0) { load_xe(b); load_xg(b); }
            double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < NSQ; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ae[s], x[s], acc, 0, 0, 0);
            if (MI_LOGIT_PREFETCH && b + 1 < NB) load_xe(b + 1);
#if MI_LOGIT_EXTRA_NOPS
            asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
#endif
            lds_part[buf][q][0][lane] = acc[0]; lds_part[buf][q][1][lane] = acc[1];
            lds_part[buf][q][2][lane] = acc[2]; lds_part[buf][q][3][lane] = acc[3];
            __syncthreads();
            {   // row group q: rows 16b + 4q + j4
                const double eta = ((lds_part[buf][0][q][lane] + lds_part[buf][1][q][lane]) + lds_part[buf][2][q][lane])
                                   + lds_part[buf][3][q][lane];
                const uint32_t row = 16 * b + 4 * q + j4;
                const double yv = prm.ypad[row];
                const bool valid = row < prm.n_rows;
                // softplus / sigmoid share e = exp(-|eta|) (the oracle evaluates it once per function; same bits)
                const double e = det_exp(eta > 0.0 ? -eta : eta);
                const double l1p = det_log(1.0 + e);
                const double sp = (eta > 0.0) ? (eta + l1p) : l1p;
                const double sg = (eta >= 0.0) ? (1.0 / (1.0 + e)) : (e / (1.0 + e));
                lds_rt[buf][q][0][lane] = valid ? (yv - sg) : 0.0;
                lds_rt[buf][q][1][lane] = valid ? (yv * eta - sp) : 0.0;
            }
            __syncthreads();
            double res[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { res[r] = lds_rt[buf][r][0][lane]; llq = llq + lds_rt[buf][r][1][lane]; }
#pragma unroll
            for (int t = 0; t < NTQ; ++t) {
#pragma unroll
                for (int sp = 0; sp < 4; ++sp)
                    gacc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ag[t * 4 + sp], res[sp], gacc[t], 0, 0, 0);
            }
            if (MI_LOGIT_PREFETCH && b + 1 < NB) load_xg(b + 1);
        }
        llq = llq + __shfl_xor(llq, 32);
        llq = llq + __shfl_xor(llq, 16);
        double nrm = 0.0;
#pragma unroll
        for (int s = 0; s < NSQ; ++s) nrm = dfma(x[s], x[s], nrm);
        nrm = nrm + __shfl_xor(nrm, 32);
        nrm = nrm + __shfl_xor(nrm, 16);
        double u1 = 0.0, u2 = 0.0;
        exchange3(nrm, u1, u2);
#pragma unroll
        for (int t = 0; t < NTQ; ++t) {
            gout[4 * t + 0] = gacc[t][0] - x[4 * t + 0];
            gout[4 * t + 1] = gacc[t][1] - x[4 * t + 1];
            gout[4 * t + 2] = gacc[t][2] - x[4 * t + 2];
            gout[4 * t + 3] = gacc[t][3] - x[4 * t + 3];
        }
        return llq - 0.5 * nrm;
    };

#pragma unroll
    for (int s = 0; s < NSQ; ++s) {
        const uint32_t dim = dim_of(s);
        const double v = prm.theta[(size_t)(dim < d ? dim : 0u) * C + cld];
        be[s] = (dim < d) ? v : 0.0;
    }
    double prev_LP = evaluate(be, gr);                   // box_log_kernel(first_draw), mala.cpp:138
    uint64_t n_acc = 0;
    const uint32_t n_total = prm.n_burnin + prm.n_keep;

#pragma unroll 1
    for (uint32_t draw = 0; draw < n_total; ++draw) {
        // proposal = mala_mean_fn(prev) + eps * z   (mala.cpp:150,159)
#pragma unroll
        for (int m = 0; m < NSQ / 2; ++m) {
            double z0, z1;
            const uint32_t slot = (uint32_t)(q * DQ / 2 + 4 * m + j4);
            rng_normal_pair(prm.seed, chain, draw, slot, STREAM_NORMAL, z0, z1);
            const double za = (dim_of(2 * m) < d) ? z0 : 0.0;
            const double zb = (dim_of(2 * m + 1) < d) ? z1 : 0.0;
            bp[2 * m] = (be[2 * m] + (s2 * gr[2 * m]) / 2.0) + eps * za;           // :123, :159
            bp[2 * m + 1] = (be[2 * m + 1] + (s2 * gr[2 * m + 1]) / 2.0) + eps * zb;
            __builtin_amdgcn_sched_barrier(0);
        }
        double prop_LP = evaluate(bp, gp);               // :162
        if (!is_finite(prop_LP)) prop_LP = -INF;         // :164-166
        // mala_prop_adjustment (mala.ipp:59-64)
        double qa = 0.0, qb = 0.0, dummy = 0.0;
#pragma unroll
        for (int s = 0; s < NSQ; ++s) {
            const double mean_prop = bp[s] + (s2 * gp[s]) / 2.0;
            const double xa = be[s] - mean_prop;         // dmvnorm.hpp:37
            qa = dfma(xa, rs * xa, qa);
            const double mean_prev = be[s] + (s2 * gr[s]) / 2.0;
            const double xb = bp[s] - mean_prev;
            qb = dfma(xb, rs * xb, qb);
        }
        qa = qa + __shfl_xor(qa, 32); qa = qa + __shfl_xor(qa, 16);
        qb = qb + __shfl_xor(qb, 32); qb = qb + __shfl_xor(qb, 16);
        exchange3(qa, qb, dummy);
        const double da = prm.cons_term - 0.5 * (prm.log_det + qa);               // dmvnorm.hpp:41
        const double db = prm.cons_term - 0.5 * (prm.log_det + qb);
        const double x = prop_LP - prev_LP + (da - db);
        const double comp_val = (x < 0.01) ? x : 0.01;   // mala.cpp:170
        const double z = rng_uniform(prm.seed, chain, draw, 0u);                  // :171
        const bool accept = z < det_exp(comp_val);       // :173
        if (accept) {
#pragma unroll
            for (int s = 0; s < NSQ; ++s) { be[s] = bp[s]; gr[s] = gp[s]; }
            prev_LP = prop_LP;
        }
        if (draw >= prm.n_burnin) {
            n_acc += accept ? 1u : 0u;
            if (prm.draws != nullptr && live) {
                double* out = prm.draws + (size_t)(draw - prm.n_burnin) * d * C + cl;
#pragma unroll
                for (int s = 0; s < NSQ; ++s) {
                    const uint32_t dim = dim_of(s);
                    if (dim < d) out[(size_t)dim * C] = be[s];
                }
            }
        }
    }
    if (live) {
#pragma unroll
        for (int s = 0; s < NSQ; ++s) {
            const uint32_t dim = dim_of(s);
            if (dim < d) prm.theta[(size_t)dim * C + cl] = be[s];
        }
        if (q == 0 && j4 == 0 && prm.n_accept) prm.n_accept[cl] = n_acc;
    }
}

}  // namespace mi
