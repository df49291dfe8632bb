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
    double* state;          // workspace: current (beta, grad) of every chain, wave-local layout (see kernel)
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

// CT = chain tiles (of 16 chains) per workgroup: every X fragment fetched from L2 feeds CT MFMAs, which is
// what lifts the kernel off the X stream.  The current state (beta, grad) of the chains lives in a wave-local
// workspace ([workgroup][wave][beta|grad][tile][slice][lane], 8 KiB per chain in total, touched twice per draw);
// registers hold the proposal, its gradient accumulators and the X fragments in flight.
template <int NTQ, int CT>
__global__ __launch_bounds__(256, 1) void mala_logistic_kernel(const MalaLogitParams prm)
{
    constexpr int NSQ = 4 * NTQ, DQ = 16 * NTQ;
    __shared__ double lds_part[2][4][CT][4][64];     // partial eta tiles  [buf][wave][tile][reg][lane]
    __shared__ double lds_rt[2][4][CT][2][64];       // residual / log-lik term of row group q  [buf][q][tile][0/1][lane]
    __shared__ double lds_dot[2 * CT][4][64];        // block dot exchange [which][wave][lane]

    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int j4 = lane >> 4;
    const uint32_t d = prm.d;
    const uint64_t C = prm.C;
    const double eps = prm.eps, s2 = prm.s2, rs = prm.rs;
    const uint32_t NB = prm.NB;
    uint64_t cl[CT], chain[CT];
    bool live[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        cl[c] = ((uint64_t)blockIdx.x * CT + c) * 16 + (lane & 15);
        live[c] = cl[c] < C;
        chain[c] = prm.chain0 + cl[c];
    }
    double* const ws_wave = prm.state + ((size_t)blockIdx.x * 4 + q) * ((size_t)2 * CT * NSQ * 64) + lane;
    auto st = [&](int v, int c, int s) -> double* { return ws_wave + (((size_t)v * CT + c) * NSQ + s) * 64; };

    double bp[CT][NSQ], gp[CT][NSQ];   // proposal and its gradient (this wave's dims)
    double ae[NSQ], ag[4 * NTQ];       // X fragments in flight

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
    // ((S0 + S1) + S2) + S3 of per-wave partial dots (each already butterflied inside the wave), 2*CT values
    auto exchange = [&](double (&v)[2 * CT]) __attribute__((always_inline)) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2 * CT; ++k) lds_dot[k][q][lane] = v[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2 * CT; ++k)
            v[k] = ((lds_dot[k][0][lane] + lds_dot[k][1][lane]) + lds_dot[k][2][lane]) + lds_dot[k][3][lane];
    };

    // value and gradient at x (all CT tiles): lp[c] = log K, gout = gradient on this wave's dims
    auto evaluate = [&](const double (&x)[CT][NSQ], double (&gout)[CT][NSQ], double (&lp)[CT]) __attribute__((always_inline)) {
        double4_t gacc[CT][NTQ];
        double llq[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            llq[c] = 0.0;
#pragma unroll
            for (int t = 0; t < NTQ; ++t) gacc[c][t] = double4_t{0.0, 0.0, 0.0, 0.0};
        }
        load_xe(0);
        load_xg(0);
#pragma unroll 1
        for (uint32_t b = 0; b < NB; ++b) {
            const int buf = (int)(b & 1u);
            double4_t acc[CT];
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[c] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < NSQ; ++s) {
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(ae[s], x[c][s], acc[c], 0, 0, 0);
            }
            if (b + 1 < NB) load_xe(b + 1);
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                lds_part[buf][q][c][0][lane] = acc[c][0]; lds_part[buf][q][c][1][lane] = acc[c][1];
                lds_part[buf][q][c][2][lane] = acc[c][2]; lds_part[buf][q][c][3][lane] = acc[c][3];
            }
            __syncthreads();
            {   // row group q: rows 16b + 4q + j4
                const uint32_t row = 16 * b + 4 * q + j4;
                const double yv = prm.ypad[row];
                const bool valid = row < prm.n_rows;
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    const double eta = ((lds_part[buf][0][c][q][lane] + lds_part[buf][1][c][q][lane]) + lds_part[buf][2][c][q][lane])
                                       + lds_part[buf][3][c][q][lane];
                    // softplus / sigmoid share e = exp(-|eta|) (the oracle evaluates it once per function; same bits)
                    const double e = det_exp(eta > 0.0 ? -eta : eta);
                    const double l1p = det_log(1.0 + e);
                    const double sp = (eta > 0.0) ? (eta + l1p) : l1p;
                    const double sg = (eta >= 0.0) ? (1.0 / (1.0 + e)) : (e / (1.0 + e));
                    lds_rt[buf][q][c][0][lane] = valid ? (yv - sg) : 0.0;
                    lds_rt[buf][q][c][1][lane] = valid ? (yv * eta - sp) : 0.0;
                }
            }
            __syncthreads();
            double res[CT][4];
#pragma unroll
            for (int c = 0; c < CT; ++c) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { res[c][r] = lds_rt[buf][r][c][0][lane]; llq[c] = llq[c] + lds_rt[buf][r][c][1][lane]; }
            }
#pragma unroll
            for (int t = 0; t < NTQ; ++t) {
#pragma unroll
                for (int sp = 0; sp < 4; ++sp) {
#pragma unroll
                    for (int c = 0; c < CT; ++c)
                        gacc[c][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ag[t * 4 + sp], res[c][sp], gacc[c][t], 0, 0, 0);
                }
            }
            if (b + 1 < NB) load_xg(b + 1);
        }
        double v[2 * CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            llq[c] = llq[c] + __shfl_xor(llq[c], 32);
            llq[c] = llq[c] + __shfl_xor(llq[c], 16);
            double nrm = 0.0;
#pragma unroll
            for (int s = 0; s < NSQ; ++s) nrm = dfma(x[c][s], x[c][s], nrm);
            nrm = nrm + __shfl_xor(nrm, 32);
            nrm = nrm + __shfl_xor(nrm, 16);
            v[2 * c] = nrm; v[2 * c + 1] = 0.0;
        }
        exchange(v);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
#pragma unroll
            for (int t = 0; t < NTQ; ++t) {
                gout[c][4 * t + 0] = gacc[c][t][0] - x[c][4 * t + 0];
                gout[c][4 * t + 1] = gacc[c][t][1] - x[c][4 * t + 1];
                gout[c][4 * t + 2] = gacc[c][t][2] - x[c][4 * t + 2];
                gout[c][4 * t + 3] = gacc[c][t][3] - x[c][4 * t + 3];
            }
            lp[c] = llq[c] - 0.5 * v[2 * c];
        }
    };

    double prev_LP[CT], prop_LP[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
#pragma unroll
        for (int s = 0; s < NSQ; ++s) {
            const uint32_t dim = dim_of(s);
            const double v = prm.theta[(size_t)(dim < d ? dim : 0u) * C + (live[c] ? cl[c] : C - 1)];
            bp[c][s] = (dim < d) ? v : 0.0;
        }
    }
    evaluate(bp, gp, prev_LP);                           // box_log_kernel(first_draw), mala.cpp:138
#pragma unroll
    for (int c = 0; c < CT; ++c) {
#pragma unroll
        for (int s = 0; s < NSQ; ++s) { *st(0, c, s) = bp[c][s]; *st(1, c, s) = gp[c][s]; }
    }
    uint64_t n_acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) n_acc[c] = 0;
    const uint32_t n_total = prm.n_burnin + prm.n_keep;

#pragma unroll 1
    for (uint32_t draw = 0; draw < n_total; ++draw) {
        // proposal = mala_mean_fn(prev) + eps * z   (mala.cpp:150,159)
#pragma unroll
        for (int c = 0; c < CT; ++c) {
#pragma unroll
            for (int m = 0; m < NSQ / 2; ++m) {
                double z0, z1;
                const uint32_t slot = (uint32_t)(q * DQ / 2 + 4 * m + j4);
                rng_normal_pair(prm.seed, chain[c], draw, slot, STREAM_NORMAL, z0, z1);
                const double za = (dim_of(2 * m) < d) ? z0 : 0.0;
                const double zb = (dim_of(2 * m + 1) < d) ? z1 : 0.0;
                bp[c][2 * m] = (*st(0, c, 2 * m) + (s2 * *st(1, c, 2 * m)) / 2.0) + eps * za;           // :123, :159
                bp[c][2 * m + 1] = (*st(0, c, 2 * m + 1) + (s2 * *st(1, c, 2 * m + 1)) / 2.0) + eps * zb;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        evaluate(bp, gp, prop_LP);                       // :162
        // mala_prop_adjustment (mala.ipp:59-64)
        double qv[2 * CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            double qa = 0.0, qb = 0.0;
#pragma unroll
            for (int s = 0; s < NSQ; ++s) {
                const double be = *st(0, c, s), gr = *st(1, c, s);
                const double mean_prop = bp[c][s] + (s2 * gp[c][s]) / 2.0;
                const double xa = be - mean_prop;        // dmvnorm.hpp:37
                qa = dfma(xa, rs * xa, qa);
                const double mean_prev = be + (s2 * gr) / 2.0;
                const double xb = bp[c][s] - mean_prev;
                qb = dfma(xb, rs * xb, qb);
            }
            qa = qa + __shfl_xor(qa, 32); qa = qa + __shfl_xor(qa, 16);
            qb = qb + __shfl_xor(qb, 32); qb = qb + __shfl_xor(qb, 16);
            qv[2 * c] = qa; qv[2 * c + 1] = qb;
        }
        exchange(qv);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            double pl = prop_LP[c];
            if (!is_finite(pl)) pl = -INF;               // mala.cpp:164-166
            const double da = prm.cons_term - 0.5 * (prm.log_det + qv[2 * c]);   // dmvnorm.hpp:41
            const double db = prm.cons_term - 0.5 * (prm.log_det + qv[2 * c + 1]);
            const double x = pl - prev_LP[c] + (da - db);
            const double comp_val = (x < 0.01) ? x : 0.01;                       // mala.cpp:170
            const double z = rng_uniform(prm.seed, chain[c], draw, 0u);          // :171
            const bool accept = z < det_exp(comp_val);                           // :173
            if (accept) {
                prev_LP[c] = pl;
                if (live[c]) {
#pragma unroll
                    for (int s = 0; s < NSQ; ++s) { *st(0, c, s) = bp[c][s]; *st(1, c, s) = gp[c][s]; }
                }
            }
            if (draw >= prm.n_burnin) {
                n_acc[c] += accept ? 1u : 0u;
                if (prm.draws != nullptr && live[c]) {
                    double* out = prm.draws + (size_t)(draw - prm.n_burnin) * d * C + cl[c];
#pragma unroll
                    for (int s = 0; s < NSQ; ++s) {
                        const uint32_t dim = dim_of(s);
                        const double v = accept ? bp[c][s] : *st(0, c, s);
                        if (dim < d) out[(size_t)dim * C] = v;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        if (live[c]) {
#pragma unroll
            for (int s = 0; s < NSQ; ++s) {
                const uint32_t dim = dim_of(s);
                const double v = *st(0, c, s);
                if (dim < d) prm.theta[(size_t)dim * C + cl[c]] = v;
            }
            if (q == 0 && j4 == 0 && prm.n_accept) prm.n_accept[cl[c]] = n_acc[c];
        }
    }
}

// HMC on the same target (mcmc::internal::hmc_impl, /root/reference/src/hmc.cpp:155-205, identity preconditioner):
// the fused value+gradient evaluation above is the user callback; one evaluation per leapfrog step (the reference's
// second half-kick of step k and first of step k+1 are at the same theta), the value of the last one is prop_U.
template <int NTQ, int CT>
__global__ __launch_bounds__(256, 1) void hmc_logistic_kernel(const MalaLogitParams prm, const uint32_t n_leap_steps)
{
    constexpr int NSQ = 4 * NTQ, DQ = 16 * NTQ;
    __shared__ double lds_part[2][4][CT][4][64];     // partial eta tiles  [buf][wave][tile][reg][lane]
    __shared__ double lds_rt[2][4][CT][2][64];       // residual / log-lik term of row group q  [buf][q][tile][0/1][lane]
    __shared__ double lds_dot[2 * CT][4][64];        // block dot exchange [which][wave][lane]

    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int j4 = lane >> 4;
    const uint32_t d = prm.d;
    const uint64_t C = prm.C;
    const double eps = prm.eps;
    const uint32_t NB = prm.NB;
    uint64_t cl[CT], chain[CT];
    bool live[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        cl[c] = ((uint64_t)blockIdx.x * CT + c) * 16 + (lane & 15);
        live[c] = cl[c] < C;
        chain[c] = prm.chain0 + cl[c];
    }
    double* const ws_wave = prm.state + ((size_t)blockIdx.x * 4 + q) * ((size_t)2 * CT * NSQ * 64) + lane;
    auto st = [&](int v, int c, int s) -> double* { return ws_wave + (((size_t)v * CT + c) * NSQ + s) * 64; };

    double bp[CT][NSQ], gp[CT][NSQ];   // proposal and its gradient (this wave's dims)
    double ae[NSQ], ag[4 * NTQ];       // X fragments in flight

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
    // ((S0 + S1) + S2) + S3 of per-wave partial dots (each already butterflied inside the wave), 2*CT values
    auto exchange = [&](double (&v)[2 * CT]) __attribute__((always_inline)) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2 * CT; ++k) lds_dot[k][q][lane] = v[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2 * CT; ++k)
            v[k] = ((lds_dot[k][0][lane] + lds_dot[k][1][lane]) + lds_dot[k][2][lane]) + lds_dot[k][3][lane];
    };

    // value and gradient at x (all CT tiles): lp[c] = log K, gout = gradient on this wave's dims
    auto evaluate = [&](const double (&x)[CT][NSQ], double (&gout)[CT][NSQ], double (&lp)[CT]) __attribute__((always_inline)) {
        double4_t gacc[CT][NTQ];
        double llq[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            llq[c] = 0.0;
#pragma unroll
            for (int t = 0; t < NTQ; ++t) gacc[c][t] = double4_t{0.0, 0.0, 0.0, 0.0};
        }
        load_xe(0);
        load_xg(0);
#pragma unroll 1
        for (uint32_t b = 0; b < NB; ++b) {
            const int buf = (int)(b & 1u);
            double4_t acc[CT];
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[c] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < NSQ; ++s) {
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(ae[s], x[c][s], acc[c], 0, 0, 0);
            }
            if (b + 1 < NB) load_xe(b + 1);
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                lds_part[buf][q][c][0][lane] = acc[c][0]; lds_part[buf][q][c][1][lane] = acc[c][1];
                lds_part[buf][q][c][2][lane] = acc[c][2]; lds_part[buf][q][c][3][lane] = acc[c][3];
            }
            __syncthreads();
            {   // row group q: rows 16b + 4q + j4
                const uint32_t row = 16 * b + 4 * q + j4;
                const double yv = prm.ypad[row];
                const bool valid = row < prm.n_rows;
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    const double eta = ((lds_part[buf][0][c][q][lane] + lds_part[buf][1][c][q][lane]) + lds_part[buf][2][c][q][lane])
                                       + lds_part[buf][3][c][q][lane];
                    // softplus / sigmoid share e = exp(-|eta|) (the oracle evaluates it once per function; same bits)
                    const double e = det_exp(eta > 0.0 ? -eta : eta);
                    const double l1p = det_log(1.0 + e);
                    const double sp = (eta > 0.0) ? (eta + l1p) : l1p;
                    const double sg = (eta >= 0.0) ? (1.0 / (1.0 + e)) : (e / (1.0 + e));
                    lds_rt[buf][q][c][0][lane] = valid ? (yv - sg) : 0.0;
                    lds_rt[buf][q][c][1][lane] = valid ? (yv * eta - sp) : 0.0;
                }
            }
            __syncthreads();
            double res[CT][4];
#pragma unroll
            for (int c = 0; c < CT; ++c) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { res[c][r] = lds_rt[buf][r][c][0][lane]; llq[c] = llq[c] + lds_rt[buf][r][c][1][lane]; }
            }
#pragma unroll
            for (int t = 0; t < NTQ; ++t) {
#pragma unroll
                for (int sp = 0; sp < 4; ++sp) {
#pragma unroll
                    for (int c = 0; c < CT; ++c)
                        gacc[c][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ag[t * 4 + sp], res[c][sp], gacc[c][t], 0, 0, 0);
                }
            }
            if (b + 1 < NB) load_xg(b + 1);
        }
        double v[2 * CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            llq[c] = llq[c] + __shfl_xor(llq[c], 32);
            llq[c] = llq[c] + __shfl_xor(llq[c], 16);
            double nrm = 0.0;
#pragma unroll
            for (int s = 0; s < NSQ; ++s) nrm = dfma(x[c][s], x[c][s], nrm);
            nrm = nrm + __shfl_xor(nrm, 32);
            nrm = nrm + __shfl_xor(nrm, 16);
            v[2 * c] = nrm; v[2 * c + 1] = 0.0;
        }
        exchange(v);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
#pragma unroll
            for (int t = 0; t < NTQ; ++t) {
                gout[c][4 * t + 0] = gacc[c][t][0] - x[c][4 * t + 0];
                gout[c][4 * t + 1] = gacc[c][t][1] - x[c][4 * t + 1];
                gout[c][4 * t + 2] = gacc[c][t][2] - x[c][4 * t + 2];
                gout[c][4 * t + 3] = gacc[c][t][3] - x[c][4 * t + 3];
            }
            lp[c] = llq[c] - 0.5 * v[2 * c];
        }
    };

    // bp = position, gp = gradient at bp, pm = momentum (this wave's dims); state slots: accepted (theta, grad)
    double pm[CT][NSQ];
    double prev_U[CT], lp[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
#pragma unroll
        for (int s = 0; s < NSQ; ++s) {
            const uint32_t dim = dim_of(s);
            const double v = prm.theta[(size_t)(dim < d ? dim : 0u) * C + (live[c] ? cl[c] : C - 1)];
            bp[c][s] = (dim < d) ? v : 0.0;
        }
    }
    evaluate(bp, gp, lp);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        prev_U[c] = -lp[c];                              // -box_log_kernel(first_draw), hmc.cpp:140
#pragma unroll
        for (int s = 0; s < NSQ; ++s) { *st(0, c, s) = bp[c][s]; *st(1, c, s) = gp[c][s]; }
    }
    uint64_t n_acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) n_acc[c] = 0;
    const uint32_t n_total = prm.n_burnin + prm.n_keep;
    // K = p . p / 2 over all dimensions: ((S0+S1)+S2)+S3 of the waves' 4-strided partial dots (hmc.cpp:160,184)
    auto kinetic_all = [&](double (&kout)[CT]) __attribute__((always_inline)) {
        double v[2 * CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            double q = 0.0;
#pragma unroll
            for (int s = 0; s < NSQ; ++s) q = dfma(pm[c][s], pm[c][s], q);
            q = q + __shfl_xor(q, 32);
            q = q + __shfl_xor(q, 16);
            v[2 * c] = q; v[2 * c + 1] = 0.0;
        }
        exchange(v);
#pragma unroll
        for (int c = 0; c < CT; ++c) kout[c] = v[2 * c] / 2.0;
    };

#pragma unroll 1
    for (uint32_t draw = 0; draw < n_total; ++draw) {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
#pragma unroll
            for (int m = 0; m < NSQ / 2; ++m) {          // momentum ~ N(0, I), hmc.cpp:156-158
                double z0, z1;
                const uint32_t slot = (uint32_t)(q * DQ / 2 + 4 * m + j4);
                rng_normal_pair(prm.seed, chain[c], draw, slot, STREAM_NORMAL, z0, z1);
                pm[c][2 * m] = (dim_of(2 * m) < d) ? z0 : 0.0;
                pm[c][2 * m + 1] = (dim_of(2 * m + 1) < d) ? z1 : 0.0;
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int s = 0; s < NSQ; ++s) { bp[c][s] = *st(0, c, s); gp[c][s] = *st(1, c, s); }   // new_draw = prev_draw (:162)
        }
        double prev_K[CT], prop_K[CT];
        kinetic_all(prev_K);
#pragma unroll 1
        for (uint32_t k = 0; k < n_leap_steps; ++k) {    // hmc.cpp:164-176
#pragma unroll
            for (int c = 0; c < CT; ++c) {
#pragma unroll
                for (int s = 0; s < NSQ; ++s) {
                    pm[c][s] = pm[c][s] + (eps * gp[c][s]) / 2.0;        // first half-step (:126)
                    bp[c][s] = bp[c][s] + eps * pm[c][s];                // (:171)
                }
            }
            evaluate(bp, gp, lp);
#pragma unroll
            for (int c = 0; c < CT; ++c) {
#pragma unroll
                for (int s = 0; s < NSQ; ++s) pm[c][s] = pm[c][s] + (eps * gp[c][s]) / 2.0;   // second half-step (:175)
            }
        }
        if (n_leap_steps == 0) evaluate(bp, gp, lp);     // the value callback of :178 at the unchanged position
        kinetic_all(prop_K);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            double prop_U = -lp[c];                      // :178
            if (!is_finite(prop_U)) prop_U = INF;        // :180-182
            const double x = -(prop_U + prop_K[c]) + (prev_U[c] + prev_K[c]);
            const double comp_val = (x < 0.01) ? x : 0.01;                       // :188
            const double z = rng_uniform(prm.seed, chain[c], draw, 0u);          // :189
            const bool accept = z < det_exp(comp_val);                           // :191
            if (accept) {
                prev_U[c] = prop_U;
                if (live[c]) {
#pragma unroll
                    for (int s = 0; s < NSQ; ++s) { *st(0, c, s) = bp[c][s]; *st(1, c, s) = gp[c][s]; }
                }
            }
            if (draw >= prm.n_burnin) {
                n_acc[c] += accept ? 1u : 0u;
                if (prm.draws != nullptr && live[c]) {
                    double* out = prm.draws + (size_t)(draw - prm.n_burnin) * d * C + cl[c];
#pragma unroll
                    for (int s = 0; s < NSQ; ++s) {
                        const uint32_t dim = dim_of(s);
                        const double v = accept ? bp[c][s] : *st(0, c, s);
                        if (dim < d) out[(size_t)dim * C] = v;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        if (live[c]) {
#pragma unroll
            for (int s = 0; s < NSQ; ++s) {
                const uint32_t dim = dim_of(s);
                const double v = *st(0, c, s);
                if (dim < d) prm.theta[(size_t)dim * C + cl[c]] = v;
            }
            if (q == 0 && j4 == 0 && prm.n_accept) prm.n_accept[cl[c]] = n_acc[c];
        }
    }
}


}  // namespace mi
