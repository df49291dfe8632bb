// callback_mode.hpp -- mcmc::hmc with a HOST std::function target (the reference's own contract,
// /root/reference/include/mcmc/hmc.hpp:42-48) for one chain.
//
// The user's log-kernel callback can only run on the host, so here the host drives the draw loop of
// /root/reference/src/hmc.cpp:155-205 and calls the callback exactly where the reference does
// (mntm_update_fn :110/:124 twice per leapfrog step, box_log_kernel :178 once per draw), while
// every arithmetic stage of the sampler -- momentum draw (Philox), half-kick, drift, kinetic energy,
// Metropolis test, row store -- runs on the GPU on device-resident state.  It is the plumbing path
// of BASELINE config[0]; throughput lives in the fused kernels.
#pragma once

#include "det_math.hpp"

namespace mi {

// one wave, lane-per-dimension-class: 4 strided fma chains (lanes 0..3) + (q0+q2)+(q1+q3), the same
// canonical dot as the fused kernels / the oracle's orc_dot(W = 4)
__device__ __forceinline__ double cb_dot4(const double* x, const double* y, uint32_t d)
{
    const int lane = threadIdx.x;
    double q = 0.0;
    if (lane < 4)
        for (uint32_t i = lane; i < d; i += 4) q = dfma(x[i], y[i], q);
    const double q0 = __shfl(q, 0), q1 = __shfl(q, 1), q2 = __shfl(q, 2), q3 = __shfl(q, 3);
    return (q0 + q2) + (q1 + q3);
}

// new momentum (hmc.cpp:156-160): p = z, K = p.p/2; new_draw = prev_draw (:162)
__global__ void cb_begin_draw(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t d,
                              const double* prev_draw, double* new_draw, double* mntm, double* scal /*[0]=prev_K*/)
{
    const uint32_t nslots = 4 * ((d + 7) / 8);
    for (uint32_t slot = threadIdx.x; slot < nslots; slot += blockDim.x) {
        const uint32_t b = slot / 4, j = slot % 4, i0 = 8 * b + j, i1 = i0 + 4;
        double z0, z1;
        rng_normal_pair(seed, chain, draw, slot, STREAM_NORMAL, z0, z1);
        if (i0 < d) mntm[i0] = z0;
        if (i1 < d) mntm[i1] = z1;
    }
    for (uint32_t i = threadIdx.x; i < d; i += blockDim.x) new_draw[i] = prev_draw[i];
    __syncthreads();
    const double k = cb_dot4(mntm, mntm, d) / 2.0;
    if (threadIdx.x == 0) scal[0] = k;
}

// mntm = mntm + (eps * grad) / 2   (hmc.cpp:126)
__global__ void cb_half_kick(uint32_t d, double eps, const double* grad, double* mntm)
{
    for (uint32_t i = threadIdx.x; i < d; i += blockDim.x) mntm[i] = mntm[i] + (eps * grad[i]) / 2.0;
}

// new_draw += eps * mntm   (hmc.cpp:171, Minv = I)
__global__ void cb_drift(uint32_t d, double eps, const double* mntm, double* new_draw)
{
    for (uint32_t i = threadIdx.x; i < d; i += blockDim.x) new_draw[i] = new_draw[i] + eps * mntm[i];
}

// energies + Metropolis test + row store (hmc.cpp:178-204). scal: [0]=prev_K [1]=prev_U (in/out) [2]=prop_U (in)
__global__ void cb_accept(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t d, uint32_t n_burnin,
                          const double* new_draw, const double* mntm, double* prev_draw, double* scal,
                          double* draws_row /* this draw's row or nullptr */, uint64_t row_stride,
                          unsigned long long* n_accept)
{
    double prop_U = scal[2];
    if (!is_finite(prop_U)) prop_U = INF;
    const double prop_K = cb_dot4(mntm, mntm, d) / 2.0;
    const double x = -(prop_U + prop_K) + (scal[1] + scal[0]);
    const double comp_val = (x < 0.01) ? x : 0.01;
    const double z = rng_uniform(seed, chain, draw, 0u);
    const bool accept = z < det_exp(comp_val);
    __syncthreads();
    if (accept)
        for (uint32_t i = threadIdx.x; i < d; i += blockDim.x) prev_draw[i] = new_draw[i];
    __syncthreads();
    if (draws_row)
        for (uint32_t i = threadIdx.x; i < d; i += blockDim.x) draws_row[(size_t)i * row_stride] = prev_draw[i];
    if (threadIdx.x == 0) {
        if (accept) scal[1] = prop_U;
        if (accept && draw >= n_burnin) n_accept[0] += 1ull;
    }
}

// ---- vector operations of the host-callback forms of mcmc::mala and mcmc::nuts (one chain, state resident in HBM): one wave,
//      the element-wise expressions of the oracle operator for operator, dots in the canonical four-chain order (cb_dot4)

// dst = a + (e * b) / 2          (mntm_update_fn: src/nuts.cpp:126; mala_mean_fn with e = eps^2: src/mala.cpp:123)
__global__ void cb_add_half(uint32_t d, double e, const double* a, const double* b, double* dst)
{
    for (uint32_t i = threadIdx.x; i < d; i += blockDim.x) dst[i] = a[i] + (e * b[i]) / 2.0;
}
// dst = a + e * b                (leap_frog_fn drift: src/nuts.cpp:146; mala proposal: src/mala.cpp:159)
__global__ void cb_add_scaled(uint32_t d, double e, const double* a, const double* b, double* dst)
{
    for (uint32_t i = threadIdx.x; i < d; i += blockDim.x) dst[i] = a[i] + e * b[i];
}
// dst[0..d) = N(0, I) of (chain, draw, stream)   (src/mala.cpp:150, src/nuts.cpp:166,200)
__global__ void cb_normals(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t stream, uint32_t d, double* dst)
{
    const uint32_t nslots = 4 * ((d + 7) / 8);
    for (uint32_t slot = threadIdx.x; slot < nslots; slot += blockDim.x) {
        const uint32_t b = slot / 4, j = slot % 4, i0 = 8 * b + j, i1 = i0 + 4;
        double z0, z1;
        rng_normal_pair(seed, chain, draw, slot, stream, z0, z1);
        if (i0 < d) dst[i0] = z0;
        if (i1 < d) dst[i1] = z1;
    }
}
// out[0] = uniform of (chain, draw, slot)   (src/mala.cpp:171; src/nuts.cpp:206,233,261; nuts.ipp:213)
__global__ void cb_uniform(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t slot, double* out)
{
    if (threadIdx.x == 0) out[0] = rng_uniform(seed, chain, draw, slot);
}
// out[0] = x . y
__global__ void cb_dot(uint32_t d, const double* x, const double* y, double* out)
{
    const double q = cb_dot4(x, y, d);
    if (threadIdx.x == 0) out[0] = q;
}
// out[0] = (a - b) . p1, out[1] = (a - b) . p2     (U-turn tests: nuts.ipp:226-227, src/nuts.cpp:286-287); tmp: d values
__global__ void cb_diff_dots(uint32_t d, const double* a, const double* b, const double* p1, const double* p2, double* tmp, double* out)
{
    for (uint32_t i = threadIdx.x; i < d; i += blockDim.x) tmp[i] = a[i] - b[i];
    __syncthreads();
    const double q1 = cb_dot4(tmp, p1, d), q2 = cb_dot4(tmp, p2, d);
    if (threadIdx.x == 0) { out[0] = q1; out[1] = q2; }
}
// out[0] = (x - mu) . (rs * (x - mu))      (dmvnorm.hpp:37-39 with INV(eps^2 I) = diag(rs)); tmp: 2 d values
__global__ void cb_quad_form(uint32_t d, double rs, const double* x, const double* mu, double* tmp, double* out)
{
    for (uint32_t i = threadIdx.x; i < d; i += blockDim.x) { const double c = x[i] - mu[i]; tmp[i] = c; tmp[d + i] = rs * c; }
    __syncthreads();
    const double q = cb_dot4(tmp, tmp + d, d);
    if (threadIdx.x == 0) out[0] = q;
}

}  // namespace mi
