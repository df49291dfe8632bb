// hmc_dense.hpp -- fused many-chain HMC for Gaussian targets whose gradient is a dense
// mat-vec, on the gfx950 fp64 matrix cores.
//
// Replaces, for C independent chains, the sampling loop of mcmc::internal::hmc_impl
// (/root/reference/src/hmc.cpp:155-205) together with the BaseMatrixOps calls inside it
// (rnorm_vec_inplace :156, L*z :158, DOT_PROD :160,184, Minv*p :171, runif :189) for the
// identity preconditioner (precond_mat unset -> BMO_MATOPS_EYE, :57).
//
// Mapping (one wavefront = 16 chains, whole trajectory register-resident):
//   G[d x 16] = P[d x d] * Theta[d x 16] is issued as v_mfma_f64_16x16x4_f64 tiles:
//     A = P fragment   lane l holds P[16t + (l&15)][4s + (l>>4)]      (read from LDS)
//     B = Theta slice  lane l holds theta[4s + (l>>4)] of chain (l&15) (VGPR)
//     D = G tile       lane l holds g[16t + 4r + (l>>4)] of chain (l&15), r = 0..3
//   The D layout of tile t register r IS the B layout of slice s = 4t + r, so the momentum
//   kick / position drift after every gradient are per-lane register updates with no
//   cross-lane traffic, and the 2^k-free leapfrog never leaves the register file.
//   P (d_pad^2 * 8 B, 128 KiB at d = 128) is staged once per workgroup in LDS in fragment
//   order: one conflict-free ds_read_b64 per MFMA.
//
// Arithmetic is specified operation by operation (compile with -ffp-contract=off):
//   each gradient row is one sequential fma chain over k ascending (the MFMA accumulates
//   its four k in order); dot products are 4 strided fma chains (one per lane group)
//   combined as (q0+q2)+(q1+q3); elementwise updates follow the reference expressions
//   p + (eps*g)/2 (:126) and theta + eps*p (:171) with one rounding per operator.
#pragma once

#include "det_math.hpp"

namespace mi {

typedef double double4_t __attribute__((ext_vector_type(4)));

struct HmcParams {
    const double* P;        // device, d x d row-major precision
    uint32_t d;
    uint64_t C;             // chains in this launch
    uint64_t chain0;        // global id of local chain 0
    double* theta;          // [d][C] in/out: always the last accepted state
    double* wsave;          // [n_waves][2][NS][64] workspace: last accepted theta and P*theta
    double* draws;          // [n_keep][d][C] or nullptr
    uint64_t* n_accept;     // [C] or nullptr
    uint64_t* n_leap;       // [C] or nullptr
    uint64_t seed;
    uint32_t n_burnin, n_keep, n_leap_steps;
    double eps;
    uint32_t ablate;        // profiling only: 1 = skip kick/drift, 2 = skip mat-vec (results meaningless)
    uint32_t stagger;       // start delay of the second wave of each SIMD, in s_sleep(127) units
};

template <int NS>
__device__ __forceinline__ double dot4(const double (&x)[NS], const double (&y)[NS])
{
    double q = 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s) q = dfma(x[s], y[s], q);
    q = q + __shfl_xor(q, 32);
    q = q + __shfl_xor(q, 16);
    return q;
}

// w = P * th for the wave's 16 chains. afrag points at this lane's column of the LDS fragments.
// Software pipeline: the NT fragments of slice s+1 are read from LDS while the NT MFMAs of slice
// s issue (NT*64 cycles of matrix pipe cover the LDS latency); sched_barriers pin that order.
template <int NT>
__device__ __forceinline__ void matvec_mfma(const double* __restrict__ afrag, const double (&th)[4 * NT],
                                            double (&w)[4 * NT])
{
    constexpr int NS = 4 * NT;
    double4_t acc[NT];
    double a_cur[NT], a_nxt[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
        a_cur[t] = afrag[(t * NS) * 64];
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s + 1 < NS) {
#pragma unroll
            for (int t = 0; t < NT; ++t) a_nxt[t] = afrag[(t * NS + s + 1) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[t], th[s], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) a_cur[t] = a_nxt[t];
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        w[4 * t + 0] = acc[t][0];
        w[4 * t + 1] = acc[t][1];
        w[4 * t + 2] = acc[t][2];
        w[4 * t + 3] = acc[t][3];
    }
}

// Stage P into LDS in MFMA A-fragment order: frag f = t*NS + s, lane l -> P[16t + (l&15)][4s + (l>>4)].
template <int NT>
__device__ __forceinline__ void stage_precision(const double* __restrict__ P, uint32_t d, double* lds)
{
    constexpr int NS = 4 * NT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int f = wave; f < NT * NS; f += nw) {
        const int t = f / NS, s = f % NS;
        const uint32_t row = 16 * t + (lane & 15), col = 4 * s + (lane >> 4);
        lds[f * 64 + lane] = (row < d && col < d) ? P[(size_t)row * d + col] : 0.0;
    }
    __syncthreads();
}

// WPB = waves per workgroup: 4 (one wave per SIMD, 512-register budget) or 8 (two waves per SIMD,
// 256 registers each: one wave's VALU phases -- kick/drift, RNG, accept -- hide under the other's MFMAs).
template <int NT, int WPB>
__global__ __launch_bounds__(64 * WPB, WPB / 4) void hmc_gauss_mfma_kernel(const HmcParams prm)
{
    constexpr int NS = 4 * NT;
    extern __shared__ __attribute__((aligned(16))) double lds_P[];
    stage_precision<NT>(prm.P, prm.d, lds_P);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane >> 4;
    const uint64_t cl = ((uint64_t)blockIdx.x * WPB + wave) * 16 + (lane & 15);
    const bool live = cl < prm.C;
    const uint64_t cld = live ? cl : prm.C - 1;       // clamped index for loads
    const uint64_t chain = prm.chain0 + cl;           // global chain id (Philox counter)
    const uint32_t d = prm.d;
    const uint64_t C = prm.C;
    const double eps = prm.eps;
    const double* afrag = lds_P + lane;

    // Register-resident state of the wave's 16 chains: position, momentum, P*position.
    // The last accepted (theta, P*theta) lives in HBM (prm.theta / prm.wsave): written on accept,
    // re-read on reject, so a rejection costs 2 KiB of traffic per chain instead of 128 VGPRs.
    double th[NS], pm[NS], w[NS];
    // addresses = wave-uniform row base (SGPR) + one per-lane element offset (VGPR)
    const size_t lane_off = (size_t)j * C + cld;
    // last accepted (theta, P*theta): wave-local contiguous [wave][2][NS][64 lanes] (512-B coalesced per slice)
    double* const ws_wave = prm.wsave + ((size_t)blockIdx.x * WPB + wave) * ((size_t)2 * NS * 64) + lane;
    auto th_mem = [&](int s) -> double* { return ws_wave + (size_t)s * 64; };
    auto w_mem = [&](int s) -> double* { return ws_wave + (size_t)(NS + s) * 64; };

#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const uint32_t dim = 4 * s + j;
        const double v = prm.theta[(size_t)(dim < d ? dim : 0u) * C + cld];   // clamped row: unconditional load
        th[s] = (dim < d) ? v : 0.0;
    }
    matvec_mfma<NT>(afrag, th, w);
    if (live) {
#pragma unroll
        for (int s = 0; s < NS; ++s) { *th_mem(s) = th[s]; *w_mem(s) = w[s]; }
    }
    double prev_U = 0.5 * dot4<NS>(th, w);              // -box_log_kernel(first_draw), hmc.cpp:140
    uint64_t n_acc = 0;
    const uint32_t n_total = prm.n_burnin + prm.n_keep;

    // Two waves share each SIMD's matrix pipe.  Started together they stay in lock-step and their
    // VALU phases (RNG, accept, kick/drift) coincide; a one-off start offset is self-preserving under
    // round-robin MFMA issue, so one wave's VALU work then always sits under the other's MFMAs.
    if (WPB > 4 && wave >= 4)
        for (uint32_t k = 0; k < prm.stagger; ++k) __builtin_amdgcn_s_sleep(127);

#pragma unroll 1
    for (uint32_t draw = 0; draw < n_total; ++draw) {
        // momentum ~ N(0, I): hmc.cpp:156-158 (L = chol(I) = I)
#pragma unroll
        for (int b = 0; b < NS / 2; ++b) {
            double z0, z1;
            rng_normal_pair(prm.seed, chain, draw, (uint32_t)(4 * b + j), STREAM_NORMAL, z0, z1);
            pm[2 * b] = (8u * b + j < d) ? z0 : 0.0;
            pm[2 * b + 1] = (8u * b + 4 + j < d) ? z1 : 0.0;
            __builtin_amdgcn_sched_barrier(0);
        }
        const double prev_K = dot4<NS>(pm, pm) / 2.0;   // hmc.cpp:160

#pragma unroll 1
        for (uint32_t k = 0; k < prm.n_leap_steps; ++k) {   // hmc.cpp:164-176, grad = -w
            if (prm.ablate != 1) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                pm[s] = pm[s] - (eps * w[s]) / 2.0;     // first half-step (:167,126)
                th[s] = th[s] + eps * pm[s];            // (:171)
            }
            }
            if (prm.ablate != 2) matvec_mfma<NT>(afrag, th, w);
            if (prm.ablate != 1) {
#pragma unroll
            for (int s = 0; s < NS; ++s) pm[s] = pm[s] - (eps * w[s]) / 2.0;   // second half-step (:175)
            }
        }

        double prop_U = 0.5 * dot4<NS>(th, w);          // -box_log_kernel(new_draw), hmc.cpp:178
        if (!is_finite(prop_U)) prop_U = INF;           // :180-182
        const double prop_K = dot4<NS>(pm, pm) / 2.0;   // :184
        const double x = -(prop_U + prop_K) + (prev_U + prev_K);
        const double comp_val = (x < 0.01) ? x : 0.01;  // std::min(0.01, x), :188
        const double z = rng_uniform(prm.seed, chain, draw, 0u);   // :189
        const bool accept = z < det_exp(comp_val);      // :191
        if (accept) {                                   // prev_draw = new_draw (:192-194)
            prev_U = prop_U;
            if (live) {
#pragma unroll
                for (int s = 0; s < NS; ++s) { *th_mem(s) = th[s]; *w_mem(s) = w[s]; }
            }
        } else {                                        // keep prev_draw: reload it (padded rows: no predicates)
#pragma unroll
            for (int s = 0; s < NS; ++s) { th[s] = *th_mem(s); w[s] = *w_mem(s); }
        }
        if (draw >= prm.n_burnin) {                     // :196-204
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
            if (dim < d) prm.theta[(size_t)dim * C + cl] = th[s];
        }
    }
    if (live && j == 0) {
        if (prm.n_accept) prm.n_accept[cl] = n_acc;                            // hmc.cpp:220-222
        if (prm.n_leap) prm.n_leap[cl] = (uint64_t)n_total * prm.n_leap_steps;
    }
}

}  // namespace mi
