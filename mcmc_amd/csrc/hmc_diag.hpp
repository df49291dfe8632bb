// hmc_diag.hpp -- many-chain HMC for separable Gaussian targets (iso / diagonal precision), any d.
//
// Replaces the draw loop of mcmc::internal::hmc_impl (/root/reference/src/hmc.cpp:155-205) for
// targets without a contraction (BASELINE config 5: d = 1024 ill-conditioned diagonal Gaussian).
// With a diagonal precision and the identity (or a diagonal) preconditioner every dimension's (theta_i, p_i)
// trajectory is independent of the others through all L leapfrog steps; only the energies couple
// them.  Two kernels, same arithmetic, picked by the number of chains (hmc_diag_pick_lanes):
// hmc_diag4_kernel spreads a chain over FOUR lanes (lane = 16 j + chain-in-wave, the layout of the MFMA kernels): lane j owns the
// dimensions i = j (mod 4), i.e. of every block of 8 dimensions the pair (8b + j, 8b + 4 + j) -- exactly the two normals of
// Philox slot 4b + j, so no lane generates a number it does not use; hmc_diag1_kernel keeps one chain per lane.
// Each lane runs all L steps of its trajectories in
// registers and touches HBM once per dimension per draw: read theta_i, write the proposal.  State is [d][C] (chain contiguous ->
// 128-byte segments per j), the proposal is written straight into the slab it will live in if accepted (the kept-draw slab,
// or a ping-pong scratch slab during burn-in); a rejection copies instead.
//
// Why four lanes per chain: the bound is the fp64 VALU (7 instructions per chain.dim.step, ~2 B of HBM per unit at L = 32),
// and ONE wave issues an fp64 VALU instruction only every ~8 cycles whatever its instruction-level parallelism; a SIMD reaches
// ~6 cycles per instruction with two resident waves and ~5 with four or more (measured, tools/valu_rate.hip).  With one lane
// per chain a run of C chains has C/64 waves for 1024 SIMDs (config 5: two per SIMD); four lanes per chain give four times
// as many (config 5: eight per SIMD, 10 % faster, and a chain count that would leave SIMDs empty fills them).
//
// Arithmetic identical to the oracle / the MFMA kernel: w_i = prec_i * theta_i, kicks p - (eps*w)/2,
// drift theta + eps*p, dot products as 4 strided fma chains (dimension i -> chain i mod 4, ascending: lane j's own chain)
// combined (q0+q2)+(q1+q3) by two shuffles (xor 32, xor 16), as dot4 of hmc_dense.hpp.
#pragma once

#include "det_math.hpp"

namespace mi {

struct HmcDiagParams {
    const double* prec;     // device, d precisions; nullptr = isotropic (all ones)
    uint32_t d;
    uint64_t C, chain0;
    double* theta;          // [d][C] in/out
    double* scratch;        // [2][d][C] ping-pong slabs
    double* draws;          // [n_keep][d][C] or nullptr
    uint64_t* n_accept;
    uint64_t* n_leap;
    uint64_t seed;
    uint32_t n_burnin, n_keep, n_leap_steps;
    double eps;
    uint32_t draw0;         // index of this call's first draw in the chains' random streams (mi_chains.draw0)
    const double* m_sqrt;   // PRECOND: diagonal of CHOL_LOWER(precond_mat) (device, d values)
    const double* m_inv;    // PRECOND: diagonal of INV(precond_mat)
    uint32_t m_per_chain;   // PRECOND: 0 = one mass for all chains (m_sqrt / m_inv are [d]); 1 = per-chain masses (mi_chains.mass_diag): [d][C]
    uint32_t* nf_flag;      // [C + 1] or nullptr: chains that reached the non-finite regime are flagged and left to literal.hpp
};

constexpr int HMC_DIAG_CHAINS_PER_BLOCK = 64;     // 4 waves x 16 chains

// (q0 + q2) + (q1 + q3) over the four lanes of a chain; every lane of the chain gets the same bits
__device__ __forceinline__ double diag_combine(double q)
{
    q = q + __shfl_xor(q, 32);
    q = q + __shfl_xor(q, 16);
    return q;
}

// PRECOND: a diagonal precond_mat M (hmc.cpp:57-59,158-160,171,184): p = sqrt(M) z, theta += eps (Minv p), K = p.(Minv p)/2 --
// still one independent trajectory per dimension.  The NaN poisoning of the reference's dense `inv_precond_matrix * mntm`
//  couples the dimensions: these kernels detect the regime (a non-finite energy is its necessary
// consequence), flag the chain in prm.nf_flag and leave theta / n_accept untouched; literal.hpp replays the chain.
#ifndef MI_DIAG4_TRAJ
#define MI_DIAG4_TRAJ 4        // trajectories in flight per lane: 4 (two Philox slots per iteration) or 2 (one)
#endif
#ifndef MI_DIAG4_MINW
#define MI_DIAG4_MINW 1        // waves per SIMD the register allocation must allow (launch bound)
#endif
template <bool PRECOND>
__global__ __launch_bounds__(256, MI_DIAG4_MINW) void hmc_diag4_kernel(const HmcDiagParams prm)
{
    constexpr int TR = MI_DIAG4_TRAJ;
    static_assert(TR == 2 || TR == 4, "one or two Philox slots per iteration");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t j = (uint32_t)(lane >> 4);
    const uint64_t cl = ((uint64_t)blockIdx.x * 4 + wave) * 16 + (uint64_t)(lane & 15);
    if (((uint64_t)blockIdx.x * 4 + wave) * 16 >= prm.C) return;          // whole wave past the end
    const bool live = cl < prm.C;
    const uint64_t c = live ? cl : prm.C - 1;           // dead lanes shadow the last chain (loads only; shuffles stay wave-wide)
    const uint64_t chain = prm.chain0 + c;
    const uint32_t d = prm.d;
    const uint64_t C = prm.C;
    const double eps = prm.eps;
    const double* prec = prm.prec;
    const uint32_t L = prm.n_leap_steps;
    const size_t slab = (size_t)d * C;

    // PRECOND tables: element i of this chain's mass is at [i * mstr] behind msq / miv (one shared table, or the chain's column of [d][C])
    [[maybe_unused]] const size_t mstr = (PRECOND && prm.m_per_chain) ? (size_t)C : (size_t)1;
    [[maybe_unused]] const double* const msq = PRECOND ? prm.m_sqrt + ((prm.m_per_chain) ? c : 0) : nullptr;
    [[maybe_unused]] const double* const miv = PRECOND ? prm.m_inv + ((prm.m_per_chain) ? c : 0) : nullptr;

    const double* cur = prm.theta + c;                  // this chain's column of the slab holding prev_draw
    // prev_U = -box_log_kernel(first_draw)  (hmc.cpp:140)
    double prev_U;
    {
        double q = 0.0;
        for (uint32_t i = j; i < d; i += 4) {
            const double t = cur[(size_t)i * C];
            const double w = (prec ? prec[i] : 1.0) * t;
            q = dfma(t, w, q);
        }
        prev_U = 0.5 * diag_combine(q);
    }
    uint64_t n_acc = 0;
    const uint32_t n_total = prm.n_burnin + prm.n_keep;
    uint32_t pp = 0;                                    // ping-pong index of the next free scratch slab
    bool nf_seen = false;

    for (uint32_t draw = 0; draw < n_total; ++draw) {
        const bool kept = draw >= prm.n_burnin;
        double* dst = (kept && prm.draws) ? prm.draws + (size_t)(draw - prm.n_burnin) * slab + c
                                          : prm.scratch + (size_t)pp * slab + c;
        if (dst == cur) { pp ^= 1u; dst = prm.scratch + (size_t)pp * slab + c; }   // never overwrite prev_draw
        double qk0 = 0.0, qu1 = 0.0, qk1 = 0.0;
        // two blocks of 8 dimensions per iteration: 4 trajectories in flight per lane (dims 8b+j, 8b+4+j, 8b+8+j, 8b+12+j)
        for (uint32_t b = 0; b * 8 < d; b += TR / 2) {
            double z[TR], th[TR], pm[TR], lam[TR], w[TR], mi_[PRECOND ? TR : 1];
            rng_normal_pair(prm.seed, chain, draw + prm.draw0, 4 * b + j, STREAM_NORMAL, z[0], z[1]);
            if constexpr (TR == 4) rng_normal_pair(prm.seed, chain, draw + prm.draw0, 4 * (b + 1) + j, STREAM_NORMAL, z[TR - 2], z[TR - 1]);
#pragma unroll
            for (int r = 0; r < TR; ++r) {
                const uint32_t i = 8 * b + 4 * (uint32_t)r + j;
                const uint32_t ic = i < d ? i : d - 1;                    // clamped: unconditional loads
                th[r] = cur[(size_t)ic * C];
                lam[r] = prec ? prec[ic] : 1.0;
                if constexpr (PRECOND) { pm[r] = msq[(size_t)ic * mstr] * z[r]; mi_[r] = miv[(size_t)ic * mstr]; }   // p = L z (hmc.cpp:158)
                else pm[r] = z[r];                                        // L = I
                w[r] = lam[r] * th[r];
            }
#pragma unroll
            for (int r = 0; r < TR; ++r)
                if (8 * b + 4 * (uint32_t)r + j < d) qk0 = dfma(pm[r], PRECOND ? mi_[PRECOND ? r : 0] * pm[r] : pm[r], qk0);
            // hmc.cpp:164-176.  The second half-step of step k and the first of step k+1 see the same gradient, hence the same
            // (eps*w)/2: formed once, subtracted twice (the reference's two roundings) -- 7 instead of 9 operations per step.
            if (L > 0) {
#pragma unroll
                for (int r = 0; r < TR; ++r) pm[r] = pm[r] - (eps * w[r]) / 2.0;                    // first half-step of step 0
            }
            // the step counter lives in a scalar register (a vector counter is 2 of the 30 VALU instructions of this VALU-bound loop)
            for (uint32_t kk = (uint32_t)__builtin_amdgcn_readfirstlane((int)((L > 0) ? L - 1 : 0u)); kk != 0u; kk = (uint32_t)__builtin_amdgcn_readfirstlane((int)(kk - 1u))) {
#pragma unroll
                for (int r = 0; r < TR; ++r) {
                    if constexpr (PRECOND) th[r] = th[r] + eps * (mi_[PRECOND ? r : 0] * pm[r]);   // :171
                    else th[r] = th[r] + eps * pm[r];
                    w[r] = lam[r] * th[r];
                    const double t = (eps * w[r]) / 2.0;
                    pm[r] = pm[r] - t;                                                             // second half-step of step k
                    pm[r] = pm[r] - t;                                                             // first half-step of step k+1
                }
            }
            if (L > 0) {
#pragma unroll
                for (int r = 0; r < TR; ++r) {
                    if constexpr (PRECOND) th[r] = th[r] + eps * (mi_[PRECOND ? r : 0] * pm[r]);
                    else th[r] = th[r] + eps * pm[r];
                    w[r] = lam[r] * th[r];
                    pm[r] = pm[r] - (eps * w[r]) / 2.0;                                            // second half-step of the last step
                }
            }
#pragma unroll
            for (int r = 0; r < TR; ++r) {
                const uint32_t i = 8 * b + 4 * (uint32_t)r + j;
                if (i < d) {
                    qu1 = dfma(th[r], w[r], qu1);
                    qk1 = dfma(pm[r], PRECOND ? mi_[PRECOND ? r : 0] * pm[r] : pm[r], qk1);
                    if (live) dst[(size_t)i * C] = th[r];
                }
            }
        }
        const double prev_K = diag_combine(qk0) / 2.0;                    // hmc.cpp:160
        double prop_U = 0.5 * diag_combine(qu1);                          // :178
        const bool u_nf = !is_finite(prop_U);
        if (u_nf) prop_U = INF;
        const double prop_K = diag_combine(qk1) / 2.0;                    // :184
        nf_seen |= u_nf | !is_finite(prop_K);
        const double x = -(prop_U + prop_K) + (prev_U + prev_K);
        const double comp_val = (x < 0.01) ? x : 0.01;
        const double zu = rng_uniform(prm.seed, chain, draw + prm.draw0, 0u);
        const bool accept = zu < det_exp(comp_val);                       // the same bits on the four lanes of the chain
        if (accept) {
            cur = dst;
            prev_U = prop_U;
            if (!(kept && prm.draws)) pp ^= 1u;
        } else if (kept && prm.draws) {
            if (live)
                for (uint32_t i = j; i < d; i += 4) dst[(size_t)i * C] = cur[(size_t)i * C];   // row = prev_draw (:202)
            cur = dst;
        }
        if (kept) n_acc += accept ? 1u : 0u;
    }
    if (!live) return;
    if (nf_seen && prm.nf_flag != nullptr) {             // literal.hpp replays this chain from its untouched initial values
        if (j == 0) { prm.nf_flag[c] = 1u; prm.nf_flag[C] = 1u; }
        return;
    }
    double* out = prm.theta + c;
    if (cur != out)
        for (uint32_t i = j; i < d; i += 4) out[(size_t)i * C] = cur[(size_t)i * C];
    if (j == 0) {
        if (prm.n_accept) prm.n_accept[c] = n_acc;
        if (prm.n_leap) prm.n_leap[c] = (uint64_t)n_total * L;
    }
}

// One lane per chain: the lane walks its d dimensions in blocks of 8 (= 4 Philox slots) with 8 trajectories in flight.  Fewer
// instructions per unit than the four-lane kernel (one uniform / exp / pointer set per chain instead of four); used once the
// chains alone fill every SIMD with waves, see hmc_diag_pick_lanes.
template <bool PRECOND>
__global__ __launch_bounds__(256) void hmc_diag1_kernel(const HmcDiagParams prm)
{
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= prm.C) return;
    const uint64_t chain = prm.chain0 + c;
    const uint32_t d = prm.d;
    const uint64_t C = prm.C;
    const double eps = prm.eps;
    const double* prec = prm.prec;
    const uint32_t L = prm.n_leap_steps;
    const size_t slab = (size_t)d * C;

    [[maybe_unused]] const size_t mstr = (PRECOND && prm.m_per_chain) ? (size_t)C : (size_t)1;
    [[maybe_unused]] const double* const msq = PRECOND ? prm.m_sqrt + ((prm.m_per_chain) ? c : 0) : nullptr;
    [[maybe_unused]] const double* const miv = PRECOND ? prm.m_inv + ((prm.m_per_chain) ? c : 0) : nullptr;

    const double* cur = prm.theta + c;                  // this chain's column of the slab holding prev_draw
    // prev_U = -box_log_kernel(first_draw)  (hmc.cpp:140)
    double prev_U;
    {
        double q[4] = {0.0, 0.0, 0.0, 0.0};
        for (uint32_t i = 0; i < d; ++i) {
            const double t = cur[(size_t)i * C];
            const double w = (prec ? prec[i] : 1.0) * t;
            q[i & 3] = dfma(t, w, q[i & 3]);
        }
        prev_U = 0.5 * ((q[0] + q[2]) + (q[1] + q[3]));
    }
    uint64_t n_acc = 0;
    const uint32_t n_total = prm.n_burnin + prm.n_keep;
    uint32_t pp = 0;                                    // ping-pong index of the next free scratch slab
    bool nf_seen = false;

    for (uint32_t draw = 0; draw < n_total; ++draw) {
        const bool kept = draw >= prm.n_burnin;
        double* dst = (kept && prm.draws) ? prm.draws + (size_t)(draw - prm.n_burnin) * slab + c
                                          : prm.scratch + (size_t)pp * slab + c;
        if (dst == cur) { pp ^= 1u; dst = prm.scratch + (size_t)pp * slab + c; }   // never overwrite prev_draw
        double qk0[4] = {0, 0, 0, 0}, qu1[4] = {0, 0, 0, 0}, qk1[4] = {0, 0, 0, 0};
        for (uint32_t b = 0; b * 8 < d; ++b) {
            double z[8], th[8], pm[8], lam[8], w[8], mi_[PRECOND ? 8 : 1];
#pragma unroll
            for (int j = 0; j < 4; ++j) rng_normal_pair(prm.seed, chain, draw + prm.draw0, 4 * b + j, STREAM_NORMAL, z[j], z[4 + j]);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const uint32_t i = 8 * b + r;
                const uint32_t ic = i < d ? i : d - 1;                    // clamped: unconditional loads
                th[r] = cur[(size_t)ic * C];
                lam[r] = prec ? prec[ic] : 1.0;
                if constexpr (PRECOND) { pm[r] = msq[(size_t)ic * mstr] * z[r]; mi_[r] = miv[(size_t)ic * mstr]; }   // p = L z (hmc.cpp:158)
                else pm[r] = z[r];                                        // L = I
                w[r] = lam[r] * th[r];
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) if (8 * b + r < d) qk0[r & 3] = dfma(pm[r], PRECOND ? mi_[PRECOND ? r : 0] * pm[r] : pm[r], qk0[r & 3]);
            // hmc.cpp:164-176.  The second half-step of step k and the first of step k+1 see the same gradient, hence the same
            // (eps*w)/2: formed once, subtracted twice (the reference's two roundings) -- 7 instead of 9 operations per step.
            if (L > 0) {
#pragma unroll
                for (int r = 0; r < 8; ++r) pm[r] = pm[r] - (eps * w[r]) / 2.0;                    // first half-step of step 0
            }
            for (uint32_t k = 0; k + 1 < L; ++k) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if constexpr (PRECOND) th[r] = th[r] + eps * (mi_[PRECOND ? r : 0] * pm[r]);   // :171
                    else th[r] = th[r] + eps * pm[r];
                    w[r] = lam[r] * th[r];
                    const double t = (eps * w[r]) / 2.0;
                    pm[r] = pm[r] - t;                                                             // second half-step of step k
                    pm[r] = pm[r] - t;                                                             // first half-step of step k+1
                }
            }
            if (L > 0) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if constexpr (PRECOND) th[r] = th[r] + eps * (mi_[PRECOND ? r : 0] * pm[r]);
                    else th[r] = th[r] + eps * pm[r];
                    w[r] = lam[r] * th[r];
                    pm[r] = pm[r] - (eps * w[r]) / 2.0;                                            // second half-step of the last step
                }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const uint32_t i = 8 * b + r;
                if (i < d) {
                    qu1[r & 3] = dfma(th[r], w[r], qu1[r & 3]);
                    qk1[r & 3] = dfma(pm[r], PRECOND ? mi_[PRECOND ? r : 0] * pm[r] : pm[r], qk1[r & 3]);
                    dst[(size_t)i * C] = th[r];
                }
            }
        }
        const double prev_K = ((qk0[0] + qk0[2]) + (qk0[1] + qk0[3])) / 2.0;   // hmc.cpp:160
        double prop_U = 0.5 * ((qu1[0] + qu1[2]) + (qu1[1] + qu1[3]));         // :178
        const bool u_nf = !is_finite(prop_U);
        if (u_nf) prop_U = INF;
        const double prop_K = ((qk1[0] + qk1[2]) + (qk1[1] + qk1[3])) / 2.0;   // :184
        nf_seen |= u_nf | !is_finite(prop_K);
        const double x = -(prop_U + prop_K) + (prev_U + prev_K);
        const double comp_val = (x < 0.01) ? x : 0.01;
        const double zu = rng_uniform(prm.seed, chain, draw + prm.draw0, 0u);
        const bool accept = zu < det_exp(comp_val);
        if (accept) {
            cur = dst;
            prev_U = prop_U;
            if (!(kept && prm.draws)) pp ^= 1u;
        } else if (kept && prm.draws) {
            for (uint32_t i = 0; i < d; ++i) dst[(size_t)i * C] = cur[(size_t)i * C];   // row = prev_draw (:202)
            cur = dst;
        }
        if (kept) n_acc += accept ? 1u : 0u;
    }
    if (nf_seen && prm.nf_flag != nullptr) { prm.nf_flag[c] = 1u; prm.nf_flag[C] = 1u; return; }   // replayed by literal.hpp
    double* out = prm.theta + c;
    if (cur != out)
        for (uint32_t i = 0; i < d; ++i) out[(size_t)i * C] = cur[(size_t)i * C];
    if (prm.n_accept) prm.n_accept[c] = n_acc;
    if (prm.n_leap) prm.n_leap[c] = (uint64_t)n_total * L;
}

// lanes per chain: 4 until one lane per chain alone gives every SIMD eight waves (C >= 8 x 1024 x 64), then 1.
// Measured on one box, config 5 (131 072 chains, two waves per SIMD with one lane per chain): 4 lanes 52-55 ms, 1 lane 58-60 ms.
inline int hmc_diag_pick_lanes(uint64_t C) { return C >= 8ull * 1024ull * 64ull ? 1 : 4; }

}  // namespace mi
