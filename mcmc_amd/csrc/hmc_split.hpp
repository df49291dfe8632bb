// hmc_split.hpp -- mcmc::hmc on dense-gradient Gaussian targets when there are FEWER chain tiles than SIMDs: the strong-scaling
// end of BASELINE configs[1] (65 536 chains over 8 GPUs = 8 192 chains = 512 tiles of 16 chains for 1 024 SIMDs per GPU).
//
// Same algorithm and the same bits as hmc_gauss_mfma_kernel (hmc_dense.hpp; reference: /root/reference/src/hmc.cpp:155-205),
// plain case (unbounded, identity precond_mat).  There one wave owns a tile of 16 chains and all d rows of P * theta; with 512
// tiles that is one busy SIMD in two.  Here SPLIT waves share a tile: wave h owns the row tiles t in [h NT/SPLIT, (h+1) NT/SPLIT)
// of the mat-vec -- by the D-layout-equals-B-layout identity of hmc_dense.hpp exactly the slices s in [h NS/SPLIT, (h+1) NS/SPLIT)
// of theta, p and P*theta, which it keeps in registers, kicks, drifts, draws normals for and stores.  The one thing a wave lacks
// is the OTHER waves' theta as MFMA B operands: after every drift each wave publishes its slices in an LDS exchange buffer
// (16 KB per tile) and reads the rest -- two workgroup barriers per leapfrog step, against NT/SPLIT * NS MFMAs of 64 cycles each.
// Dot products keep the oracle's order (four strided fma chains over the dimensions, i ascending, then (q0+q2)+(q1+q3)): the
// chain of lane class j runs through wave 0's slices, is handed to wave 1 through LDS, and so on; three such relays per draw.
// LDS: the 128 KB of fragments + 16 KB of exchange buffer per tile; two tiles per workgroup fill the 160 KB of a CU: SPLIT = 2
// with four waves (one per SIMD), or SPLIT = 4 with eight (two per SIMD: a wave's exchange, kick, drift and normals then run
// under its neighbour's MFMAs).
#pragma once

#include "hmc_dense.hpp"

namespace mi {

template <int NT, int SPLIT, int WPB = 4>
constexpr size_t hmc_split_lds_bytes()
{
    constexpr int NS = 4 * NT, TILES = WPB / SPLIT;
    return ((size_t)NT * NS * 64 + (size_t)TILES * NS * 64) * sizeof(double);       // d = 128, SPLIT = 2: exactly the 160 KB of a CU
}

// The role h of a wave inside its tile is a RUN-TIME, wave-uniform value that enters addresses only: the wave keeps its own slices in
// `own[NSO]` (compile-time indices) and takes every B operand of the mat-vec -- its own slices included -- from the LDS exchange
// buffer, one read ahead of the MFMAs that use it.  So there is ONE body for all roles and every barrier sits on the common path that
// all waves of the workgroup execute (round 3 instantiated the body per role, each copy with its own barriers, and relied on
// s_barrier counting waves rather than program counters).
template <int NT, int SPLIT, int WPB>
__device__ __forceinline__ void hmc_split_body(const HmcParams& prm, double* lds_P, const int h)
{
    constexpr int NS = 4 * NT;
    constexpr int NSO = NS / SPLIT;          // slices this wave owns
    constexpr int NTO = NT / SPLIT;          // row tiles this wave owns
    constexpr int TILES = WPB / SPLIT;       // chain tiles per workgroup (WPB = 4: one wave per SIMD; 8: two, one's exchange / kick / drift under the other's MFMAs)
    double* const lds_x = lds_P + NT * NS * 64;            // [TILES][NS][64]: theta slices of every tile, published by their owners

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = wave / SPLIT;
    const int j = lane >> 4;
    const uint64_t cl = ((uint64_t)blockIdx.x * TILES + tile) * 16 + (lane & 15);
    const bool live = cl < prm.C;
    const uint64_t cld = live ? cl : prm.C - 1;
    const uint64_t chain = prm.chain0 + cl;
    const uint32_t d = prm.d;
    const uint64_t C = prm.C;
    const double eps = prm.eps;
    const int s0 = h * NSO;                                // first own slice (addresses only: register indices take the role constant)
    double* const xt = lds_x + (size_t)tile * NS * 64 + lane;
    // own A fragments: (t, s) with t = h NTO + tt at ((h NTO + tt) NS + s) * 64 doubles: one per-wave base, immediates below 64 KB
    typedef const double __attribute__((address_space(3)))* lds_cptr;
    uint32_t a_off = (uint32_t)(uintptr_t)(lds_cptr)(lds_P + lane) + (uint32_t)(h * NTO * NS * 64 * 8);
    asm volatile("" : "+v"(a_off));
    const lds_cptr afrag = (lds_cptr)(uintptr_t)a_off;

    double own[NSO];           // own slices of theta
    double pm[NSO], w[NSO];    // own slices of the momentum and of P * theta
    const size_t lane_off = (size_t)j * C + cld;
    // last accepted (theta, P*theta), own slices: [tile][2][NS][64 lanes], as hmc_dense.hpp
    double* const ws_tile = prm.wsave + ((size_t)blockIdx.x * TILES + tile) * ((size_t)3 * NS * 64) + lane;
    auto th_mem = [&](int k) -> double* { return ws_tile + (size_t)(s0 + k) * 64; };
    auto w_mem = [&](int k) -> double* { return ws_tile + (size_t)(NS + s0 + k) * 64; };

    // publish the own slices of theta; the B operands of the mat-vec are read back from the buffer (all slices, the own ones too)
    auto exchange = [&]() __attribute__((always_inline)) {
        __syncthreads();                                   // everyone is done reading the previous contents
#pragma unroll
        for (int k = 0; k < NSO; ++k) xt[(s0 + k) * 64] = own[k];
        __syncthreads();
    };
    // w(own rows) = P(own rows, :) * theta
    auto gradient = [&]() __attribute__((always_inline)) {
        exchange();
        double4_t acc[NTO];
        double a_cur[NTO], a_nxt[NTO];
        double b_cur = xt[0], b_nxt = 0.0;
#pragma unroll
        for (int tt = 0; tt < NTO; ++tt) {
            acc[tt] = double4_t{0.0, 0.0, 0.0, 0.0};
            a_cur[tt] = afrag[(tt * NS + 0) * 64];
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s + 1 < NS) {
#pragma unroll
                for (int tt = 0; tt < NTO; ++tt) a_nxt[tt] = afrag[(tt * NS + s + 1) * 64];
                b_nxt = xt[(s + 1) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tt = 0; tt < NTO; ++tt)
                acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[tt], b_cur, acc[tt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tt = 0; tt < NTO; ++tt) a_cur[tt] = a_nxt[tt];
            b_cur = b_nxt;
        }
#pragma unroll
        for (int tt = 0; tt < NTO; ++tt) {
            w[4 * tt + 0] = acc[tt][0]; w[4 * tt + 1] = acc[tt][1]; w[4 * tt + 2] = acc[tt][2]; w[4 * tt + 3] = acc[tt][3];
        }
    };
    // dot4 of hmc_dense.hpp over ALL slices: the fma chain of every lane runs through the waves of the tile in slice order
    // (the relay cell is the first slice of the tile's exchange buffer, idle outside exchange(): a barrier on entry fences it)
    auto chain_dot = [&](auto&& xk, auto&& yk) __attribute__((always_inline)) -> double {
        double* const r = xt;
        __syncthreads();
#pragma unroll
        for (int hh = 0; hh < SPLIT; ++hh) {
            if (h == hh) {
                double q = (hh == 0) ? 0.0 : *r;
#pragma unroll
                for (int k = 0; k < NSO; ++k) q = dfma(xk(k), yk(k), q);
                *r = q;
            }
            __syncthreads();
        }
        double q = *r;
        q = q + __shfl_xor(q, 32);
        q = q + __shfl_xor(q, 16);
        return q;
    };
    auto kinetic = [&]() __attribute__((always_inline)) -> double {
        return chain_dot([&](int k) { return pm[k]; }, [&](int k) { return pm[k]; }) / 2.0;
    };
    auto potential = [&]() __attribute__((always_inline)) -> double {
        return 0.5 * chain_dot([&](int k) { return own[k]; }, [&](int k) { return w[k]; });
    };

#pragma unroll
    for (int k = 0; k < NSO; ++k) {
        const uint32_t dim = 4 * (s0 + k) + j;
        const double v = prm.theta[(size_t)(dim < d ? dim : 0u) * C + cld];   // clamped row: unconditional load
        own[k] = (dim < d) ? v : 0.0;
    }
    gradient();
    if (live) {
#pragma unroll
        for (int k = 0; k < NSO; ++k) { *th_mem(k) = own[k]; *w_mem(k) = w[k]; }
    }
    double prev_U = potential();                        // -box_log_kernel(first_draw), hmc.cpp:140
    uint64_t n_acc = 0;
    const uint32_t n_total = prm.n_burnin + prm.n_keep;
    const uint32_t L = prm.n_leap_steps;
    bool nf_seen = false;                               // non-finite regime: detect, flag, replay (hmc_dense.hpp; the energies are tile-uniform)

#pragma unroll 1
    for (uint32_t draw = 0; draw < n_total; ++draw) {
        // momentum ~ N(0, I): hmc.cpp:156-158; this wave draws the normals of its own dimensions only
#pragma unroll
        for (int b = 0; b < NSO / 2; ++b) {
            const int bb = s0 / 2 + b;                  // Philox block of the chain: dimensions 8 bb + j and 8 bb + 4 + j
            double z0, z1;
            rng_normal_pair_at(prm.seed, chain, draw + prm.draw0, (uint32_t)(4 * bb), (uint32_t)j, STREAM_NORMAL, z0, z1);
            pm[2 * b] = (8u * bb + j < d) ? z0 : 0.0;
            pm[2 * b + 1] = (8u * bb + 4 + j < d) ? z1 : 0.0;
            __builtin_amdgcn_sched_barrier(0);
        }
        const double prev_K = kinetic();                // hmc.cpp:160
        // hmc.cpp:164-176, grad = -w; adjacent half-steps share (eps*w)/2 (two roundings, as the reference's two statements)
        if (L > 0) {
#pragma unroll
            for (int k = 0; k < NSO; ++k) {
                pm[k] = pm[k] - (eps * w[k]) / 2.0;
                own[k] = own[k] + eps * pm[k];
            }
        }
#pragma unroll 1
        for (uint32_t st = 0; st + 1 < L; ++st) {
            gradient();
#pragma unroll
            for (int k = 0; k < NSO; ++k) {
                const double t = (eps * w[k]) / 2.0;
                pm[k] = pm[k] - t;
                pm[k] = pm[k] - t;
                own[k] = own[k] + eps * pm[k];
            }
        }
        if (L > 0) {
            gradient();
#pragma unroll
            for (int k = 0; k < NSO; ++k) pm[k] = pm[k] - (eps * w[k]) / 2.0;
        }
        double prop_U = potential();                    // hmc.cpp:178
        const bool u_nf = !is_finite(prop_U);
        if (u_nf) prop_U = INF;                         // :180-182
        const double prop_K = kinetic();                // :184
        nf_seen |= u_nf | !is_finite(prop_K);
        const double x = -(prop_U + prop_K) + (prev_U + prev_K);
        const double comp_val = (x < 0.01) ? x : 0.01;  // :188
        const double z = rng_uniform(prm.seed, chain, draw + prm.draw0, 0u);   // :189
        const bool accept = z < det_exp(comp_val);      // :191
        if (accept) {
            prev_U = prop_U;
            if (live) {
#pragma unroll
                for (int k = 0; k < NSO; ++k) { *th_mem(k) = own[k]; *w_mem(k) = w[k]; }
            }
        } else {
#pragma unroll
            for (int k = 0; k < NSO; ++k) { own[k] = *th_mem(k); w[k] = *w_mem(k); }
        }
        if (draw >= prm.n_burnin) {                     // :196-204
            n_acc += accept ? 1u : 0u;
            if (prm.draws != nullptr && live) {
                double* out = prm.draws + (size_t)(draw - prm.n_burnin) * d * C;
#pragma unroll
                for (int k = 0; k < NSO; ++k) {
                    const uint32_t dim = 4 * (s0 + k) + j;
                    if (dim < d) (out + (size_t)(4 * (s0 + k)) * C)[lane_off] = own[k];
                }
            }
        }
    }

    const bool replay = nf_seen && prm.nf_flag != nullptr;
    if (live && replay && j == 0 && h == 0) { prm.nf_flag[cl] = 1u; prm.nf_flag[C] = 1u; }
    if (live && !replay) {
#pragma unroll
        for (int k = 0; k < NSO; ++k) {
            const uint32_t dim = 4 * (s0 + k) + j;
            if (dim < d) prm.theta[(size_t)dim * C + cl] = own[k];
        }
    }
    if (live && !replay && j == 0 && h == 0) {
        if (prm.n_accept) prm.n_accept[cl] = n_acc;                            // hmc.cpp:220-222
        if (prm.n_leap) prm.n_leap[cl] = (uint64_t)n_total * prm.n_leap_steps;
    }
}

template <int NT, int SPLIT, int WPB = 4>
__global__ MI_NO_DS_MERGE __launch_bounds__(64 * WPB, WPB / 4) void hmc_gauss_split_kernel(const HmcParams prm)
{
    static_assert(SPLIT == 2 || SPLIT == 4, "a tile is shared by 2 or 4 waves");
    static_assert(NT % SPLIT == 0, "row tiles divide evenly over the waves of a tile");
    static_assert(WPB == 4 || WPB == 8, "one or two waves per SIMD");
    extern __shared__ __attribute__((aligned(16))) double lds_P[];
    stage_precision<NT>(prm.P, prm.d, lds_P);
    // one body for every role: h enters addresses only (see hmc_split_body), every barrier is on the common path
    const int h = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6) % SPLIT);
    hmc_split_body<NT, SPLIT, WPB>(prm, lds_P, h);
}

}  // namespace mi
