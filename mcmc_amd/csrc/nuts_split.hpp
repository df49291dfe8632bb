// nuts_split.hpp -- many-chain NUTS of the plain case (unbounded, identity precond_mat; BASELINE configs[3]) with every 16-chain tile
// SPLIT OVER TWO WAVES, two tiles per SIMD: one tile's record round trips run under the other tile's MFMAs.
//
// Same algorithm, arithmetic, tree derivation, record layout and bits as nuts_gauss_reg_kernel (nuts_reg.hpp; reference:
// /root/reference/src/nuts.cpp:30-332, include/mcmc/nuts.ipp:30-241; the iterative leaf-indexed tree is derived in nuts_dense.hpp).
// What nuts_reg.hpp measured: a tick is 18 k cycles of mat-vec + ~28 k cycles of record traffic, Philox and
// bookkeeping, and with ONE wave per SIMD (it needs all 512 registers for the register-carried leaf state) the two never overlap:
// matrix pipe 31 % busy, 2.9 TB/s of HBM traffic, both waiting for each other.  Here:
//   * wave h in {0, 1} of a tile owns the slices s in [16 h, 16 h + 16) of theta, p, P*theta (hmc_split.hpp: by the
//     D-layout-is-B-layout identity exactly the row tiles t in [4 h, 4 h + 4) of the mat-vec): 96 registers of leaf state and 64 of
//     U-turn operands instead of 192 + 128, so a wave fits 256 registers and a SIMD holds two -- of DIFFERENT tiles (wave = 2 tile + h,
//     SIMD = wave mod 4), whose ticks are independent;
//   * the one thing a wave lacks is its partner's half of theta as MFMA B operands: every leapfrog publishes the own half in an LDS
//     exchange buffer (8 KB per wave) and reads the other half from there during the mat-vec;
//   * dot products keep the oracle's order (four strided fma chains over the dimensions, i ascending, then (q0+q2)+(q1+q3)): the chain
//     of lane class j runs through wave 0's slices, is handed to wave 1 through LDS, finished there, and the result handed back --
//     one such relay per tick for its four dot products (d.p(b), d.p(b2), theta.P theta, p.p), one more when a doubling completes;
//   * the two waves of a tile synchronise PAIR-WISE through LDS sequence counters (a workgroup barrier would put the four tiles of a
//     workgroup in lock step, which is what the asynchronous state machine exists to avoid); both run the same control flow on the same
//     per-chain scalars, which they compute redundantly from the same relayed results;
//   * LDS: 64 KB of exchange buffers next to the precision do not fit 160 KB, so each wave keeps the fragments of two of its four
//     row tiles in LDS (64 KB for the workgroup) and streams the other two from L2 in fragment order (the stream of matvec_mfma_g,
//     hmc_dense.hpp: every wave of the launch reads the same 64 KB; issued three slices ahead, the first ones before the exchange).
// Per-level tree scalars are shared by the two waves of a tile (both write the same values); what is read-modify-written (n of
// nuts.cpp:283, the dual-averaging state) lives in a small per-wave table instead.
#pragma once

#include <type_traits>

#include "nuts_reg.hpp"

namespace mi {

// TPW = chain tiles per workgroup (2 TPW waves): 4 = two tiles per SIMD; 2 (1) = one wave per SIMD (on half the SIMDs), the shapes for
// fewer tiles than the chip has SIMD pairs
template <int NT, int TPW>
constexpr size_t nuts_split_lds_bytes()
{
    constexpr int NS = 4 * NT;
    // fragments of two row tiles per wave role, exchange buffers, levels 1..10, per-wave scalars, non-finite flags, counters
    return ((size_t)(NT / 2) * NS * 64 + (size_t)2 * TPW * (NS / 2 + 1) * 64 + (size_t)NUTS_MAX_DEPTH * 4 * 16 * TPW + (size_t)2 * TPW * 7 * 16 + 8 * TPW + TPW) * sizeof(double);
}

// P (d x d row-major) -> MFMA A-fragment order [t][s][lane] = P[16 t + (lane & 15)][4 s + (lane >> 4)], zero padded to 16 NT
// (what stage_precision writes to LDS); one launch per run, 128 KB at NT = 8
template <int NT>
__global__ void pack_precision_fragments_kernel(const double* __restrict__ P, uint32_t d, double* __restrict__ out)
{
    constexpr int NS = 4 * NT;
    const int f = blockIdx.x, lane = threadIdx.x;       // f = t NS + s
    const int t = f / NS, s = f % NS;
    const uint32_t row = 16 * t + (lane & 15), col = 4 * s + (lane >> 4);
    out[(size_t)f * 64 + lane] = (row < d && col < d) ? P[(size_t)row * d + col] : 0.0;
}

template <int NT, int TPW>
__global__ MI_NO_DS_MERGE __launch_bounds__(128 * TPW, 2) void nuts_gauss_split_kernel(const NutsParams prm)
{
    static_assert(TPW == 1 || TPW == 2 || TPW == 4, "1, 2 or 4 tiles per workgroup");
    static_assert(NT == 8, "the split kernel is the d = 128 shape (smaller precisions leave LDS for nuts_gauss_reg_kernel's own second workgroup)");
    constexpr int NS = 4 * NT;
    constexpr int NSO = NS / 2;                          // slices a wave owns
    constexpr int NTO = NT / 2;                          // row tiles a wave owns
    constexpr int NTL = NTO / 2;                         // ... of which in LDS; the others stream from L2
    constexpr int NTG = NTO - NTL;
    constexpr int WS_NVEC = NUTS_NVEC_ASYNC;
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    double* const lds_P = lds_all;                                   // [2 roles][NTL][NS][64]
    constexpr int XS = NSO + 1;                                      // exchange slots per wave: its theta slices + one early partial sum
    double* const lds_x = lds_P + 2 * NTL * NS * 64;                 // [tiles][2 roles][XS][64]
    constexpr int CW = 16 * TPW;                                     // chains per workgroup
    double* const lds_lvl = lds_x + 2 * TPW * XS * 64;               // [levels 1..10][4][CW]
    double* const lds_pw = lds_lvl + NUTS_MAX_DEPTH * 4 * CW;        // [waves][7][16]: per-wave copies of the read-modify-written scalars
    uint32_t* const lds_nf = reinterpret_cast<uint32_t*>(lds_pw + 2 * TPW * 7 * 16);   // [CW]
    uint32_t* const lds_sig = lds_nf + CW;                           // [tiles][2 roles] sequence counters

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {   // stage the LDS-resident fragments from the packed copy (coalesced), zero the counters
        const int nw = blockDim.x >> 6;
        for (int f = wave; f < 2 * NTL * NS; f += nw) {
            const int hh = f / (NTL * NS), r = f % (NTL * NS);       // role, (tt, s)
            lds_P[f * 64 + lane] = prm.Pfrag[((size_t)(hh * NTO) * NS + r) * 64 + lane];
        }
        if (threadIdx.x < 2 * TPW) lds_sig[threadIdx.x] = 0u;
        __syncthreads();                                             // the only workgroup barrier of the kernel
    }
    const int tile = __builtin_amdgcn_readfirstlane(wave >> 1);
    const int h = __builtin_amdgcn_readfirstlane(wave & 1);
    const int j4 = lane >> 4;
    const int ct = tile * 16 + (lane & 15);                          // chain within the workgroup
    const uint64_t cl = ((uint64_t)blockIdx.x * TPW + tile) * 16 + (lane & 15);
    const bool live = cl < prm.C;
    if (__ballot(live) == 0ull) return;                              // (both waves of the tile agree)
#ifdef MI_NUTS_SPLIT_PROF
    if (prm.prof != nullptr && prm.prof[95] != 0ull && (unsigned long long)tile >= prm.prof[95]) return;    // experiment: fewer tiles per workgroup (results void)
#endif
    const uint64_t cld = live ? cl : prm.C - 1;
    const uint64_t chain = prm.chain0 + cl;
    const uint32_t d = prm.d;
    const uint64_t C = prm.C;
    const int s0 = h * NSO;                                          // first own slice
    const size_t lane_off = (size_t)j4 * C + cld;

    auto lvl = [&](int l, int f) -> double& { return lds_lvl[((l - 1) * 4 + f) * CW + ct]; };        // l = 1 .. 10
    auto pw = [&](int k) -> double& { return lds_pw[(wave * 7 + k) * 16 + (lane & 15)]; };
    // exchange buffers of the tile: the own half (written here, read by the partner) and the partner's
    double* const x_own = lds_x + ((size_t)(tile * 2 + h) * XS) * 64 + lane;
    double* const x_par = lds_x + ((size_t)(tile * 2 + (1 - h)) * XS) * 64 + lane;
    // ---- pair-wise synchronisation.  Every synchronisation point is passed by both waves of the tile in the same order (their control
    // flow is identical); at point n a wave publishes n in its own counter when its side of the hand-over is in LDS and waits for the
    // partner's counter to reach n before it touches what the partner handed over.
    uint32_t nsync = 0;
    uint32_t* const sig_own = lds_sig + tile * 2 + h;
    uint32_t* const sig_par = lds_sig + tile * 2 + (1 - h);
    // (LDS operations of a wave execute in order; the waitcnt + compiler barrier make the hand-over visible before the counter moves.
    //  No memory-model fence: at workgroup scope it would also drain the global stores in flight, microseconds on the critical path.)
    auto signal = [&](uint32_t n) __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __hip_atomic_store(sig_own, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // (A partner that never arrives would hang the GPU: after ~1 s of polling a wave stops waiting for good and runs out on whatever
    //  it reads -- a broken build fails its parity tests instead of the box.  A healthy wait lasts a phase of a tick, microseconds.)
#ifdef MI_NUTS_SPLIT_PROF   // phase clocks of workgroup 0, every wave (tools/nuts_prof.py; a variant build, never the shipped library)
    unsigned long long pc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tmark = clock64();
#define MI_SPROF(k) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long tn_ = clock64(); pc[k] += tn_ - tmark; tmark = tn_; }
#define MI_SPROF_WAIT0 const unsigned long long tw0_ = clock64();
#define MI_SPROF_WAIT1 pc[10] += clock64() - tw0_;
#else
#define MI_SPROF(k)
#define MI_SPROF_WAIT0
#define MI_SPROF_WAIT1
#endif
    bool sync_lost = false;
    auto await = [&](uint32_t n) __attribute__((always_inline)) {
        MI_SPROF_WAIT0
        uint32_t spins = 0;
        while (!sync_lost && __hip_atomic_load(sig_par, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < n) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 24)) sync_lost = true;
        }
        asm volatile("" ::: "memory");
        MI_SPROF_WAIT1
    };
    // workspace: [tile][vector] blocks of NS * 512 bytes, wave-uniform base + one 32-bit byte offset per access; inside a vector
    // [chain][pair of slices][j4] in 16-byte granules (nuts_async.hpp): wave h moves the second / first 512 bytes of a chain's KB
    const int tile_u = tile;
    char* const ws_tile_u = reinterpret_cast<char*>(__builtin_assume_aligned(prm.ws, 256)) + ((size_t)blockIdx.x * TPW + tile_u) * ((size_t)WS_NVEC * NS * 512);
    uint32_t lane_b = (uint32_t)(lane & 15) * (uint32_t)(NS * 32) + (uint32_t)j4 * 16u + (uint32_t)h * (uint32_t)(NSO * 32);   // redefined (opaquely) at the top of every tick
    auto wsp = [&](int v, int k) -> double* {                // k even: the pair (k, k + 1) of this lane's OWN slices
        return reinterpret_cast<double*>(ws_tile_u + ((uint32_t)v * (uint32_t)(NS * 512) + lane_b + (uint32_t)(k >> 1) * 64u));
    };
    auto ld_row = [&](int v, auto& dst) __attribute__((always_inline)) {
        constexpr int N = (int)(sizeof(dst) / sizeof(double));
        static_assert(N == NSO, "rows are the wave's own half");
#pragma unroll
        for (int k = 0; k < N; k += 2) {
            const double2 t = *reinterpret_cast<const double2*>(wsp(v, k));
            dst[k] = t.x; dst[k + 1] = t.y;
        }
    };
    auto st_row = [&](int v, const auto& src) __attribute__((always_inline)) {
        constexpr int N = (int)(sizeof(src) / sizeof(double));
        static_assert(N == NSO, "rows are the wave's own half");
#pragma unroll
        for (int k = 0; k < N; k += 2) *reinterpret_cast<double2*>(wsp(v, k)) = double2{src[k], src[k + 1]};
    };
    auto st_pair = [&](int v, int k, double a, double b) __attribute__((always_inline)) {
        *reinterpret_cast<double2*>(wsp(v, k)) = double2{a, b};
    };
    auto dim_ok = [&](int k) -> bool { return (uint32_t)(4 * (s0 + k) + j4) < d; };

    // the chain's last leaf, own slices: position, momentum, P * position (MFMA B / D layout).  Loop-carried (nuts_reg.hpp).
    double th[NSO], pm[NSO], w[NSO];

    // ---- fragment sources of the own row tiles: tt < NTL in LDS, the others in the packed global copy
    typedef const double __attribute__((address_space(3)))* lds_cptr;
    uint32_t a_off = (uint32_t)(uintptr_t)(lds_cptr)(lds_P + lane) + (uint32_t)(h * NTL * NS * 64 * 8);
    const double* const gfrag_own = prm.Pfrag + ((size_t)(h * NTO + NTL) * NS) * 64;       // wave-uniform

    // w(own rows) = P(own rows, :) * theta, theta = (own half in registers, partner's half through LDS).  Synchronisation point.
    // INVARIANT the exchange relies on: between two calls there is at least one relay() -- its hand-overs order the partner's reads of
    // this wave's buffer (during ITS mat-vec) before this wave's next write, and this wave's reads before the partner's writes.
    constexpr int PD = 3;                                            // global fragments in flight, in slices
    auto gradient = [&](double* early, auto&& early_body) __attribute__((always_inline)) {
        const uint32_t n = ++nsync;
        uint32_t gb_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)gfrag_own);
        uint32_t gb_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uintptr_t)gfrag_own >> 32));
        asm volatile("" : "+s"(gb_lo), "+s"(gb_hi));
        asm volatile("" : "+v"(a_off));
        const lds_cptr afr = (lds_cptr)(uintptr_t)a_off;
        // (an explicitly GLOBAL pointer: rebuilt from integers it would be a flat one, and behind an outstanding flat load every LDS
        //  wait also waits for memory)
        typedef const double __attribute__((address_space(1)))* glb_cptr;
        const glb_cptr gbase = (glb_cptr)(((uintptr_t)gb_hi << 32) | (uintptr_t)gb_lo);
        const uint32_t lane_ = (uint32_t)(threadIdx.x & 63);
        auto gfr = [&](int tt, int s) -> double { return gbase[(uint32_t)(((tt - NTL) * NS + s) * 64) + lane_]; };
        auto lfr = [&](int tt, int s) -> double { return afr[(tt * NS + s) * 64]; };
        double ag[PD + 1][NTG > 0 ? NTG : 1];
#pragma unroll
        for (int p = 0; p < PD; ++p)                                  // the first global fragments: requested before the exchange
#pragma unroll
            for (int tt = NTL; tt < NTO; ++tt) ag[p][tt - NTL] = gfr(tt, p);
#pragma unroll
        for (int k = 0; k < NSO; ++k) x_own[k * 64] = th[k];
        if (early != nullptr && h == 0) { *early = 0.0; early_body(*early); x_own[NSO * 64] = *early; }
        signal(n);
        double4_t acc[NTO];
        double al_cur[NTL], al_nxt[NTL];
#pragma unroll
        for (int tt = 0; tt < NTO; ++tt) acc[tt] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int tt = 0; tt < NTL; ++tt) al_cur[tt] = lfr(tt, 0);
        await(n);
#ifdef MI_NUTS_SPLIT_PROF
        { const unsigned long long tn_ = clock64(); pc[9] += tn_ - tmark; tmark = tn_; }     // (exchange + wait for the partner, booked under "loop head")
#endif
        if (early != nullptr && h == 1) { *early = x_par[NSO * 64]; early_body(*early); }
        auto half = [&](auto first_c, auto own_c) __attribute__((always_inline)) {
            constexpr int sb = decltype(first_c)::value ? 0 : NSO;
            constexpr bool own = decltype(own_c)::value;
            double b_cur = own ? 0.0 : x_par[0], b_nxt = 0.0;
#pragma unroll
            for (int k = 0; k < NSO; ++k) {
                const int s = sb + k;
                if (s + PD < NS) {
#pragma unroll
                    for (int tt = NTL; tt < NTO; ++tt) ag[(s + PD) % (PD + 1)][tt - NTL] = gfr(tt, s + PD);
                }
                if (s + 1 < NS) {
#pragma unroll
                    for (int tt = 0; tt < NTL; ++tt) al_nxt[tt] = lfr(tt, s + 1);
                }
                if (!own && k + 1 < NSO) b_nxt = x_par[(k + 1) * 64];
                __builtin_amdgcn_sched_barrier(0);
                const double b = own ? th[k] : b_cur;
#pragma unroll
                for (int tt = 0; tt < NTL; ++tt) acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(al_cur[tt], b, acc[tt], 0, 0, 0);
#pragma unroll
                for (int tt = NTL; tt < NTO; ++tt) acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ag[s % (PD + 1)][tt - NTL], b, acc[tt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tt = 0; tt < NTL; ++tt) al_cur[tt] = al_nxt[tt];
                b_cur = b_nxt;
            }
        };
        if (h == 0) { half(std::true_type{}, std::true_type{}); half(std::false_type{}, std::false_type{}); }
        else { half(std::true_type{}, std::false_type{}); half(std::false_type{}, std::true_type{}); }
#pragma unroll
        for (int tt = 0; tt < NTO; ++tt) {
            w[4 * tt + 0] = acc[tt][0]; w[4 * tt + 1] = acc[tt][1]; w[4 * tt + 2] = acc[tt][2]; w[4 * tt + 3] = acc[tt][3];
        }
    };
    // N dot products in the order of dot4 (hmc_dense.hpp) over ALL slices: `body(q)` continues the N fma chains of this lane over the
    // wave's own slices.  Wave 0 starts them at zero and leaves them in the PARTNER's exchange buffer (which it has finished reading:
    // its mat-vec is behind it); wave 1 continues, adds the lane classes (q0+q2)+(q1+q3) and leaves the results in wave 0's buffer.
    // Synchronisation point; both waves return the same bits.
    auto relay = [&](auto& q, auto&& body, double* fin1 = nullptr) __attribute__((always_inline)) {
        constexpr int N = (int)(sizeof(q) / sizeof(double));
        static_assert(N + 1 <= XS, "relay cells live in the exchange buffers");
        const uint32_t n = ++nsync;
        if (h == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i) q[i] = 0.0;
        } else {
            await(n);
#pragma unroll
            for (int i = 0; i < N; ++i) q[i] = x_own[i * 64];
        }
        body(q);
        if (h == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i) x_par[i * 64] = q[i];
            signal(n);
            await(n);
#pragma unroll
            for (int i = 0; i < N; ++i) q[i] = x_own[i * 64];
            if (fin1 != nullptr) *fin1 = x_own[N * 64];
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                q[i] = q[i] + __shfl_xor(q[i], 32);
                q[i] = q[i] + __shfl_xor(q[i], 16);
                x_par[i * 64] = q[i];
            }
            if (fin1 != nullptr) {                       // a chain wave 1 finished earlier (gradient's early chain): same treatment
                double f = *fin1;
                f = f + __shfl_xor(f, 32);
                f = f + __shfl_xor(f, 16);
                x_par[N * 64] = f;
                *fin1 = f;
            }
            signal(n);
        }
    };

    // ---------------------------------------------------------------- setup (nuts.cpp:156-195), all chains together
#pragma unroll
    for (int k = 0; k < NSO; ++k) {
        const uint32_t dimc = dim_ok(k) ? (uint32_t)(4 * (s0 + k) + j4) : 0u;
        const double v = prm.theta[(size_t)dimc * C + cld];
        th[k] = dim_ok(k) ? v : 0.0;
    }
    auto no_early = [](double&) {};
    gradient(nullptr, no_early);
    if (live) { st_row(V_PREV, th); st_row(V_WPREV, w); }
    double prev_U;
    {
        double q[1];
        relay(q, [&](auto& q_) {
#pragma unroll
            for (int k = 0; k < NSO; ++k) q_[0] = dfma(th[k], w[k], q_[0]);
        });
        prev_U = 0.5 * q[0];                             // nuts.cpp:181 (no finiteness guard there)
    }
    // Non-finite regime: detected through the energies, flagged, replayed by the general variant (nuts_reg.hpp)
    lds_nf[ct] = is_finite(prev_U) ? 0u : 1u;
    auto note_nonfinite = [&](bool bad) __attribute__((always_inline)) { if (__ballot(bad) != 0ull) { if (bad) lds_nf[ct] = 1u; } };
    // K = p.p / 2 and (optionally) U = theta . P theta / 2 of the register state: one relay
    auto kinetic = [&]() __attribute__((always_inline)) -> double {
        double q[1];
        relay(q, [&](auto& q_) {
#pragma unroll
            for (int k = 0; k < NSO; ++k) q_[0] = dfma(pm[k], pm[k], q_[0]);
        });
        return q[0] / 2.0;
    };

    uint64_t n_leap = 0;
    double eps;
    if (prm.draw0 == 0) {   // nuts_find_initial_step_size (nuts.ipp:30-93) from (first_draw, z_init), nuts.cpp:166-172
        auto leapfrog = [&](double e) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < NSO; ++k) pm[k] = pm[k] - (e * w[k]) / 2.0;
#pragma unroll
            for (int k = 0; k < NSO; ++k) th[k] = th[k] + e * pm[k];
            gradient(nullptr, no_early);
#pragma unroll
            for (int k = 0; k < NSO; ++k) pm[k] = pm[k] - (e * w[k]) / 2.0;
        };
        auto energy = [&]() __attribute__((always_inline)) -> double {
            double q[2];
            relay(q, [&](auto& q_) {
#pragma unroll
                for (int k = 0; k < NSO; ++k) { q_[0] = dfma(th[k], w[k], q_[0]); q_[1] = dfma(pm[k], pm[k], q_[1]); }
            });
            double u = 0.5 * q[0];
            if (!is_finite(u)) u = INF;
            return u + q[1] / 2.0;
        };
#pragma unroll
        for (int b = 0; b < NSO / 2; ++b) {
            const int bb = s0 / 2 + b;                   // Philox block of the chain: dimensions 8 bb + j and 8 bb + 4 + j
            double z0, z1;
            rng_normal_pair_at(prm.seed, chain, 0u, (uint32_t)(4 * bb), (uint32_t)j4, STREAM_INIT, z0, z1);
            pm[2 * b] = (8u * bb + j4 < d) ? z0 : 0.0;
            pm[2 * b + 1] = (8u * bb + 4 + j4 < d) ? z1 : 0.0;
            __builtin_amdgcn_sched_barrier(0);
        }
        double U0 = prev_U;
        if (!is_finite(U0)) U0 = INF;
        const double K0 = kinetic();
        const double log_half = det_log(0.5), neg_log2 = -det_log(2.0);
        eps = 1.0;
        leapfrog(eps);
        n_leap++;
        double dH = -energy() + (U0 + K0);
        note_nonfinite(!is_finite(dH));
        int a_val = 2 * (dH > log_half ? 1 : 0) - 1;
        bool cond = dH > neg_log2;
        while (__ballot(cond) != 0ull) {
            const double e_new = eps * ((a_val == 1) ? 2.0 : 0.5);
            if (cond) { eps = e_new; n_leap++; }
            leapfrog(eps);
            const double dH2 = -energy() + (U0 + K0);
            note_nonfinite(cond && !is_finite(dH2));
            if (cond) {
                a_val = 2 * (dH2 > log_half ? 1 : 0) - 1;
                cond = dH2 > neg_log2;
            }
        }
    } else {                // continuation of an adapted run (mi_chains.draw0 > n_adapt_draws): the step size comes back in
        eps = (live && prm.step_out) ? prm.step_out[cl] : 1.0;
    }
    // per-wave scalars: touched once per doubling or per draw, so not in registers
    auto h_val_ = [&]() -> double& { return pw(0); };
    auto eps_bar_ = [&]() -> double& { return pw(1); };
    auto mu_val_ = [&]() -> double& { return pw(2); };
    auto prev_K_ = [&]() -> double& { return pw(3); };
    auto n_val_ = [&]() -> double& { return pw(4); };
    auto alpha_ = [&]() -> double& { return pw(5); };
    auto n_alpha_ = [&]() -> double& { return pw(6); };
    mu_val_() = det_log(10 * eps);                       // nuts.cpp:174
    h_val_() = 0.0;
    eps_bar_() = (prm.draw0 == 0) ? prm.eps_bar0 : eps;
    if (prm.draw0 > 0 && prm.draw0 <= prm.n_adapt && prm.adapt_state != nullptr) {      // a continuation inside the adaptation window
        h_val_() = prm.adapt_state[cld]; eps_bar_() = prm.adapt_state[C + cld]; mu_val_() = prm.adapt_state[2 * C + cld];
    }
    uint64_t n_acc = 0;
    const uint32_t n_total = prm.n_burnin + prm.n_keep;
    const uint32_t n_adapt = prm.n_adapt;
    const uint32_t max_depth = prm.max_depth;

    // ---------------------------------------------------------------- per-chain state (nuts_reg.hpp: the same machine)
    int state = (n_total > 0) ? NS_NEED_DRAW : NS_DONE;
    uint32_t draw = 0, jd = 0, li = 0, uslot = 0;
    int vdir = 1;
    double e_signed = 0.0, H0 = 0.0, log_u = 0.0;
    int good_round = 0;
    uint32_t utpre = 0;
    int mv = V_MNTM, mvn = V_MNTM2;
    int pb = 0, pb0 = 0;
    bool mom_ready = false;
    double next_K = 0.0, next_lu = 0.0;
    bool row_pend = false, row2_pend = false;
    uint32_t row_draw = 0;
    bool pos_init = true, neg_init = true;
    auto pvec = [](int b) -> int { return b ? V_PREVB : V_PREV; };
    auto wvec = [](int b) -> int { return b ? V_WPREVB : V_WPREV; };

    auto begin_doubling = [&](bool p) __attribute__((always_inline)) {      // direction draw, nuts.cpp:233-235
        const double zdir = rng_uniform(prm.seed, chain, draw + prm.draw0, uslot);
        if (p) {
            uslot++;
            vdir = (zdir <= 0.5) ? -1 : 1;
            e_signed = (double)vdir * eps;
            H0 = prev_U + prev_K_();
            li = 0;
        }
    };
    auto end_draw = [&](bool p, uint32_t my_depth) __attribute__((always_inline)) {     // dual averaging nuts.cpp:294-302
        if (p && prm.depth_trace && live && j4 == 0 && h == 0) prm.depth_trace[(size_t)draw * C + cl] = my_depth;
        if (__ballot(p && draw + prm.draw0 < n_adapt) != 0ull) {
            if (p && draw + prm.draw0 < n_adapt) {
                const double it = (double)(draw + prm.draw0 + 1);
                const double h_new = h_val_() + (1.0 / (it + prm.t0)) * (prm.delta - (alpha_() / n_alpha_()) - h_val_());
                h_val_() = h_new;
                eps = det_exp(mu_val_() - h_new * __builtin_sqrt(it) / prm.gamma);
                const double eb = eps_bar_();
                eps_bar_() = eb * det_exp(det_pow(it, -prm.kappa) * (det_log(eps) - det_log(eb)));
            }
        }
        if (p && !(draw + prm.draw0 < n_adapt)) eps = eps_bar_();
        const bool kept = p && draw >= prm.n_burnin;
        if (kept) n_acc += (uint64_t)good_round;
        if (p) {
            row2_pend = kept && prm.draws != nullptr;
            draw++;
        }
    };
    auto store_row = [&](bool p, int vec, uint32_t idx) __attribute__((always_inline)) {     // kept row (nuts.cpp:306-309), own slices
        if (__ballot(p) == 0ull) return;
        if (p && live) {
            double* out = prm.draws + (size_t)(idx - prm.n_burnin) * d * C;
            double tmp[NSO];
            ld_row(vec, tmp);
#pragma unroll
            for (int k = 0; k < NSO; ++k)
                if (dim_ok(k)) (out + (size_t)(4 * (s0 + k)) * C)[lane_off] = tmp[k];
        }
    };
    auto roll_state = [&](bool p) __attribute__((always_inline)) {          // enter the next draw (nuts.cpp:200-219)
        if (p) {
            const int t_ = mv; mv = mvn; mvn = t_;
            prev_K_() = next_K;
            log_u = next_lu - prev_U - next_K;            // :206
            mom_ready = false;
            row_pend = row2_pend; row_draw = draw - 1u; row2_pend = false;
            pb0 = pb; pos_init = true; neg_init = true;
            uslot = 1;
            jd = 0; n_val_() = 1.0; alpha_() = 0.0; n_alpha_() = 0.0; good_round = 0;
            state = NS_TREE;
        }
    };

#pragma unroll 1
    while (__ballot(state != NS_DONE) != 0ull) {
        asm volatile("" : "+v"(lane_b));
        MI_SPROF(9)
        // ------------------------------------------------------------ A. the phase: rows, momenta ahead, waiting chains start
        if (__ballot(state == NS_NEED_DRAW) != 0ull) {
            store_row(row_pend, pvec(pb0), row_draw);
            store_row(row2_pend, pvec(pb), draw - 1u);
            row_pend = false; row2_pend = false;
            if (state == NS_NEED_DRAW && draw >= n_total) state = NS_DONE;
            const uint32_t nidx = draw + ((state == NS_TREE) ? 1u : 0u);     // the draw the momentum is for
            const bool gen = state != NS_DONE && !mom_ready && nidx < n_total;
            double kq[1];
            relay(kq, [&](auto& q_) {                         // the chain of :204 runs through the normals as they are drawn
#pragma unroll 1
                for (int b = 0; b < NSO / 2; ++b) {           // nuts.cpp:200-202, own dimensions
                    const int bb = s0 / 2 + b;
                    double z0, z1;
                    rng_normal_pair(prm.seed, chain, nidx + prm.draw0, (uint32_t)(4 * bb + j4), STREAM_NORMAL, z0, z1);
                    const double pa = (8u * bb + j4 < d) ? z0 : 0.0;
                    const double pb_ = (8u * bb + 4 + j4 < d) ? z1 : 0.0;
                    q_[0] = dfma(pa, pa, q_[0]);
                    q_[0] = dfma(pb_, pb_, q_[0]);
                    if (gen && live) st_pair(mvn, 2 * b, pa, pb_);
                }
            });
            const double lu = det_log(rng_uniform(prm.seed, chain, nidx + prm.draw0, 0u));
            if (gen) { next_K = kq[0] / 2.0; next_lu = lu; mom_ready = true; }     // :204
            const bool p = state == NS_NEED_DRAW;
            roll_state(p);
            if (max_depth > 0) begin_doubling(p);
            else { end_draw(p, 0u); if (p) state = NS_NEED_DRAW; }              // while-loop of :227 never entered
        }
        const bool run = state == NS_TREE;
        MI_SPROF(0)
        if (__ballot(run) == 0ull) continue;
#ifdef MI_NUTS_SPLIT_PROF
        pc[11]++;
#endif

        // ------------------------------------------------------------ B. one leaf for every running chain
        auto slot_of = [&](uint32_t k) -> int { return (k == 0) ? 0 : (__builtin_ctz(k) + 1); };
        const int slot_i = slot_of(li);
        const int rec_t = V_LEAF0 + 3 * slot_i, rec_p = rec_t + 1, rec_w = rec_t + 2;    // this leaf's record (even leaves only)
        const bool odd = (li & 1u) != 0u;
        // EAGER U-turn tests (nuts_reg.hpp): an even leaf li > 0 is the first leaf b2 of the second half of the level-(ctz(li) + 1) node
        // whose first leaf is b = li - 2^ctz(li); an odd leaf is b2 of its own level-1 node with b = the start state of this leapfrog
        const uint32_t cz_i = (li == 0) ? 0u : (uint32_t)__builtin_ctz(li);
        const bool eager = run && !odd && li != 0u && (cz_i + 1u <= jd);
        const uint32_t bleaf = li - (1u << cz_i);
        const int sb = (!eager || bleaf == 0) ? 0 : (__builtin_ctz(bleaf) + 1);
        const int eb_t = V_LEAF0 + 3 * sb, eb_p = eb_t + 1;           // (theta, p) of leaf b for the eager lanes
        // U-turn operands of the tick: d = theta(b2) - theta(b) and p(b); after the relay the staging registers of record copies
        double dd[NSO], Lp[NSO];
#pragma unroll
        for (int k = 0; k < NSO; ++k) { dd[k] = 0.0; Lp[k] = 0.0; }
        {   // ONE round trip at the top of the tick: start records (li == 0 or ctz(li) >= 2; otherwise the registers hold the previous
            // leaf) and the eager operands -- theta(b) into the registers of d, p(b) into Lp.  (nuts_reg.hpp hides the operands' latency
            // under its mat-vec; here the other tile of the SIMD runs meanwhile, and nothing may be in flight in front of the mat-vec's
            // fragment stream: vmcnt is in order.)
            const int cz = (li == 0) ? 0 : __builtin_ctz(li);
            const bool need = run && (li == 0 || cz >= 2);
            if (__ballot(need) != 0ull) {
                const int vt = (li == 0) ? pvec(pb) : V_LEAF0 + 3 * cz;                  // leaf li - 2^(cz-1) sits in slot cz
                const int vp = (li == 0) ? mv : V_LEAF0 + 3 * cz + 1;
                const int vw = (li == 0) ? wvec(pb) : V_LEAF0 + 3 * cz + 2;
                if (need) { ld_row(vt, th); ld_row(vp, pm); ld_row(vw, w); }
            }
            if (__ballot(eager) != 0ull) { if (eager) { ld_row(eb_t, dd); ld_row(eb_p, Lp); } }
        }
        MI_SPROF(1)
        // one leapfrog of signed size e (nuts.ipp:132, nuts.cpp:139-154), grad = -w;  d = theta(b2) - theta(b) (by direction), Lp = p(b).
        // d . p(b) needs nothing of the mat-vec: its chain rides the exchange of theta (wave 0's partial sum travels with its slices,
        // wave 1 finishes it before its MFMAs), so p(b) is dead across the mat-vec -- 32 registers the kernel does not have.
#pragma unroll
        for (int k = 0; k < NSO; ++k) {
            const double p0 = pm[k], t0 = th[k];
            pm[k] = p0 - (e_signed * w[k]) / 2.0;
            th[k] = t0 + e_signed * pm[k];
            const double tb = eager ? dd[k] : t0;
            dd[k] = (vdir > 0) ? (th[k] - tb) : (tb - th[k]);
            Lp[k] = eager ? Lp[k] : p0;
        }
        MI_SPROF(2)
        double q1c = 0.0;                                // wave 0: its partial chain; wave 1: the finished chain, lane classes not yet added
        gradient(&q1c, [&](double& q_) {
#pragma unroll
            for (int k = 0; k < NSO; ++k) q_ = dfma(dd[k], Lp[k], q_);
        });
        MI_SPROF(3)
#pragma unroll
        for (int k = 0; k < NSO; ++k) pm[k] = pm[k] - (e_signed * w[k]) / 2.0;
        double q4[4];                                    // d.p(b2), theta.P theta, p.p by relay; d.p(b) rides along finished
        {
            double q3[3];
            relay(q3, [&](auto& q_) {
#pragma unroll
                for (int k = 0; k < NSO; ++k) {
                    q_[0] = dfma(dd[k], pm[k], q_[0]);
                    q_[1] = dfma(th[k], w[k], q_[1]);
                    q_[2] = dfma(pm[k], pm[k], q_[2]);
                }
            }, &q1c);
            q4[0] = q1c; q4[1] = q3[0]; q4[2] = q3[1]; q4[3] = q3[2];
        }
        MI_SPROF(4)
        double pU = 0.5 * q4[2];                         // nuts.ipp:134-138
        const double pK = q4[3] / 2.0;                   // :140
        if (!is_finite(pU)) pU = INF;
        const bool ut_now = (q4[0] >= 0.0) && (q4[1] >= 0.0);  // odd leaf: its level-1 test; eager even leaf: the test of level ctz(li) + 1
        if (eager) utpre = (utpre & ~(1u << (cz_i + 1u))) | ((ut_now ? 1u : 0u) << (cz_i + 1u));
        if (run && live && !odd) {                       // even leaves are the records later leaves and tests read
            st_row(rec_t, th); st_row(rec_p, pm); st_row(rec_w, w);
        }
        // the tree's far edge is what a successful doubling leaves in draw_pos / draw_neg (src/nuts.cpp:241-256)
        const bool at_edge = run && (li == ((jd == 0u) ? 0u : (1u << (jd - 1))));
        if (at_edge && live) {
            const int et = (vdir > 0) ? V_TPOS_T : V_TNEG_T, ep = (vdir > 0) ? V_TPOS_P : V_TNEG_P;
            st_row(et, th); st_row(ep, pm);
        }
        if (at_edge) { if (vdir > 0) pos_init = false; else neg_init = false; }
        double cn = (log_u <= -pU - pK) ? 1.0 : 0.0;     // :146
        const bool cs = log_u < 1000.0 - pU - pK;        // :147
        const double dH = -(pU + pK) + H0;
        note_nonfinite(run && !is_finite(dH));
        double ca = det_exp((dH < 0.0) ? dH : 0.0);      // :157
        double cna = 1.0;
        double cU = pU;
        bool cref_regs = true;                           // carried proposal: this leaf (registers) ...
        int cref_t = rec_t, cref_w = rec_w;              // ... or a record (theta, P*theta)
        if (run) n_leap++;
        MI_SPROF(5)
        // ---- unwind (nuts.ipp:212-229), per-chain leaf index
        bool failed = run && !cs;
        bool walking = run;
        uint32_t pend_level = jd + 1;
#pragma unroll 1
        for (uint32_t l = 1; l <= (uint32_t)NUTS_MAX_DEPTH; ++l) {
            if (walking && l > jd) walking = false;                      // reached the root of its own tree
            const bool bit = ((li >> (l - 1)) & 1u) != 0u;
            if (walking && !failed && !bit) { pend_level = l; walking = false; }   // first half: wait here
            if (__ballot(walking) == 0ull) break;
            const bool mrg = walking && bit;
            if (__ballot(mrg) == 0ull) continue;
            const double z = rng_uniform(prm.seed, chain, draw + prm.draw0, uslot);  // :213
            if (mrg) {
                uslot++;
                const double p_n = lvl((int)l, 0), p_a = lvl((int)l, 1), p_na = lvl((int)l, 2), p_U = lvl((int)l, 3);
                const double prob = cn / (p_n + cn);                     // :212
                if (!(z < prob)) {                                       // keep new_draw_p (:215-217)
                    const int ps = slot_of(li - 1);                      // level 1: the previous (even) leaf's record
                    cref_regs = false;
                    cref_t = (l == 1) ? V_LEAF0 + 3 * ps : V_PP0 + (int)l;
                    cref_w = (l == 1) ? V_LEAF0 + 3 * ps + 2 : V_PPW0 + (int)l;
                    cU = p_U;
                }
                cn = p_n + cn;                                           // :220-222
                ca = p_a + ca;
                cna = p_na + cna;
            }
            const bool need_ut = mrg && !failed;
            const bool ok = (l == 1) ? ut_now : (((utpre >> l) & 1u) != 0u);      // :226-227
            if (need_ut && !ok) failed = true;                                   // :229
        }
        MI_SPROF(6)
        // ---- end of the doubling? top-level accept first (src/nuts.cpp:260-279)
        const bool keep = run && !failed;
        const bool complete = keep && (li == (1u << jd) - 1u);
        const bool fin = run && (failed || complete);
        bool take = false;
        if (__ballot(complete) != 0ull) {
            const double z = rng_uniform(prm.seed, chain, draw + prm.draw0, uslot);  // :261
            if (complete) {
                uslot++;
                take = z < cn / n_val_();                                   // :263
                if (take) { prev_U = cU; good_round = 1; pb = 1 - pb0; }  // :264-277; the proposal goes to pvec(pb) below
            }
        }
        // ---- pending first half: proposal and its P*theta by value, scalars to LDS (both waves of the tile write the same values)
        if (keep && !complete) {
            lvl((int)pend_level, 0) = cn; lvl((int)pend_level, 1) = ca;
            lvl((int)pend_level, 2) = cna; lvl((int)pend_level, 3) = cU;
        }
        {
            const bool do_store = keep && (complete ? take : (pend_level > 1u)) && live;
            if (__ballot(do_store) != 0ull) {
                const int pl = do_store ? (int)pend_level : 1;
                const int dst_t = take ? pvec(1 - pb0) : V_PP0 + pl;
                const int dst_w = take ? wvec(1 - pb0) : V_PPW0 + pl;
                if (do_store && cref_regs) { st_row(dst_t, th); st_row(dst_w, w); }
                const bool do_copy = do_store && !cref_regs;
                if (__ballot(do_copy) != 0ull) {
                    if (do_copy) {       // both rows in ONE round trip, through the registers of d and p(b) (dead since the relay)
                        ld_row(cref_t, dd); ld_row(cref_w, Lp);
                        st_row(dst_t, dd); st_row(dst_w, Lp);
                    }
                }
            }
        }
        MI_SPROF(7)
        if (__ballot(fin) != 0ull) {
            if (fin) { alpha_() = ca; n_alpha_() = cna; n_val_() = n_val_() + cn; }   // :246,255 ; :283
            bool s_ok = false;
            if (__ballot(complete) != 0ull) {
                const int en_t = neg_init ? pvec(pb0) : V_TNEG_T, en_p = neg_init ? mv : V_TNEG_P;
                const int ep_t = pos_init ? pvec(pb0) : V_TPOS_T, ep_p = pos_init ? mv : V_TPOS_P;
                // [ (pos - neg) . p_neg >= 0 ] * [ (pos - neg) . p_pos >= 0 ] (:286-289): the leaf state of a chain whose doubling is
                // complete is dead (the next doubling starts from prev_draw), so its registers take the operands in ONE round trip
                if (complete) { ld_row(en_t, th); ld_row(en_p, pm); ld_row(ep_t, w); ld_row(ep_p, dd); }
                double q2[2];
                relay(q2, [&](auto& q_) {
#pragma unroll
                    for (int k = 0; k < NSO; ++k) {
                        const double dd_ = w[k] - th[k];
                        q_[0] = dfma(dd_, pm[k], q_[0]);
                        q_[1] = dfma(dd_, dd[k], q_[1]);
                    }
                });
                s_ok = complete && (q2[0] >= 0.0) && (q2[1] >= 0.0);
            }
            const bool more = fin && s_ok && (jd + 1 < max_depth);
            if (fin) jd = jd + 1;                                        // :284
            const bool ended = fin && !more;
            bool roll = false;
            if (__ballot(ended) != 0ull) {
                end_draw(ended, jd);
                roll = ended && draw < n_total && mom_ready && !row_pend;
                if (ended && !roll) state = NS_NEED_DRAW;
                roll_state(roll);
            }
            begin_doubling(more || roll);
        }
        if (run && !fin) li = li + 1;
        MI_SPROF(8)
    }
#ifdef MI_NUTS_SPLIT_PROF
    if (prm.prof && blockIdx.x == 0 && lane == 0)
        for (int k = 0; k < 12; ++k) prm.prof[wave * 12 + k] = pc[k];
#endif

    // the pair-wise wait timed out (a broken build, never seen): the run is void.  The status word behind the flags makes it LOUD: the host enqueues
    // trap_if_sync_lost_kernel behind this launch, which aborts the queue -- the caller's next synchronisation fails instead of returning garbage
    if (sync_lost && prm.nf_flag != nullptr && lane == 0) prm.nf_flag[C + 1] = 0xdeadu;
    const bool replay = lds_nf[ct] != 0u && prm.nf_flag != nullptr;
    if (live && replay && j4 == 0 && h == 0) { prm.nf_flag[cl] = 1u; prm.nf_flag[C] = 1u; }
    if (live && !replay) {
        double tmp[NSO];
        ld_row(pvec(pb), tmp);
#pragma unroll
        for (int k = 0; k < NSO; ++k)
            if (dim_ok(k)) prm.theta[(size_t)(4 * (s0 + k)) * C + lane_off] = tmp[k];
        if (j4 == 0 && h == 0) {
            if (prm.n_accept) prm.n_accept[cl] = n_acc;
            if (prm.n_leap) prm.n_leap[cl] = n_leap;
            if (prm.step_out) prm.step_out[cl] = eps;
            if (prm.adapt_state) { prm.adapt_state[cl] = h_val_(); prm.adapt_state[C + cl] = eps_bar_(); prm.adapt_state[2 * C + cl] = mu_val_(); }
        }
    }
}

}  // namespace mi
