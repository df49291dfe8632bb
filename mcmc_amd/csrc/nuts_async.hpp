// nuts_async.hpp -- many-chain NUTS, asynchronous per-chain state machine (the production NUTS kernel).
//
// Same algorithm, arithmetic and workspace layout as nuts_dense.hpp (see the derivation of the
// iterative tree there; reference: /root/reference/src/nuts.cpp:30-332, include/mcmc/nuts.ipp:30-241),
// but the 16 chains of a wavefront no longer wait for each other at draw or doubling boundaries.
// Every chain carries its own (draw, doubling depth j, leaf index i, direction, step size, RNG slot);
// one "tick" of the wave = one leapfrog (one MFMA mat-vec) for every chain that is inside a tree,
// each on its own state.  Tree sizes differ by up to 2^9 leapfrogs between chains and draws
// (measured: lock-step per draw keeps only 17 % of the lanes busy at BASELINE config 4); here a chain
// that finishes its tree starts its next doubling / next draw on the following tick.
// Momentum refresh (16 Philox blocks + Box-Muller per lane) is batched: chains wait at the draw
// boundary until `refresh_batch` of them are waiting (or nothing else can run).
// P*theta of every pending proposal is stored next to it, so accepting a proposal needs no extra
// mat-vec at the next doubling.
#pragma once

#include "nuts_dense.hpp"

#ifndef MI_NUTS_CHU
#define MI_NUTS_CHU 32
#endif
#ifndef MI_NUTS_CH
#define MI_NUTS_CH 32
#endif
// chunk sizes of the two places where several vectors are in flight per chunk (a chunk that does not fit the 256 arch VGPRs
// is spilled to scratch, and a scratch reload waits behind every store of the tick: vmcnt is in order)
#ifndef MI_NUTS_CHF
#define MI_NUTS_CHF 32     // end-of-doubling U-turn dots: 4 vectors, one round trip (the tick-local state is dead there)
#endif
#ifndef MI_NUTS_CHC
#define MI_NUTS_CHC 16     // proposal copies: 2 vectors
#endif

namespace mi {

enum : int { V_PPW0 = 52, NUTS_NVEC_ASYNC = 64 };   // P*theta of the pending proposal of level l at 52 + l
enum : int { NS_NEED_DRAW = 0, NS_TREE = 1, NS_DONE = 2 };

// GENERAL: settings.vals_bound and / or a diagonal precond_mat, with the arithmetic of the general HMC variant
// (hmc_dense.hpp; the reference's NUTS shares mntm_update_fn / leap_frog_fn / box_log_kernel with HMC, nuts.cpp:84-154):
// x = inv_transform(theta) feeds the mat-vec, the kick uses J^-1_ii (P x)_i, the drift M^-1 p, K = p.(M^-1 p)/2,
// U = -(K(x) + log_jacobian(theta)) summed over dimensions in order, p = sqrt(M) z; the tree lives in the transformed space
// (the U-turn dots are plain), draws are reported through inv_transform.  Identity tables reproduce the plain kernel's bits.
// DENSE_M (with GENERAL): a dense precond_mat, as in hmc_dense.hpp: INV(M) and CHOL_LOWER(M) from the host as two more sets of
// MFMA A-fragments in LDS (d <= 64), p = L z, Minv p and the kinetic energy as mat-vecs.
template <int NT, bool GENERAL = false, bool DENSE_M = false>
__global__ MI_NO_DS_MERGE __launch_bounds__(256, 1) void nuts_gauss_async_kernel(const NutsParams prm, const uint32_t refresh_batch)
{
    static_assert(!DENSE_M || GENERAL, "the dense preconditioner rides the general variant");
    if (prm.replay_flag != nullptr && prm.replay_flag[prm.C] == 0u) return;    // a replay launch with nothing flagged
    constexpr int NS = 4 * NT;
    constexpr int WS_NVEC = NUTS_NVEC_ASYNC;
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    double* lds_P = lds_all;
    double* lds_lvl = lds_all + NT * NS * 64;            // [NUTS_LVLS][4][64]
    double* const lds_da = lds_lvl + NUTS_LVLS * 4 * 64;  // [3][64]: the dual-averaging state (h, epsilon_bar, mu), touched once per draw
    double* const lds_lb = lds_da + 3 * 64;
    double* const lds_ub = lds_lb + 16 * NT;
    double* const lds_ms = lds_ub + 16 * NT;
    double* const lds_mi = lds_ms + 16 * NT;
    int* const lds_bt = reinterpret_cast<int*>(lds_mi + 16 * NT);
    double* const lds_Minv = lds_mi + 16 * NT + 8 * NT;   // after the int table, DENSE_M only
    double* const lds_L = lds_Minv + NT * NS * 64;
    if constexpr (DENSE_M && !dense_m_from_global<NT>()) {
        stage_precision<NT>(prm.Minv, prm.d, lds_Minv);
        stage_precision<NT>(prm.Lchol, prm.d, lds_L);
    }
    if constexpr (GENERAL) {
        for (int i = threadIdx.x; i < 16 * NT; i += blockDim.x) {
            const bool in = (uint32_t)i < prm.d;
            lds_lb[i] = in ? prm.lb[i] : 0.0;
            lds_ub[i] = in ? prm.ub[i] : 0.0;
            lds_bt[i] = in ? prm.btype[i] : 1;
            lds_ms[i] = in ? prm.m_sqrt[i] : 1.0;
            lds_mi[i] = in ? prm.m_inv[i] : 1.0;
        }
    }
    stage_precision<NT>(prm.P, prm.d, lds_P);            // ends with a barrier

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // bit s: slice s (dims 4s .. 4s+3) has a bounded dimension (wave-uniform).  The log-Jacobian sum of box_log_kernel
    // (log_jacobian.hpp:36-57, a scalar loop over i ascending = 4 shuffles per slice) runs over those slices only, and not at all
    // without vals_bound (nuts.cpp:84-95 adds the term only then)
    uint32_t bslices = 0;
    if constexpr (GENERAL) {
        if (prm.vals_bound)
            for (int s_ = 0; s_ < 4 * NT; ++s_) {
                const bool any = lds_bt[4 * s_] != 1 || lds_bt[4 * s_ + 1] != 1 || lds_bt[4 * s_ + 2] != 1 || lds_bt[4 * s_ + 3] != 1;
                bslices |= (any ? 1u : 0u) << s_;
            }
        bslices = (uint32_t)__builtin_amdgcn_readfirstlane((int)bslices);
    }
    const int j4 = lane >> 4;
    const int cw = wave * 16 + (lane & 15);
    const uint64_t cl = ((uint64_t)blockIdx.x * 4 + wave) * 16 + (lane & 15);
    // replay of the chains nuts_gauss_memo_kernel flagged (non-finite regime): the others are not `live` -- they compute along from
    // whatever theta holds and store nothing -- and a wave without a flagged chain leaves (no barrier follows)
    const bool live = cl < prm.C && (prm.replay_flag == nullptr || prm.replay_flag[cl] != 0u);
    if (prm.replay_flag != nullptr && __ballot(live) == 0ull) return;
    const uint64_t cld = cl < prm.C ? cl : prm.C - 1;
    const uint64_t chain = prm.chain0 + cl;
    const uint32_t d = prm.d;
    const uint64_t C = prm.C;
    const double* afrag = lds_P + lane;
    // d > 64: prm.Minv / prm.Lchol are in fragment order in global memory (hmc_dense.hpp: matvec_mfma_g)
    [[maybe_unused]] const double* afrag_minv = (DENSE_M && dense_m_from_global<NT>()) ? prm.Minv + lane : lds_Minv + lane;
    [[maybe_unused]] const double* afrag_l = (DENSE_M && dense_m_from_global<NT>()) ? prm.Lchol + lane : lds_L + lane;
    const size_t lane_off = (size_t)j4 * C + cld;

    // 1 / m of this lane's dimension of slice s.  The lane's part of the index is opaque where the table is read: the 4 * NT values are
    // loop invariants of the tick loop otherwise, registers the general kernel does not have
    auto mi_tab = [&]() __attribute__((always_inline)) -> const double* {
        int jo = (int)((threadIdx.x & 63) >> 4);
        asm volatile("" : "+v"(jo));
        return lds_mi + jo;
    };
    auto lvl = [&](int l, int f) -> double& { return lds_lvl[(l * 4 + f) * 64 + cw]; };
    // workspace layout [wave][vec][slice][lane]: a vector of a wave's 16 chains is 16 KiB contiguous (one 512-B
    // coalesced access per slice, 4 pages per vector) instead of 128 fragments 4*C*8 bytes apart
    // Addressing: wave-uniform base (SGPR pair) + one 32-bit byte offset per access, so that every access is the
    // `global_load/store v, v_off, s[base]` form.  With a per-lane 64-bit base the compiler precomputed a 64-bit address
    // pair per (vector, slice) -- hundreds of VGPRs of loop invariants, spilled, and reloaded from scratch in front of every
    // load (a scratch reload waits behind every store in flight: vmcnt is in order).
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    char* const ws_wave_u = reinterpret_cast<char*>(__builtin_assume_aligned(prm.ws, 256)) + ((size_t)blockIdx.x * 4 + wave_u) * ((size_t)WS_NVEC * NS * 512);
    // [wave][vector][chain][pair of slices][j4] in 16-byte granules: a chain's values of a vector are one contiguous block of
    // NS * 32 bytes, its four lanes move 64 consecutive bytes per instruction, and a chain that is masked off touches no cache
    // line at all.  With [vector][slice][lane] every access moved whole 128-byte lines of 16 chains although only ~9 of a wave's
    // 16 chains are inside a tree on an average tick (25 KB per chain-leaf for 13 KB of records): config 4 went from 3.75 s to
    // 2.03 s on that change of layout alone (round 1; the same happens with [vector][pair][lane], 0.87 -> 1.55 s in round 2).
    // One row per LANE ([lane][slice], round 1) moves the same bytes with four times the cache lines per instruction: 6 % slower.
    uint32_t lane_b = (uint32_t)(lane & 15) * (uint32_t)(NS * 32) + (uint32_t)(lane >> 4) * 16u;   // redefined (opaquely) at the top of every tick
    auto wsp = [&](int v, int s) -> double* {                // s even: the pair (s, s + 1) of this lane
        return reinterpret_cast<double*>(ws_wave_u + ((uint32_t)v * (uint32_t)(NS * 512) + lane_b + (uint32_t)(s >> 1) * 64u));
    };
    auto ld_row = [&](int v, int s0, auto& dst) __attribute__((always_inline)) {      // dst[0..N) <- slices s0.. of vector v
        constexpr int N = (int)(sizeof(dst) / sizeof(double));
        static_assert(N % 2 == 0, "rows move in pairs of slices");
#pragma unroll
        for (int k = 0; k < N; k += 2) {
            const double2 t = *reinterpret_cast<const double2*>(wsp(v, s0 + k));
            dst[k] = t.x; dst[k + 1] = t.y;
        }
    };
    auto st_row = [&](int v, int s0, const auto& src) __attribute__((always_inline)) {
        constexpr int N = (int)(sizeof(src) / sizeof(double));
        static_assert(N % 2 == 0, "rows move in pairs of slices");
#pragma unroll
        for (int k = 0; k < N; k += 2) *reinterpret_cast<double2*>(wsp(v, s0 + k)) = double2{src[k], src[k + 1]};
    };
    auto st_pair = [&](int v, int s0, double a, double b) __attribute__((always_inline)) {
        *reinterpret_cast<double2*>(wsp(v, s0)) = double2{a, b};
    };
    auto dim_ok = [&](int s) -> bool { return (uint32_t)(4 * s + j4) < d; };

    double th[NS], pm[NS], w[NS];

    auto store_vec = [&](int v, const double (&x)[NS], bool pred) __attribute__((always_inline)) {
        if (pred && live) {
            st_row(v, 0, x);
        }
    };
    // vector ops touch 8 slices at a time (16 VGPRs in flight): the register file is full of theta / p / P*theta
    constexpr int CH = (NS < MI_NUTS_CH) ? NS : MI_NUTS_CH;
    constexpr int CHF = (NS < MI_NUTS_CHF) ? NS : MI_NUTS_CHF;
    constexpr int CHC = (NS < MI_NUTS_CHC) ? NS : MI_NUTS_CHC;
    constexpr int CHU = (NS < MI_NUTS_CHU) ? NS : MI_NUTS_CHU;   // U-turn operands: 4 vectors in flight per chunk
    auto copy_vec = [&](int vsrc, int vdst, bool pred) __attribute__((always_inline)) {
        if (pred && live) {
#pragma unroll
            for (int c0 = 0; c0 < NS; c0 += CH) {
                double tmp[CH];
                ld_row(vsrc, c0, tmp);
                st_row(vdst, c0, tmp);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // GENERAL helpers on the register-resident (th, pm, w): x = inv_transform(theta), w = P x, kick vector J^-1 w
    auto general_gradient = [&](double (&xs_)[GENERAL ? NS : 1]) __attribute__((always_inline)) {
        if constexpr (GENERAL) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int i = 4 * s + j4;
                if ((bslices >> s) & 1u) xs_[s] = ((uint32_t)i < d) ? box_inv_transform(th[s], lds_bt[i], lds_lb[i], lds_ub[i]) : 0.0;
                else xs_[s] = ((uint32_t)i < d) ? th[s] : 0.0;      // (uniform branch: the transform code of an unbounded slice is never fetched)
            }
            target_times<NT>(afrag, prm.P, d, prm.sep_target != 0u, xs_, w);
        }
    };
    auto kick = [&](double e) __attribute__((always_inline)) {         // p += e [J] grad / 2, grad = -w (nuts.cpp:108-135)
        if constexpr (GENERAL) {
            double kw[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int i = 4 * s + j4;
                if ((bslices >> s) & 1u) kw[s] = box_inv_jacobian(th[s], lds_bt[i], lds_lb[i], lds_ub[i]) * w[s];
                else kw[s] = 1.0 * w[s];
            }
            if (prm.vals_bound) dense_product_poison<NS>(w, kw, j4, d);   // jacob_matrix * grad_obj is a dense product
#pragma unroll
            for (int s = 0; s < NS; ++s) pm[s] = pm[s] - (e * kw[s]) / 2.0;
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s) pm[s] = pm[s] - (e * w[s]) / 2.0;
        }
    };
    auto drift = [&](double e) __attribute__((always_inline)) {        // theta += e Minv p (a dense product in the reference)
        if constexpr (GENERAL) {
            double mp[NS];
            if constexpr (DENSE_M) {
                matvec_m2<NT>(afrag_minv, pm, mp);
            } else {
const double* const mi_t = mi_tab();
#pragma unroll
                for (int s = 0; s < NS; ++s) mp[s] = mi_t[4 * s] * pm[s];
                dense_product_poison<NS>(pm, mp, j4, d);
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) th[s] = th[s] + e * mp[s];
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s) th[s] = th[s] + e * pm[s];
        }
    };
    double xs[GENERAL ? NS : 1];                                       // setup phase: x of the register-resident theta
    auto leapfrog = [&](double e) __attribute__((always_inline)) {
        kick(e);
        drift(e);
        if constexpr (GENERAL) general_gradient(xs); else matvec_mfma<NT>(afrag, th, w);
        kick(e);
    };
    // U = -box_log_kernel(theta) of the register-resident state whose x / P x are (xs_, w) (nuts.cpp:84-95)
    auto potential_raw = [&](const double (&xs_)[GENERAL ? NS : 1]) __attribute__((always_inline)) -> double {
        if constexpr (GENERAL) {
            const double kval = -0.5 * dot4<NS>(xs_, w);
            double lj = 0.0;                             // log_jacobian.hpp:36-57: scalar loop, i ascending
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (((bslices >> s) & 1u) == 0u) continue;
                const int i0 = 4 * s;
                const double term = box_log_jacobian_term(th[s], lds_bt[i0 + j4], lds_lb[i0 + j4], lds_ub[i0 + j4]);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const double tg = __shfl(term, (lane & 15) + 16 * g);
                    if ((uint32_t)(i0 + g) < d && lds_bt[i0 + g] != 1) lj = lj + tg;
                }
            }
            return -(kval + lj);
        } else {
            return 0.5 * dot4<NS>(th, w);
        }
    };
    auto potential = [&]() __attribute__((always_inline)) -> double {
        double u = potential_raw(xs);
        if (!is_finite(u)) u = INF;
        return u;
    };
    auto kinetic = [&]() __attribute__((always_inline)) -> double {    // K = p . (Minv p) / 2
        if constexpr (GENERAL) {
            double mp[NS];
            if constexpr (DENSE_M) {
                matvec_m2<NT>(afrag_minv, pm, mp);
            } else {
const double* const mi_t = mi_tab();
#pragma unroll
                for (int s = 0; s < NS; ++s) mp[s] = mi_t[4 * s] * pm[s];
                dense_product_poison<NS>(pm, mp, j4, d);
            }
            double q = 0.0;
#pragma unroll
            for (int s = 0; s < NS; ++s) q = dfma(pm[s], mp[s], q);
            q = q + __shfl_xor(q, 32);
            q = q + __shfl_xor(q, 16);
            return q / 2.0;
        } else {
            return dot4<NS>(pm, pm) / 2.0;
        }
    };
    // [ (pos - neg) . p_1 >= 0 ] * [ (pos - neg) . p_2 >= 0 ], pos/neg = (t2,t1) for v=+1, (t1,t2) for v=-1;
    // all four operands come from the level records in the workspace
    auto uturn_ok = [&](bool pred, int vt1, int vp1, int vt2, int vp2, int vdir) __attribute__((always_inline)) -> bool {
        double q1 = 0.0, q2 = 0.0;
        if (pred) {
#pragma unroll
            for (int c0 = 0; c0 < NS; c0 += CHU) {
                double t1[CHU], p1[CHU], t2[CHU], p2[CHU];
                ld_row(vt1, c0, t1); ld_row(vp1, c0, p1); ld_row(vt2, c0, t2); ld_row(vp2, c0, p2);
#pragma unroll
                for (int k = 0; k < CHU; ++k) {
                    const double dd = (vdir > 0) ? (t2[k] - t1[k]) : (t1[k] - t2[k]);
                    q1 = dfma(dd, p1[k], q1);
                    q2 = dfma(dd, p2[k], q2);
                }
                if (CHU < NS) __builtin_amdgcn_sched_barrier(0);
            }
        }
        q1 = q1 + __shfl_xor(q1, 32); q1 = q1 + __shfl_xor(q1, 16);
        q2 = q2 + __shfl_xor(q2, 32); q2 = q2 + __shfl_xor(q2, 16);
        return (q1 >= 0.0) && (q2 >= 0.0);
    };

    // ---------------------------------------------------------------- setup (nuts.cpp:156-195), all chains together
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const uint32_t dimc = dim_ok(s) ? (uint32_t)(4 * s + j4) : 0u;
        const double v = prm.theta[(size_t)dimc * C + cld];
        if constexpr (GENERAL) th[s] = dim_ok(s) ? box_transform(v, lds_bt[dimc], lds_lb[dimc], lds_ub[dimc]) : 0.0;   // nuts.cpp:160-162
        else th[s] = dim_ok(s) ? v : 0.0;
    }
    if constexpr (GENERAL) general_gradient(xs); else matvec_mfma<NT>(afrag, th, w);
    store_vec(V_PREV, th, true);
    store_vec(V_WPREV, w, true);
    double prev_U = potential_raw(xs);                   // nuts.cpp:181

    uint64_t n_leap = 0;
    double eps;
    if (prm.draw0 == 0) {   // nuts_find_initial_step_size (nuts.ipp:30-93) from (first_draw, L z_init), nuts.cpp:166-172
#pragma unroll
        for (int b = 0; b < NS / 2; ++b) {
            double z0, z1;
            rng_normal_pair(prm.seed, chain, 0u, (uint32_t)(4 * b + j4), STREAM_INIT, z0, z1);
            pm[2 * b] = (8u * b + j4 < d) ? z0 : 0.0;
            pm[2 * b + 1] = (8u * b + 4 + j4 < d) ? z1 : 0.0;
            if constexpr (GENERAL && !DENSE_M) {         // L z with a diagonal L (nuts.cpp:170)
                pm[2 * b] = lds_ms[8 * b + j4] * pm[2 * b];
                pm[2 * b + 1] = lds_ms[8 * b + 4 + j4] * pm[2 * b + 1];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (DENSE_M) {                         // L z (nuts.cpp:170)
            double zz[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) zz[s] = pm[s];
            matvec_m2<NT>(afrag_l, zz, pm);
        }
        double U0 = prev_U;
        if (!is_finite(U0)) U0 = INF;
        const double K0 = kinetic();
        const double log_half = det_log(0.5), neg_log2 = -det_log(2.0);
        eps = 1.0;
        leapfrog(eps);
        n_leap++;
        double dH = -(potential() + kinetic()) + (U0 + K0);
        int a_val = 2 * (dH > log_half ? 1 : 0) - 1;
        bool cond = dH > neg_log2;
        if (prm.replay_flag != nullptr && !live) cond = false;
        while (__ballot(cond) != 0ull) {
            const double e_new = eps * ((a_val == 1) ? 2.0 : 0.5);
            if (cond) { eps = e_new; n_leap++; }
            leapfrog(eps);
            const double dH2 = -(potential() + kinetic()) + (U0 + K0);
            if (cond) {
                a_val = 2 * (dH2 > log_half ? 1 : 0) - 1;
                cond = dH2 > neg_log2;
            }
        }
    } else {                // continuation of an adapted run (mi_chains.draw0 > n_adapt_draws): the step size comes back in
        eps = (live && prm.step_out) ? prm.step_out[cl] : 1.0;
    }
    // (in LDS, not in registers: this kernel's general instantiation at d > 64 has none to spare -- with the three scalars loop-carried
    //  in VGPRs next to the adaptation-state load and store, hipcc 7.2 produced code that returned the chain index as step size for
    //  64 < d < 128 with bounds; tests/test_gpu_parity_nuts.py pins those shapes)
    auto h_val_ = [&]() -> double& { return lds_da[cw]; };
    auto eps_bar_ = [&]() -> double& { return lds_da[64 + cw]; };
    auto mu_val_ = [&]() -> double& { return lds_da[128 + cw]; };
    mu_val_() = det_log(10 * eps);                       // nuts.cpp:174
    h_val_() = 0.0;
    eps_bar_() = (prm.draw0 == 0) ? prm.eps_bar0 : eps;
    if (prm.draw0 > 0 && prm.draw0 <= prm.n_adapt && prm.adapt_state != nullptr) {      // a continuation inside the adaptation window
        h_val_() = prm.adapt_state[cld]; eps_bar_() = prm.adapt_state[C + cld]; mu_val_() = prm.adapt_state[2 * C + cld];
    }
    uint64_t n_acc = 0;
    const uint32_t n_total = prm.n_burnin + prm.n_keep;
    const uint32_t n_adapt = prm.n_adapt;                // the run's window in GLOBAL draw indices (the clamp of nuts.cpp:54 is immaterial: it only
                                                         // matters when every draw adapts)
    const uint32_t max_depth = prm.max_depth;

    // ---------------------------------------------------------------- per-chain state
    int state = (n_total > 0 && (live || prm.replay_flag == nullptr)) ? NS_NEED_DRAW : NS_DONE;   // replay: only the flagged chains run
    uint32_t draw = 0;           // this chain's draw index
    uint32_t jd = 0;             // depth of the doubling in progress
    uint32_t li = 0;             // next leaf of that doubling
    uint32_t uslot = 0;
    int vdir = 1;
    double e_signed = 0.0, H0 = 0.0, prev_K = 0.0, log_u = 0.0, n_val = 1.0;
    double alpha_val = 0.0, n_alpha_val = 0.0;
    int good_round = 0;
    bool fin_pending = false;    // the draw's epilogue (dual averaging, row store) is done in the next refresh phase
    uint32_t fin_depth = 0;

    // start doubling jd (direction draw, nuts.cpp:233-235) for lanes with `p`
    auto begin_doubling = [&](bool p) __attribute__((always_inline)) {
        const double zdir = rng_uniform(prm.seed, chain, draw + prm.draw0, uslot);
        if (p) {
            uslot++;
            vdir = (zdir <= 0.5) ? -1 : 1;
            e_signed = (double)vdir * eps;
            H0 = prev_U + prev_K;
            li = 0;
        }
    };
    // end of a draw (dual averaging nuts.cpp:294-302, row store :306-309) for lanes with `p`
    auto finish_draw = [&](bool p, uint32_t my_depth) __attribute__((always_inline)) {
        if (__ballot(p) == 0ull) return;
        if (p && prm.depth_trace && live && j4 == 0) prm.depth_trace[(size_t)draw * C + cl] = my_depth;
        if (p) fin_pending = false;
        if (p) {
            if (draw + prm.draw0 < n_adapt) {
                const double it = (double)(draw + prm.draw0 + 1);
                const double h_new = h_val_() + (1.0 / (it + prm.t0)) * (prm.delta - (alpha_val / n_alpha_val) - h_val_());
                h_val_() = h_new;
                eps = det_exp(mu_val_() - h_new * __builtin_sqrt(it) / prm.gamma);
                const double eb = eps_bar_();
                eps_bar_() = eb * det_exp(det_pow(it, -prm.kappa) * (det_log(eps) - det_log(eb)));
            } else {
                eps = eps_bar_();
            }
        }
        const bool kept = p && draw >= prm.n_burnin;
        if (kept) n_acc += (uint64_t)good_round;
        if (__ballot(kept && prm.draws != nullptr) != 0ull) {
            if (kept && prm.draws != nullptr && live) {
                double* out = prm.draws + (size_t)(draw - prm.n_burnin) * d * C;
#pragma unroll
                for (int c0 = 0; c0 < NS; c0 += CH) {
                    double tmp[CH];
                    ld_row(V_PREV, c0, tmp);
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        if constexpr (GENERAL) {             // the stored row goes through inv_transform (nuts.cpp:320-327)
                            const int i = 4 * (c0 + k) + j4;
                            if (dim_ok(c0 + k)) tmp[k] = box_inv_transform(tmp[k], lds_bt[i], lds_lb[i], lds_ub[i]);
                        }
                        if (dim_ok(c0 + k)) (out + (size_t)(4 * (c0 + k)) * C)[lane_off] = tmp[k];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (p) {
            draw++;
            state = (draw < n_total) ? NS_NEED_DRAW : NS_DONE;
        }
    };

#ifdef MI_NUTS_ASYNC_PROF     // phase clocks of block 0, wave 0 (tools/nuts_prof.py; a variant build, never the shipped library)
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long n_ticks = 0, n_active = 0, n_refresh = 0, n_finblk = 0;
    unsigned long long tmark = clock64();
#define MI_PROF(k) { const unsigned long long tn_ = clock64(); pc[k] += tn_ - tmark; tmark = tn_; }
#define MI_PROF_COUNT(stmt) stmt
#else
#define MI_PROF(k)
#define MI_PROF_COUNT(stmt)
#endif
#pragma unroll 1
    while (__ballot(state != NS_DONE) != 0ull) {
        asm volatile("" : "+v"(lane_b));
        MI_PROF(7)
        // ------------------------------------------------------------ A. momentum refresh for waiting chains
        const unsigned n_wait = __popcll(__ballot(state == NS_NEED_DRAW)) / 4;
        const unsigned n_run = __popcll(__ballot(state == NS_TREE)) / 4;
        if (n_wait >= refresh_batch || (n_run == 0 && n_wait > 0)) {
            MI_PROF_COUNT(n_refresh++;)
            finish_draw(state == NS_NEED_DRAW && fin_pending, fin_depth);   // epilogue of the draws that just ended
            const bool p = state == NS_NEED_DRAW;
            double kq = 0.0;
            if constexpr (DENSE_M) {                          // p = L z and K = p . (Minv p) / 2 as mat-vecs (nuts.cpp:200-204)
                double zz[NS], pp[NS], mp[NS];
#pragma unroll 1
                for (int b = 0; b < NS / 2; ++b) {
                    double z0, z1;
                    rng_normal_pair(prm.seed, chain, draw + prm.draw0, (uint32_t)(4 * b + j4), STREAM_NORMAL, z0, z1);
                    zz[2 * b] = (8u * b + j4 < d) ? z0 : 0.0;
                    zz[2 * b + 1] = (8u * b + 4 + j4 < d) ? z1 : 0.0;
                }
                matvec_m2<NT>(afrag_l, zz, pp);
                matvec_m2<NT>(afrag_minv, pp, mp);
#pragma unroll
                for (int s = 0; s < NS; ++s) kq = dfma(pp[s], mp[s], kq);
                if (p && live) { st_row(V_MNTM, 0, pp); st_row(V_TPOS_P, 0, pp); st_row(V_TNEG_P, 0, pp); }
            } else {
#pragma unroll 1
            for (int b = 0; b < NS / 2; ++b) {               // nuts.cpp:200-202, this chain's own draw index
                double z0, z1;
                rng_normal_pair(prm.seed, chain, draw + prm.draw0, (uint32_t)(4 * b + j4), STREAM_NORMAL, z0, z1);
                double pa = (8u * b + j4 < d) ? z0 : 0.0;
                double pb = (8u * b + 4 + j4 < d) ? z1 : 0.0;
                if constexpr (GENERAL) {                      // p = L z, K = p . (Minv p) / 2
                    pa = lds_ms[8 * b + j4] * pa;
                    pb = lds_ms[8 * b + 4 + j4] * pb;
                    kq = dfma(pa, lds_mi[8 * b + j4] * pa, kq);
                    kq = dfma(pb, lds_mi[8 * b + 4 + j4] * pb, kq);
                } else {
                    kq = dfma(pa, pa, kq);
                    kq = dfma(pb, pb, kq);
                }
                if (p && live) {                              // mntm_vec, mntm_pos, mntm_neg (:202, :214-215)
                    st_pair(V_MNTM, 2 * b, pa, pb);
                    st_pair(V_TPOS_P, 2 * b, pa, pb);
                    st_pair(V_TNEG_P, 2 * b, pa, pb);
                }
            }
            }
            kq = kq + __shfl_xor(kq, 32);
            kq = kq + __shfl_xor(kq, 16);
            const double kk = kq / 2.0;                       // :204
            const double lu = det_log(rng_uniform(prm.seed, chain, draw + prm.draw0, 0u));
            copy_vec(V_PREV, V_TPOS_T, p);                    // draw_pos = draw_neg = prev_draw (:212-213)
            copy_vec(V_PREV, V_TNEG_T, p);
            if (p) {
                prev_K = kk;
                log_u = lu - prev_U - prev_K;                 // :206
                uslot = 1;
                jd = 0; n_val = 1.0; alpha_val = 0.0; n_alpha_val = 0.0; good_round = 0;
                state = NS_TREE;
            }
            if (max_depth > 0) begin_doubling(p);
            else if (p) { fin_pending = true; fin_depth = 0u; state = NS_NEED_DRAW; }   // while-loop of :227 never entered
        }
        MI_PROF(0)
        const bool run = state == NS_TREE;
        if (__ballot(run) == 0ull) continue;

        // ------------------------------------------------------------ B. one leaf for every running chain
        // The register-resident state is TICK-LOCAL: start state <- workspace record, one leapfrog, energies,
        // leaf record -> workspace.  Nothing large is live across the tree bookkeeping below (which reads the
        // records), so the register file is not spilled around it.  slot_of(k): record slot of leaf k.
        auto slot_of = [&](uint32_t k) -> int { return (k == 0) ? 0 : (__builtin_ctz(k) + 1); };
        const int slot_i = slot_of(li);
        double pU, pK;
        {
            double th[NS], pm[NS], w[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) { th[s] = 0.0; pm[s] = 0.0; w[s] = 0.0; }
            {
                const int cz = (li == 0) ? 0 : __builtin_ctz(li);
                const int sslot = (li == 0) ? 0 : ((cz >= 2) ? cz : slot_of(li - 1));   // leaf li - 2^(cz-1), or leaf li - 1
                const int vt = (li == 0) ? V_PREV : V_LEAF0 + 3 * sslot;
                const int vp = (li == 0) ? V_MNTM : V_LEAF0 + 3 * sslot + 1;
                const int vw = (li == 0) ? V_WPREV : V_LEAF0 + 3 * sslot + 2;
                if (run) {
                    ld_row(vt, 0, th); ld_row(vp, 0, pm); ld_row(vw, 0, w);
                }
            }
            MI_PROF(1)
            MI_PROF_COUNT(n_ticks++; n_active += __popcll(__ballot(run)) / 4;)
            // one leapfrog of signed size e (nuts.ipp:132, nuts.cpp:139-154), grad = -w
            double xl[GENERAL ? NS : 1];
            auto kick_l = [&]() __attribute__((always_inline)) {
                if constexpr (GENERAL) {
                    double kw[NS];
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const int i = 4 * s + j4;
                        if ((bslices >> s) & 1u) kw[s] = box_inv_jacobian(th[s], lds_bt[i], lds_lb[i], lds_ub[i]) * w[s];
                        else kw[s] = 1.0 * w[s];
                    }
                    if (prm.vals_bound) dense_product_poison<NS>(w, kw, j4, d);
#pragma unroll
                    for (int s = 0; s < NS; ++s) pm[s] = pm[s] - (e_signed * kw[s]) / 2.0;
                } else {
#pragma unroll
                    for (int s = 0; s < NS; ++s) pm[s] = pm[s] - (e_signed * w[s]) / 2.0;
                }
            };
            kick_l();
            if constexpr (GENERAL) {
                double mp[NS];
                if constexpr (DENSE_M) {
                    matvec_m2<NT>(afrag_minv, pm, mp);
                } else {
const double* const mi_t = mi_tab();
#pragma unroll
                    for (int s = 0; s < NS; ++s) mp[s] = mi_t[4 * s] * pm[s];
                    dense_product_poison<NS>(pm, mp, j4, d);
                }
#pragma unroll
                for (int s = 0; s < NS; ++s) th[s] = th[s] + e_signed * mp[s];
            } else {
#pragma unroll
                for (int s = 0; s < NS; ++s) th[s] = th[s] + e_signed * pm[s];
            }
            if constexpr (GENERAL) {
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int i = 4 * s + j4;
                    if ((bslices >> s) & 1u) xl[s] = ((uint32_t)i < d) ? box_inv_transform(th[s], lds_bt[i], lds_lb[i], lds_ub[i]) : 0.0;
                    else xl[s] = ((uint32_t)i < d) ? th[s] : 0.0;
                }
                target_times<NT>(afrag, prm.P, d, prm.sep_target != 0u, xl, w);
            } else {
                matvec_mfma<NT>(afrag, th, w);
            }
            kick_l();
            MI_PROF(2)
            if constexpr (GENERAL) {                     // nuts.ipp:134-140 with box_log_kernel and K = p.(Minv p)/2
                const double kval = -0.5 * dot4<NS>(xl, w);
                double lj = 0.0;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    if (((bslices >> s) & 1u) == 0u) continue;
                    const int i0 = 4 * s;
                    const double term = box_log_jacobian_term(th[s], lds_bt[i0 + j4], lds_lb[i0 + j4], lds_ub[i0 + j4]);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const double tg = __shfl(term, (lane & 15) + 16 * g);
                        if ((uint32_t)(i0 + g) < d && lds_bt[i0 + g] != 1) lj = lj + tg;
                    }
                }
                pU = -(kval + lj);
                double mp[NS];
                if constexpr (DENSE_M) {
                    matvec_m2<NT>(afrag_minv, pm, mp);
                } else {
const double* const mi_t = mi_tab();
#pragma unroll
                    for (int s = 0; s < NS; ++s) mp[s] = mi_t[4 * s] * pm[s];
                    dense_product_poison<NS>(pm, mp, j4, d);
                }
                double q = 0.0;
#pragma unroll
                for (int s = 0; s < NS; ++s) q = dfma(pm[s], mp[s], q);
                q = q + __shfl_xor(q, 32);
                q = q + __shfl_xor(q, 16);
                pK = q / 2.0;
            } else {
                pU = 0.5 * dot4<NS>(th, w);              // nuts.ipp:134-138
                pK = dot4<NS>(pm, pm) / 2.0;             // :140
            }
            if (!is_finite(pU)) pU = INF;
            if (run && live) {                           // leaf record (every leaf: odd ones live in slot 1 for one tick)
                st_row(V_LEAF0 + 3 * slot_i, 0, th); st_row(V_LEAF0 + 3 * slot_i + 1, 0, pm); st_row(V_LEAF0 + 3 * slot_i + 2, 0, w);
            }
            // the tree's far edge (= near edge of its second half, or the leaf itself at depth 0) is what a
            // successful doubling leaves in draw_pos / draw_neg (src/nuts.cpp:241-256); a failed one ends the draw,
            // so it can be written in place as soon as that leaf exists
            const bool st_edge = run && live && (li == ((jd == 0u) ? 0u : (1u << (jd - 1))));
            if (st_edge) {
                const int et = (vdir > 0) ? V_TPOS_T : V_TNEG_T, ep = (vdir > 0) ? V_TPOS_P : V_TNEG_P;
                st_row(et, 0, th); st_row(ep, 0, pm);
            }
        }
        double cn = (log_u <= -pU - pK) ? 1.0 : 0.0;     // :146
        const bool cs = log_u < 1000.0 - pU - pK;        // :147
        const double dd = -(pU + pK) + H0;
        double ca = det_exp((dd < 0.0) ? dd : 0.0);      // :157
        double cna = 1.0;
        double cU = pU;
        int cref_t = V_LEAF0 + 3 * slot_i;               // carried proposal: this leaf's record (theta, P*theta)
        int cref_w = V_LEAF0 + 3 * slot_i + 2;
        if (run) n_leap++;
        MI_PROF(3)
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
                const double p_n = lvl(l, 0), p_a = lvl(l, 1), p_na = lvl(l, 2), p_U = lvl(l, 3);
                const double prob = cn / (p_n + cn);                     // :212
                if (!(z < prob)) {                                       // keep new_draw_p (:215-217)
                    const int ps = slot_of(li - 1);                      // level 1: the previous (even) leaf's record
                    cref_t = (l == 1) ? V_LEAF0 + 3 * ps : V_PP0 + (int)l;
                    cref_w = (l == 1) ? V_LEAF0 + 3 * ps + 2 : V_PPW0 + (int)l;
                    cU = p_U;
                }
                cn = p_n + cn;                                           // :220-222
                ca = p_a + ca;
                cna = p_na + cna;
            }
            const bool need_ut = mrg && !failed;
            if (__ballot(need_ut) != 0ull) {
                const uint32_t b = li - (1u << l) + 1;                   // first leaf of the node (valid where need_ut)
                const int slot1 = (!need_ut || b == 0) ? 0 : (__builtin_ctz(b) + 1);
                const int slot2 = (l == 1) ? slot_i : (int)l;            // first leaf of the second half
                const bool ok = uturn_ok(need_ut, V_LEAF0 + 3 * slot1, V_LEAF0 + 3 * slot1 + 1,
                                         V_LEAF0 + 3 * slot2, V_LEAF0 + 3 * slot2 + 1, vdir);     // :226-227
                if (need_ut && !ok) failed = true;                       // :229
            }
        }
        MI_PROF(4)
        // ---- end of the doubling? top-level accept first (src/nuts.cpp:260-279), so that an accepted
        //      proposal goes straight to prev_draw instead of through a pending slot
        const bool keep = run && !failed;
        const bool complete = keep && (li == (1u << jd) - 1u);
        const bool fin = run && (failed || complete);
        bool take = false;
        if (__ballot(complete) != 0ull) {
            const double z = rng_uniform(prm.seed, chain, draw + prm.draw0, uslot);  // :261
            if (complete) {
                uslot++;
                take = z < cn / n_val;                                   // :263
                if (take) { prev_U = cU; good_round = 1; }               // :264-277
            }
        }
        // ---- pending first half: proposal and its P*theta by value, scalars to LDS
        if (keep && !complete) {
            lvl((int)pend_level, 0) = cn; lvl((int)pend_level, 1) = ca;
            lvl((int)pend_level, 2) = cna; lvl((int)pend_level, 3) = cU;
        }
        {
            // a pending first half at level 1 IS the leaf record just written (referenced, not copied); deeper
            // levels and accepted proposals are copied record -> slot
            const bool do_store = keep && (complete ? take : (pend_level > 1u));
            if (__ballot(do_store) != 0ull) {
                const int pl = do_store ? (int)pend_level : 1;
                const int dst_t = take ? V_PREV : V_PP0 + pl;
                const int dst_w = take ? V_WPREV : V_PPW0 + pl;
                if (do_store && live) {
#pragma unroll
                    for (int c0 = 0; c0 < NS; c0 += CHC) {
                        double t1[CHC], t2[CHC];
                        ld_row(cref_t, c0, t1); ld_row(cref_w, c0, t2);
                        st_row(dst_t, c0, t1); st_row(dst_w, c0, t2);
                        if (CHC < NS) __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        MI_PROF(5)
        if (__ballot(fin) != 0ull) {
            MI_PROF_COUNT(n_finblk++;)
            if (fin) { alpha_val = ca; n_alpha_val = cna; n_val = n_val + cn; }   // :246,255 ; :283
            bool s_ok = false;
            if (__ballot(complete) != 0ull) {
                double q1 = 0.0, q2 = 0.0;
                if (complete) {
#pragma unroll
                    for (int c0 = 0; c0 < NS; c0 += CHF) {
                        double tp[CHF], tn[CHF], pp[CHF], pn[CHF];
                        ld_row(V_TPOS_T, c0, tp); ld_row(V_TNEG_T, c0, tn); ld_row(V_TPOS_P, c0, pp); ld_row(V_TNEG_P, c0, pn);
#pragma unroll
                        for (int k = 0; k < CHF; ++k) {
                            const double df = tp[k] - tn[k];
                            q1 = dfma(df, pn[k], q1);                    // :286
                            q2 = dfma(df, pp[k], q2);                    // :287
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                q1 = q1 + __shfl_xor(q1, 32); q1 = q1 + __shfl_xor(q1, 16);
                q2 = q2 + __shfl_xor(q2, 32); q2 = q2 + __shfl_xor(q2, 16);
                s_ok = complete && (q1 >= 0.0) && (q2 >= 0.0);           // :289
            }
            const bool more = fin && s_ok && (jd + 1 < max_depth);
            if (fin) jd = jd + 1;                                        // :284
            begin_doubling(more);
            if (fin && !more) { fin_pending = true; fin_depth = jd; state = NS_NEED_DRAW; }
        }
        if (run && !fin) li = li + 1;
        MI_PROF(6)
    }
#ifdef MI_NUTS_ASYNC_PROF
    if (prm.prof && blockIdx.x == 0 && threadIdx.x == 0)
    {
        for (int k = 0; k < 8; ++k) prm.prof[k] = pc[k];
        prm.prof[12] = n_ticks; prm.prof[13] = n_active; prm.prof[14] = n_refresh; prm.prof[15] = n_finblk;
    }
#endif

    if (live) {
        double tmp[NS];
        ld_row(V_PREV, 0, tmp);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if constexpr (GENERAL) {
                const int i = 4 * s + j4;
                if (dim_ok(s)) tmp[s] = box_inv_transform(tmp[s], lds_bt[i], lds_lb[i], lds_ub[i]);
            }
            if (dim_ok(s)) prm.theta[(size_t)(4 * s) * C + lane_off] = tmp[s];
        }
        if (j4 == 0) {
            if (prm.n_accept) prm.n_accept[cl] = n_acc;
            if (prm.n_leap) prm.n_leap[cl] = n_leap;
            if (prm.step_out) prm.step_out[cl] = eps;
            if (prm.adapt_state) { prm.adapt_state[cl] = h_val_(); prm.adapt_state[C + cl] = eps_bar_(); prm.adapt_state[2 * C + cl] = mu_val_(); }
        }
    }
}

}  // namespace mi
