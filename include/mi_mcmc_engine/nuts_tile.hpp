// nuts_tile.hpp -- mcmc::nuts for USER-DEFINED targets on the tiled (MFMA-layout) engine: the asynchronous per-chain tree state machine
// of rounds 2-4's built-in Gaussian kernel (nuts_reg.hpp, since replaced by nuts_memo.hpp; DESIGN.md 4.4: register-carried leaf state, eager U-turn tests, momenta generated
// ahead; the iterative leaf-indexed tree is derived in mcmc_amd/csrc/nuts_dense.hpp) with the gradient behind the tile functor of
// tile_samplers.hpp instead of the dense mat-vec.
//
// Replaces, for C independent chains, mcmc::internal::nuts_impl with nuts_find_initial_step_size and the recursive nuts_build_tree
// (ref: src/nuts.cpp:30-332, include/mcmc/nuts.ipp:30-241; the callback contract ref: include/mcmc/nuts.hpp:65-72 is the tile functor:
// one grad_tile call per leaf gives the gradient both half-kicks around the position use AND the value box_log_kernel returns there --
// the reference's three callbacks per leaf are deterministic repeats), identity precond_mat, no bounds.
//
// A leaf's record is (theta, p, grad); pending proposals carry (theta, grad) and their U = -value as a per-level scalar.  With the
// built-in dense Gaussian written as a tile target (grad = -(P theta), value = -theta.P theta / 2) every operation has the bits of
// nuts_gauss_reg_kernel's: p + (e grad) / 2 == p - (e P theta) / 2 exactly.  The identity `inv_precond_matrix * mntm` is applied
// element-wise with the NaN rule of the reference's dense product (dense_product_poison, diag_quadratic), so the non-finite regime
// needs no replay on this route.
#pragma once

#include "tile_samplers.hpp"

namespace mi {

namespace tile_nuts {
// workspace vectors of a chain (the numbering of nuts_dense.hpp / nuts_async.hpp)
enum : int {
    V_PREV = 0, V_WPREV = 1, V_MNTM = 2, V_TPOS_T = 3, V_TPOS_P = 4, V_TNEG_T = 5, V_TNEG_P = 6,
    V_LEAF0 = 7,             // slot k: theta 7+3k, p 8+3k, grad 9+3k, k = 0..10 (even leaves only: slot 0's p / grad rows are free, see below)
    V_PP0 = 40,              // pending proposal of level l at 40 + l, its gradient at 52 + l
    V_PPW0 = 52,
    NUTS_NVEC = 64,
    NUTS_MAX_DEPTH = 10,
    NUTS_LVLS = 12
};
enum : int { V_MNTM2 = V_LEAF0 + 3, V_PREVB = V_LEAF0 + 4, V_WPREVB = V_LEAF0 + 5 };
enum : int { NS_NEED_DRAW = 0, NS_TREE = 1, NS_DONE = 2 };
constexpr size_t lds_doubles() { return (size_t)NUTS_LVLS * 4 * 64 + 3 * 64; }     // behind the target's own LDS (GEN: TileGen's tables follow)

// (the kernel lives in this namespace so that its vector numbering is found before the built-in kernels' mi:: enums, which a
//  translation unit of the engine sees as well)
// GEN: settings.vals_bound and / or a diagonal precond_mat (TileGen, tile_samplers.hpp): the tree lives in the transformed space (the
// U-turn dots are plain), rows are reported through inv_transform
template <class T, bool GEN = false>
__global__ MI_NO_DS_MERGE __launch_bounds__(256, 1) void nuts_tile_kernel(const TileParams prm, const T tgt)
{
    constexpr int NT = T::NT, NS = 4 * NT;
    constexpr int WS_NVEC = NUTS_NVEC;
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    double* const lds_t = lds_all;                                   // the target's own LDS (its matrices in fragment order)
    double* const lds_lvl = lds_all + prm.lds_user_doubles;         // [NUTS_LVLS][4][64]
    double* const lds_da = lds_lvl + NUTS_LVLS * 4 * 64;            // [3][64]: the dual-averaging state (h, epsilon_bar, mu)
    [[maybe_unused]] double* const lds_gen = lds_da + 3 * 64;        // GEN: the bounds / mass tables
    tgt.stage(lds_t);
    if constexpr (GEN) TileGen<T::NT>::stage(lds_gen, prm);
    __syncthreads();
    [[maybe_unused]] TileGen<T::NT> tg;
    if constexpr (GEN) tg.use(lds_gen, prm);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j4 = lane >> 4;
    const int cw = wave * 16 + (lane & 15);
    const uint64_t cl = ((uint64_t)blockIdx.x * 4 + wave) * 16 + (lane & 15);
    const bool live = cl < prm.C;
    const uint64_t cld = live ? cl : prm.C - 1;
    const uint64_t chain = prm.chain0 + cl;
    const uint32_t d = prm.d;
    const uint64_t C = prm.C;
    const size_t lane_off = (size_t)j4 * C + cld;

    auto lvl = [&](int l, int f) -> double& { return lds_lvl[(l * 4 + f) * 64 + cw]; };
    // workspace: [wave][vector] blocks of NS * 512 bytes, wave-uniform base + one 32-bit byte offset per access (nuts_async.hpp)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    char* const ws_wave_u = reinterpret_cast<char*>(__builtin_assume_aligned(prm.ws, 256)) + ((size_t)blockIdx.x * 4 + wave_u) * ((size_t)WS_NVEC * NS * 512);
    // inside a vector: [chain][pair of slices][j4] in 16-byte granules (nuts_async.hpp: why)
    uint32_t lane_b = (uint32_t)(lane & 15) * (uint32_t)(NS * 32) + (uint32_t)j4 * 16u;     // redefined (opaquely) at the top of every tick
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
    // K = p . (I p) / 2 with the identity as the dense product it is in the reference (nuts.cpp leap_frog_fn / nuts.ipp:51,66,140)
    auto kinetic_of = [&](const double (&p)[NS]) __attribute__((always_inline)) -> double {
        if constexpr (GEN) return tg.kinetic(p);
        else return diag_quadratic<NS>(p, 1.0, j4, d) / 2.0;
    };
    double val = 0.0;                                    // log kernel at the register-resident position
    constexpr int CHC = (NS < 16) ? NS : 16;             // record copies: 2 vectors per chunk
    // the chain's last leaf: position, momentum, GRADIENT of the log kernel at the position (MFMA B / D layout).  Loop-carried.
    double th[NS], pm[NS], w[NS];
    // value and gradient at the register-resident position (GEN: at x = inv_transform(theta), hmc.cpp:108-110)
    auto eval = [&]() __attribute__((always_inline)) {
        if constexpr (GEN) {
            double xs[NS];
            tg.x_of(th, xs);
            tgt.grad_tile(lds_t, xs, w, val, true);
        } else {
            tgt.grad_tile(lds_t, th, w, val, true);
        }
    };
    auto potential_of = [&]() __attribute__((always_inline)) -> double {       // -box_log_kernel(theta), nuts.cpp:84-95
        if constexpr (GEN) return tg.potential(val, th);
        else return -val;
    };
    // p += (e [J^-1] grad) / 2 (nuts.cpp:108-135)
    auto kick = [&](double e) __attribute__((always_inline)) {
        if constexpr (GEN) {
            double t[NS];
            tg.kick_terms(th, w, e, t);
#pragma unroll
            for (int s = 0; s < NS; ++s) pm[s] = pm[s] + t[s];
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s) pm[s] = pm[s] + (e * w[s]) / 2.0;
        }
    };
    // theta += e (Minv p), a dense product in the reference (nuts.cpp:139-154)
    auto drift = [&](double e) __attribute__((always_inline)) {
        if constexpr (GEN) {
            double mp[NS];
            tg.minv_p(pm, mp);
#pragma unroll
            for (int s = 0; s < NS; ++s) th[s] = th[s] + e * mp[s];
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s) th[s] = th[s] + e * pm[s];
            dense_product_poison<NS>(pm, th, j4, d);
        }
    };

    // ---------------------------------------------------------------- setup (nuts.cpp:156-195), all chains together
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const uint32_t dimc = dim_ok(s) ? (uint32_t)(4 * s + j4) : 0u;
        const double v = prm.theta[(size_t)dimc * C + cld];
        if constexpr (GEN) th[s] = dim_ok(s) ? tg.enter(v, dimc) : 0.0;        // nuts.cpp:160-162
        else th[s] = dim_ok(s) ? v : 0.0;
    }
    eval();
    if (live) { st_row(V_PREV, 0, th); st_row(V_WPREV, 0, w); }
    double prev_U = potential_of();                      // nuts.cpp:181 (no finiteness guard there)
    // Non-finite regime: the identity `inv_precond_matrix * mntm` is applied element-wise WITH the NaN rule of the dense product
    // (dense_product_poison, diag_quadratic), so this route needs no replay (tile_samplers.hpp)
    uint64_t n_leap = 0;
    double eps;
    if (prm.draw0 == 0) {   // nuts_find_initial_step_size (nuts.ipp:30-93) from (first_draw, z_init), nuts.cpp:166-172
        auto leapfrog = [&](double e) __attribute__((always_inline)) {
            kick(e);
            drift(e);
            eval();
            kick(e);
        };
        auto energy = [&]() __attribute__((always_inline)) -> double {
            double u = potential_of();
            if (!is_finite(u)) u = INF;
            return u + kinetic_of(pm);
        };
#pragma unroll
        for (int b = 0; b < NS / 2; ++b) {
            double z0, z1;
            rng_normal_pair(prm.seed, chain, 0u, (uint32_t)(4 * b + j4), STREAM_INIT, z0, z1);
            pm[2 * b] = (8u * b + j4 < d) ? z0 : 0.0;
            pm[2 * b + 1] = (8u * b + 4 + j4 < d) ? z1 : 0.0;
            if constexpr (GEN) {                         // mntm_vec = sqrt_precond_matrix * rand_vec (nuts.cpp:168)
                pm[2 * b] = tg.msqrt(8 * b + j4) * pm[2 * b];
                pm[2 * b + 1] = tg.msqrt(8 * b + 4 + j4) * pm[2 * b + 1];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        double U0 = prev_U;
        if (!is_finite(U0)) U0 = INF;
        const double K0 = kinetic_of(pm);
        const double log_half = det_log(0.5), neg_log2 = -det_log(2.0);
        eps = 1.0;
        leapfrog(eps);
        n_leap++;
        double dH = -energy() + (U0 + K0);
        int a_val = 2 * (dH > log_half ? 1 : 0) - 1;
        bool cond = dH > neg_log2;
        while (__ballot(cond) != 0ull) {
            const double e_new = eps * ((a_val == 1) ? 2.0 : 0.5);
            if (cond) { eps = e_new; n_leap++; }
            leapfrog(eps);
            const double dH2 = -energy() + (U0 + K0);
            if (cond) {
                a_val = 2 * (dH2 > log_half ? 1 : 0) - 1;
                cond = dH2 > neg_log2;
            }
        }
    } else {                // continuation of an adapted run (mi_chains.draw0 > n_adapt_draws): the step size comes back in
        eps = (live && prm.step_out) ? prm.step_out[cl] : 1.0;
    }
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
    int state = (n_total > 0) ? NS_NEED_DRAW : NS_DONE;
    uint32_t draw = 0;           // this chain's draw index
    uint32_t jd = 0;             // depth of the doubling in progress
    uint32_t li = 0;             // next leaf of that doubling
    uint32_t uslot = 0;
    int vdir = 1;
    double e_signed = 0.0, H0 = 0.0, log_u = 0.0;
    // per-chain scalars that are touched once per doubling or per draw live in row 0 of the level table (levels start at 1)
    // instead of registers: the kinetic energy of the draw's momentum, n (nuts.cpp:283), alpha and n_alpha (:246,255)
    auto prev_K_ = [&]() -> double& { return lvl(0, 0); };
    auto n_val_ = [&]() -> double& { return lvl(0, 1); };
    auto alpha_ = [&]() -> double& { return lvl(0, 2); };
    auto n_alpha_ = [&]() -> double& { return lvl(0, 3); };
    int good_round = 0;
    uint32_t utpre = 0;          // bit l: the U-turn test of the open level-l node passed (set when the first leaf of its second half ran)
    // Draw boundaries without waiting.  The lanes of a chain that waits at a draw boundary are dead weight in every wave-wide
    // phase of a tick (the MFMA mat-vec alone is half of a tick), and with one refresh phase per batch of waiting chains the
    // cohort that was refreshed together waits for its slowest member at the next boundary: 9.4 of a wave's 16 chains were
    // inside a tree on an average tick.  What the next draw needs and does not depend on the chain's state -- the momentum
    // (nuts.cpp:200-202), its kinetic energy (:204) and the slice uniform (:206): functions of (seed, chain, draw index) -- is
    // therefore generated AHEAD, by a phase that serves every chain of the wave that lacks it (a phase costs a wave's time
    // whatever the number of chains it serves), into the momentum vector the running draw does not use.  A chain that ends
    // a draw runs the epilogue on the spot (dual averaging :294-302) and starts the next draw in the same tick; the kept row
    // (:306-309) is written by the next phase, from a vector that stays intact meanwhile:
    //   * prev_draw alternates between two vectors: an accepted proposal (:264-277) goes to the one that did NOT hold prev_draw
    //     when the draw started, so that vector is, during the whole next draw, both the row still to be written and the
    //     initial draw_pos = draw_neg (:212-213);
    //   * mntm_pos = mntm_neg = mntm_vec (:214-215) likewise: the edges are the draw's initial vectors until a doubling has
    //     written that side (pos_init / neg_init), no copies at the start of a draw.
    // A chain waits (NS_NEED_DRAW) only if its next momentum is not there or its previous row is still unwritten, and any waiting
    // chain triggers the phase: one phase per draw and chain, shared by (nearly) all 16 chains of the wave.
    int mv = V_MNTM, mvn = V_MNTM2;          // momentum vector of the running draw / of the next one
    int pb = 0, pb0 = 0;                     // which of the two vectors holds prev_draw now / held it when the draw started
    bool mom_ready = false;                  // the next draw's momentum is in mvn (kinetic energy, log slice uniform: next_K, next_lu)
    double next_K = 0.0, next_lu = 0.0;
    bool row_pend = false, row2_pend = false;   // kept rows still to be written: draw row_draw from pvec(pb0); draw - 1 from pvec(pb)
    uint32_t row_draw = 0;
    bool pos_init = true, neg_init = true;
    auto pvec = [](int b) -> int { return b ? V_PREVB : V_PREV; };
    auto wvec = [](int b) -> int { return b ? V_WPREVB : V_WPREV; };

    // start doubling jd (direction draw, nuts.cpp:233-235) for lanes with `p`
    auto begin_doubling = [&](bool p) __attribute__((always_inline)) {
        const double zdir = rng_uniform(prm.seed, chain, draw + prm.draw0, uslot);
        if (p) {
            uslot++;
            vdir = (zdir <= 0.5) ? -1 : 1;
            e_signed = (double)vdir * eps;
            H0 = prev_U + prev_K_();
            li = 0;
        }
    };
    // end of a draw for lanes with `p` (dual averaging nuts.cpp:294-302; the row store :306-309 is left to the next phase)
    auto end_draw = [&](bool p, uint32_t my_depth) __attribute__((always_inline)) {
        if (p && prm.depth_trace && live && j4 == 0) prm.depth_trace[(size_t)draw * C + cl] = my_depth;
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
    // kept row `idx` of lanes with `p` from workspace vector `vec`
    auto store_row = [&](bool p, int vec, uint32_t idx) __attribute__((always_inline)) {
        if (__ballot(p) == 0ull) return;
        if (p && live) {
            double* out = prm.draws + (size_t)(idx - prm.n_burnin) * d * C;
#pragma unroll
            for (int c0 = 0; c0 < NS; c0 += CHC) {
                double tmp[CHC];
                ld_row(vec, c0, tmp);
#pragma unroll
                for (int k = 0; k < CHC; ++k) {
                    if constexpr (GEN) tmp[k] = tg.leave(tmp[k], c0 + k);      // rows are reported in the constrained space
                    if (dim_ok(c0 + k)) (out + (size_t)(4 * (c0 + k)) * C)[lane_off] = tmp[k];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // lanes with `p` (next momentum ready, no older row pending) enter their next draw (nuts.cpp:200-219)
    auto roll_state = [&](bool p) __attribute__((always_inline)) {
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
        // ------------------------------------------------------------ A. the phase: rows, momenta ahead, waiting chains start
        if (__ballot(state == NS_NEED_DRAW) != 0ull) {
            store_row(row_pend, pvec(pb0), row_draw);
            store_row(row2_pend, pvec(pb), draw - 1u);
            row_pend = false; row2_pend = false;
            if (state == NS_NEED_DRAW && draw >= n_total) state = NS_DONE;
            const uint32_t nidx = draw + ((state == NS_TREE) ? 1u : 0u);     // the draw the momentum is for
            const bool gen = state != NS_DONE && !mom_ready && nidx < n_total;
            double kq = 0.0;
#pragma unroll 1
            for (int b = 0; b < NS / 2; ++b) {               // nuts.cpp:200-202, this chain's own draw index
                double z0, z1;
                rng_normal_pair(prm.seed, chain, nidx + prm.draw0, (uint32_t)(4 * b + j4), STREAM_NORMAL, z0, z1);
                double pa = (8u * b + j4 < d) ? z0 : 0.0;
                double pb_ = (8u * b + 4 + j4 < d) ? z1 : 0.0;
                if constexpr (GEN) {                         // :202 and :204 with the diagonal matrices
                    pa = tg.msqrt(8 * b + j4) * pa; pb_ = tg.msqrt(8 * b + 4 + j4) * pb_;
                    kq = dfma(pa, tg.mi[8 * b + j4] * pa, kq);
                    kq = dfma(pb_, tg.mi[8 * b + 4 + j4] * pb_, kq);
                } else {
                    kq = dfma(pa, 1.0 * pa, kq);             // (finite normals: the identity product needs no NaN rule here)
                    kq = dfma(pb_, 1.0 * pb_, kq);
                }
                if (gen && live) st_pair(mvn, 2 * b, pa, pb_);
            }
            kq = kq + __shfl_xor(kq, 32);
            kq = kq + __shfl_xor(kq, 16);
            const double lu = det_log(rng_uniform(prm.seed, chain, nidx + prm.draw0, 0u));
            if (gen) { next_K = kq / 2.0; next_lu = lu; mom_ready = true; }     // :204
            const bool p = state == NS_NEED_DRAW;             // (all of them have a momentum now and no row pending)
            roll_state(p);
            if (max_depth > 0) begin_doubling(p);
            else { end_draw(p, 0u); if (p) state = NS_NEED_DRAW; }              // while-loop of :227 never entered
        }
        const bool run = state == NS_TREE;
        if (__ballot(run) == 0ull) continue;

        // ------------------------------------------------------------ B. one leaf for every running chain
        auto slot_of = [&](uint32_t k) -> int { return (k == 0) ? 0 : (__builtin_ctz(k) + 1); };
        const int slot_i = slot_of(li);
        const int rec_t = V_LEAF0 + 3 * slot_i, rec_p = rec_t + 1, rec_w = rec_t + 2;    // this leaf's record (even leaves only)
        const bool odd = (li & 1u) != 0u;
        {   // start state: the registers hold the previous leaf (li odd, or ctz(li) == 1); otherwise a record
            const int cz = (li == 0) ? 0 : __builtin_ctz(li);
            const bool need = run && (li == 0 || cz >= 2);
            if (__ballot(need) != 0ull) {
                const int vt = (li == 0) ? pvec(pb) : V_LEAF0 + 3 * cz;                  // leaf li - 2^(cz-1) sits in slot cz
                const int vp = (li == 0) ? mv : V_LEAF0 + 3 * cz + 1;
                const int vw = (li == 0) ? wvec(pb) : V_LEAF0 + 3 * cz + 2;
                if (need) { ld_row(vt, 0, th); ld_row(vp, 0, pm); ld_row(vw, 0, w); }
            }
        }
        // EAGER U-turn tests.  The test of a level-l node (nuts.ipp:226-227) uses its first leaf b and the first leaf of its second
        // half, b2 = b + 2^(l-1) (nuts_dense.hpp) -- both exist as soon as b2 does, 2^(l-1) - 1 ticks before the node closes.  An
        // even leaf li > 0 is that b2 for exactly one node, level l = ctz(li) + 1 (if l <= jd), with b = li - 2^ctz(li).  So the
        // test is evaluated HERE, with (theta, p)(b2) in registers and (theta, p)(b) fetched together with the start records
        // (one round trip at the top of the tick, two vectors instead of four), and its bit kept for the tick that closes the
        // node: the unwind below touches no memory.  An odd leaf is b2 of its own level-1 node with b = li - 1 = the start of this
        // leapfrog: the same expressions with the start state as (theta, p)(b).
        const uint32_t cz_i = (li == 0) ? 0u : (uint32_t)__builtin_ctz(li);
        const bool eager = run && !odd && li != 0u && (cz_i + 1u <= jd);
        const bool any_eager = __ballot(eager) != 0ull;
        const uint32_t bleaf = li - (1u << cz_i);
        const int sb = (!eager || bleaf == 0) ? 0 : (__builtin_ctz(bleaf) + 1);
        const int eb_t = V_LEAF0 + 3 * sb, eb_p = eb_t + 1;           // (theta, p) of leaf b for the eager lanes
        // one leapfrog of signed size e (nuts.ipp:132, nuts.cpp:139-154), grad = w;  d = theta(b2) - theta(b) (by direction).
        // Odd leaf: b is the start state, d and q1 = d . p(b) fall out of the kick / drift loop.  Eager even leaf: the rows of leaf b
        // are requested here and used AFTER the mat-vec (8.6 us of matrix-pipe time in which the wave has nothing else in
        // flight), so their latency costs nothing.  q2 = d . p(b2) comes out of the second kick in both cases.
        double dd[NS];           // d; on an eager lane it first receives theta(b) (after the loop that writes d on every lane: a lane is
        double Lp[NS];           // either odd or eager, and the load may not be pending when the VALU writes the register); p(b)
        double q1 = 0.0, q2 = 0.0;
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            Lp[s_] = pm[s_];                             // p(b) of an odd leaf: the start momentum (an eager lane's load overwrites it below)
            dd[s_] = th[s_];                             // t0, until the drift (with its NaN rule) is known
        }
        kick(e_signed);
        drift(e_signed);
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            dd[s_] = (vdir > 0) ? (th[s_] - dd[s_]) : (dd[s_] - th[s_]);
            q1 = dfma(dd[s_], Lp[s_], q1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (any_eager) { if (eager) { ld_row(eb_t, 0, dd); ld_row(eb_p, 0, Lp); } }
        eval();
        if (any_eager) {
            if (eager) {
                double q1e = 0.0;
#pragma unroll
                for (int s_ = 0; s_ < NS; ++s_) {
                    dd[s_] = (vdir > 0) ? (th[s_] - dd[s_]) : (dd[s_] - th[s_]);
                    q1e = dfma(dd[s_], Lp[s_], q1e);
                }
                q1 = q1e;
            }
        }
        kick(e_signed);
#pragma unroll
        for (int s = 0; s < NS; ++s) q2 = dfma(dd[s], pm[s], q2);
        double pU = potential_of();                      // nuts.ipp:134-138
        const double pK = kinetic_of(pm);                // :140
        if (!is_finite(pU)) pU = INF;
        q1 = q1 + __shfl_xor(q1, 32); q1 = q1 + __shfl_xor(q1, 16);
        q2 = q2 + __shfl_xor(q2, 32); q2 = q2 + __shfl_xor(q2, 16);
        const bool ut_now = (q1 >= 0.0) && (q2 >= 0.0);  // odd leaf: its level-1 test; eager even leaf: the test of level ctz(li) + 1
        if (eager) utpre = (utpre & ~(1u << (cz_i + 1u))) | ((ut_now ? 1u : 0u) << (cz_i + 1u));
        if (run && live && !odd) {                       // even leaves are the records later leaves and tests read
            st_row(rec_t, 0, th); st_row(rec_p, 0, pm); st_row(rec_w, 0, w);
        }
        // the tree's far edge (= near edge of its second half, or the leaf itself at depth 0) is what a successful doubling
        // leaves in draw_pos / draw_neg (src/nuts.cpp:241-256); a failed one ends the draw, so it is written in place
        const bool st_edge = run && live && (li == ((jd == 0u) ? 0u : (1u << (jd - 1))));
        if (st_edge) {
            const int et = (vdir > 0) ? V_TPOS_T : V_TNEG_T, ep = (vdir > 0) ? V_TPOS_P : V_TNEG_P;
            st_row(et, 0, th); st_row(ep, 0, pm);
        }
        if (run && (li == ((jd == 0u) ? 0u : (1u << (jd - 1))))) { if (vdir > 0) pos_init = false; else neg_init = false; }
        double cn = (log_u <= -pU - pK) ? 1.0 : 0.0;     // :146
        const bool cs = log_u < 1000.0 - pU - pK;        // :147
        const double dH = -(pU + pK) + H0;
        double ca = det_exp((dH < 0.0) ? dH : 0.0);      // :157
        double cna = 1.0;
        double cU = pU;
        bool cref_regs = true;                           // carried proposal: this leaf (registers) ...
        int cref_t = rec_t, cref_w = rec_w;              // ... or a record (theta, P*theta)
        if (run) n_leap++;
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
            const bool ok = (l == 1) ? ut_now : (((utpre >> l) & 1u) != 0u);      // :226-227, evaluated when its second operand appeared
            if (need_ut && !ok) failed = true;                                   // :229
        }
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
                take = z < cn / n_val_();                                   // :263
                if (take) { prev_U = cU; good_round = 1; pb = 1 - pb0; }  // :264-277; the proposal goes to pvec(pb) below
            }
        }
        // ---- pending first half: proposal and its P*theta by value, scalars to LDS
        if (keep && !complete) {
            lvl((int)pend_level, 0) = cn; lvl((int)pend_level, 1) = ca;
            lvl((int)pend_level, 2) = cna; lvl((int)pend_level, 3) = cU;
        }
        {
            // a pending first half at level 1 IS the (even) leaf's record just written (referenced, not copied); deeper levels
            // and accepted proposals are written to their slot: from the registers when the carried proposal is this leaf,
            // record -> slot otherwise
            const bool do_store = keep && (complete ? take : (pend_level > 1u)) && live;
            if (__ballot(do_store) != 0ull) {
                const int pl = do_store ? (int)pend_level : 1;
                const int dst_t = take ? pvec(1 - pb0) : V_PP0 + pl;
                const int dst_w = take ? wvec(1 - pb0) : V_PPW0 + pl;
                if (do_store && cref_regs) { st_row(dst_t, 0, th); st_row(dst_w, 0, w); }
                const bool do_copy = do_store && !cref_regs;
                if (__ballot(do_copy) != 0ull) {
                    if (do_copy) {       // both rows in ONE round trip, through the registers of d and p(b) (dead since the second kick)
                        ld_row(cref_t, 0, dd); ld_row(cref_w, 0, Lp);
                        st_row(dst_t, 0, dd); st_row(dst_w, 0, Lp);
                    }
                }
            }
        }
        if (__ballot(fin) != 0ull) {
            if (fin) { alpha_() = ca; n_alpha_() = cna; n_val_() = n_val_() + cn; }   // :246,255 ; :283
            bool s_ok = false;
            if (__ballot(complete) != 0ull) {
                const int en_t = neg_init ? pvec(pb0) : V_TNEG_T, en_p = neg_init ? mv : V_TNEG_P;
                const int ep_t = pos_init ? pvec(pb0) : V_TPOS_T, ep_p = pos_init ? mv : V_TPOS_P;
                // [ (pos - neg) . p_neg >= 0 ] * [ (pos - neg) . p_pos >= 0 ] (:286-289).  The leaf state of a chain whose doubling
                // is complete is dead (the next doubling starts from prev_draw), so its registers take the four operands in
                // ONE round trip (with chains out of step, some chain of the wave is here in three ticks out of four)
                double x4[NS];
                double q1 = 0.0, q2 = 0.0;
                if (complete) {
                    ld_row(en_t, 0, th); ld_row(en_p, 0, pm); ld_row(ep_t, 0, w); ld_row(ep_p, 0, x4);
#pragma unroll
                    for (int k = 0; k < NS; ++k) {
                        const double dd_ = w[k] - th[k];
                        q1 = dfma(dd_, pm[k], q1);
                        q2 = dfma(dd_, x4[k], q2);
                    }
                }
                q1 = q1 + __shfl_xor(q1, 32); q1 = q1 + __shfl_xor(q1, 16);
                q2 = q2 + __shfl_xor(q2, 32); q2 = q2 + __shfl_xor(q2, 16);
                s_ok = complete && (q1 >= 0.0) && (q2 >= 0.0);
            }
            const bool more = fin && s_ok && (jd + 1 < max_depth);
            if (fin) jd = jd + 1;                                        // :284
            const bool ended = fin && !more;
            bool roll = false;
            if (__ballot(ended) != 0ull) {
                end_draw(ended, jd);
                roll = ended && draw < n_total && mom_ready && !row_pend;
                if (ended && !roll) state = NS_NEED_DRAW;                // the phase: its row, its next momentum, or the end of its run
                roll_state(roll);
            }
            begin_doubling(more || roll);
        }
        if (run && !fin) li = li + 1;
    }

    if (live) {
#pragma unroll
        for (int c0 = 0; c0 < NS; c0 += CHC) {
            double tmp[CHC];
            ld_row(pvec(pb), c0, tmp);
#pragma unroll
            for (int k = 0; k < CHC; ++k) {
                if constexpr (GEN) tmp[k] = tg.leave(tmp[k], c0 + k);
                if (dim_ok(c0 + k)) prm.theta[(size_t)(4 * (c0 + k)) * C + lane_off] = tmp[k];
            }
        }
        if (j4 == 0) {
            if (prm.n_accept) prm.n_accept[cl] = n_acc;
            if (prm.n_leap) prm.n_leap[cl] = n_leap;
            if (prm.step_out) prm.step_out[cl] = eps;
            if (prm.adapt_state) { prm.adapt_state[cl] = h_val_(); prm.adapt_state[C + cl] = eps_bar_(); prm.adapt_state[2 * C + cl] = mu_val_(); }
        }
    }
}


}  // namespace tile_nuts
using tile_nuts::nuts_tile_kernel;

}  // namespace mi
