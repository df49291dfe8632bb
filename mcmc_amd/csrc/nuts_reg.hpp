// nuts_reg.hpp -- many-chain NUTS, asynchronous per-chain state machine with REGISTER-CARRIED leaf state (the production
// kernel of the plain case: unbounded, identity precond_mat; BASELINE configs[3]).
//
// Same algorithm, arithmetic, tree derivation and record layout as nuts_async.hpp (reference:
// /root/reference/src/nuts.cpp:30-332, include/mcmc/nuts.ipp:30-241; the iterative leaf-indexed tree is derived in
// nuts_dense.hpp), and the same bits.  nuts_async.hpp keeps the state of a leaf tick-local: every tick loads a start record
// (theta, p, P theta: 3 KB per chain) and stores the leaf's record (3 KB), and the level-1 U-turn test of an odd leaf reloads
// four vectors (4 KB) -- measured 11.3 KB of HBM traffic per chain-leaf, 3.7 TB/s (profiles/r2_c4_pmc.json): the kernel sits
// on memory traffic, not on the matrix pipe (24 % busy).  But three leaves in four start from the RESULT OF THE PREVIOUS LEAF
// (leaf i starts from leaf i-1 unless ctz(i) >= 2, nuts_dense.hpp), which is in registers when the tick ends.  So here:
//   * (theta, p, P theta) of a chain's last leaf stay in registers across ticks; a tick loads a start record only for the
//     lanes with li == 0 or ctz(li) >= 2;
//   * ODD leaves never go to memory: their record is only ever the next leaf's start (registers), the second operand of
//     their own level-1 U-turn test (registers: the test's first operand, leaf li-1, IS the start state of this leapfrog, so
//     d = theta_end - theta_start and d.p_start fall out of the kick / drift loop and d.p_end out of the second kick), and
//     possibly the carried proposal (stored to its destination straight from registers);
//   * the U-turn test of a level-l node (l >= 2) is evaluated EAGERLY, at the tick of the first leaf of its second half (an even
//     leaf, in registers) against the node's first leaf (two vectors from memory, fetched together with the start records at
//     the top of the tick) instead of four vectors in a dependent round trip per level when the node closes: the unwind of a
//     tick (nuts.ipp:212-229) then runs on LDS scalars and a bit mask;
//   * pending copies, edges and the top-level test of a doubling use the records in memory as before.
#pragma once

#include "nuts_async.hpp"

#ifndef MI_NUTS_R_CHU
#define MI_NUTS_R_CHU 8      // top-level U-turn test of a doubling: 4 vectors per chunk
#endif
#ifndef MI_NUTS_R_CHC
#define MI_NUTS_R_CHC 16     // record copies: 2 vectors per chunk
#endif
#ifndef MI_NUTS_R_CHE
#define MI_NUTS_R_CHE 16     // eager U-turn operands (theta, p of the node's first leaf): slices per streamed chunk, two chunks in flight
#endif

namespace mi {

template <int NT>
__global__ MI_NO_DS_MERGE __launch_bounds__(256, 1) void nuts_gauss_reg_kernel(const NutsParams prm, const uint32_t refresh_batch)
{
    constexpr int NS = 4 * NT;
    constexpr int WS_NVEC = NUTS_NVEC_ASYNC;
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    double* lds_P = lds_all;
    double* lds_lvl = lds_all + NT * NS * 64;            // [NUTS_LVLS][4][64]
    stage_precision<NT>(prm.P, prm.d, lds_P);            // ends with a barrier

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j4 = lane >> 4;
    const int cw = wave * 16 + (lane & 15);
    const uint64_t cl = ((uint64_t)blockIdx.x * 4 + wave) * 16 + (lane & 15);
    const bool live = cl < prm.C;
    const uint64_t cld = live ? cl : prm.C - 1;
    const uint64_t chain = prm.chain0 + cl;
    const uint32_t d = prm.d;
    const uint64_t C = prm.C;
    const double* afrag = lds_P + lane;
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

    constexpr int CHU = (NS < MI_NUTS_R_CHU) ? NS : MI_NUTS_R_CHU;
    constexpr int CHC = (NS < MI_NUTS_R_CHC) ? NS : MI_NUTS_R_CHC;
    auto copy_vec = [&](int vsrc, int vdst, bool pred) __attribute__((always_inline)) {
        if (pred && live) {
#pragma unroll
            for (int c0 = 0; c0 < NS; c0 += CHC) {
                double tmp[CHC];
                ld_row(vsrc, c0, tmp);
                st_row(vdst, c0, tmp);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // [ (pos - neg) . p_1 >= 0 ] * [ (pos - neg) . p_2 >= 0 ], pos/neg = (t2,t1) for v=+1, (t1,t2) for v=-1; operands from the records
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

    // the chain's last leaf: position, momentum, P * position (MFMA B / D layout).  Loop-carried: see the header.
    double th[NS], pm[NS], w[NS];

    // ---------------------------------------------------------------- setup (nuts.cpp:156-195), all chains together
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const uint32_t dimc = dim_ok(s) ? (uint32_t)(4 * s + j4) : 0u;
        const double v = prm.theta[(size_t)dimc * C + cld];
        th[s] = dim_ok(s) ? v : 0.0;
    }
    matvec_mfma<NT>(afrag, th, w);
    if (live) { st_row(V_PREV, 0, th); st_row(V_WPREV, 0, w); }
    double prev_U = 0.5 * dot4<NS>(th, w);               // nuts.cpp:181 (no finiteness guard there)

    uint64_t n_leap = 0;
    double eps;
    if (prm.draw0 == 0) {   // nuts_find_initial_step_size (nuts.ipp:30-93) from (first_draw, z_init), nuts.cpp:166-172
        auto leapfrog = [&](double e) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < NS; ++s) pm[s] = pm[s] - (e * w[s]) / 2.0;
#pragma unroll
            for (int s = 0; s < NS; ++s) th[s] = th[s] + e * pm[s];
            matvec_mfma<NT>(afrag, th, w);
#pragma unroll
            for (int s = 0; s < NS; ++s) pm[s] = pm[s] - (e * w[s]) / 2.0;
        };
        auto energy = [&]() __attribute__((always_inline)) -> double {
            double u = 0.5 * dot4<NS>(th, w);
            if (!is_finite(u)) u = INF;
            return u + dot4<NS>(pm, pm) / 2.0;
        };
#pragma unroll
        for (int b = 0; b < NS / 2; ++b) {
            double z0, z1;
            rng_normal_pair(prm.seed, chain, 0u, (uint32_t)(4 * b + j4), STREAM_INIT, z0, z1);
            pm[2 * b] = (8u * b + j4 < d) ? z0 : 0.0;
            pm[2 * b + 1] = (8u * b + 4 + j4 < d) ? z1 : 0.0;
            __builtin_amdgcn_sched_barrier(0);
        }
        double U0 = prev_U;
        if (!is_finite(U0)) U0 = INF;
        const double K0 = dot4<NS>(pm, pm) / 2.0;
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
    const double mu_val = det_log(10 * eps);             // nuts.cpp:174
    double h_val = 0.0;
    double eps_bar = (prm.draw0 == 0) ? prm.eps_bar0 : eps;
    uint64_t n_acc = 0;
    const uint32_t n_total = prm.n_burnin + prm.n_keep;
    const uint32_t n_adapt = prm.n_adapt <= n_total ? prm.n_adapt : n_total;
    const uint32_t max_depth = prm.max_depth;

    // ---------------------------------------------------------------- per-chain state
    int state = (n_total > 0) ? NS_NEED_DRAW : NS_DONE;
    uint32_t draw = 0;           // this chain's draw index
    uint32_t jd = 0;             // depth of the doubling in progress
    uint32_t li = 0;             // next leaf of that doubling
    uint32_t uslot = 0;
    int vdir = 1;
    double e_signed = 0.0, H0 = 0.0, prev_K = 0.0, log_u = 0.0, n_val = 1.0;
    double alpha_val = 0.0, n_alpha_val = 0.0;
    int good_round = 0;
    uint32_t utpre = 0;          // bit l: the U-turn test of the open level-l node passed (set when the first leaf of its second half ran)
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
            if (draw < n_adapt) {
                const double it = (double)(draw + 1);
                h_val = h_val + (1.0 / (it + prm.t0)) * (prm.delta - (alpha_val / n_alpha_val) - h_val);
                eps = det_exp(mu_val - h_val * __builtin_sqrt(it) / prm.gamma);
                eps_bar = eps_bar * det_exp(det_pow(it, -prm.kappa) * (det_log(eps) - det_log(eps_bar)));
            } else {
                eps = eps_bar;
            }
        }
        const bool kept = p && draw >= prm.n_burnin;
        if (kept) n_acc += (uint64_t)good_round;
        if (__ballot(kept && prm.draws != nullptr) != 0ull) {
            if (kept && prm.draws != nullptr && live) {
                double* out = prm.draws + (size_t)(draw - prm.n_burnin) * d * C;
#pragma unroll
                for (int c0 = 0; c0 < NS; c0 += CHC) {
                    double tmp[CHC];
                    ld_row(V_PREV, c0, tmp);
#pragma unroll
                    for (int k = 0; k < CHC; ++k)
                        if (dim_ok(c0 + k)) (out + (size_t)(4 * (c0 + k)) * C)[lane_off] = tmp[k];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (p) {
            draw++;
            state = (draw < n_total) ? NS_NEED_DRAW : NS_DONE;
        }
    };

#pragma unroll 1
    while (__ballot(state != NS_DONE) != 0ull) {
        asm volatile("" : "+v"(lane_b));
        // ------------------------------------------------------------ A. momentum refresh for waiting chains
        const unsigned n_wait = __popcll(__ballot(state == NS_NEED_DRAW)) / 4;
        const unsigned n_run = __popcll(__ballot(state == NS_TREE)) / 4;
        if (n_wait >= refresh_batch || (n_run == 0 && n_wait > 0)) {
            finish_draw(state == NS_NEED_DRAW && fin_pending, fin_depth);   // epilogue of the draws that just ended
            const bool p = state == NS_NEED_DRAW;
            double kq = 0.0;
#pragma unroll 1
            for (int b = 0; b < NS / 2; ++b) {               // nuts.cpp:200-202, this chain's own draw index
                double z0, z1;
                rng_normal_pair(prm.seed, chain, draw + prm.draw0, (uint32_t)(4 * b + j4), STREAM_NORMAL, z0, z1);
                const double pa = (8u * b + j4 < d) ? z0 : 0.0;
                const double pb = (8u * b + 4 + j4 < d) ? z1 : 0.0;
                kq = dfma(pa, pa, kq);
                kq = dfma(pb, pb, kq);
                if (p && live) {                              // mntm_vec, mntm_pos, mntm_neg (:202, :214-215)
                    st_pair(V_MNTM, 2 * b, pa, pb);
                    st_pair(V_TPOS_P, 2 * b, pa, pb);
                    st_pair(V_TNEG_P, 2 * b, pa, pb);
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
                const int vt = (li == 0) ? V_PREV : V_LEAF0 + 3 * cz;                    // leaf li - 2^(cz-1) sits in slot cz
                const int vp = (li == 0) ? V_MNTM : V_LEAF0 + 3 * cz + 1;
                const int vw = (li == 0) ? V_WPREV : V_LEAF0 + 3 * cz + 2;
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
        // one leapfrog of signed size e (nuts.ipp:132, nuts.cpp:139-154), grad = -w;  d = theta(b2) - theta(b) (by direction),
        // q1 = d . p(b) fall out of the kick / drift loop, q2 = d . p(b2) out of the second kick.  The rows of leaf b stream
        // through in chunks of CHE slices, one chunk ahead of its use.
        double dd[NS];
        double q1 = 0.0, q2 = 0.0;
        {
            constexpr int CHE = (NS < MI_NUTS_R_CHE) ? NS : MI_NUTS_R_CHE;
            constexpr int NCE = NS / CHE;
            double ra_t[CHE], ra_p[CHE], rn_t[CHE], rn_p[CHE];
#pragma unroll
            for (int k = 0; k < CHE; ++k) { ra_t[k] = 0.0; ra_p[k] = 0.0; rn_t[k] = 0.0; rn_p[k] = 0.0; }
            if (any_eager) { if (eager) { ld_row(eb_t, 0, ra_t); ld_row(eb_p, 0, ra_p); } }
#pragma unroll
            for (int c = 0; c < NCE; ++c) {
                if (c + 1 < NCE && any_eager) { if (eager) { ld_row(eb_t, (c + 1) * CHE, rn_t); ld_row(eb_p, (c + 1) * CHE, rn_p); } }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < CHE; ++k) {
                    const int s_ = c * CHE + k;
                    const double p0 = pm[s_], t0 = th[s_];
                    pm[s_] = p0 - (e_signed * w[s_]) / 2.0;
                    th[s_] = t0 + e_signed * pm[s_];
                    const double rt = odd ? t0 : ra_t[k], rp = odd ? p0 : ra_p[k];
                    dd[s_] = (vdir > 0) ? (th[s_] - rt) : (rt - th[s_]);
                    q1 = dfma(dd[s_], rp, q1);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < CHE; ++k) { ra_t[k] = rn_t[k]; ra_p[k] = rn_p[k]; }
            }
        }
        matvec_mfma<NT>(afrag, th, w);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            pm[s] = pm[s] - (e_signed * w[s]) / 2.0;
            q2 = dfma(dd[s], pm[s], q2);
        }
        double pU = 0.5 * dot4<NS>(th, w);               // nuts.ipp:134-138
        const double pK = dot4<NS>(pm, pm) / 2.0;        // :140
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
            // a pending first half at level 1 IS the (even) leaf's record just written (referenced, not copied); deeper levels
            // and accepted proposals are written to their slot: from the registers when the carried proposal is this leaf,
            // record -> slot otherwise
            const bool do_store = keep && (complete ? take : (pend_level > 1u)) && live;
            if (__ballot(do_store) != 0ull) {
                const int pl = do_store ? (int)pend_level : 1;
                const int dst_t = take ? V_PREV : V_PP0 + pl;
                const int dst_w = take ? V_WPREV : V_PPW0 + pl;
                if (do_store && cref_regs) { st_row(dst_t, 0, th); st_row(dst_w, 0, w); }
                const bool do_copy = do_store && !cref_regs;
                if (__ballot(do_copy) != 0ull) {
                    if (do_copy) {
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
        }
        if (__ballot(fin) != 0ull) {
            if (fin) { alpha_val = ca; n_alpha_val = cna; n_val = n_val + cn; }   // :246,255 ; :283
            bool s_ok = false;
            if (__ballot(complete) != 0ull)
                s_ok = uturn_ok(complete, V_TNEG_T, V_TNEG_P, V_TPOS_T, V_TPOS_P, 1) && complete;   // :286-289
            const bool more = fin && s_ok && (jd + 1 < max_depth);
            if (fin) jd = jd + 1;                                        // :284
            begin_doubling(more);
            if (fin && !more) { fin_pending = true; fin_depth = jd; state = NS_NEED_DRAW; }
        }
        if (run && !fin) li = li + 1;
    }

    if (live) {
#pragma unroll
        for (int c0 = 0; c0 < NS; c0 += CHC) {
            double tmp[CHC];
            ld_row(V_PREV, c0, tmp);
#pragma unroll
            for (int k = 0; k < CHC; ++k)
                if (dim_ok(c0 + k)) prm.theta[(size_t)(4 * (c0 + k)) * C + lane_off] = tmp[k];
        }
        if (j4 == 0) {
            if (prm.n_accept) prm.n_accept[cl] = n_acc;
            if (prm.n_leap) prm.n_leap[cl] = n_leap;
            if (prm.step_out) prm.step_out[cl] = eps;
        }
    }
}

}  // namespace mi
