// nuts_dense.hpp -- many-chain NUTS for dense-gradient Gaussian targets on the fp64 matrix cores.
//
// Replaces, for C independent chains, mcmc::internal::nuts_impl
// (/root/reference/src/nuts.cpp:30-332) with nuts_find_initial_step_size and the recursive
// nuts_build_tree (/root/reference/include/mcmc/nuts.ipp:30-93, :97-241), identity preconditioner.
//
// Same wavefront mapping as hmc_dense.hpp (16 chains per wave, theta/p/P*theta register-resident
// in the MFMA B/D layout).  The recursion is run iteratively, leaf by leaf, and reproduces the
// reference's control flow exactly, including its deviations from Hoffman & Gelman:
//   * every doubling restarts from (prev_draw, mntm_vec) (src/nuts.cpp:241-256), and prev_draw /
//     prev_U may already have been replaced earlier in the same draw (:272-273);
//   * the second-half recursive calls cross their edge outputs (nuts.ipp:195,207).  With
//     F = edge on the v side, N = the other edge, that makes for a node with children 1, 2:
//         start(child 2) = F(child 1),   F(node) = N(child 2),   N(node) = N(child 1),
//     and N(subtree) is the state after the subtree's FIRST leaf.  Hence leaf i (i > 0, l = ctz(i))
//     starts from the result of leaf i-1 (l <= 1) or of leaf i - 2^(l-1) (l >= 2), and the U-turn
//     test of the level-l node closing at leaf i uses the results of leaves i-2^l+1 and
//     i-2^(l-1)+1;
//   * one runif per completed second half in post-order; after a failure (s = 0) the pending
//     second-half merges on the unwind path still consume theirs (nuts.ipp:213), first halves
//     do not; alpha / n_alpha are those of the LAST doubling only (src/nuts.cpp:246,255);
//   * the step-size search only ever doubles (nuts.ipp:70-89).
// All alive chains of a wave are at the same doubling depth and leaf index (they start every draw
// together), so slot indices and merge levels are wave-uniform; chains whose tree stopped are
// masked ("wavefront-divergent build_tree": __ballot decides when the wave is done).
//
// Per-chain vectors that do not fit in registers live in an HBM workspace ws[vec][d][C]:
// last accepted state, draw momentum, top-level edges, one (theta,p,P*theta) record per tree level
// (even leaves only) and one pending proposal per level; per-level scalars (n', alpha', n_alpha',
// U of the proposal) live in LDS next to the staged precision matrix.
#pragma once

#include "hmc_dense.hpp"

namespace mi {

struct NutsParams {
    const double* P;
    uint32_t d;
    uint64_t C, chain0;
    double* theta;          // [d][C] in: initial_vals, out: last state
    double* ws;             // [n_waves][64 vectors][NS][64 lanes] workspace, wave-local and contiguous
    double* draws;          // [n_keep][d][C] or nullptr
    uint64_t* n_accept;     // [C] or nullptr
    uint64_t* n_leap;       // [C] or nullptr
    double* step_out;       // [C] or nullptr: final step size
    uint32_t* depth_trace;  // [n_total][C] or nullptr (tests)
    unsigned long long* prof;   // [16] or nullptr: section cycle counters of workgroup 0 wave 0 (profiling builds)
    uint64_t seed;
    uint32_t n_burnin, n_keep, n_adapt, max_depth;
    double delta, eps_bar0, gamma, t0, kappa;
    // general variant of the asynchronous kernel (settings.vals_bound and / or a diagonal precond_mat), see nuts_async.hpp
    uint32_t draw0;         // asynchronous kernel: index of this call's first draw (mi_chains.draw0); > 0 = continuation after the
                            // adaptation window, step sizes come in through step_out
    int vals_bound;         // settings.vals_bound (0: only a diagonal precond_mat)
    const int* btype;
    const double* lb;
    const double* ub;
    const double* m_sqrt;   // diagonal of CHOL_LOWER(precond_mat)
    const double* m_inv;    // diagonal of INV(precond_mat)
    const double* Minv;     // DENSE_M: INV(precond_mat), d*d row-major (device)
    const double* Lchol;    // DENSE_M: CHOL_LOWER(precond_mat), d*d row-major (device)
    uint32_t* nf_flag;      // nuts_gauss_memo_kernel: [C + 1] or nullptr.  A chain that saw a non-finite energy is flagged (nf_flag[c] = 1,
                            // nf_flag[C] = 1) and its outputs are left untouched: the general variant, which reproduces the reference's
                            // dense products in the non-finite regime, replays it
    const uint32_t* replay_flag;   // nuts_gauss_async_kernel as that replay: only the chains with a non-zero entry run (and write)
    uint32_t* next_chain;   // nuts_gauss_memo_kernel (nuts_memo.hpp): chains handed out beyond the first gridDim.x * 64, zeroed by the launcher
    double* adapt_state;    // [3][C] or nullptr (mi_chains.nuts_adapt_state): the dual-averaging state (h, epsilon_bar, mu; nuts.cpp:174-176,
                            // 294-302) -- written at the end of every call, read at the start of a continuation that begins inside the
                            // adaptation window (0 < draw0 <= n_adapt); n_adapt is the RUN's window, draw indices are global (draw0 + i)
    uint64_t* n_exec;       // [C] or nullptr (mi_chains.n_leapfrogs_executed): leapfrogs really computed -- nuts_gauss_memo_kernel writes it; every
                            // other kernel executes what it counts in n_leap, and the host copies that
    // nuts_gauss_memo_kernel<., ., true> (nuts_memo.hpp): the momenta of every draw of every chain, filled by nuts_momenta_kernel before the launch.
    // mom: [n_total][C] blocks of 16 NT doubles in the granule order of a workspace row; msc: [n_total][C] (kinetic energy, log of the slice uniform)
    double* mom;
    double* msc;
    // nuts_gauss_memo_kernel, GaussMemoPolicy::SPLIT (nuts_memo_core.hpp): n_pieces > 1 cuts every chain's run into pieces of piece_len draws handed out as
    // separate work items; piece_q [n_pieces - 1][C] (0xffffffff = not published) and piece_tail [n_pieces] are zeroed / filled by the launcher, which then
    // also makes sure n_accept, n_leap, n_exec, step_out and adapt_state exist (the hand-over goes through them)
    uint32_t n_pieces, piece_len;
    uint32_t* piece_q;
    uint32_t* piece_tail;
    void* split_ws;         // nuts_split_workspace_bytes(C) bytes for the two arrays above and stand-ins for the outputs the caller did not ask for, or nullptr (no pieces)
    uint32_t sep_target;    // general variants: P is the expanded diagonal of an ISO / DIAG target: its gradient is taken element-wise (hmc_dense.hpp: target_times)
};

enum : int {
    V_PREV = 0, V_WPREV = 1, V_MNTM = 2, V_TPOS_T = 3, V_TPOS_P = 4, V_TNEG_T = 5, V_TNEG_P = 6,
    V_LEAF0 = 7,             // slot k: theta 7+3k, p 8+3k, P*theta 9+3k, k = 0..10
    V_PP0 = 40,              // pending proposal of level l at 40 + l, l = 1..11
    NUTS_NVEC = 52,
    NUTS_MAX_DEPTH = 10,
    NUTS_LVLS = 12
};

template <int NT>
__global__ __launch_bounds__(256, 1) void nuts_gauss_mfma_kernel(const NutsParams prm)
{
    constexpr int NS = 4 * NT;
    constexpr int WS_NVEC = 64;     // same allocation as the asynchronous kernel
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    double* lds_P = lds_all;
    double* lds_lvl = lds_all + NT * NS * 64;            // [NUTS_LVLS][4][64]
    stage_precision<NT>(prm.P, prm.d, lds_P);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j4 = lane >> 4;
    const int cw = wave * 16 + (lane & 15);              // chain within the workgroup
    const uint64_t cl = ((uint64_t)blockIdx.x * 4 + wave) * 16 + (lane & 15);
    const bool live = cl < prm.C;
    const uint64_t cld = live ? cl : prm.C - 1;
    const uint64_t chain = prm.chain0 + cl;
    const uint32_t d = prm.d;
    const uint64_t C = prm.C;
    const double* afrag = lds_P + lane;
    const size_t lane_off = (size_t)j4 * C + cld;

    auto lvl = [&](int l, int f) -> double& { return lds_lvl[(l * 4 + f) * 64 + cw]; };
    // element (4s + j4) of workspace vector v of this lane's chain
    // workspace layout [wave][vec][slice][lane]: a vector of a wave's 16 chains is 16 KiB contiguous (one 512-B
    // coalesced access per slice, 4 pages per vector) instead of 128 fragments 4*C*8 bytes apart
    double* const ws_wave = prm.ws + ((size_t)blockIdx.x * 4 + wave) * ((size_t)WS_NVEC * NS * 64) + lane;
    auto wsp = [&](int v, int s) -> double* { return ws_wave + ((size_t)v * NS + s) * 64; };
    auto dim_ok = [&](int s) -> bool { return (uint32_t)(4 * s + j4) < d; };

    double th[NS], pm[NS], w[NS];

    // workspace vectors: unconditional, fully pipelined loads (rows are padded, dead lanes read a clamped chain)
    auto load_vec = [&](int v, double (&x)[NS]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < NS; ++s) x[s] = *wsp(v, s);
    };
    auto store_vec = [&](int v, const double (&x)[NS], bool pred) __attribute__((always_inline)) {
        if (pred && live) {
#pragma unroll
            for (int s = 0; s < NS; ++s) *wsp(v, s) = x[s];
        }
    };
    // per-lane source/destination vector ids
    auto copy_vec = [&](int vsrc, int vdst, bool pred) __attribute__((always_inline)) {
        double tmp[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) tmp[s] = *wsp(vsrc, s);
        if (pred && live) {
#pragma unroll
            for (int s = 0; s < NS; ++s) *wsp(vdst, s) = tmp[s];
        }
    };
    // one leapfrog step of signed size e (nuts.cpp:139-154), grad = -w
    auto leapfrog = [&](double e) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            pm[s] = pm[s] - (e * w[s]) / 2.0;
            th[s] = th[s] + e * pm[s];
        }
        matvec_mfma<NT>(afrag, th, w);
#pragma unroll
        for (int s = 0; s < NS; ++s) pm[s] = pm[s] - (e * w[s]) / 2.0;
    };
    auto potential = [&]() __attribute__((always_inline)) -> double {                   // -box_log_kernel_fn(theta), non-finite -> +inf
        double u = 0.5 * dot4<NS>(th, w);
        if (!is_finite(u)) u = INF;
        return u;
    };
    auto kinetic = [&]() __attribute__((always_inline)) -> double { return dot4<NS>(pm, pm) / 2.0; };
    auto draw_momentum = [&](uint32_t draw, uint32_t stream) __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < NS / 2; ++b) {
            double z0, z1;
            rng_normal_pair(prm.seed, chain, draw, (uint32_t)(4 * b + j4), stream, z0, z1);
            pm[2 * b] = (8u * b + j4 < d) ? z0 : 0.0;
            pm[2 * b + 1] = (8u * b + 4 + j4 < d) ? z1 : 0.0;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // [ (pos - neg) . p_a >= 0 ] * [ (pos - neg) . p_b >= 0 ] with pos/neg = (t2,t1) for v=+1, (t1,t2) for v=-1
    auto uturn_ok = [&](int vt1, int vp1, bool n2_in_regs, int vt2, int vp2, int vdir) __attribute__((always_inline)) -> bool {
        double q1 = 0.0, q2 = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const double t1 = *wsp(vt1, s);
            const double p1 = *wsp(vp1, s);
            const double t2 = n2_in_regs ? th[s] : *wsp(vt2, s);
            const double p2 = n2_in_regs ? pm[s] : *wsp(vp2, s);
            const double dd = (vdir > 0) ? (t2 - t1) : (t1 - t2);
            q1 = dfma(dd, p1, q1);
            q2 = dfma(dd, p2, q2);
        }
        q1 = q1 + __shfl_xor(q1, 32); q1 = q1 + __shfl_xor(q1, 16);
        q2 = q2 + __shfl_xor(q2, 32); q2 = q2 + __shfl_xor(q2, 16);
        return (q1 >= 0.0) && (q2 >= 0.0);
    };

    // ---------------------------------------------------------------- setup (nuts.cpp:156-195)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const uint32_t dimc = dim_ok(s) ? (uint32_t)(4 * s + j4) : 0u;        // clamped row: unconditional load
        const double v = prm.theta[(size_t)dimc * C + cld];
        th[s] = dim_ok(s) ? v : 0.0;
    }
    matvec_mfma<NT>(afrag, th, w);
    store_vec(V_PREV, th, true);
    store_vec(V_WPREV, w, true);
    double prev_U = 0.5 * dot4<NS>(th, w);               // nuts.cpp:181 (no finiteness guard there)

    // nuts_find_initial_step_size (nuts.ipp:30-93) from (first_draw, L z_init)
    draw_momentum(0u, STREAM_INIT);                      // nuts.cpp:166-168
    double eps;
    uint64_t n_leap = 0;
    {
        double U0 = prev_U;
        if (!is_finite(U0)) U0 = INF;                    // nuts.ipp:44-49
        const double K0 = kinetic();                     // :51
        const double log_half = det_log(0.5), neg_log2 = -det_log(2.0);
        eps = 1.0;                                       // :40
        leapfrog(eps);                                   // :58
        n_leap++;
        double dH = -(potential() + kinetic()) + (U0 + K0);
        int a_val = 2 * (dH > log_half ? 1 : 0) - 1;     // :70
        bool cond = dH > neg_log2;                       // :71
        while (__ballot(cond) != 0ull) {
            const double e_new = eps * ((a_val == 1) ? 2.0 : 0.5);   // std::pow(2, a_val), :74
            if (cond) { eps = e_new; n_leap++; }
            leapfrog(eps);                               // continues from the moved state (:76)
            const double dH2 = -(potential() + kinetic()) + (U0 + K0);
            if (cond) {
                a_val = 2 * (dH2 > log_half ? 1 : 0) - 1;            // :88
                cond = dH2 > neg_log2;                               // :89
            }
        }
    }
    const double mu_val = det_log(10 * eps);             // nuts.cpp:174
    double h_val = 0.0;
    double eps_bar = prm.eps_bar0;                       // :59
    uint64_t n_acc = 0;
    const uint32_t n_total = prm.n_burnin + prm.n_keep;
    const uint32_t n_adapt = prm.n_adapt <= n_total ? prm.n_adapt : n_total;    // :54
    const uint32_t max_depth = prm.max_depth;
    bool wprev_dirty = false;                            // V_WPREV is stale w.r.t. V_PREV for this chain

    // ---------------------------------------------------------------- draws (nuts.cpp:199-310)
#pragma unroll 1
    for (uint32_t draw = 0; draw < n_total; ++draw) {
        uint32_t uslot = 0;
        draw_momentum(draw, STREAM_NORMAL);              // :200-202
        store_vec(V_MNTM, pm, true);
        const double prev_K = kinetic();                 // :204
        const double log_u = det_log(rng_uniform(prm.seed, chain, draw, uslot++)) - prev_U - prev_K;   // :206
        // draw_pos = draw_neg = prev_draw, mntm_pos = mntm_neg = mntm_vec (:212-215)
        store_vec(V_TPOS_P, pm, true);
        store_vec(V_TNEG_P, pm, true);
        copy_vec(V_PREV, V_TPOS_T, true);
        copy_vec(V_PREV, V_TNEG_T, true);

        uint32_t depth = 0;                              // wave-uniform for alive chains
        uint32_t my_depth = 0;                           // tree_depth of this chain when its loop ended
        double n_val = 1.0;
        bool alive = max_depth > 0;                      // s_val == 1 && tree_depth < max_tree_depth (:227)
        double alpha_val = 0.0, n_alpha_val = 0.0;       // :221-222
        int good_round = 0;

#pragma unroll 1
        while (__ballot(alive) != 0ull) {
            // ---- one doubling: nuts_build_tree(v, eps, log_u, prev_U, prev_K, prev_draw, mntm_vec, depth)
            const double zdir = rng_uniform(prm.seed, chain, draw, uslot);       // :233
            if (alive) uslot++;
            const int vdir = (zdir <= 0.5) ? -1 : 1;                             // :235
            const double e_signed = (double)vdir * eps;
            const double H0 = prev_U + prev_K;

            if (__ballot(wprev_dirty) != 0ull) {                                 // refresh P*prev_draw
                load_vec(V_PREV, th);
                matvec_mfma<NT>(afrag, th, w);
                store_vec(V_WPREV, w, wprev_dirty);
                wprev_dirty = false;
            }

            const uint32_t n_leaves = 1u << depth;
            bool tact = alive;                           // tree still being built for this chain
            double res_n = 0.0, res_a = 0.0, res_na = 0.0, res_U = 0.0;
            bool res_s = false;

#pragma unroll 1
            for (uint32_t i = 0; i < n_leaves; ++i) {
                if (__ballot(tact) == 0ull) break;
                // ---- start state of leaf i
                if (i == 0) {
                    load_vec(V_PREV, th);
                    load_vec(V_MNTM, pm);
                    load_vec(V_WPREV, w);
                } else {
                    const int l = __builtin_ctz(i);
                    if (l >= 2) {                        // result of leaf i - 2^(l-1): slot l
                        load_vec(V_LEAF0 + 3 * l, th);
                        load_vec(V_LEAF0 + 3 * l + 1, pm);
                        load_vec(V_LEAF0 + 3 * l + 2, w);
                    }
                }
                // ---- depth-0 tree (nuts.ipp:126-158)
                leapfrog(e_signed);                      // :132
                const double pU = potential();           // :134-138
                const double pK = kinetic();             // :140
                double cn = (log_u <= -pU - pK) ? 1.0 : 0.0;              // :146
                bool cs = log_u < 1000.0 - pU - pK;                       // :147
                const double dd = -(pU + pK) + H0;
                double ca = det_exp((dd < 0.0) ? dd : 0.0);               // std::min(0, dd), :157
                double cna = 1.0;
                double cU = pU;
                int cref = -1;                           // proposal: -1 = this leaf (registers), else workspace vector id
                if (tact) n_leap++;
                if ((i & 1u) == 0u || depth == 1u) {     // even leaves are somebody's near edge / restart point;
                                                         // leaf 1 of a depth-1 tree is the tree's far edge
                    const int slot = (i == 0) ? 0 : (__builtin_ctz(i) + 1);
                    store_vec(V_LEAF0 + 3 * slot, th, tact);
                    store_vec(V_LEAF0 + 3 * slot + 1, pm, tact);
                    store_vec(V_LEAF0 + 3 * slot + 2, w, tact);
                }
                // ---- unwind: merges of every level whose second half this leaf completes
                bool failed = tact && !cs;
                bool walking = tact;
                uint32_t pend_level = depth + 1;         // root reached unless stopped earlier
                for (uint32_t l = 1; l <= depth; ++l) {
                    const bool bit = ((i >> (l - 1)) & 1u) != 0u;
                    if (!bit) {                          // first half of the level-l node: non-failed chains wait here
                        if (walking && !failed) { pend_level = l; walking = false; }
                        if (__ballot(walking) == 0ull) break;
                        continue;
                    }
                    if (__ballot(walking) == 0ull) break;
                    // second half complete: merge with the pending first half (nuts.ipp:212-229)
                    const double z = rng_uniform(prm.seed, chain, draw, uslot);  // :213
                    if (walking) {
                        uslot++;
                        const double p_n = lvl(l, 0), p_a = lvl(l, 1), p_na = lvl(l, 2), p_U = lvl(l, 3);
                        const double prob = cn / (p_n + cn);                     // :212
                        if (!(z < prob)) { cref = V_PP0 + (int)l; cU = p_U; }    // keep new_draw_p (:215-217)
                        cn = p_n + cn;                                           // :220-222
                        ca = p_a + ca;
                        cna = p_na + cna;
                    }
                    const bool need_uturn = walking && !failed;
                    if (__ballot(need_uturn) != 0ull) {
                        const uint32_t b = i - (1u << l) + 1;                    // first leaf of the node
                        const int slot1 = (b == 0) ? 0 : (__builtin_ctz(b) + 1);
                        const bool ok = uturn_ok(V_LEAF0 + 3 * slot1, V_LEAF0 + 3 * slot1 + 1, l == 1,
                                                 V_LEAF0 + 3 * (int)l, V_LEAF0 + 3 * (int)l + 1, vdir);   // :226-227
                        if (need_uturn && !ok) failed = true;                    // s' = s'' * check1 * check2 (:229)
                    }
                }
                // ---- bookkeeping for chains still in the tree
                if (tact && !failed) {
                    // carried subtree becomes the pending first half of level pend_level (or the tree's
                    // result when pend_level == depth + 1): proposal by value, scalars to LDS
                    lvl((int)pend_level, 0) = cn; lvl((int)pend_level, 1) = ca;
                    lvl((int)pend_level, 2) = cna; lvl((int)pend_level, 3) = cU;
                }
                {
                    const bool do_store = tact && !failed && live;
                    if (__ballot(do_store) != 0ull) {
                        const int vdst = V_PP0 + (int)pend_level;
                        const int csrc = (cref < 0) ? V_PREV : cref;    // any valid vector when the leaf itself is kept
                        double tmp[NS];
#pragma unroll
                        for (int s = 0; s < NS; ++s) tmp[s] = *wsp(csrc, s);
                        if (do_store) {
#pragma unroll
                            for (int s = 0; s < NS; ++s) *wsp(vdst, s) = (cref < 0) ? th[s] : tmp[s];
                        }
                    }
                }
                if (tact && failed) {                    // tree returns s' = 0
                    res_n = cn; res_a = ca; res_na = cna; res_s = false; tact = false;
                }
                if (tact && i == n_leaves - 1) {         // root complete, s' = 1
                    res_n = cn; res_a = ca; res_na = cna; res_U = cU; res_s = true; tact = false;
                }
            }

            // ---- back in nuts_impl (src/nuts.cpp:258-289)
            if (alive) { alpha_val = res_a; n_alpha_val = res_na; }              // overwritten by every doubling
            bool s_ok = false;
            if (alive && res_s) {
                const double z = rng_uniform(prm.seed, chain, draw, uslot++);    // :261
                const bool take = z < res_n / n_val;                             // :263
                if (take) {
                    prev_U = res_U;                                              // :264-273 (U recomputed = same bits)
                    good_round = 1;                                              // :277
                    wprev_dirty = true;
                }
                copy_vec(V_PP0 + (int)depth + 1, V_PREV, take);
                // far edge of the tree = near edge of its second half (slot depth), or the leaf itself
                const int fslot = (depth == 0) ? 0 : (int)depth;
                const int vt = (vdir > 0) ? V_TPOS_T : V_TNEG_T, vp = (vdir > 0) ? V_TPOS_P : V_TNEG_P;
                copy_vec(V_LEAF0 + 3 * fslot, vt, true);
                copy_vec(V_LEAF0 + 3 * fslot + 1, vp, true);
            }
            if (alive) n_val = n_val + res_n;                                    // :283
            if (__ballot(alive && res_s) != 0ull) {
                double q1 = 0.0, q2 = 0.0;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const double tp = *wsp(V_TPOS_T, s), tn = *wsp(V_TNEG_T, s);
                    const double pp = *wsp(V_TPOS_P, s), pn = *wsp(V_TNEG_P, s);
                    const double dd = tp - tn;
                    q1 = dfma(dd, pn, q1);                                       // :286
                    q2 = dfma(dd, pp, q2);                                       // :287
                }
                q1 = q1 + __shfl_xor(q1, 32); q1 = q1 + __shfl_xor(q1, 16);
                q2 = q2 + __shfl_xor(q2, 32); q2 = q2 + __shfl_xor(q2, 16);
                s_ok = res_s && (q1 >= 0.0) && (q2 >= 0.0);                      // :289
            }
            depth += 1;                                                          // :284
            if (alive) my_depth = depth;
            alive = alive && s_ok && (depth < max_depth);
        }

        // ---- dual averaging (src/nuts.cpp:294-302)
        if (prm.depth_trace && live && j4 == 0) prm.depth_trace[(size_t)draw * C + cl] = my_depth;
        if (draw < n_adapt) {
            const double it = (double)(draw + 1);
            h_val = h_val + (1.0 / (it + prm.t0)) * (prm.delta - (alpha_val / n_alpha_val) - h_val);
            eps = det_exp(mu_val - h_val * __builtin_sqrt(it) / prm.gamma);
            eps_bar = eps_bar * det_exp(det_pow(it, -prm.kappa) * (det_log(eps) - det_log(eps_bar)));
        } else {
            eps = eps_bar;
        }
        if (draw >= prm.n_burnin) {                      // :306-309
            n_acc += (uint64_t)good_round;
            if (prm.draws != nullptr && live) {
                double* out = prm.draws + (size_t)(draw - prm.n_burnin) * d * C;
                double tmp[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) tmp[s] = *wsp(V_PREV, s);
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    if (dim_ok(s)) (out + (size_t)(4 * s) * C)[lane_off] = tmp[s];
            }
        }
    }

    if (live) {
        double tmp[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) tmp[s] = *wsp(V_PREV, s);
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (dim_ok(s)) prm.theta[(size_t)(4 * s) * C + lane_off] = tmp[s];
        if (j4 == 0) {
            if (prm.n_accept) prm.n_accept[cl] = n_acc;
            if (prm.n_leap) prm.n_leap[cl] = n_leap;
            if (prm.step_out) prm.step_out[cl] = eps;
        }
    }
}

}  // namespace mi
