// tile_samplers.hpp -- mcmc::hmc / mcmc::mala for USER-DEFINED targets on the tiled (MFMA-layout) engine: the draw loops of
// hmc_dense.hpp / mala_dense.hpp with the gradient behind a compile-time policy instead of the built-in dense mat-vec.
//
// The reference takes the target as a callback (ref: include/mcmc/hmc.hpp:42-48: fp_t (const ColVec_t& vals, ColVec_t* grad_out,
// void* data)).  Its device form at the dimensions this engine is built for is a TILE functor: one call evaluates value and gradient
// for the 16 chains of a wavefront, whose vectors sit in registers in the MFMA B / D lane layout
//     lane l, register s  <->  dimension 4 s + (l >> 4) of chain (l & 15),      s = 0 .. NS - 1,  NS = 4 NT,  d <= 16 NT
// (hmc_dense.hpp: the D layout of a 16 x 16 x 4 fp64 MFMA tile is the B layout of the next product, so a dense mat-vec maps vectors
// in this layout to vectors in this layout with no cross-lane traffic).  include/mi_mcmc_tile_target.hpp is the user-facing side.
//
//   struct T {
//       static constexpr int NT = 8;                                   // 1, 2, 4 or 8
//       static constexpr int WPB = 8;                                  // optional: waves per workgroup, 4 (default) or 8
//       size_t lds_doubles() const;                                    // host: LDS the target wants (its matrices in fragment order)
//       __device__ void stage(double* lds) const;                      // once per workgroup, every thread; a barrier follows
//       __device__ void grad_tile(const double* lds, const double (&theta)[4 * NT], double (&grad)[4 * NT], double& value, bool want_value) const;
//   };
// grad_tile returns the gradient and, when want_value (a compile-time constant at every call site: the leapfrog loop needs the value
// only after its last step), the log kernel (the same bits in the four lanes of a chain: reduce with mi::dot4); padding
// dimensions (>= d) hold zeros on entry and must get a zero gradient.  Arithmetic is the user's statement: a host function with the
// same operation order, handed to the oracle as the reference's callback, reproduces the device draws bit for bit
// (examples/user_tile_target.hip, tests/test_user_tile_target.py).
//
// The reference's dense `inv_precond_matrix * mntm` / `precond_matrix * grad_obj` (identity here: no precond_mat, no bounds on this
// route) are applied element-wise with the explicit NaN rule of hmc_dense.hpp (dense_product_poison), so the non-finite regime needs
// no replay on this route.
#pragma once

#include <type_traits>

#include "hmc_dense.hpp"

namespace mi {

struct TileParams {
    uint32_t d;
    uint64_t C, chain0;
    double* theta;          // [d][C] in/out
    double* wsave;          // [n_tiles][3][NS][64]: last accepted theta and gradient
    double* draws;          // [n_keep][d][C] or nullptr
    uint64_t* n_accept;
    uint64_t* n_leap;
    uint64_t seed;
    uint32_t n_burnin, n_keep, n_leap_steps, draw0;
    double eps;             // step_size
    double s2, rs, cons_term, log_det;      // mala: eps^2, 1 / eps^2, -d log(2 pi) / 2, LOG_DET(eps^2 I) (host, the oracle's order)
    // nuts (nuts_tile.hpp)
    double* ws;             // nuts: memo::memo_wave_bytes(NS) per wave (16 chain slots): point records + scalar table (nuts_memo_core.hpp)
    double* step_out;       // [C] or nullptr: final step size (in: the adapted step sizes of a continuation, draw0 > 0)
    uint32_t* depth_trace;  // [n_total][C] or nullptr
    double* adapt_state;    // [3][C] or nullptr: dual-averaging state (h, epsilon_bar, mu), mi_chains.nuts_adapt_state
    uint32_t n_adapt, max_depth;
    double delta, eps_bar0, gamma, t0, kappa;
    uint32_t lds_user_doubles;              // the target's own LDS (T::lds_doubles()); the sampler's tables follow it
    // general runs (hmc, nuts): settings.vals_bound and / or a DIAGONAL precond_mat (TileGen below); all device, d values each
    int vals_bound;
    const int* btype;       // determine_bounds_type.hpp:27-57
    const double* lb;
    const double* ub;
    const double* m_sqrt;   // diag of CHOL_LOWER(precond_mat) (ones = identity)
    const double* m_inv;    // diag of INV(precond_mat)
    // nuts (nuts_memo_core.hpp): leapfrogs really computed [C] or nullptr; the fields of the built-in kernel's dynamic hand-out and replay, unused
    // on this route (every chain has its own slot; the tile policy applies the reference's NaN rules itself): nullptr
    uint64_t* n_exec;
    uint32_t* next_chain;   // nuts: the chain counter of the persistent grid (nuts_grid workgroups; chains beyond its 64 nuts_grid slots are handed out as slots
                            // fall free), or nullptr: one workgroup per 64 chains, every chain in its own slot
    uint32_t* nf_flag;
    unsigned long long* prof;
    uint32_t nuts_grid;     // nuts: workgroups to launch (the engine: min(tiles, CUs) -- the workspace, 148 vectors per chain SLOT at max_tree_depth 10, is sized
                            // by the grid, not by the chains); 0 = (C + 63) / 64
    // mala with a DIAGONAL precond_mat, no bounds (mala_tile_kernel<T, true>; round 6): the diagonal of precond_mat and of INV(eps^2 precond_mat), d values each on
    // the device (m_sqrt above: of CHOL_LOWER(precond_mat)); log_det carries LOG_DET(eps^2 precond_mat).  nullptr: the identity
    const double* m;
    const double* s_inv;
    // nuts on a persistent grid (next_chain set) with more chains than chain slots: the runs cut into n_pieces pieces of piece_len draws that migrate between slots
    // (nuts_memo_core.hpp, SPLIT); piece_q [n_pieces - 1][C] and piece_tail [n_pieces] set up by the engine (launch_common.hpp: memo_setup_pieces), which then
    // also provides n_accept, n_leap, n_exec, step_out and adapt_state (the hand-over goes through them).  n_pieces <= 1: whole chains per slot
    uint32_t n_pieces, piece_len;
    uint32_t* piece_q;
    uint32_t* piece_tail;
};

// ---- settings.vals_bound and / or a diagonal precond_mat on the tile route, with the arithmetic of the general built-in kernels
// (hmc_dense.hpp; the reference's NUTS shares mntm_update_fn / leap_frog_fn / box_log_kernel with HMC, ref: src/hmc.cpp:84-128,
// src/nuts.cpp:84-154): the chain lives in the transformed space, the target is evaluated at x = inv_transform(theta), the kick uses
// [J^-1] grad (a dense product in the reference: NaN rule), the drift Minv p (likewise), K = p.(Minv p) / 2, U = -(K(x) +
// log_jacobian(theta)) with the log-Jacobian summed over the bounded dimensions in order, p = sqrt(M) z; rows are reported through
// inv_transform.  Identity tables reproduce the plain kernels' bits.  Tables: 5 x 16 NT doubles + 16 NT ints of LDS.
template <int NT>
struct TileGen {
    static constexpr int NS = 4 * NT;
    static constexpr size_t lds_doubles() { return (size_t)16 * NT * 4 + 8 * NT; }
    const double* lb; const double* ub; const double* ms; const double* mi; const int* bt;
    uint32_t bslices, d;
    int vals_bound, j, lane;

    // every thread of the workgroup; a barrier must follow before use()
    __device__ __forceinline__ static void stage(double* tab, const TileParams& prm)
    {
        double* l = tab; double* u = l + 16 * NT; double* s_ = u + 16 * NT; double* i_ = s_ + 16 * NT;
        int* b = reinterpret_cast<int*>(i_ + 16 * NT);
        for (int k = threadIdx.x; k < 16 * NT; k += blockDim.x) {
            const bool in = (uint32_t)k < prm.d;
            l[k] = in ? prm.lb[k] : 0.0; u[k] = in ? prm.ub[k] : 0.0; b[k] = in ? prm.btype[k] : 1;
            s_[k] = in ? prm.m_sqrt[k] : 1.0; i_[k] = in ? prm.m_inv[k] : 1.0;
        }
    }
    __device__ __forceinline__ void use(double* tab, const TileParams& prm)
    {
        lb = tab; ub = lb + 16 * NT; ms = ub + 16 * NT; mi = ms + 16 * NT; bt = reinterpret_cast<const int*>(mi + 16 * NT);
        d = prm.d; vals_bound = prm.vals_bound; lane = threadIdx.x & 63; j = lane >> 4;
        uint32_t m = 0;
        if (vals_bound)
            for (int s = 0; s < NS; ++s) {
                const bool any = bt[4 * s] != 1 || bt[4 * s + 1] != 1 || bt[4 * s + 2] != 1 || bt[4 * s + 3] != 1;
                m |= (any ? 1u : 0u) << s;
            }
        bslices = (uint32_t)__builtin_amdgcn_readfirstlane((int)m);     // bit s: slice s holds a bounded dimension (wave-uniform)
    }
    __device__ __forceinline__ bool bounded(int s) const { return ((bslices >> s) & 1u) != 0u; }
    __device__ __forceinline__ double enter(double v, uint32_t dim) const { return box_transform(v, bt[dim], lb[dim], ub[dim]); }       // hmc.cpp:134-136
    __device__ __forceinline__ double leave(double v, int s) const                                                                    // :211-218
    {
        const int dim = 4 * s + j;
        return bounded(s) ? box_inv_transform(v, bt[dim], lb[dim], ub[dim]) : v;
    }
    __device__ __forceinline__ void x_of(const double (&th)[NS], double (&xs)[NS]) const                                                // :108
    {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = 4 * s + j;
            if (bounded(s)) xs[s] = ((uint32_t)i < d) ? box_inv_transform(th[s], bt[i], lb[i], ub[i]) : 0.0;
            else xs[s] = ((uint32_t)i < d) ? th[s] : 0.0;
        }
    }
    // t = (e * ([J^-1] grad)) / 2 at (theta, grad)  (hmc.cpp:122,126)
    __device__ __forceinline__ void kick_terms(const double (&th)[NS], const double (&g)[NS], double e, double (&t)[NS]) const
    {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = 4 * s + j;
            if (bounded(s)) t[s] = box_inv_jacobian(th[s], bt[i], lb[i], ub[i]) * g[s];
            else t[s] = 1.0 * g[s];
        }
        if (vals_bound) dense_product_poison<NS>(g, t, j, d);
#pragma unroll
        for (int s = 0; s < NS; ++s) t[s] = (e * t[s]) / 2.0;
    }
    __device__ __forceinline__ void minv_p(const double (&pm)[NS], double (&mp)[NS]) const                                              // inv_precond_matrix * mntm
    {
        const double* mic = mi + j;
        asm volatile("" : "+v"(mic));
#pragma unroll
        for (int s = 0; s < NS; ++s) mp[s] = mic[4 * s] * pm[s];
        dense_product_poison<NS>(pm, mp, j, d);
    }
    __device__ __forceinline__ double kinetic(const double (&pm)[NS]) const                                                             // :160,184
    {
        double mp[NS];
        minv_p(pm, mp);
        double q = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) q = dfma(pm[s], mp[s], q);
        q = q + __shfl_xor(q, 32);
        q = q + __shfl_xor(q, 16);
        return q / 2.0;
    }
    // U = -box_log_kernel(theta) given the log kernel `val` at x(theta) (hmc.cpp:84-95; log_jacobian.hpp:36-57: a scalar loop, i ascending)
    __device__ __forceinline__ double potential(double val, const double (&th)[NS]) const
    {
        double lj = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (!bounded(s)) continue;
            const int i0 = 4 * s;
            const double term = box_log_jacobian_term(th[s], bt[i0 + j], lb[i0 + j], ub[i0 + j]);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const double tg = __shfl(term, (lane & 15) + 16 * g);
                if ((uint32_t)(i0 + g) < d && bt[i0 + g] != 1) lj = lj + tg;
            }
        }
        return -(val + lj);
    }
    __device__ __forceinline__ double msqrt(int dim) const { return ms[dim]; }
};

template <class T, class = void> struct tile_wpb_of { static constexpr int value = 4; };
template <class T> struct tile_wpb_of<T, std::void_t<decltype(T::WPB)>> { static constexpr int value = T::WPB; };
template <class T> constexpr int tile_wpb() { return tile_wpb_of<T>::value; }

// x . (D x) for a diagonal D applied as the dense product it is in the reference (dg == 1: the identity; else the constant dg on
// the diagonal): the element-wise sum unless an entry of x is non-finite -- then every OTHER row of D x is NaN (0 * inf), and so is
// the sum whenever the chain has a second dimension.  The rare branch forms the poisoned product literally.
template <int NS>
__device__ __forceinline__ double diag_quadratic(const double (&x)[NS], double dg, int j, uint32_t d)
{
    double q = 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s) q = dfma(x[s], dg * x[s], q);
    q = q + __shfl_xor(q, 32);
    q = q + __shfl_xor(q, 16);
    if (__ballot(!is_finite(q)) != 0ull) {
        double y[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) y[s] = dg * x[s];
        dense_product_poison<NS>(x, y, j, d);
        double q2 = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) q2 = dfma(x[s], y[s], q2);
        q2 = q2 + __shfl_xor(q2, 32);
        q2 = q2 + __shfl_xor(q2, 16);
        q = q2;
    }
    return q;
}

// ... with a diagonal matrix diag(dg) (slice s of this lane: dg[s]) instead of dg * I
template <int NS>
__device__ __forceinline__ double diag_quadratic_v(const double (&x)[NS], const double (&dg)[NS], int j, uint32_t d)
{
    double q = 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s) q = dfma(x[s], dg[s] * x[s], q);
    q = q + __shfl_xor(q, 32);
    q = q + __shfl_xor(q, 16);
    if (__ballot(!is_finite(q)) != 0ull) {
        double y[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) y[s] = dg[s] * x[s];
        dense_product_poison<NS>(x, y, j, d);
        double q2 = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) q2 = dfma(x[s], y[s], q2);
        q2 = q2 + __shfl_xor(q2, 32);
        q2 = q2 + __shfl_xor(q2, 16);
        q = q2;
    }
    return q;
}

// mcmc::hmc (ref: src/hmc.cpp:155-205), identity preconditioner, no bounds
template <class T, int WPB>
__global__ MI_NO_DS_MERGE __launch_bounds__(64 * WPB, WPB / 4) void hmc_tile_kernel(const TileParams prm, const T tgt)
{
    constexpr int NT = T::NT, NS = 4 * NT;
    extern __shared__ __attribute__((aligned(16))) double lds_t[];
    tgt.stage(lds_t);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane >> 4;
    const uint64_t cl = ((uint64_t)blockIdx.x * WPB + wave) * 16 + (lane & 15);
    const bool live = cl < prm.C;
    const uint64_t cld = live ? cl : prm.C - 1;
    const uint64_t chain = prm.chain0 + cl;
    const uint32_t d = prm.d;
    const uint64_t C = prm.C;
    const double eps = prm.eps;
    const size_t lane_off = (size_t)j * C + cld;
    double* const ws_wave = prm.wsave + ((size_t)blockIdx.x * WPB + wave) * ((size_t)3 * NS * 64) + lane;
    auto ws_group = [&](int k) -> double* {             // opaque base per group of 8 slices (hmc_dense.hpp: why)
        double* b = ws_wave + (size_t)(k & ~7) * 64;
        asm volatile("" : "+v"(b));
        return b + (k & 7) * 64;
    };
    double th[NS], pm[NS], g[NS];
    double val;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const uint32_t dim = 4 * s + j;
        const double v = prm.theta[(size_t)(dim < d ? dim : 0u) * C + cld];
        th[s] = (dim < d) ? v : 0.0;
    }
    tgt.grad_tile(lds_t, th, g, val, true);
    if (live) {
#pragma unroll
        for (int s = 0; s < NS; ++s) { *ws_group(s) = th[s]; *ws_group(NS + s) = g[s]; }
    }
    double prev_U = -val;                               // hmc.cpp:140
    uint64_t n_acc = 0;
    const uint32_t n_total = prm.n_burnin + prm.n_keep;
    const uint32_t L = prm.n_leap_steps;
    auto kinetic = [&]() __attribute__((always_inline)) -> double {       // p . (I p) / 2 (:160,184)
        return diag_quadratic<NS>(pm, 1.0, j, d) / 2.0;
    };
    auto drift = [&]() __attribute__((always_inline)) {                   // theta += eps * (I p) (:171): theta_i + eps * NaN where poisoned
#pragma unroll
        for (int s = 0; s < NS; ++s) th[s] = th[s] + eps * pm[s];
        dense_product_poison<NS>(pm, th, j, d);
    };
#pragma unroll 1
    for (uint32_t draw = 0; draw < n_total; ++draw) {
#pragma unroll
        for (int b = 0; b < NS / 2; ++b) {               // :156-158
            double z0, z1;
            rng_normal_pair_at(prm.seed, chain, draw + prm.draw0, (uint32_t)(4 * b), (uint32_t)j, STREAM_NORMAL, z0, z1);
            pm[2 * b] = (8u * b + j < d) ? z0 : 0.0;
            pm[2 * b + 1] = (8u * b + 4 + j < d) ? z1 : 0.0;
            __builtin_amdgcn_sched_barrier(0);
        }
        const double prev_K = kinetic();
        if (L > 0) {                                     // first half-step of step 0 (:167, :126: mntm + step * grad / 2)
#pragma unroll
            for (int s = 0; s < NS; ++s) pm[s] = pm[s] + (eps * g[s]) / 2.0;
            drift();
        }
#pragma unroll 1
        for (uint32_t k = 0; k + 1 < L; ++k) {
            double unused;
            tgt.grad_tile(lds_t, th, g, unused, false);
#pragma unroll
            for (int s = 0; s < NS; ++s) {               // second half-step of step k and first of step k + 1: the same gradient
                const double t = (eps * g[s]) / 2.0;
                pm[s] = pm[s] + t;
                pm[s] = pm[s] + t;
            }
            drift();
        }
        if (L > 0) {
            tgt.grad_tile(lds_t, th, g, val, true);
#pragma unroll
            for (int s = 0; s < NS; ++s) pm[s] = pm[s] + (eps * g[s]) / 2.0;
        }
        double prop_U = -val;                            // :178
        if (!is_finite(prop_U)) prop_U = INF;            // :180-182
        const double prop_K = kinetic();                 // :184
        const double x = -(prop_U + prop_K) + (prev_U + prev_K);
        const double comp_val = (x < 0.01) ? x : 0.01;   // :188
        const double z = rng_uniform(prm.seed, chain, draw + prm.draw0, 0u);
        const bool accept = z < det_exp(comp_val);       // :191
        if (accept) {
            prev_U = prop_U;
            if (live) {
#pragma unroll
                for (int s = 0; s < NS; ++s) { *ws_group(s) = th[s]; *ws_group(NS + s) = g[s]; }
            }
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s) { th[s] = *ws_group(s); g[s] = *ws_group(NS + s); }
            val = -prev_U;                               // (n_leap_steps = 0: the value at the unchanged position)
        }
        if (draw >= prm.n_burnin) {
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
        if (j == 0) {
            if (prm.n_accept) prm.n_accept[cl] = n_acc;
            if (prm.n_leap) prm.n_leap[cl] = (uint64_t)n_total * L;
        }
    }
}

// mcmc::hmc with settings.vals_bound and / or a diagonal precond_mat (TileGen): the draw loop of the general built-in variant
// (hmc_dense.hpp, BOUNDED) on the tile functor.  One wave per SIMD (a fourth register-resident vector).
template <class T>
__global__ MI_NO_DS_MERGE __launch_bounds__(256, 1) void hmc_tile_gen_kernel(const TileParams prm, const T tgt)
{
    constexpr int WPB = 4;
    constexpr int NT = T::NT, NS = 4 * NT;
    extern __shared__ __attribute__((aligned(16))) double lds_t[];
    tgt.stage(lds_t);
    TileGen<NT>::stage(lds_t + prm.lds_user_doubles, prm);
    __syncthreads();
    TileGen<NT> gen;
    gen.use(lds_t + prm.lds_user_doubles, prm);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane >> 4;
    const uint64_t cl = ((uint64_t)blockIdx.x * WPB + wave) * 16 + (lane & 15);
    const bool live = cl < prm.C;
    const uint64_t cld = live ? cl : prm.C - 1;
    const uint64_t chain = prm.chain0 + cl;
    const uint32_t d = prm.d;
    const uint64_t C = prm.C;
    const double eps = prm.eps;
    const size_t lane_off = (size_t)j * C + cld;
    double* const ws_wave = prm.wsave + ((size_t)blockIdx.x * WPB + wave) * ((size_t)3 * NS * 64) + lane;
    auto ws_group = [&](int k) -> double* {
        double* b = ws_wave + (size_t)(k & ~7) * 64;
        asm volatile("" : "+v"(b));
        return b + (k & 7) * 64;
    };
    double th[NS], pm[NS], g[NS];
    double val;
    auto eval = [&]() __attribute__((always_inline)) {   // value and gradient at x = inv_transform(theta) (hmc.cpp:108-110)
        double xs[NS];
        gen.x_of(th, xs);
        tgt.grad_tile(lds_t, xs, g, val, true);
    };
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const uint32_t dim = 4 * s + j;
        const double v = prm.theta[(size_t)(dim < d ? dim : 0u) * C + cld];
        th[s] = (dim < d) ? gen.enter(v, dim) : 0.0;
    }
    eval();
    if (live) {
#pragma unroll
        for (int s = 0; s < NS; ++s) { *ws_group(s) = th[s]; *ws_group(NS + s) = g[s]; }
    }
    double prev_U = gen.potential(val, th);              // hmc.cpp:140
    double val_prev = val;
    uint64_t n_acc = 0;
    const uint32_t n_total = prm.n_burnin + prm.n_keep;
    const uint32_t L = prm.n_leap_steps;
    auto drift = [&]() __attribute__((always_inline)) {  // theta += eps * (Minv p) (:171)
        double mp[NS];
        gen.minv_p(pm, mp);
#pragma unroll
        for (int s = 0; s < NS; ++s) th[s] = th[s] + eps * mp[s];
    };
#pragma unroll 1
    for (uint32_t draw = 0; draw < n_total; ++draw) {
#pragma unroll
        for (int b = 0; b < NS / 2; ++b) {               // :156-158: p = L z with a diagonal L
            double z0, z1;
            rng_normal_pair_at(prm.seed, chain, draw + prm.draw0, (uint32_t)(4 * b), (uint32_t)j, STREAM_NORMAL, z0, z1);
            pm[2 * b] = gen.msqrt(8 * b + j) * ((8u * b + j < d) ? z0 : 0.0);
            pm[2 * b + 1] = gen.msqrt(8 * b + 4 + j) * ((8u * b + 4 + j < d) ? z1 : 0.0);
            __builtin_amdgcn_sched_barrier(0);
        }
        const double prev_K = gen.kinetic(pm);
        if (L > 0) {
            double t[NS];
            gen.kick_terms(th, g, eps, t);
#pragma unroll
            for (int s = 0; s < NS; ++s) pm[s] = pm[s] + t[s];
            drift();
        }
#pragma unroll 1
        for (uint32_t k = 0; k + 1 < L; ++k) {
            eval();
            double t[NS];
            gen.kick_terms(th, g, eps, t);
#pragma unroll
            for (int s = 0; s < NS; ++s) { pm[s] = pm[s] + t[s]; pm[s] = pm[s] + t[s]; }
            drift();
        }
        if (L > 0) {
            eval();
            double t[NS];
            gen.kick_terms(th, g, eps, t);
#pragma unroll
            for (int s = 0; s < NS; ++s) pm[s] = pm[s] + t[s];
        }
        double prop_U = gen.potential(val, th);          // :178
        if (!is_finite(prop_U)) prop_U = INF;            // :180-182
        const double prop_K = gen.kinetic(pm);           // :184
        const double x = -(prop_U + prop_K) + (prev_U + prev_K);
        const double comp_val = (x < 0.01) ? x : 0.01;   // :188
        const double z = rng_uniform(prm.seed, chain, draw + prm.draw0, 0u);
        const bool accept = z < det_exp(comp_val);       // :191
        if (accept) {
            prev_U = prop_U; val_prev = val;
            if (live) {
#pragma unroll
                for (int s = 0; s < NS; ++s) { *ws_group(s) = th[s]; *ws_group(NS + s) = g[s]; }
            }
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s) { th[s] = *ws_group(s); g[s] = *ws_group(NS + s); }
            val = val_prev;
        }
        if (draw >= prm.n_burnin) {
            n_acc += accept ? 1u : 0u;
            if (prm.draws != nullptr && live) {
                double* out = prm.draws + (size_t)(draw - prm.n_burnin) * d * C;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const uint32_t dim = 4 * s + j;
                    if (dim < d) (out + (size_t)(4 * s) * C)[lane_off] = gen.leave(th[s], s);     // :211-218
                }
            }
        }
    }
    if (live) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const uint32_t dim = 4 * s + j;
            if (dim < d) prm.theta[(size_t)dim * C + cl] = gen.leave(th[s], s);
        }
        if (j == 0) {
            if (prm.n_accept) prm.n_accept[cl] = n_acc;
            if (prm.n_leap) prm.n_leap[cl] = (uint64_t)n_total * L;
        }
    }
}

// mcmc::mala (ref: src/mala.cpp:149-186, include/mcmc/mala.ipp:59-64, include/stats/dmvnorm.hpp:28-54), no
// bounds: one evaluation per draw (at the proposal), the current gradient cached -- the reference's three gradient calls are
// deterministic repeats (mala_dense.hpp).  Identity preconditioner, or (GEN, round 6) a DIAGONAL precond_mat M (mala.cpp:57-58,123,159; mala.ipp:58-64):
// mu(v) = v + (eps^2 (M g)) / 2, proposal = mu + eps (sqrt(M) z), Sigma = eps^2 M in both dmvnorm terms -- INV(Sigma) = diag(1 / (eps^2 m)) and
// LOG_DET(Sigma) from the host, what the oracle's Gauss-Jordan / Cholesky give for a diagonal matrix --, every dense product of the reference
// applied element-wise with its NaN rule (dense_product_poison), as on the identity route: no replay.  Tables: 3 x 16 NT doubles of LDS behind the target's.
template <class T, bool GEN = false>
__global__ MI_NO_DS_MERGE __launch_bounds__(256, 1) void mala_tile_kernel(const TileParams prm, const T tgt)
{
    constexpr int WPB = 4;                              // one wave per SIMD: four register-resident vectors per chain tile
    constexpr int NT = T::NT, NS = 4 * NT;
    extern __shared__ __attribute__((aligned(16))) double lds_t[];
    tgt.stage(lds_t);
    [[maybe_unused]] double* const tab_m = lds_t + prm.lds_user_doubles;     // GEN: m | sqrt(m) | 1 / (eps^2 m), padded with ones
    if constexpr (GEN) {
        for (int k = threadIdx.x; k < 16 * NT; k += blockDim.x) {
            const bool in = (uint32_t)k < prm.d;
            tab_m[k] = in ? prm.m[k] : 1.0; tab_m[16 * NT + k] = in ? prm.m_sqrt[k] : 1.0; tab_m[32 * NT + k] = in ? prm.s_inv[k] : 1.0;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane >> 4;
    const uint64_t cl = ((uint64_t)blockIdx.x * WPB + wave) * 16 + (lane & 15);
    const bool live = cl < prm.C;
    const uint64_t cld = live ? cl : prm.C - 1;
    const uint64_t chain = prm.chain0 + cl;
    const uint32_t d = prm.d;
    const uint64_t C = prm.C;
    const double eps = prm.eps, s2 = prm.s2, rs = prm.rs;
    const size_t lane_off = (size_t)j * C + cld;
    double th[NS], g[NS], tp[NS], gp[NS];
    double val, valp;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const uint32_t dim = 4 * s + j;
        const double v = prm.theta[(size_t)(dim < d ? dim : 0u) * C + cld];
        th[s] = (dim < d) ? v : 0.0;
    }
    tgt.grad_tile(lds_t, th, g, val, true);
    double prev_LP = val;                               // mala.cpp:138
    uint64_t n_acc = 0;
    const uint32_t n_total = prm.n_burnin + prm.n_keep;
    // this lane's column of a table (entry of slice s at [4 s]), re-derived opaquely where it is used: as loop invariants the entries would be kept in registers
    [[maybe_unused]] auto tcol = [&](int t) __attribute__((always_inline)) -> const double* {
        const double* p = tab_m + 16 * NT * t + j;
        asm volatile("" : "+v"(p));
        return p;
    };
    // mala_mean_fn (:123): v + eps^2 (M grad) / 2, M (the identity, or diagonal) as the dense product it is
    auto mean_of = [&](const double (&v)[NS], const double (&gr)[NS], double (&out)[NS]) __attribute__((always_inline)) {
        if constexpr (GEN) {
            const double* mc = tcol(0);
#pragma unroll
            for (int s = 0; s < NS; ++s) out[s] = v[s] + (s2 * (mc[4 * s] * gr[s])) / 2.0;
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s) out[s] = v[s] + (s2 * gr[s]) / 2.0;
        }
        dense_product_poison<NS>(gr, out, j, d);         // v_i + (eps^2 NaN) / 2 where (M grad)_i is poisoned
    };
    // (x - mu)' INV(eps^2 M) (x - mu) (dmvnorm.hpp:37-39)
    auto quad = [&](const double (&xv)[NS], const double (&mu)[NS]) __attribute__((always_inline)) -> double {
        double xc[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) xc[s] = xv[s] - mu[s];
        if constexpr (GEN) {
            const double* sc = tcol(2);
            double dg[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) dg[s] = sc[4 * s];
            return diag_quadratic_v<NS>(xc, dg, j, d);
        } else return diag_quadratic<NS>(xc, rs, j, d);
    };
#pragma unroll 1
    for (uint32_t draw = 0; draw < n_total; ++draw) {
        double mean_prev[NS];
        mean_of(th, g, mean_prev);
#pragma unroll
        for (int b = 0; b < NS / 2; ++b) {               // proposal = mean + eps * (sqrt(M) z) (:150,159)
            double z0, z1;
            rng_normal_pair_at(prm.seed, chain, draw + prm.draw0, (uint32_t)(4 * b), (uint32_t)j, STREAM_NORMAL, z0, z1);
            double za = (8u * b + j < d) ? z0 : 0.0, zb = (8u * b + 4 + j < d) ? z1 : 0.0;
            if constexpr (GEN) { const double* lc = tcol(1); za = lc[8 * b] * za; zb = lc[8 * b + 4] * zb; }      // (z is finite: no NaN rule to apply)
            tp[2 * b] = mean_prev[2 * b] + eps * za;
            tp[2 * b + 1] = mean_prev[2 * b + 1] + eps * zb;
            __builtin_amdgcn_sched_barrier(0);
        }
        tgt.grad_tile(lds_t, tp, gp, valp, true);
        double prop_LP = valp;                           // :162
        if (!is_finite(prop_LP)) prop_LP = -INF;         // :164-166
        double mean_prop[NS];
        mean_of(tp, gp, mean_prop);
        const double da = prm.cons_term - 0.5 * (prm.log_det + quad(th, mean_prop));      // dmvnorm(prev | mu(prop))
        const double db = prm.cons_term - 0.5 * (prm.log_det + quad(tp, mean_prev));      // dmvnorm(prop | mu(prev))
        const double x = prop_LP - prev_LP + (da - db);
        const double comp_val = (x < 0.01) ? x : 0.01;   // :170
        const double z = rng_uniform(prm.seed, chain, draw + prm.draw0, 0u);
        const bool accept = z < det_exp(comp_val);       // :173
        if (accept) {
#pragma unroll
            for (int s = 0; s < NS; ++s) { th[s] = tp[s]; g[s] = gp[s]; }
            prev_LP = prop_LP;
        }
        if (draw >= prm.n_burnin) {
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
            if (dim < d) prm.theta[(size_t)(4 * s) * C + lane_off] = th[s];
        }
        if (j == 0 && prm.n_accept) prm.n_accept[cl] = n_acc;
    }
}

}  // namespace mi
