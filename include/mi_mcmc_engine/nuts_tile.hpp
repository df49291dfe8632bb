// nuts_tile.hpp -- mcmc::nuts for USER-DEFINED targets on the tiled (MFMA-layout) engine: the tick of the built-in Gaussian kernel -- every
// doubling on a MEMOISED trajectory, nuts_memo_core.hpp: each distinct state of a doubling is computed once (one grad_tile call), the tree is
// walked on scalars -- with the gradient behind the tile functor of tile_samplers.hpp instead of the dense mat-vec.  (Rounds 3-4 ran a copy of the
// register-carried tick of nuts_reg.hpp here: one grad_tile call per LEAF.  Same draws, bit for bit; on a target whose trees grow deep ~40 % fewer
// gradient evaluations.)
//
// Replaces, for C independent chains, mcmc::internal::nuts_impl with nuts_find_initial_step_size and the recursive nuts_build_tree
// (ref: src/nuts.cpp:30-332, include/mcmc/nuts.ipp:30-241; the callback contract ref: include/mcmc/nuts.hpp:65-72 is the tile functor:
// one grad_tile call per state gives the gradient both half-kicks around the position use AND the value box_log_kernel returns there --
// the reference's three callbacks per leaf are deterministic repeats), identity precond_mat and no bounds, or (GEN) settings.vals_bound and / or a
// diagonal precond_mat through TileGen.
//
// A point's record is (theta, p, grad).  With the built-in dense Gaussian written as a tile target (grad = -(P theta), value = -theta.P theta / 2)
// every operation has the bits of nuts_gauss_memo_kernel's: p + (e grad) / 2 == p - (e P theta) / 2 exactly.  The identity
// `inv_precond_matrix * mntm` is applied element-wise WITH the NaN rule of the reference's dense product (dense_product_poison,
// diag_quadratic), so the non-finite regime needs no replay on this route (TileMemoPolicy::REPLAY = false: chains are never flagged, and the
// lanes of a tick that do not step are left alone instead of stepping by e = 0).  Every chain has its own slot (no dynamic hand-out: the
// launch is the target library's, one workgroup per 64 chains).
#pragma once

#include "tile_samplers.hpp"
#include "nuts_memo_core.hpp"

namespace mi {

namespace tile_nuts {
enum : int { NUTS_MAX_DEPTH = memo::MEMO_MAX_DEPTH };
// LDS of the sampler behind the target's own (GEN: TileGen's tables follow): the tick's rows and test table
constexpr size_t lds_doubles() { return memo::lds_bytes() / sizeof(double); }
// workspace bytes of a launch of n_chains: one workgroup of four waves per 64 chains
inline size_t ws_bytes(uint64_t n_chains, int nt) { return (size_t)((n_chains + 63) / 64) * 4 * memo::memo_wave_bytes(4 * nt); }
// ... of a persistent grid of n_wg workgroups (TileParams::nuts_grid, ::next_chain): 4 waves x 16 chain slots each
inline size_t ws_bytes_grid(uint64_t n_wg, int nt) { return (size_t)n_wg * 4 * memo::memo_wave_bytes(4 * nt); }

// GEN: settings.vals_bound and / or a diagonal precond_mat (TileGen, tile_samplers.hpp): the tree lives in the transformed space (the
// U-turn dots are plain), rows are reported through inv_transform
template <class T, bool GEN>
struct TileMemoPolicy {
    static constexpr int NT = T::NT, NS = 4 * NT;
    static constexpr bool REPLAY = false;
    static constexpr bool SPLIT = true;          // the engine may cut the runs into pieces (TileParams::n_pieces); with bounds the hand-over carries theta in the transformed space
    static constexpr bool PRE_MOM = false;       // momenta generated inside the tick (the table of nuts_memo.hpp belongs to the built-in kernel's launcher)
#ifndef MI_TILE_LANE_WALK
#define MI_TILE_LANE_WALK 1
#endif
    static constexpr bool LANE_WALK = MI_TILE_LANE_WALK != 0;
    const T& tgt;
    double* lds_t;               // the target's own LDS (its matrices in fragment order)
    const TileGen<T::NT>& tg;    // GEN: bounds / mass tables
    int j4;
    uint32_t d;
    __device__ __forceinline__ double enter(double v, int dim) const { if constexpr (GEN) return tg.enter(v, (uint32_t)dim); else return v; }      // nuts.cpp:160-162
    __device__ __forceinline__ double leave(double v, int s) const { if constexpr (GEN) return tg.leave(v, s); else return v; }     // rows: the constrained space
    __device__ __forceinline__ double msqrt_times(double z, int dim) const { if constexpr (GEN) return tg.msqrt(dim) * z; else return z; }
    __device__ __forceinline__ double minv_times(double p, int dim) const { if constexpr (GEN) return tg.mi[dim] * p; else return 1.0 * p; }
    // p += (e [J^-1] grad) / 2 (nuts.cpp:108-135) on the lanes that step
    __device__ __forceinline__ void kick(const double (&th)[NS], double (&pm)[NS], const double (&w)[NS], double e, bool act) const
    {
        if constexpr (GEN) {
            double t[NS];
            tg.kick_terms(th, w, e, t);
#pragma unroll
            for (int s = 0; s < NS; ++s) pm[s] = act ? pm[s] + t[s] : pm[s];
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s) pm[s] = act ? pm[s] + (e * w[s]) / 2.0 : pm[s];
        }
    }
    // theta += e (Minv p), a dense product in the reference (nuts.cpp:139-154)
    __device__ __forceinline__ void drift(double (&th)[NS], const double (&pm)[NS], double e, bool act) const
    {
        if constexpr (GEN) {
            double mp[NS];
            tg.minv_p(pm, mp);
#pragma unroll
            for (int s = 0; s < NS; ++s) th[s] = act ? th[s] + e * mp[s] : th[s];
        } else {
            double t[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) t[s] = th[s] + e * pm[s];
            dense_product_poison<NS>(pm, t, j4, d);
#pragma unroll
            for (int s = 0; s < NS; ++s) th[s] = act ? t[s] : th[s];
        }
    }
    // value and gradient at the register-resident position (GEN: at x = inv_transform(theta), hmc.cpp:108-110)
    __device__ __forceinline__ void eval(const double (&th)[NS], double (&w)[NS], double& val) const
    {
        if constexpr (GEN) {
            double xs[NS];
            tg.x_of(th, xs);
            tgt.grad_tile(lds_t, xs, w, val, true);
        } else {
            tgt.grad_tile(lds_t, th, w, val, true);
        }
    }
    __device__ __forceinline__ double potential(const double (&th)[NS], const double (&)[NS], double val) const       // -box_log_kernel(theta), nuts.cpp:84-95
    {
        if constexpr (GEN) return tg.potential(val, th);
        else return -val;
    }
    // K = p . (Minv p) / 2 with the matrices as the dense products they are in the reference (nuts.cpp leap_frog_fn / nuts.ipp:51,66,140)
    __device__ __forceinline__ double kinetic(const double (&p)[NS]) const
    {
        if constexpr (GEN) return tg.kinetic(p);
        else return diag_quadratic<NS>(p, 1.0, j4, d) / 2.0;
    }
};

// (the kernel lives in this namespace so that a translation unit of the engine, which sees the built-in kernels' mi:: enums as well, finds these names first)
template <class T, bool GEN = false>
__global__ MI_NO_DS_MERGE __launch_bounds__(256, 1) void nuts_tile_kernel(const TileParams prm, const T tgt)
{
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    double* const lds_t = lds_all;                                   // the target's own LDS (its matrices in fragment order)
    char* const lds_rows = reinterpret_cast<char*>(lds_all + prm.lds_user_doubles);
    uint16_t* const lds_pm = reinterpret_cast<uint16_t*>(lds_rows + memo::R_END * 512);
    [[maybe_unused]] double* const lds_gen = reinterpret_cast<double*>(lds_rows + memo::lds_bytes());        // GEN: the bounds / mass tables
    tgt.stage(lds_t);
    if constexpr (GEN) TileGen<T::NT>::stage(lds_gen, prm);
    __syncthreads();
    TileGen<T::NT> tg;
    if constexpr (GEN) tg.use(lds_gen, prm);
    TileMemoPolicy<T, GEN> pol{tgt, lds_t, tg, (int)((threadIdx.x & 63) >> 4), prm.d};
    memo::nuts_memo_run<T::NT>(prm, pol, lds_rows, lds_pm);
}

}  // namespace tile_nuts
using tile_nuts::nuts_tile_kernel;

}  // namespace mi
