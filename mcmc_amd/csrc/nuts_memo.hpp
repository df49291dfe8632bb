// nuts_memo.hpp -- many-chain NUTS for the plain case (unbounded; identity or DIAGONAL precond_mat; d <= 128, max_tree_depth <= 10) with every
// doubling evaluated on a MEMOISED TRAJECTORY: same draws, accepts, tree depths, leapfrog counts and step sizes as the tick-local kernel of
// nuts_async.hpp (and as rounds 2-4's register-carried kernels nuts_reg.hpp / nuts_dyn.hpp / nuts_split.hpp, which this one replaced), bit for bit,
// with ~40 % fewer leapfrogs executed on BASELINE configs[3].
//
// Replaces what they replace: mcmc::internal::nuts_impl with nuts_find_initial_step_size and nuts_build_tree
// (/root/reference/src/nuts.cpp:30-332, include/mcmc/nuts.ipp:30-241).
//
// Why.  The reference's second-half calls cross their edge outputs (nuts.ipp:195,207) and every doubling restarts from (prev_draw, mntm_vec)
// (src/nuts.cpp:241-256).  nuts_dense.hpp derives from that plumbing that leaf i of a doubling (c = ctz i) starts from the result of leaf
// i - 1 (c <= 1) or of leaf i - 2^(c-1), i.e. the state after leaf i is
//        s(i) = LF^{n(i)}(prev_draw, mntm_vec),        n(i) = 1 + sum over the set bits k of i of (k + 1):
// the 2^j leaves of a doubling lie on ONE trajectory and visit only 1 + j (j + 1) / 2 distinct points of it (46 of 512 at j = 9), and the
// U-turn test of a level-l node whose first leaf sits at point n1 (nuts.ipp:224-229) compares the points n1 and n1 + l.  A leaf's n', s',
// alpha (nuts.ipp:146-157) depend on its point only.  The kernels before this one execute every leaf's leapfrog: on configs[3] (65 536
// chains, 100 adapting + 100 kept draws) 42 % of the executed leapfrogs recompute a state the same doubling has already computed (the
// burn-in grows depth-8..10 trees while the step size adapts; measured with oracle/mcmc_oracle.c: orc_nuts_memo, which is this algorithm on
// the CPU and is checked against the recursion bit for bit in tests/test_oracle_memo.py).
//
// The tick.  Chains are handed to 16 slots per wave dynamically: the grid is persistent (as many workgroups as the chip holds), a slot whose chain
// has finished its draws takes the next chain index from a global counter (one atomic per wave and tick with a free slot), and a new chain enters
// the tick loop in two more states -- INIT (P theta at its initial values, nuts.cpp:181, with z_init as momentum so that the tick's kinetic energy is
// K0) and SEARCH (one leapfrog of nuts_find_initial_step_size per tick); results cannot depend on the slot (counter-based RNG on the global chain id).  Per tick every chain inside a tree
//   1. computes the NEXT POINT of its doubling's trajectory -- one leapfrog from the point in registers (the trajectory is sequential: no
//      start-record loads except the origin at the start of a doubling), P theta on the matrix cores --, its energies and the leaf scalars
//      (n', s' as bits, alpha and U in a per-chain scalar table), and stores the point's record (theta, p, P theta);
//   2. evaluates the U-turn tests whose SECOND point this is: for every level l with a node whose first leaf sits at point m - l (a table
//      in LDS says which), (theta, p)(m - l) from memory against (theta, p)(m) in registers -- the first one is requested before the
//      mat-vec and lands under it; the results are bits in an LDS table [level][point];
//   3. WALKS the leaves that this point unblocks: leaf after leaf exactly as the recursion returns through them (nuts.ipp:212-239: the same
//      merges in the same order, one uniform per merge from the same Philox slot, the same early exit), on the memoised scalars -- no
//      vector work at all -- until the chain needs a point that does not exist yet, or its doubling ends;
//   4. ends the doubling as before (top-level accept src/nuts.cpp:260-279 -- the proposal is a point record --, the test :286-289, dual
//      averaging, next doubling / next draw).
// n_leapfrogs reports the REFERENCE's count (one per leaf walked: what mcmc::nuts executes); NutsParams::n_exec the leapfrogs really made.
//
// Workspace per wave: 10 fixed vectors (prev_draw x 2 with P theta, momentum x 2, the four edge vectors) + 3 vectors per point (46 points at
// max_tree_depth 10) in the record layout of nuts_async.hpp, then the scalar table [47][64 lanes][alpha, U].  Non-finite regime: detected,
// flagged (LDS), the chain leaves its slot at once, and the general variant (nuts_async.hpp) replays it from its initial state.

#pragma once

#include "nuts_async.hpp"
#include "nuts_memo_core.hpp"

namespace mi {

using memo::memo_wave_bytes;

// The built-in Gaussian: gradient = -(P theta) with P's MFMA A-fragments resident in LDS (w holds P theta), U = theta . P theta / 2; identity or
// (DIAGM) a DIAGONAL precond_mat without bounds (nuts.cpp:139-154 with hmc.cpp's leap_frog_fn: p = sqrt(m) z, theta += e (p / m),
// K = p . (p / m) / 2; two tables in LDS, read where they are used).  The identity / diagonal products are applied element-wise, which is NOT the
// reference's dense product once a component is non-finite: REPLAY -- flagged chains are replayed by the general variant (nuts_async.hpp).
template <int NT, bool DIAGM, bool PRE = false>
struct GaussMemoPolicy {
    static constexpr int NS = 4 * NT;
    static constexpr bool REPLAY = true;
    static constexpr bool PRE_MOM = PRE;         // the momenta come from the table nuts_momenta_kernel filled (below)
    static constexpr bool SPLIT = true;          // the launcher may cut the runs into pieces (NutsParams::n_pieces; nuts_launch.hip decides)
    // the lane-parallel walk everywhere but at d = 128 with the two mass tables: that instantiation has no register left for a round's operands
    // (100 B of scratch and 718 ms against 638 ms with the level loop on configs[3] with a diagonal precond_mat, same bits)
    static constexpr bool LANE_WALK = !(DIAGM && NT == 8);
    const double* afrag;         // this lane's column of P's fragments
    const double* lds_ms;        // DIAGM: [16 NT] sqrt(m), [16 NT] 1 / m
    const double* lds_mi;
    // this lane's column of a mass table (entry of slice s at [4 s]), re-derived opaquely where it is used: as loop invariants the compiler
    // would keep the entries in registers this kernel does not have
    __device__ __forceinline__ const double* mcol(const double* tab) const
    {
        const double* p = tab + ((threadIdx.x & 63) >> 4);
        asm volatile("" : "+v"(p));
        return p;
    }
    __device__ __forceinline__ double enter(double v, int) const { return v; }
    __device__ __forceinline__ double leave(double v, int) const { return v; }
    __device__ __forceinline__ double msqrt_times(double z, int dim) const { if constexpr (DIAGM) return lds_ms[dim] * z; else return z; }
    __device__ __forceinline__ double minv_times(double p, int dim) const { if constexpr (DIAGM) return lds_mi[dim] * p; else return p; }
    // (the lanes that do not step pass e = 0: pm - (0 w) / 2 and theta + 0 p are pm and theta bit for bit while everything is finite, and a chain
    //  where it is not is flagged and replayed)
    __device__ __forceinline__ void kick(const double (&)[NS], double (&pm)[NS], const double (&w)[NS], double e, bool) const
    {
#pragma unroll
        for (int s = 0; s < NS; ++s) pm[s] = pm[s] - (e * w[s]) / 2.0;
    }
    __device__ __forceinline__ void drift(double (&th)[NS], const double (&pm)[NS], double e, bool) const
    {
        if constexpr (DIAGM) {
            const double* mic = mcol(lds_mi);
#pragma unroll
            for (int s = 0; s < NS; ++s) th[s] = th[s] + e * (mic[4 * s] * pm[s]);
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s) th[s] = th[s] + e * pm[s];
        }
    }
    __device__ __forceinline__ void eval(const double (&th)[NS], double (&w)[NS], double&) const { matvec_mfma<NT>(afrag, th, w); }
    __device__ __forceinline__ double potential(const double (&th)[NS], const double (&w)[NS], double) const { return 0.5 * dot4<NS>(th, w); }
    __device__ __forceinline__ double kinetic(const double (&p)[NS]) const      // K = p . (Minv p) / 2 (nuts.cpp leap_frog_fn / nuts.ipp:51,66,140)
    {
        if constexpr (DIAGM) {
            const double* mic = mcol(lds_mi);
            double q = 0.0;
#pragma unroll
            for (int s = 0; s < NS; ++s) q = dfma(p[s], mic[4 * s] * p[s], q);
            q = q + __shfl_xor(q, 32);
            q = q + __shfl_xor(q, 16);
            return q / 2.0;
        } else {
            return dot4<NS>(p, p) / 2.0;
        }
    }
};

// LDS: P's fragments | the tick's rows and test table (memo::lds_bytes()) | DIAGM: the two mass tables
// The momenta of a run, ahead of it: mntm_vec = sqrt(M) z of every draw of every chain (src/nuts.cpp:200-202), its kinetic energy
// (:204) and the log of the draw's slice uniform (:206) are pure functions of (seed, chain, draw) -- Philox counters -- so they need not be made inside
// the tick, where a wave that is alone on its SIMD walks 16 Box-Muller pairs per lane one dependent operation at a time (5.6 % of the tick on
// configs[3], and every chain of the wave waits).  One wave per (draw, 16-chain tile), the tick's own lane layout and the tick's own statements -- the
// same operations in the same order, so the same bits -- at full occupancy; the tick then reads 16 NT doubles per chain and draw.
// Table: (16 NT + 2) doubles per chain and draw (configs[3]: 13.6 GB); the launcher falls back to the in-tick generation when it would not fit.
template <int NT, bool DIAGM>
__global__ __launch_bounds__(256) void nuts_momenta_kernel(const NutsParams prm)
{
    constexpr int NS = 4 * NT;
    const int lane = threadIdx.x & 63, j4 = lane >> 4;
    const uint32_t d = prm.d;
    const uint64_t n_tiles = (prm.C + 15) / 16;
    const uint64_t wid = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t n_total = prm.n_burnin + prm.n_keep;
    if (wid >= n_tiles * (uint64_t)n_total) return;
    const uint32_t k = (uint32_t)(wid / n_tiles);                        // LOCAL draw index (the Philox counter takes k + draw0)
    const uint64_t c = (wid % n_tiles) * 16 + (uint64_t)(lane & 15);
    const bool live = c < prm.C;
    const uint64_t cc = live ? c : prm.C - 1;
    char* const base = reinterpret_cast<char*>(prm.mom) + (((size_t)k * prm.C + cc) * (size_t)(NS * 32) + (size_t)j4 * 16u);
    double kq = 0.0;
#pragma unroll 2
    for (int b = 0; b < NS / 2; ++b) {                   // the statements of the tick's phase (nuts_memo_core.hpp), this chain's own draw index
        double z0, z1;
        rng_normal_pair(prm.seed, prm.chain0 + cc, k + prm.draw0, (uint32_t)(4 * b + j4), STREAM_NORMAL, z0, z1);
        const uint32_t da = 8u * b + j4, db = 8u * b + 4u + j4;
        double pa = (da < d) ? z0 : 0.0;
        double pb_ = (db < d) ? z1 : 0.0;
        if constexpr (DIAGM) {
            const double msa = (da < d) ? prm.m_sqrt[da] : 1.0, msb = (db < d) ? prm.m_sqrt[db] : 1.0;
            const double mia = (da < d) ? prm.m_inv[da] : 1.0, mib = (db < d) ? prm.m_inv[db] : 1.0;
            pa = msa * pa; pb_ = msb * pb_;                                  // :202: p = sqrt(M) z
            kq = dfma(pa, mia * pa, kq);                                     // :204: K = p . (Minv p) / 2
            kq = dfma(pb_, mib * pb_, kq);
        } else {
            kq = dfma(pa, pa, kq);
            kq = dfma(pb_, pb_, kq);
        }
        if (live) *reinterpret_cast<double2*>(base + (size_t)b * 64u) = double2{pa, pb_};
    }
    kq = kq + __shfl_xor(kq, 32);
    kq = kq + __shfl_xor(kq, 16);
    const double lu = det_log(rng_uniform(prm.seed, prm.chain0 + cc, k + prm.draw0, 0u));
    if (live && j4 == 0) reinterpret_cast<double2*>(prm.msc)[(size_t)k * prm.C + c] = double2{kq / 2.0, lu};
}
__host__ __device__ constexpr size_t memo_momenta_bytes(uint64_t C, uint32_t n_total, int NS)
{
    return (size_t)n_total * C * ((size_t)NS * 32 + 16);
}

template <int NT, bool DIAGM = false, bool PRE = false>
__global__ MI_NO_DS_MERGE __launch_bounds__(256, 1) void nuts_gauss_memo_kernel(const NutsParams prm)
{
    constexpr int NS = 4 * NT;
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    double* lds_P = lds_all;
    char* const lds_rows = reinterpret_cast<char*>(lds_all + NT * NS * 64);
    uint16_t* const lds_pm = reinterpret_cast<uint16_t*>(lds_rows + memo::R_END * 512);
    double* const lds_ms = reinterpret_cast<double*>(lds_pm + 10 * 48);      // (10 * 48 * 2 = 960 bytes: 8-aligned)
    double* const lds_mi = lds_ms + 16 * NT;
    if (DIAGM) {
        for (int i = threadIdx.x; i < 16 * NT; i += blockDim.x) {
            const bool in = (uint32_t)i < prm.d;
            lds_ms[i] = in ? prm.m_sqrt[i] : 1.0;
            lds_mi[i] = in ? prm.m_inv[i] : 1.0;
        }
    }
    stage_precision<NT>(prm.P, prm.d, lds_P);            // ends with a barrier
    GaussMemoPolicy<NT, DIAGM, PRE> pol{lds_P + (threadIdx.x & 63), lds_ms, lds_mi};
    memo::nuts_memo_run<NT>(prm, pol, lds_rows, lds_pm);
}

}  // namespace mi
