// nuts_lds.hpp -- mcmc::nuts on the LDS-streamed evaluation (logistic_lds.hpp): the logistic-regression target with d <= 512 and the
// dense Gaussian with 128 < d <= 512, where one evaluation of the target is a pass of the whole design / precision matrix through
// LDS, shared by the 2 chain tiles x 4 dimension quarters of a workgroup.
//
// Replaces, for C independent chains, mcmc::internal::nuts_impl with nuts_find_initial_step_size and the recursive nuts_build_tree
// (ref: src/nuts.cpp:30-332, include/mcmc/nuts.ipp:30-241; leap_frog_fn src/nuts.cpp:139-154), identity or DIAGONAL precond_mat, with or without
// settings.vals_bound (lds_box.hpp).
//
// The sampler evaluates every doubling on a MEMOISED TRAJECTORY (round 6; the derivation is nuts_memo.hpp's): the reference's second-half
// calls cross their edge outputs (nuts.ipp:195,207) and every doubling restarts from (prev_draw, mntm_vec) (src/nuts.cpp:241-256), so leaf i of a doubling
// is the state LF^{n(i)}(prev_draw, mntm_vec), n(i) = 1 + the sum over the set bits k of i of (k + 1): the 2^j leaves of a doubling visit only
// 1 + j (j + 1) / 2 distinct points of ONE leapfrog trajectory, and the U-turn test of a level-l node whose first leaf sits at point n1 compares the points
// n1 and n1 + l.  Per tick a running chain computes the NEXT POINT of its trajectory (one leapfrog = one evaluation of the target, which here is everything:
// 4 N d flops), evaluates the tests whose second point this is, and then WALKS the leaves the point unblocks exactly as the recursion returns through them
// (nuts.ipp:212-239: the same merges in the same order, one uniform per merge from the same Philox slot, the same early exit) on memoised scalars -- no
// vector work.  Same draws, accepts, depths, leapfrog counts and step sizes as the leaf-per-tick machine of round 4 (and as the recursive oracle and
// literal_kernel<2>), bit for bit; on bench.py's nuts-on-configs[2]'s-target leg 22 % of the leapfrogs the reference counts are distinct: 3 826 -> 1 012 ms.
// What the evaluation forces:
//   * a chain's vectors are split over the FOUR waves of its tile (wave q: dims [q DQ, (q+1) DQ)), every per-chain scalar is replicated
//     in the four waves, and every dot product over dimensions is ((S0 + S1) + S2) + S3 of the waves' 4-strided partial dots through
//     LDS -- the order the hmc / mala kernels of logistic_lds.hpp and the oracle's blocked reductions use (orc_dot_b);
//   * the evaluation is a WORKGROUP collective (its block stream is shared by both tiles), so a tick -- one point for every running
//     chain -- is taken by the 32 chains of a workgroup together and every decision that guards a collective is a workgroup vote: the point's first
//     test shares the pass over the registers and the exchange with the kinetic energy, every further test of the point (one point in six has a second
//     one) and the whole tree's test (:286-289) are a pass and an exchange of their own, announced by flags that travel with the exchange before.
//     The walk has no collective: the four waves of a tile walk the same leaves on their own copies of the scalars.
//     Chains stay asynchronous inside that: each is at its own point of its own doubling of its own draw.
// Registers hold (theta, p, grad) of the chain's last point; the momentum crosses the evaluation through the workspace when the tile is
// wide (NTQ >= 6: the evaluation's accumulators take its registers).  Point records (3 vectors per point, 46 points), the fixed vectors and the per-chain
// scalars -- pending first halves by level (the proposal BY REFERENCE: a point index), bit masks over points for n', s' and the tests, alpha and U of every
// point -- live in global memory (LDS is full of matrix): vectors chain-major inside a wave's block so that a chain's row is contiguous whatever
// record each chain of the wave addresses.  n_leap_out reports the REFERENCE's count (one per leaf walked), n_exec_out the leapfrogs really made.
//
// Chains are handed to lanes DYNAMICALLY.  A workgroup has 32 chain slots (2 tiles x 16 lanes); the grid is persistent (as many workgroups as
// the chip holds at once, or fewer if the chains are few) and a slot whose chain has finished all its draws fetches the next chain index
// from a global counter at the next vote.  A new chain enters the same tick loop in two more states -- INIT (the evaluation at its initial
// values: nuts.cpp:181) and SEARCH (one leapfrog of nuts_find_initial_step_size per tick, nuts.ipp:30-93) -- so a workgroup never waits
// for its slowest chain while there are chains left, and the workspace is sized by the slots, not by the chains.  Results do not depend on
// the slot a chain runs in: its random numbers are counter-based on the global chain index and its arithmetic never crosses lanes of
// other chains.
//
// Non-finite regime: detected through the energies of every leaf; the chain is flagged and replayed by
// literal_kernel<2> (the reference's dense products as written), like the hmc / mala kernels of logistic_lds.hpp do.
#pragma once

#include "logistic_launch.hpp"
#include "lds_box.hpp"

// -DMI_NUTS_LDS_PROF: shader-clock totals per section of the tick (workgroup 0, every wave), printed by the kernel when it ends -- timing
// experiments only (tools/build_variant.sh)
#ifdef MI_NUTS_LDS_PROF
#define MI_LPROF(i) do { const unsigned long long t_ = clock64(); prof_[i] += t_ - tp_; tp_ = t_; } while (0)
#elif defined(MI_NUTS_LDS_SECTIONS)
#define MI_LPROF(i) __builtin_amdgcn_sched_barrier(0)       // (experiment: the sections of the profile as scheduling regions)
#else
#define MI_LPROF(i) do { } while (0)
#endif

namespace mi {

namespace lds_nuts {
// workspace vectors of a chain
enum : int {
    V_PREV = 0, V_WPREV = 1, V_MNTM = 2, V_TPOS_T = 3, V_TPOS_P = 4, V_TNEG_T = 5, V_TNEG_P = 6,
    V_MNTM2 = 7, V_PREVB = 8, V_WPREVB = 9,
    V_TMPP = 10,             // the momentum across an evaluation (wide tiles); INIT: z_init
    V_TMPT = 11,             // BOUNDS: theta across an evaluation (the registers hold x = inv_transform(theta) meanwhile)
    V_XT = 12, V_XW = 13, V_XP = 14,   // DENSEM: theta / gradient / momentum of the last point while a streamed product of the preconditioner has the registers
    V_PT0 = 15,              // point n (1 ..) of the doubling's trajectory: theta at V_PT0 + 3 (n - 1), p at + 1, the gradient at + 2
    MAXPTS = 46,             // 1 + 9 * 10 / 2: the deepest doubling of max_tree_depth = 10 has depth 9
    NVEC = V_PT0 + 3 * MAXPTS,
    MAX_DEPTH = 10,
    // doubles of per-chain scalars: [12 levels][4] pending first halves (n', alpha', n_alpha', proposal point; level 0: the draw's kinetic energy, n, alpha,
    // n_alpha) | the dual-averaging state | 12 bit masks over points | alpha and U of every point
    SC_DA = 48, SC_OKB = 52, SC_ALPHA = 64, SC_U = 112,
    SC_PER_CHAIN = 160
};
enum : int { NS_NEED_DRAW = 0, NS_TREE = 1, NS_DONE = 2, NS_INIT = 3, NS_SEARCH = 4,
             NS_WAIT = 5 };      // pieces (below): the slot holds the ticket of a later piece whose chain has not been published yet
// runs cut into pieces (nuts_lds_body): entries of the piece queues -- not published yet / the chain was flagged before it got here -- and the loads and stores of
// what one slot hands to another THROUGH MEMORY inside a launch: agent scope, so that a slot on another XCD sees them without a fence's write-back of its whole L2
enum : uint32_t { PQ_EMPTY = 0xffffffffu, PQ_GONE = 0xfffffffeu };
template <class T> __device__ __forceinline__ T coh_ld(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ void coh_st(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// doubles of workspace per workgroup (8 waves): vectors, then scalars
__host__ __device__ constexpr size_t vec_doubles_per_wave(int NSQ) { return (size_t)NVEC * NSQ * 64; }
__host__ __device__ constexpr size_t sc_doubles_per_wave() { return (size_t)16 * SC_PER_CHAIN; }

// The memoised trajectory (nuts_memo.hpp, DESIGN.md 4.4e: leaf i of a doubling is the state LF^{n(i)}(prev_draw, mntm_vec), n(i) = 1 + the sum over the set
// bits k of i of (k + 1); the U-turn test of a level-l node whose first leaf sits at point n1 compares the points n1 and n1 + l).
// point of leaf i
__host__ __device__ constexpr uint32_t npt_of(uint32_t i)
{
    uint32_t n = 1;
    for (uint32_t k = 0; k < 10; ++k) if ((i >> k) & 1u) n += k + 1;
    return n;
}
// is there a level-l node in a doubling of depth j whose first leaf sits at point n1?  (n1 - 1 must be a sum of distinct integers of {l + 1 .. j})
__host__ __device__ constexpr bool pair_used(int l, int n1, int j)
{
    const int m = n1 - 1;
    for (int t = 0; t <= j - l; ++t) {
        const int lo = t * (l + 1) + t * (t - 1) / 2, hi = t * j - t * (t - 1) / 2;
        if (m >= lo && m <= hi) return true;
    }
    return false;
}
// bit l of [j][m]: point m of a depth-j doubling closes a level-l test (against point m - l)
struct PmTable { uint16_t v[10][48]; };
constexpr PmTable make_pm_table()
{
    PmTable t{};
    for (int j = 0; j < 10; ++j)
        for (int m = 0; m < 48; ++m) {
            uint32_t bits = 0;
            for (int l = 1; l <= j; ++l)
                if (m - l >= 1 && pair_used(l, m - l, j)) bits |= 1u << l;
            t.v[j][m] = (uint16_t)bits;
        }
    return t;
}
__device__ const PmTable pm_table = make_pm_table();
// npt_of on the device: 1 + popc(i) + sum_b 2^b popc(i & M_b), M_b = the bit positions k with bit b of k set
__device__ __forceinline__ uint32_t npt_of_dev(uint32_t i)
{
    return 1u + (uint32_t)__builtin_popcount(i) + (uint32_t)__builtin_popcount(i & 0x2AAu) + 2u * (uint32_t)__builtin_popcount(i & 0xCCu)
         + 4u * (uint32_t)__builtin_popcount(i & 0xF0u) + 8u * (uint32_t)__builtin_popcount(i & 0x300u);
}
}  // namespace lds_nuts

// DIAGM: a DIAGONAL precond_mat (nuts.cpp:57-59,168,202-204,139-154: p = sqrt(m) z, K = p.(p / m) / 2, theta += e (p / m); the U-turn dots
// are plain), tables read from global memory where they are used (prm.m_sqrt, prm.m_inv: padded with ones to 64 NTQ entries).
// BOUNDS: settings.vals_bound (lds_box.hpp): the tree lives in the transformed space (the U-turn dots are plain), the target is evaluated at
// x = inv_transform(theta), the kicks carry the inverse Jacobian, the potential the log-Jacobian, rows are reported through inv_transform.
// Always together with DIAGM (tables of ones for the identity: 1.0 * p is p).
// DENSEM (round 6): a DENSE precond_mat without bounds (nuts.cpp:57-59: inv_precond_matrix = INV(M), sqrt_precond_matrix = CHOL_LOWER(M); :168,202
// p = L z; leap_frog_fn :139-154 theta += e (Minv p); nuts.ipp:51,66,140 and nuts.cpp:204 K = p . (Minv p) / 2; the U-turn dots are plain).  `dm`:
//     dm.product(img, next, x, acc) -- acc = A x for the matrix with block images img (prm.Lp / prm.Mip), streamed through LDS like the target's
//                                      matrix by the whole workgroup; `next` = the images whose block 0 its last block prefetches
//     dm.reload0(img)               -- block 0 of img into the buffer the next product starts from (the prefetch guessed another matrix)
//     dm.next(img)                  -- what the NEXT evaluation of the target prefetches behind its last block
// Per point: Minv p for the drift, the evaluation, Minv p for the kinetic energy -- three streamed products instead of one, each one fma chain per
// element over the columns in ascending order (the oracle's orc_gemv).  Per draw: L z and Minv (L z).  A product needs the registers of two
// vectors, so on the wide tiles the point's theta / gradient wait in the workspace meanwhile (V_XT, V_XW).
template <int NTQ, bool DIAGM, bool BOUNDS, bool DENSEM = false, class Eval, class DM>
__device__ __forceinline__ void nuts_lds_body(const LogitParams& prm, Eval& evaluate, double* const part_all, [[maybe_unused]] DM& dm)
{
    static_assert(!DENSEM || (!DIAGM && !BOUNDS), "a dense precond_mat: without bounds");
    using namespace lds_nuts;
    constexpr int NS = 4 * NTQ, DQ = 16 * NTQ;
    constexpr bool PM_MEM = NTQ >= 6;                    // the momentum leaves the registers for the evaluation
#ifndef MI_NUTS_LDS_CH_WIDE
#define MI_NUTS_LDS_CH_WIDE 8      // (timing experiments: chunk size of the wide tiles)
#endif
    constexpr int CH = (NTQ >= 6) ? MI_NUTS_LDS_CH_WIDE : ((NS % 8 == 0) ? 8 : 4);   // slices per chunk of a row that passes through temporaries

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int g = wv >> 2, q = wv & 3;
    const int j4 = lane >> 4;
    const uint32_t d = prm.d;
    const uint64_t C = prm.C;
    const uint64_t n_slots = (uint64_t)gridDim.x * 32;   // chains [0, n_slots) start in their own slot; the counter hands out the rest
    double* const part = part_all + g * (4 * 4 * 64);
    constexpr int FLAG_AT = (3 * 4) * 64;                // slot 3 of a tile's exchange area: the tile's flags, then the 16 chain indices of an assignment
    // this slot's chain (replicated in the four waves of the tile); cl >= C: none
    uint64_t cl = ((uint64_t)blockIdx.x * 2 + g) * 16 + (lane & 15);
    bool exhausted = false;                              // the counter has run past the last chain: this slot asks no more
    // Runs cut into PIECES (the launcher: prm.n_pieces > 1 when there are more chains than chain slots; nuts_memo_core.hpp has the reasoning and the protocol):
    // item v < C is piece 0 of chain v, item v >= C the ticket for entry v - C of the piece queues prm.piece_q [n_pieces - 1][C], where chains are published in
    // the order in which their previous piece ended; a later piece continues its chain exactly as a continuation call does.  What crosses between slots inside
    // the launch is written / read at agent scope (lds_nuts::coh_st / coh_ld).  With bounds the hand-over carries theta in the TRANSFORMED space, as the chain holds it
    // (through inv_transform and transform it would be rounded twice): prm.theta holds constrained values again when a chain's last piece has ended.
    uint32_t n_pieces = 1u, piece_len = 0xffffffffu;
    if (prm.n_pieces > 1u) { n_pieces = prm.n_pieces; piece_len = prm.piece_len; }
    const bool pieces = n_pieces > 1u;
    const uint64_t n_items = C * (uint64_t)n_pieces;
    bool piece_done = false;     // this chain's piece ended with the draw it just finished
    bool pub_pend = false;       // ... and it left its slot: wave 0 of the tile publishes it at the next vote, behind the barrier that orders the four waves' stores

    // ---- workgroup collectives.  ((S0 + S1) + S2) + S3 of per-wave partial sums (each already butterflied inside the wave), K <= 3 values
    // at a time, and the OR of a flag word over the workgroup's waves (the four waves of a tile hold the same flags: wave 0 of each tile
    // speaks).  Every wave of the workgroup calls these in the same order.
    auto read_flags = [&]() __attribute__((always_inline)) -> uint32_t {      // (uniform: the block stream's buffer parity is loop-carried behind these votes)
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)part_all[FLAG_AT] | (uint32_t)part_all[(4 * 4 * 64) + FLAG_AT]));
    };
    auto exchange = [&](auto& v, uint32_t flags) __attribute__((always_inline)) -> uint32_t {
        constexpr int K = (int)(sizeof(v) / sizeof(double));
        static_assert(K <= 3, "slot 3 of the exchange area carries the flags");
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; ++k) part[(k * 4 + q) * 64 + lane] = v[k];
        if (q == 0 && lane == 0) part[FLAG_AT] = (double)flags;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; ++k)
            v[k] = ((part[(k * 4 + 0) * 64 + lane] + part[(k * 4 + 1) * 64 + lane]) + part[(k * 4 + 2) * 64 + lane])
                   + part[(k * 4 + 3) * 64 + lane];
        return read_flags();
    };
    auto wg_or = [&](uint32_t flags) __attribute__((always_inline)) -> uint32_t {
        __syncthreads();
        if (q == 0 && lane == 0) part[FLAG_AT] = (double)flags;
        __syncthreads();
        return read_flags();
    };
    auto any = [&](bool p) -> bool { return __ballot(p) != 0ull; };
    auto fold = [&](double a) __attribute__((always_inline)) -> double {     // the 4-strided chains of a block: (q0 + q2) + (q1 + q3)
        a = a + __shfl_xor(a, 32);
        a = a + __shfl_xor(a, 16);
        return a;
    };

    // ---- workspace.  Vectors: [wave][vector] blocks of NS * 512 bytes, inside a vector [chain][pair of slices][j4] in 16-byte granules
    // (nuts_async.hpp: a chain's row is contiguous, so lanes that address different records still fetch whole lines); wave-uniform
    // base + one 32-bit byte offset per access.
    char* const ws_wave_u = reinterpret_cast<char*>(__builtin_assume_aligned(prm.nuts_ws, 256))
                            + ((size_t)blockIdx.x * 8 + wv) * (vec_doubles_per_wave(NS) * sizeof(double));
    uint32_t lane_b = (uint32_t)(lane & 15) * (uint32_t)(NS * 32) + (uint32_t)j4 * 16u;     // redefined (opaquely) at the top of every tick
    auto wsp = [&](int v, int s) -> double* {            // s even: the pair (s, s + 1) of this lane
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
#ifndef MI_LDS_NUTS_NT
#define MI_LDS_NUTS_NT 0         // (timing experiment, as MI_MEMO_NT of nuts_memo_core.hpp) the gradient row (1) / every row (2) of a point's record stored non-temporal: measured, nothing
                                 // on this kernel (profiles/r6_nuts_lds_nt_ab.log: the streamed evaluation is two thirds of its tick)
#endif
    [[maybe_unused]] auto st_row_nt = [&](int v, int s0, const auto& src) __attribute__((always_inline)) {
        constexpr int N = (int)(sizeof(src) / sizeof(double));
        typedef double d2v_ __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int k = 0; k < N; k += 2) { d2v_ t = {src[k], src[k + 1]}; __builtin_nontemporal_store(t, reinterpret_cast<d2v_*>(wsp(v, s0 + k))); }
    };
    auto st_pair = [&](int v, int s0, double a, double b) __attribute__((always_inline)) {
        *reinterpret_cast<double2*>(wsp(v, s0)) = double2{a, b};
    };
    auto cp_row = [&](int src, int dst) __attribute__((always_inline)) {      // a whole row, CH slices at a time
#pragma unroll
        for (int c0 = 0; c0 < NS; c0 += CH) { double t[CH]; ld_row(src, c0, t); st_row(dst, c0, t); }
    };
    // per-chain scalars: chain-major, one private copy per wave (the four waves of a tile compute the same values)
    double* const sc_chain = prm.nuts_sc + ((size_t)blockIdx.x * 8 + wv) * sc_doubles_per_wave() + (size_t)(lane & 15) * SC_PER_CHAIN;
    auto lvl = [&](int l, int f) -> double& { return sc_chain[l * 4 + f]; };
    auto h_val_ = [&]() -> double& { return sc_chain[SC_DA]; };
    auto eps_bar_ = [&]() -> double& { return sc_chain[SC_DA + 1]; };
    auto mu_val_ = [&]() -> double& { return sc_chain[SC_DA + 2]; };
    // bit masks over the points of the doubling in progress: row 0: n' of point n; rows 1..10: the U-turn test of the level-l node whose first leaf sits at
    // point n1 passed; row 11: s' of point n.  alpha (nuts.ipp:157) and U of every point
    auto okb = [&](int r) -> unsigned long long& { return reinterpret_cast<unsigned long long*>(sc_chain)[SC_OKB + r]; };
    auto pt_alpha = [&](uint32_t n) -> double& { return sc_chain[SC_ALPHA + n]; };
    auto pt_U = [&](uint32_t n) -> double& { return sc_chain[SC_U + n]; };

    auto dim_of = [&](int s) -> uint32_t {               // opaque on purpose (logistic_lds.hpp: why)
        uint32_t j = (uint32_t)j4;
        asm volatile("" : "+v"(j));
        return (uint32_t)(q * DQ + 4 * s) + j;
    };

    // DIAGM: entry of slice s of a mass table for this lane (logistic_lds.hpp: mass_at)
    [[maybe_unused]] auto mass_at = [&](const double* tab, int s) __attribute__((always_inline)) -> double {
        uint32_t off = (uint32_t)j4 * 8u;
        asm volatile("" : "+v"(off));
        return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(tab + (q * DQ + 4 * s)) + off);
    };

    // the chain's last leaf on this wave's dims: position, momentum, GRADIENT of the log kernel at the position (MFMA B / D layout)
    double th[NS], pm[NS], w[NS];
    double val = 0.0;
    static_assert(!BOUNDS || DIAGM, "the bounded variant reads the mass tables");
    [[maybe_unused]] LdsBox<NTQ> box;
    if constexpr (BOUNDS) box.init(prm.btype, prm.lb, prm.ub, d, q, lane);
    [[maybe_unused]] double* const lj_rel = part + FLAG_AT + 32;             // the log-Jacobian relay of the tile: 16 doubles
    bool nf = false;                                     // the chain reached the non-finite regime: flagged, replayed by literal.hpp
    auto kick = [&](double e) __attribute__((always_inline)) {               // p += (e grad) / 2 (nuts.cpp:108-135)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if constexpr (BOUNDS) pm[s] = pm[s] + (e * box.jgrad(th[s], w[s], s)) / 2.0;     // (hmc.cpp:122,126 / nuts.cpp:108-135)
            else pm[s] = pm[s] + (e * w[s]) / 2.0;
        }
    };
    auto drift = [&](double e) __attribute__((always_inline)) {              // theta += e (Minv p), Minv = I or diagonal (nuts.cpp:139-154)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if constexpr (DIAGM) th[s] = th[s] + e * (mass_at(prm.m_inv, s) * pm[s]);
            else th[s] = th[s] + e * pm[s];
        }
    };

    const uint32_t n_total = prm.n_burnin + prm.n_keep;
    const uint32_t n_adapt = prm.n_adapt;                // the run's window in GLOBAL draw indices
    const uint32_t max_depth = prm.max_depth;            // 1 .. MAX_DEPTH (the host routes everything else to literal.hpp)
    const double log_half = det_log(0.5), neg_log2 = -det_log(2.0);

    // ---------------------------------------------------------------- per-chain state (replicated in the four waves of the tile)
    int state = (cl < C) ? NS_INIT : NS_DONE;
    uint64_t n_leap = 0, n_exec = 0, n_acc = 0;          // leapfrogs as the reference counts them (one per leaf) / leapfrogs made (one per distinct point)
    double eps = 1.0, prev_U = 0.0;
    uint32_t draw = 0;           // this chain's draw index
    uint32_t jd = 0;             // depth of the doubling in progress
    uint32_t li = 0;             // next leaf of that doubling to be walked
    uint32_t npts = 0;           // points of its trajectory that exist (the registers hold point npts; 0: the origin has to be loaded)
    uint32_t uslot = 0;
    int vdir = 1;                // direction of the doubling; SEARCH: a of nuts.ipp:75
    bool s_first = true;         // SEARCH: the leapfrog of nuts.ipp:62-72 (before the loop)
    double e_signed = 0.0, H0 = 0.0, log_u = 0.0;        // H0: energy at the start of the draw; SEARCH: U0 + K0 of the initial state
    auto prev_K_ = [&]() -> double& { return lvl(0, 0); };
    auto n_val_ = [&]() -> double& { return lvl(0, 1); };
    auto alpha_ = [&]() -> double& { return lvl(0, 2); };
    auto n_alpha_ = [&]() -> double& { return lvl(0, 3); };
    int good_round = 0;
    int mv = V_MNTM, mvn = V_MNTM2;          // momentum vector of the running draw / of the next one
    int pb = 0, pb0 = 0;                     // which of the two vectors holds prev_draw now / held it when the draw started
    bool mom_ready = false;
    double next_K = 0.0, next_lu = 0.0;
    bool row_pend = false, row2_pend = false;
    uint32_t row_draw = 0;
    bool pos_init = true, neg_init = true;
    auto pvec = [](int b) -> int { return b ? V_PREVB : V_PREV; };
    auto wvec = [](int b) -> int { return b ? V_WPREVB : V_WPREV; };

    auto begin_doubling = [&](bool p) __attribute__((always_inline)) {       // direction draw, nuts.cpp:233-235
        const double zdir = rng_uniform(prm.seed, prm.chain0 + cl, draw + prm.draw0, uslot);
        if (p) {
            uslot++;
            vdir = (zdir <= 0.5) ? -1 : 1;
            e_signed = (double)vdir * eps;
            H0 = prev_U + prev_K_();
            li = 0; npts = 0;
        }
    };
    auto end_draw = [&](bool p, uint32_t my_depth) __attribute__((always_inline)) {   // dual averaging nuts.cpp:294-302
        if (p && prm.depth_trace && q == 0 && j4 == 0) prm.depth_trace[(size_t)draw * C + cl] = my_depth;
        if (any(p && draw + prm.draw0 < n_adapt)) {
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
        if (pieces && p && draw < n_total && draw % piece_len == 0u) piece_done = true;
    };
    auto store_row = [&](bool p, int vec, uint32_t idx) __attribute__((always_inline)) {   // kept row `idx` (nuts.cpp:306-309)
        if (!any(p)) return;
        if (p) {
            double* out = prm.draws + (size_t)(idx - prm.n_burnin) * d * C + cl;
#pragma unroll
            for (int c0 = 0; c0 < NS; c0 += CH) {
                double tmp[CH];
                ld_row(vec, c0, tmp);
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const uint32_t dim = dim_of(c0 + k);
                    if constexpr (BOUNDS) tmp[k] = box.leave(tmp[k], c0 + k);      // rows are reported in the constrained space
                    if (dim < d) out[(size_t)dim * C] = tmp[k];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto roll_state = [&](bool p) __attribute__((always_inline)) {           // enter the next draw (nuts.cpp:200-219)
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
    // a chain leaves its slot: final state, counters, step size and dual-averaging state (nuts.cpp:311-330) -- or, flagged, only its flag
    auto retire = [&](bool p) __attribute__((always_inline)) {
        if (!any(p)) return;
        const bool flagged = p && nf && prm.nf_flag != nullptr;
        if (flagged && q == 0 && j4 == 0) { prm.nf_flag[cl] = 1u; prm.nf_flag[C] = 1u; }
        if (p && !flagged) {
#pragma unroll
            for (int c0 = 0; c0 < NS; c0 += CH) {
                double tmp[CH];
                ld_row(pvec(pb), c0, tmp);
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const uint32_t dim = dim_of(c0 + k);
                    if constexpr (BOUNDS) { const double lv = box.leave(tmp[k], c0 + k); tmp[k] = (pieces && piece_done) ? tmp[k] : lv; }     // (a piece's end inside the run: the transformed value)
                    if (dim < d) { if (pieces) lds_nuts::coh_st(prm.theta + ((size_t)dim * C + cl), tmp[k]); else prm.theta[(size_t)dim * C + cl] = tmp[k]; }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (q == 0 && j4 == 0) {
                if (pieces) {            // (the launcher provides every one of these arrays when it cuts the runs into pieces)
                    lds_nuts::coh_st(prm.n_accept + cl, (uint64_t)n_acc); lds_nuts::coh_st(prm.n_leap_out + cl, (uint64_t)n_leap); lds_nuts::coh_st(prm.n_exec_out + cl, (uint64_t)n_exec);
                    lds_nuts::coh_st(prm.step_out + cl, eps);
                    lds_nuts::coh_st(prm.adapt_state + cl, h_val_()); lds_nuts::coh_st(prm.adapt_state + C + cl, eps_bar_()); lds_nuts::coh_st(prm.adapt_state + 2 * C + cl, mu_val_());
                } else {
                if (prm.n_accept) prm.n_accept[cl] = n_acc;
                if (prm.n_leap_out) prm.n_leap_out[cl] = n_leap;
                if (prm.n_exec_out) prm.n_exec_out[cl] = n_exec;
                if (prm.step_out) prm.step_out[cl] = eps;
                if (prm.adapt_state) { prm.adapt_state[cl] = h_val_(); prm.adapt_state[C + cl] = eps_bar_(); prm.adapt_state[2 * C + cl] = mu_val_(); }
                }
            }
        }
        if (pieces) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (p) pub_pend = true; }     // (published at the next vote: piece_done, nf, draw and cl stay as they are until then)
        if (p) state = NS_DONE;
    };
    // SEARCH ends (or is skipped by a continuation): the dual-averaging state of nuts.cpp:174-176, then the chain waits for its first phase
    auto start_sampling = [&](bool p) __attribute__((always_inline)) {
        if (!any(p)) return;
        if (p) {
            mu_val_() = det_log(10 * eps);                   // nuts.cpp:174
            h_val_() = 0.0;
            const uint32_t g0 = prm.draw0 + draw;        // the global index of the chain's next draw (a later piece starts at draw > 0 like a continuation call)
            eps_bar_() = (g0 == 0u) ? prm.eps_bar0 : eps;
            // a continuation inside the adaptation window -- or ANY later piece of a run: behind the window the triple is dead weight for the draws, but it is what
            // the call exports at its end (mi_chains.nuts_adapt_state), and that must not depend on the cut
            if (((g0 > 0u && g0 <= n_adapt) || (pieces && draw != 0u)) && prm.adapt_state != nullptr) {
                h_val_() = lds_nuts::coh_ld(prm.adapt_state + cl); eps_bar_() = lds_nuts::coh_ld(prm.adapt_state + C + cl); mu_val_() = lds_nuts::coh_ld(prm.adapt_state + 2 * C + cl);
            }
            state = NS_NEED_DRAW;
        }
    };

#ifdef MI_NUTS_LDS_PROF
    unsigned long long prof_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tp_ = clock64(), n_ticks_ = 0, n_phase_ = 0, n_top_ = 0, n_lane_ticks_ = 0, n_tests_ = 0, n_walk_ = 0;
#endif
#pragma unroll 1
    for (;;) {
        asm volatile("" : "+v"(lane_b));
        retire(nf && state != NS_DONE);                  // a flagged chain is replayed from its initial state: nothing of it is kept
        // ------------------------------------------------------------ the vote; free slots take the next chains
        {
            const bool want = state == NS_DONE && !exhausted;
            __syncthreads();
            if (q == 0) {
                if (pieces) {
                    // chains that left their slot since the last vote (all four waves' stores are complete: each waited, and the barrier above ordered them).  Not
                    // flagged: the chain finished a piece and is published for piece draw / piece_len (none at the end of the run); flagged (in piece p): it never
                    // continues -- PQ_GONE in EVERY later queue, so that each queue still receives its C entries
                    if (any(pub_pend)) {
                        if (pub_pend && lane < 16) {
                            const bool flagged = nf;
                            const uint32_t p_cur = draw / piece_len - (piece_done ? 1u : 0u);
                            const uint32_t first = flagged ? p_cur + 1u : draw / piece_len;
                            const uint32_t last = flagged ? n_pieces - 1u : ((draw < n_total) ? first : 0u);
                            for (uint32_t qq = (first < 1u ? 1u : first); qq <= last && qq < n_pieces; ++qq) {
                                const uint32_t t = atomicAdd(prm.piece_tail + qq, 1u);
                                lds_nuts::coh_st(prm.piece_q + ((size_t)(qq - 1u) * C + t), flagged ? (uint32_t)lds_nuts::PQ_GONE : (uint32_t)cl);
                            }
                        }
                    }
                    // tickets: one look at the queue entry per vote (never a spin)
                    uint32_t e = lds_nuts::PQ_EMPTY;
                    if (any(state == NS_WAIT)) { if (state == NS_WAIT && lane < 16) e = lds_nuts::coh_ld(prm.piece_q + (size_t)(cl - C)); }
                    if (lane < 16) part[FLAG_AT + 17 + lane] = (double)e;
                }
                const uint32_t m = (uint32_t)(__ballot(want) & 0xffffull);          // the tile's 16 slots (lanes 0..15; the j4 copies agree)
                uint32_t base = 0;
                if (m != 0u) {
                    if (lane == 0) base = atomicAdd(prm.nuts_next, (uint32_t)__builtin_popcount(m));
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                }
                const uint64_t nid = n_slots + base + (uint32_t)__builtin_popcount(m & ((1u << (lane & 15)) - 1u));
                const bool got = want && nid < n_items;
                if (lane < 16) part[FLAG_AT + 1 + lane] = got ? (double)nid : -1.0;
                // (a tile that holds nothing but unserved tickets runs empty ticks meanwhile: an early `continue` with an s_sleep here costs every instantiation
                //  hundreds of bytes of scratch, and such a tile is rare -- tickets are drawn in the order in which chains are published)
                const uint32_t fl = ((any(state != NS_DONE) || any(got)) ? 1u : 0u) | (any(state == NS_NEED_DRAW) ? 2u : 0u);
                if (lane == 0) part[FLAG_AT] = (double)fl;
            }
            __syncthreads();
            pub_pend = false;
            if (pieces && state == NS_WAIT) {            // (the ticket's entry, as wave 0 of the tile saw it)
                const uint32_t e = (uint32_t)part[FLAG_AT + 17 + (lane & 15)];
                if (e != lds_nuts::PQ_EMPTY) {
                    if (e == lds_nuts::PQ_GONE) state = NS_DONE;     // flagged before it got here: the slot takes the next item at the next vote
                    else {                               // the chain goes on where its last piece stopped: counters and draw index here, theta / step size / dual averaging in INIT
                        draw = (uint32_t)(cl / C) * piece_len;
                        cl = e;
                        n_acc = lds_nuts::coh_ld(prm.n_accept + cl); n_leap = lds_nuts::coh_ld(prm.n_leap_out + cl); n_exec = lds_nuts::coh_ld(prm.n_exec_out + cl);
                        state = NS_INIT;
                    }
                }
            }
            if (want) {
                const double nid = part[FLAG_AT + 1 + (lane & 15)];
                if (nid >= 0.0) {                        // a new chain in this slot: everything per-chain starts over
                    cl = (uint64_t)nid;                  // (nid >= C: the ticket of a later piece)
                    state = (cl < C) ? NS_INIT : NS_WAIT; nf = false; n_leap = 0; n_exec = 0; n_acc = 0; draw = 0; eps = 1.0;
                    mv = V_MNTM; mvn = V_MNTM2; pb = 0; pb0 = 0; mom_ready = false; row_pend = false; row2_pend = false; piece_done = false;
                } else exhausted = true;
            }
        }
        const uint32_t f0 = read_flags();
        MI_LPROF(0);
        if ((f0 & 1u) == 0u) break;
        // ------------------------------------------------------------ A. the phase: rows, momenta ahead, waiting chains start or leave
        if ((f0 & 2u) != 0u) {
            store_row(row_pend, pvec(pb0), row_draw);
            store_row(row2_pend, pvec(pb), draw - 1u);
            row_pend = false; row2_pend = false;
            retire(state == NS_NEED_DRAW && (draw >= n_total || piece_done));
            const uint32_t nidx = draw + ((state == NS_TREE) ? 1u : 0u);     // the draw the momentum is for
            const bool gen = (state == NS_TREE || state == NS_NEED_DRAW) && !mom_ready && nidx < n_total;
            double kq = 0.0;
            if constexpr (DENSEM) {
                // p = CHOL_LOWER(M) z (:202) and K = p . (INV(M) p) / 2 (:204): two streamed products; the last point waits in the workspace meanwhile
                st_row(V_XT, 0, th); st_row(V_XP, 0, pm); st_row(V_XW, 0, w);
                double zv[NS];
#pragma unroll
                for (int b = 0; b < NS / 2; ++b) {
                    double z0, z1;
                    rng_normal_pair(prm.seed, prm.chain0 + cl, nidx + prm.draw0, (uint32_t)(q * DQ / 2 + 4 * b + j4), STREAM_NORMAL, z0, z1);
                    zv[2 * b] = (dim_of(2 * b) < d) ? z0 : 0.0;
                    zv[2 * b + 1] = (dim_of(2 * b + 1) < d) ? z1 : 0.0;
                    __builtin_amdgcn_sched_barrier(0);
                }
                double4_t acc[NTQ];
                dm.reload0(prm.Lp);                          // (the last product prefetched INV(M) for a drift: this tick starts with a phase)
                dm.product(prm.Lp, prm.Mip, zv, acc);
#pragma unroll
                for (int s_ = 0; s_ < NS; ++s_) zv[s_] = acc[s_ >> 2][s_ & 3];
                if (gen) st_row(mvn, 0, zv);
                dm.product(prm.Mip, prm.Mip, zv, acc);       // (whatever follows -- INIT's product reloads its own block 0 -- starts from INV(M))
#pragma unroll
                for (int s_ = 0; s_ < NS; ++s_) kq = dfma(zv[s_], acc[s_ >> 2][s_ & 3], kq);
                ld_row(V_XT, 0, th); ld_row(V_XP, 0, pm); ld_row(V_XW, 0, w);
            } else {
#pragma unroll 1
            for (int b = 0; b < NS / 2; ++b) {               // nuts.cpp:200-202, this chain's own draw index
                double z0, z1;
                rng_normal_pair(prm.seed, prm.chain0 + cl, nidx + prm.draw0, (uint32_t)(q * DQ / 2 + 4 * b + j4), STREAM_NORMAL, z0, z1);
                double pa = (dim_of(2 * b) < d) ? z0 : 0.0;
                double pb_ = (dim_of(2 * b + 1) < d) ? z1 : 0.0;
                if constexpr (DIAGM) {                       // :202 and :204 with the diagonal matrices
                    pa = mass_at(prm.m_sqrt, 2 * b) * pa; pb_ = mass_at(prm.m_sqrt, 2 * b + 1) * pb_;
                    kq = dfma(pa, mass_at(prm.m_inv, 2 * b) * pa, kq);
                    kq = dfma(pb_, mass_at(prm.m_inv, 2 * b + 1) * pb_, kq);
                } else {
                    kq = dfma(pa, pa, kq);
                    kq = dfma(pb_, pb_, kq);
                }
                if (gen) st_pair(mvn, 2 * b, pa, pb_);
            }
            }
            double v1[1] = {fold(kq)};
            (void)exchange(v1, 0u);
            const double lu = det_log(rng_uniform(prm.seed, prm.chain0 + cl, nidx + prm.draw0, 0u));
            if (gen) { next_K = v1[0] / 2.0; next_lu = lu; mom_ready = true; }     // :204
            const bool p = state == NS_NEED_DRAW;
            roll_state(p);
            begin_doubling(p);
#ifdef MI_NUTS_LDS_PROF
            n_phase_++;
#endif
            MI_LPROF(1);
            if (wg_or(any(state == NS_TREE || state == NS_INIT || state == NS_SEARCH) ? 1u : 0u) == 0u) continue;
        }
        const bool run = state == NS_TREE, init = state == NS_INIT, srch = state == NS_SEARCH;
#ifdef MI_NUTS_LDS_PROF
        n_ticks_++;
        n_lane_ticks_ += (unsigned long long)__builtin_popcountll(__ballot(run || init || srch) & 0xffffull);
#endif

        // ------------------------------------------------------------ B. the next point of every running chain's trajectory.  The walk below takes every
        // leaf its points allow, so a running chain always needs the point after the one in its registers (an evaluation is the workgroup's: a chain
        // that sat one out would still pay for it)
        {   // the origin of a doubling (prev_draw, mntm_vec, its gradient: src/nuts.cpp:241-256); every later point continues from the registers
            const bool need = run && npts == 0u;
            if (any(need)) {
                if (need) { ld_row(pvec(pb), 0, th); ld_row(mv, 0, pm); ld_row(wvec(pb), 0, w); }
            }
        }
        const uint32_t mpt = npts + 1u;                  // the point this tick computes (run lanes)
        // the tests this point closes: level l against point mpt - l, lowest level first.  The first one shares the pass over the registers and the
        // exchange with the kinetic energy; the others (one in six points has a second one) take a pass and an exchange each, voted by the workgroup
        uint32_t pmask = run ? (uint32_t)pm_table.v[jd < 10u ? jd : 9u][mpt < 48u ? mpt : 0u] : 0u;
        const bool tst = pmask != 0u;
        const bool any_tst = any(tst);
        const uint32_t l_first = tst ? (uint32_t)__builtin_ctz(pmask) : 1u;
        const int eb_t = V_PT0 + 3 * ((int)mpt - (int)l_first - 1), eb_p = eb_t + 1;     // (lanes without a test: never loaded)
        MI_LPROF(2);
        // SEARCH: the step of this leapfrog (nuts.ipp:62, 80-82)
        if (srch) {
            if (!s_first) eps = eps * ((vdir == 1) ? 2.0 : 0.5);
            n_leap++; n_exec++;
        }
        // one leapfrog of size e (nuts.ipp:132 / :64, nuts.cpp:139-154), grad = w.  INIT: e = 0 and the state is set after the (idle) updates
        const double e_tick = run ? e_signed : (srch ? eps : 0.0);
        kick(e_tick);
        if constexpr (DENSEM) {                          // theta += e (INV(M) p) (nuts.cpp:148): the gradient is dead after the kick, the product takes its registers
            constexpr bool PARK_T = NTQ >= 6;
            if constexpr (PARK_T) st_row(V_XT, 0, th);
            double4_t mp[NTQ];
            dm.product(prm.Mip, prm.Xp, pm, mp);
            if constexpr (PARK_T) ld_row(V_XT, 0, th);
#pragma unroll
            for (int s_ = 0; s_ < NS; ++s_) th[s_] = th[s_] + e_tick * mp[s_ >> 2][s_ & 3];
        } else drift(e_tick);
        if (any(init)) {                                 // first_draw and z_init (nuts.cpp:160-168): an evaluation, no leapfrog
            if (init) {
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const uint32_t dim = dim_of(s);
                    const double v = pieces ? lds_nuts::coh_ld(prm.theta + ((size_t)(dim < d ? dim : 0u) * C + cl)) : prm.theta[(size_t)(dim < d ? dim : 0u) * C + cl];
                    if constexpr (BOUNDS) th[s] = (dim < d) ? ((pieces && draw != 0u) ? v : box.enter(v, s)) : 0.0;      // nuts.cpp:160-162 (a later piece: already transformed)
                    else th[s] = (dim < d) ? v : 0.0;
                }
            }
#pragma unroll 1
            for (int b = 0; b < NS / 2; ++b) {
                double z0, z1;
                rng_normal_pair(prm.seed, prm.chain0 + cl, 0u, (uint32_t)(q * DQ / 2 + 4 * b + j4), STREAM_INIT, z0, z1);
                double pa = (dim_of(2 * b) < d) ? z0 : 0.0;
                double pb_ = (dim_of(2 * b + 1) < d) ? z1 : 0.0;
                if constexpr (DIAGM) { pa = mass_at(prm.m_sqrt, 2 * b) * pa; pb_ = mass_at(prm.m_sqrt, 2 * b + 1) * pb_; }   // nuts.cpp:168
                if (init) st_pair(V_TMPP, 2 * b, pa, pb_);
            }
            if (init) ld_row(V_TMPP, 0, pm);
        }
        if constexpr (DENSEM) {
            // z_init is multiplied by CHOL_LOWER(M) too (nuts.cpp:168): a collective, so the workgroup votes on "a chain is in INIT"
            if (wg_or(any(init) ? 1u : 0u) != 0u) {
                st_row(V_XT, 0, th); st_row(V_XW, 0, w);
                double zv[NS];
                ld_row(V_TMPP, 0, zv);                       // (every lane of the wave stored its z_init slice above when some lane was in INIT; others: stale, unused)
                double4_t acc[NTQ];
                dm.reload0(prm.Lp);                          // (the drift's product prefetched the target's block 0: the evaluation comes after this one)
                dm.product(prm.Lp, prm.Xp, zv, acc);
                if (init) {
#pragma unroll
                    for (int s_ = 0; s_ < NS; ++s_) pm[s_] = acc[s_ >> 2][s_ & 3];
                }
                ld_row(V_XT, 0, th); ld_row(V_XW, 0, w);
            }
        }
        MI_LPROF(3);
        if constexpr (PM_MEM) st_row(V_TMPP, 0, pm);
        if constexpr (BOUNDS) { st_row(V_TMPT, 0, th); box.x_inplace(th); }      // the target sees x = inv_transform(theta) (hmc.cpp:108)
        if constexpr (DENSEM) dm.next(prm.Mip);          // the kinetic energy's product follows the evaluation
        evaluate(th, w, val);
        if constexpr (BOUNDS) ld_row(V_TMPT, 0, th);
        if constexpr (PM_MEM) ld_row(V_TMPP, 0, pm);
        MI_LPROF(4);
        // second half-kick, the kinetic energy and the point's first test -- d = theta(mpt) - theta(n1) (by direction), q1 = d . p(n1), q2 = d . p(mpt)
        // (nuts.ipp:224-229) -- in one pass
        double q1 = 0.0, q2 = 0.0, pk = 0.0;
        if constexpr (DENSEM) {
            // K' = p' . (INV(M) p') / 2 (nuts.ipp:140): the second half-kick first, then the product of the new momentum (theta and the new gradient wait in
            // the workspace on the wide tiles), then the dots of the pass below without its kick and kinetic terms
#pragma unroll
            for (int s_ = 0; s_ < NS; ++s_) pm[s_] = pm[s_] + (e_tick * w[s_]) / 2.0;
            constexpr bool PARK_TW = NTQ >= 4;
            if constexpr (PARK_TW) { st_row(V_XT, 0, th); st_row(V_XW, 0, w); }
            double4_t mp[NTQ];
            dm.product(prm.Mip, prm.Mip, pm, mp);            // (the next tick's drift follows; a phase or an INIT in between reloads its own block 0)
#pragma unroll
            for (int s_ = 0; s_ < NS; ++s_) pk = dfma(pm[s_], mp[s_ >> 2][s_ & 3], pk);
            if constexpr (PARK_TW) { ld_row(V_XT, 0, th); ld_row(V_XW, 0, w); }
        }
#pragma unroll
        for (int c0 = 0; c0 < NS; c0 += CH) {
            double tb[CH], pbv[CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) { tb[k] = 0.0; pbv[k] = 0.0; }
            if (any_tst) { if (tst) { ld_row(eb_t, c0, tb); ld_row(eb_p, c0, pbv); } }
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const int s = c0 + k;
                if constexpr (BOUNDS) pm[s] = pm[s] + (e_tick * box.jgrad(th[s], w[s], s)) / 2.0;
                else if constexpr (!DENSEM) pm[s] = pm[s] + (e_tick * w[s]) / 2.0;
                const double dd = (vdir > 0) ? (th[s] - tb[k]) : (tb[k] - th[s]);
                q1 = dfma(dd, pbv[k], q1);
                q2 = dfma(dd, pm[s], q2);
                if constexpr (DIAGM) pk = dfma(pm[s], mass_at(prm.m_inv, s) * pm[s], pk);
                else if constexpr (!DENSEM) pk = dfma(pm[s], pm[s], pk);
            }
        }
        MI_LPROF(5);
        // a doubling can complete in this tick only on its last point, 1 + jd (jd + 1) / 2: the whole tree's test is a collective, voted here
        const bool last_pt = run && (mpt == 1u + jd * (jd + 1u) / 2u);
        double v3[3] = {fold(q1), fold(q2), fold(pk)};
        uint32_t fl = exchange(v3, (any(last_pt) ? 1u : 0u) | (any((pmask & (pmask - 1u)) != 0u) ? 2u : 0u));
        const bool wg_complete = (fl & 1u) != 0u;
        MI_LPROF(6);
        const double pK = v3[2] / 2.0;                   // nuts.ipp:140 / :51,66
        double pU = -val;                                // nuts.ipp:134-138 / :50,65
        if constexpr (BOUNDS) pU = -(val + box.log_jacobian(th, lj_rel, [&]() { __syncthreads(); }));     // -box_log_kernel(theta), nuts.cpp:84-95
        const bool u_nf = !is_finite(pU);
        if (tst) {                                       // the test of level l_first, first leaf at point mpt - l_first
            const unsigned long long bit = 1ull << (mpt - l_first);
            const bool ok = (v3[0] >= 0.0) && (v3[1] >= 0.0);
            okb((int)l_first) = (okb((int)l_first) & ~bit) | (ok ? bit : 0ull);
            pmask &= pmask - 1u;
        }
        // ... and the point's other tests (p(mpt) is final now): the rows of point mpt - l, two dots, an exchange
        while ((fl & 2u) != 0u) {
#ifdef MI_NUTS_LDS_PROF
            n_tests_++;
#endif
            const bool t = pmask != 0u;
            const uint32_t l = t ? (uint32_t)__builtin_ctz(pmask) : 1u;
            const int vt = V_PT0 + 3 * ((int)mpt - (int)l - 1);
            double r1 = 0.0, r2 = 0.0;
            if (any(t)) {
#pragma unroll
                for (int c0 = 0; c0 < NS; c0 += CH) {
                    double tb[CH], pbv[CH];
#pragma unroll
                    for (int k = 0; k < CH; ++k) { tb[k] = 0.0; pbv[k] = 0.0; }
                    if (t) { ld_row(vt, c0, tb); ld_row(vt + 1, c0, pbv); }
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        const double dd = (vdir > 0) ? (th[c0 + k] - tb[k]) : (tb[k] - th[c0 + k]);
                        r1 = dfma(dd, pbv[k], r1);
                        r2 = dfma(dd, pm[c0 + k], r2);
                    }
                }
            }
            double v2[2] = {fold(r1), fold(r2)};
            fl = exchange(v2, any((pmask & (pmask - 1u)) != 0u) ? 2u : 0u);
            if (t) {
                const unsigned long long bit = 1ull << (mpt - l);
                const bool ok = (v2[0] >= 0.0) && (v2[1] >= 0.0);
                okb((int)l) = (okb((int)l) & ~bit) | (ok ? bit : 0ull);
                pmask &= pmask - 1u;
            }
        }
        // ---- INIT: the chain's first state is on record; SEARCH: one step of nuts_find_initial_step_size
        if (any(init)) {
            if (init) {
                st_row(V_PREV, 0, th); st_row(V_WPREV, 0, w);
                prev_U = pU;                             // nuts.cpp:181 (no finiteness guard there)
                if (u_nf || !is_finite(pK)) nf = true;
                H0 = (u_nf ? INF : pU) + pK;             // U0 + K0 (nuts.ipp:50-52)
                s_first = true;
            }
            const bool cont = prm.draw0 != 0 || draw != 0u;      // a continuation call, or a later piece of a run: the same thing
            if (init && cont) eps = prm.step_out ? lds_nuts::coh_ld(prm.step_out + cl) : 1.0;     // the step size comes back in
            start_sampling(init && cont);
            if (init && !cont) state = NS_SEARCH;
        }
        if (any(srch)) {
            if (srch) {
                if (u_nf || !is_finite(pK)) nf = true;
                const double dH = -((u_nf ? INF : pU) + pK) + H0;            // nuts.ipp:68,86
                vdir = 2 * (dH > log_half ? 1 : 0) - 1;                      // :75,88
                s_first = false;
            }
            start_sampling(srch && !(-((u_nf ? INF : pU) + pK) + H0 > neg_log2));     // :78,90: the loop ends
        }
        if (u_nf) { pU = INF; if (run) nf = true; }
        if (run && !is_finite(pK)) nf = true;
        // ---- the point's scalars (nuts.ipp:146-157): n', s' as bits, alpha and U in the chain's table; its record (theta, p, gradient); the tree's far
        //      edge -- point 1 + jd, the first leaf of the second half (the leaf itself at depth 0) -- is what a successful doubling leaves in draw_pos /
        //      draw_neg (src/nuts.cpp:241-256); a doubling that fails before that ends the draw, so it is written in place
        const double dH_pt = -(pU + pK) + H0;
        const double ca_pt = det_exp((dH_pt < 0.0) ? dH_pt : 0.0);      // :157
        if (any(run)) {
            if (run) {
                const unsigned long long bit = 1ull << mpt;
                const bool cn_b = log_u <= -pU - pK;         // :146
                const bool cs_b = log_u < 1000.0 - pU - pK;  // :147
                okb(0) = (okb(0) & ~bit) | (cn_b ? bit : 0ull);
                okb(11) = (okb(11) & ~bit) | (cs_b ? bit : 0ull);
                pt_alpha(mpt) = ca_pt; pt_U(mpt) = pU;
                n_exec++;
                const int vr = V_PT0 + 3 * ((int)mpt - 1);
#if MI_LDS_NUTS_NT == 1
                st_row(vr, 0, th); st_row(vr + 1, 0, pm); st_row_nt(vr + 2, 0, w);
#elif MI_LDS_NUTS_NT == 2
                st_row_nt(vr, 0, th); st_row_nt(vr + 1, 0, pm); st_row_nt(vr + 2, 0, w);
#else
                st_row(vr, 0, th); st_row(vr + 1, 0, pm); st_row(vr + 2, 0, w);
#endif
                npts = mpt;
            }
            const bool st_edge = run && (mpt == 1u + jd);
            if (any(st_edge)) {
                if (st_edge) {
                    const int et = (vdir > 0) ? V_TPOS_T : V_TNEG_T, ep = (vdir > 0) ? V_TPOS_P : V_TNEG_P;
                    st_row(et, 0, th); st_row(ep, 0, pm);
                    if (vdir > 0) pos_init = false; else neg_init = false;
                }
            }
        }
        MI_LPROF(7);
        // ------------------------------------------------------------ C. walk the leaves this point unblocks (nuts.ipp:146-158, 212-239): leaf after leaf as
        // the recursion returns through them -- the same merges in the same order, one uniform per merge from the same Philox slot, the same early exit --
        // on the memoised scalars; no vector work, no collective (the four waves of a tile walk the same leaves on their own copies)
        bool wl = run;                                   // (the leaf a chain waits at sits on its newest point)
        bool at_fin = false, complete = false;
        double cn = 0.0, cna = 0.0, ca = 0.0;
        uint32_t cref = 0;
        uint32_t n = npt_of_dev(li);
#pragma unroll 1
        while (any(wl)) {
#ifdef MI_NUTS_LDS_PROF
            n_walk_++;
#endif
            const uint32_t t1 = (uint32_t)__builtin_ctz(~li);
            const uint32_t nn = n + (t1 + 1u) - t1 * (t1 + 1u) / 2u;         // the point of leaf li + 1
            bool failed = false;
            if (wl) {
                const unsigned long long nbit = 1ull << n;
                cn = (okb(0) & nbit) ? 1.0 : 0.0;
                ca = pt_alpha(n);
                cna = 1.0; cref = n;
                failed = !(okb(11) & nbit);
                n_leap++;
            }
            bool walking = wl;
            uint32_t pend_level = jd + 1;
#pragma unroll 1
            for (uint32_t l = 1; l <= (uint32_t)MAX_DEPTH; ++l) {
                if (walking && l > jd) walking = false;                      // reached the root of its own tree
                const bool bit = ((li >> (l - 1)) & 1u) != 0u;
                if (walking && !failed && !bit) { pend_level = l; walking = false; }   // first half: wait here
                if (!any(walking)) break;
                const bool mrg = walking && bit;
                if (!any(mrg)) continue;
                const double z = rng_uniform(prm.seed, prm.chain0 + cl, draw + prm.draw0, uslot);  // :213
                if (mrg) {
                    uslot++;
                    const double p_n = lvl((int)l, 0), p_a = lvl((int)l, 1), p_na = lvl((int)l, 2);
                    const double prob = cn / (p_n + cn);                     // :212
                    if (!(z < prob)) cref = (uint32_t)lvl((int)l, 3);        // keep new_draw_p (:215-217): a point of the trajectory, by reference
                    cn = p_n + cn;                                           // :220-222
                    ca = p_a + ca;
                    cna = p_na + cna;
                    if (!failed) {                                           // :226-229, evaluated when its second point appeared
                        const uint32_t n1 = n - l * (l + 1u) / 2u;           // the node's first leaf: li with its l low (set) bits cleared
                        if (!((okb((int)l) >> n1) & 1ull)) failed = true;
                    }
                }
            }
            if (wl) {
                const bool keep = !failed;
                complete = keep && (li == (1u << jd) - 1u);
                if (keep && !complete) {                 // a pending first half: its scalars, the proposal by reference
                    lvl((int)pend_level, 0) = cn; lvl((int)pend_level, 1) = ca;
                    lvl((int)pend_level, 2) = cna; lvl((int)pend_level, 3) = (double)cref;
                    li = li + 1u; n = nn;
                    wl = nn <= npts;
                } else {
                    at_fin = true; wl = false;
                }
            }
        }
        MI_LPROF(8);
        // ------------------------------------------------------------ D. end of a doubling: top-level accept first (src/nuts.cpp:260-279).  The proposal is a
        // point of the trajectory: its record (theta, gradient) becomes prev_draw
        bool take = false;
        if (any(complete)) {
            const double z = rng_uniform(prm.seed, prm.chain0 + cl, draw + prm.draw0, uslot);  // :261
            if (complete) {
                uslot++;
                take = z < cn / n_val_();                                   // :263
                if (take) { prev_U = pt_U(cref); good_round = 1; pb = 1 - pb0; }  // :264-277
            }
            if (any(take)) {
                if (take) { const int vq = V_PT0 + 3 * ((int)cref - 1); cp_row(vq, pvec(1 - pb0)); cp_row(vq + 2, wvec(1 - pb0)); }
            }
        }
        MI_LPROF(9);
        // ---- the whole tree's U-turn test (:286-289): a dot product over dimensions, so every wave of the workgroup takes part
        //      whenever some chain of the workgroup was at the last point of its doubling
        bool s_ok = false;
#ifdef MI_NUTS_LDS_PROF
        if (wg_complete) n_top_++;
#endif
        if (wg_complete) {
            double r1 = 0.0, r2 = 0.0;
            if (any(complete)) {
                const int en_t = neg_init ? pvec(pb0) : V_TNEG_T, en_p = neg_init ? mv : V_TNEG_P;
                const int ep_t = pos_init ? pvec(pb0) : V_TPOS_T, ep_p = pos_init ? mv : V_TPOS_P;
                if (complete) {
#pragma unroll
                    for (int c0 = 0; c0 < NS; c0 += CH) {
                        double tn[CH], pn[CH], tp[CH], pp[CH];
                        ld_row(en_t, c0, tn); ld_row(en_p, c0, pn); ld_row(ep_t, c0, tp); ld_row(ep_p, c0, pp);
#pragma unroll
                        for (int k = 0; k < CH; ++k) {
                            const double dd_ = tp[k] - tn[k];
                            r1 = dfma(dd_, pn[k], r1);
                            r2 = dfma(dd_, pp[k], r2);
                        }
                    }
                }
            }
            double v2[2] = {fold(r1), fold(r2)};
            (void)exchange(v2, 0u);
            s_ok = complete && (v2[0] >= 0.0) && (v2[1] >= 0.0);
        }
        MI_LPROF(10);
        if (any(at_fin)) {
            if (at_fin) { alpha_() = ca; n_alpha_() = cna; n_val_() = n_val_() + cn; }   // :246,255 ; :283
            const bool more = at_fin && s_ok && (jd + 1 < max_depth);
            if (at_fin) jd = jd + 1;                                     // :284
            const bool ended = at_fin && !more;
            bool roll = false;
            if (any(ended)) {
                end_draw(ended, jd);
                roll = ended && draw < n_total && mom_ready && !row_pend && !piece_done;
                if (ended && !roll) state = NS_NEED_DRAW;                // the phase: its row, its next momentum, or the end of its run
                roll_state(roll);
            }
            begin_doubling(more || roll);
        }
        MI_LPROF(11);
    }
#ifdef MI_NUTS_LDS_PROF
#if MI_NUTS_LDS_PROF == 2
    if (wv == 0 && lane == 0) {      // every workgroup: its ticks and cycles (which one is the last to finish?)
        unsigned long long tot2 = 0;
        for (int i = 0; i < 12; ++i) tot2 += prof_[i];
        printf("[nuts_lds wg] %u %llu %llu\n", (unsigned)blockIdx.x, n_ticks_, tot2);
    }
#endif
    if (blockIdx.x == 0 && lane == 0) {
        unsigned long long tot = 0;
        for (int i = 0; i < 12; ++i) tot += prof_[i];
        printf("[nuts_lds prof] wave %d: %llu ticks (%.1f of 16 slots busy), %llu phases, %llu tree tests, %llu extra point tests, %llu walk iterations, %.1f k cycles per tick | vote %.1f%% phaseA %.1f%% origin %.1f%% kick+drift %.1f%% "
               "eval %.1f%% kick2+dots %.1f%% exchange+tests %.1f%% point record %.1f%% walk %.1f%% accept %.1f%% tree-test %.1f%% fin %.1f%%\n",
               wv, n_ticks_, (double)n_lane_ticks_ / (double)(n_ticks_ ? n_ticks_ : 1), n_phase_, n_top_, n_tests_, n_walk_, (double)tot / (double)(n_ticks_ ? n_ticks_ : 1) / 1e3,
               100.0 * prof_[0] / tot, 100.0 * prof_[1] / tot, 100.0 * prof_[2] / tot, 100.0 * prof_[3] / tot, 100.0 * prof_[4] / tot, 100.0 * prof_[5] / tot,
               100.0 * prof_[6] / tot, 100.0 * prof_[7] / tot, 100.0 * prof_[8] / tot, 100.0 * prof_[9] / tot, 100.0 * prof_[10] / tot, 100.0 * prof_[11] / tot);
    }
#endif
}

}  // namespace mi
