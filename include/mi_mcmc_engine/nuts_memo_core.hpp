// nuts_memo_core.hpp -- the tick of mcmc::nuts on a MEMOISED TRAJECTORY (see mcmc_amd/csrc/nuts_memo.hpp for the derivation and what was
// measured), as ONE device function shared by the built-in Gaussian kernel (nuts_memo.hpp: P theta on the matrix cores) and the user tile targets
// (nuts_tile.hpp: the gradient behind a functor).  What differs between them is a POLICY:
//     static constexpr bool REPLAY     -- the policy's arithmetic leaves the reference's in the non-finite regime: flag such chains for a replay
//     static constexpr bool PRE_MOM    -- the momenta of every draw of every chain (src/nuts.cpp:200-204: p = sqrt(M) z, its kinetic energy, log of the slice
//                                         uniform) come from a TABLE a pre-pass kernel filled at full occupancy (prm.mom, prm.msc: nuts_memo.hpp) instead of
//                                         being generated inside the tick, where 128 Box-Muller normals per chain and draw are pure latency of a wave that is
//                                         alone on its SIMD.  Same Philox counters, same operations, same bits.
//     static constexpr bool SPLIT      -- the launcher may cut every chain's run into prm.n_pieces PIECES of prm.piece_len draws that are handed out as separate work items
//                                         (the built-in kernel's persistent grid: a slot's last chain otherwise ends up to one whole chain after the mean load).  A piece that is
//                                         not the first continues its chain exactly as a continuation CALL does (mi_chains.draw0: step size, dual-averaging state, theta come back
//                                         from memory, the gradient is re-evaluated, no step-size search) -- the hand-over is the one tests/test_gpu_resume.py pins
//     static constexpr bool LANE_WALK  -- the merges of a leaf four levels per round, one accept decision per lane class (MI_MEMO_WALK_V2 below), or one level per
//                                         iteration (the instantiations that have no registers left for the round's operands)
//     double enter(v, dim)             -- initial_vals into the sampler's space (nuts.cpp:160-162); leave(v, slice): the way back for rows / theta
//     double msqrt_times(z, dim)       -- sqrt_precond_matrix * z (nuts.cpp:168, 202);   double minv_times(p, dim): inv_precond_matrix * p, element-wise
//     void kick(th, pm, w, e, act)     -- p += (e [J^-1] grad) / 2 (nuts.cpp:108-135) on the lanes `act`;   drift(th, pm, e, act): theta += e (Minv p) (:139-154)
//     void eval(th, w, val)            -- gradient (or what the kick consumes) and value at theta
//     double potential(th, w, val)     -- -box_log_kernel(theta) (nuts.cpp:84-95);       double kinetic(pm): p . (Minv p) / 2
// Replaces mcmc::internal::nuts_impl with nuts_find_initial_step_size and nuts_build_tree (ref: src/nuts.cpp:30-332, include/mcmc/nuts.ipp:30-241).
#pragma once

#ifndef MI_MEMO_WALK_CAP
#define MI_MEMO_WALK_CAP 8       // at most that many leaves per chain and tick (a chain with leaves left computes no point in the next tick: the other
                                 // 15 chains of its wave do not wait for a 32-leaf walk).  0 = no limit.  Measured on configs[3]: 0: 578 ms, 6: 569, 8: 563, 12: 568
#endif

#ifndef MI_MEMO_RNG_ILP
#define MI_MEMO_RNG_ILP 1        // Box-Muller pairs per iteration of the momentum loop.  Measured on configs[3] (NT = 8): 1: 518 ms; 2: 28 B of scratch; 4: 548 ms (no scratch, but the interleaved pairs push 15 more state registers through the AGPRs around the loop)
#endif

#ifndef MI_MEMO_WALK_V2
#define MI_MEMO_WALK_V2 1        // the merges of a leaf (nuts.ipp:212-229) four levels per round: a leaf with t trailing ones in its index merges at the levels
                                 // 1 .. t, and what those merges need -- the pending halves' n', n_alpha', alpha', proposal and the test bits -- is on record before the
                                 // leaf is walked, so the counts are prefix sums and only the accept decisions differ by level: lane class j4 of a chain takes the
                                 // decision of the level whose uniform it holds (no shuffles), a ballot collects them.  0: one level per iteration of a loop (round 5:
                                 // ~110 instructions per level and ~4 levels per leaf for the slowest chain of a wave; the walk was 19 % of the tick)
#endif
#ifndef MI_MEMO_APF
#define MI_MEMO_APF 0            // alpha of the leaves a walk can reach in this tick (nuts.ipp:157, on record in the scalar table) requested TOGETHER, ahead of the
                                 // walk (1: behind the mat-vec; 2: in front of it), instead of one load per walk iteration: every iteration waited out a
                                 // round trip to memory for 8 bytes (~2 k cycles of a 52 k tick x 4.8 iterations).  0: the per-iteration load.  Needs a walk cap
#endif

#include "hmc_dense.hpp"

namespace mi {
namespace memo {

enum : int { NS_NEED_DRAW = 0, NS_TREE = 1, NS_DONE = 2,
             NS_INIT = 3, NS_SEARCH = 4,            // a new chain in a slot: the gradient at its initial values, then one leapfrog of
                                                    // nuts_find_initial_step_size per tick (nuts.ipp:30-93)
             NS_WAIT = 5 };                         // SPLIT: the slot holds the ticket of a later piece whose chain has not been published yet
// SPLIT: entries of the piece queues
enum : uint32_t { PQ_EMPTY = 0xffffffffu, PQ_GONE = 0xfffffffeu };       // not published yet / the chain was flagged before it got here (nothing to continue)
// loads and stores of what one slot hands to another THROUGH MEMORY inside a launch: agent scope (write-through / L2-coherent reads), so that a slot on
// another XCD sees them without a release fence's write-back of its whole L2
template <class T> __device__ __forceinline__ T coh_ld(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ void coh_st(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// workspace vectors of a chain slot
enum : int {
    MV_PREV = 0, MV_WPREV = 1, MV_MNTM = 2, MV_TPOS_T = 3, MV_TPOS_P = 4, MV_TNEG_T = 5, MV_TNEG_P = 6,
    MV_MNTM2 = 7, MV_PREVB = 8, MV_WPREVB = 9,
    MV_PT0 = 10,                 // point n (1 ..): theta at MV_PT0 + 3 (n - 1), p at + 1, the gradient row at + 2
    MEMO_MAXPTS = 46,            // 1 + 9 * 10 / 2: the deepest doubling of max_tree_depth = 10 has depth 9
    MEMO_NVEC = MV_PT0 + 3 * MEMO_MAXPTS,
    MEMO_MAX_DEPTH = 10
};
// per-chain LDS rows of 64 eight-byte columns (column = chain slot of the workgroup), addressed from ONE opaque per-lane base + an immediate
// offset (as separate arrays the row addresses are loop invariants of the tick loop -- a dozen VGPRs the d = 128 instantiation does not have)
enum : int {
    R_LA = 0,        // [11]: alpha' of the pending first half of level l (row 0 unused)
    R_SC = 11,       // [4]: per-draw scalars: kinetic energy of the draw's momentum, n, alpha, n_alpha
    R_NF = 15,       // non-zero = the chain saw a non-finite energy
    R_DA = 16,       // [3]: the dual-averaging state (h, epsilon_bar, mu)
    R_OKB = 19,      // [12] bit masks over points: row 0: n' of point n; rows 1..10: the U-turn test of the level-l node whose first leaf sits
                     // at point n1 passed; row 11: s' of point n
    R_LP = 31,       // [11]: n' | n_alpha' << 11 | proposal point << 22 of the pending first half of level l
    R_CH = 42,       // [7]: step size, U of prev_draw, H0 and log u of the draw / doubling, the signed step, kinetic energy and log u of the NEXT draw
    R_CT = 49,       // [3] counters: leapfrogs as the reference counts them, leapfrogs executed, accepted kept draws
    R_END = 52
};
// bytes of LDS the tick needs behind the policy's own: the rows, then the test table [10][48] uint16
constexpr size_t lds_bytes() { return (size_t)R_END * 512 + 10 * 48 * 2; }

// bytes of workspace per wave (16 chain slots): the vectors, then the scalar table
__host__ __device__ constexpr size_t memo_wave_bytes(int NS) { return (size_t)MEMO_NVEC * NS * 512 + (size_t)(MEMO_MAXPTS + 1) * 64 * 16; }

// point of leaf i: n(i) = 1 + sum_{k : bit k of i} (k + 1) = 1 + popc(i) + sum_b 2^b popc(i & M_b), M_b = the bit positions k with bit b of k set
__device__ __forceinline__ uint32_t memo_npt(uint32_t i)
{
    return 1u + (uint32_t)__builtin_popcount(i) + (uint32_t)__builtin_popcount(i & 0x2AAu) + 2u * (uint32_t)__builtin_popcount(i & 0xCCu)
         + 4u * (uint32_t)__builtin_popcount(i & 0xF0u) + 8u * (uint32_t)__builtin_popcount(i & 0x300u);
}
// is there a level-l node in a doubling of depth j whose first leaf sits at point n1?  (first leaves of level-l nodes: multiples of 2^l; n1 - 1
// must be a sum of distinct integers of {l + 1 .. j}: the sums of t of these consecutive integers are the integers between the t smallest
// and the t largest)
__host__ __device__ inline bool memo_pair_used(int l, int n1, int j)
{
    const int m = n1 - 1;
    for (int t = 0; t <= j - l; ++t) {
        const int lo = t * (l + 1) + t * (t - 1) / 2, hi = t * j - t * (t - 1) / 2;
        if (m >= lo && m <= hi) return true;
    }
    return false;
}

// The whole run of the workgroup's chain slots.  `prm`: NutsParams or TileParams (the fields are the same); lds_rows / lds_pm: lds_bytes() of LDS
// behind what the policy staged; every thread of the workgroup calls this (a barrier separates the table fill from its use).
template <int NT, class POL, class PRM>
__device__ __forceinline__ void nuts_memo_run(const PRM& prm, POL& pol, char* const lds_rows, uint16_t* const lds_pm)
{
    constexpr int NS = 4 * NT;
    for (int i = threadIdx.x; i < 10 * 48; i += blockDim.x) {       // bit l of [j][m]: point m of a depth-j doubling closes a level-l test
        const int j = i / 48, m = i % 48;
        uint32_t bits = 0;
        for (int l = 1; l <= j; ++l)
            if (m - l >= 1 && memo_pair_used(l, m - l, j)) bits |= 1u << l;
        lds_pm[i] = (uint16_t)bits;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j4 = lane >> 4;
    const int cw = wave * 16 + (lane & 15);
    const uint32_t d = prm.d;
    const uint64_t C = prm.C;
    // 4, 2 or 1 waves per workgroup (the launcher: with few chains every CU gets a workgroup, and a wave the SIMD, the LDS port and the L1 to itself)
    const uint32_t nw = blockDim.x >> 6;
    const uint64_t n_slots = (uint64_t)gridDim.x * (16u * nw);   // chains [0, n_slots) start in their own slot; the counter hands out the rest
    uint64_t cl = ((uint64_t)blockIdx.x * nw + wave) * 16 + (lane & 15);     // this slot's chain (the four lanes of a chain agree); >= C: none
    bool exhausted = prm.next_chain == nullptr;          // no counter: every chain has its own slot (the tile route), nothing is handed out
    // SPLIT: work items are (piece, chain) pairs -- item v < C is piece 0 of chain v; item v >= C is the ticket for entry v - C of the piece queues
    // prm.piece_q [n_pieces - 1][C], where the chains are published in the order in which their previous piece ended
    uint32_t n_pieces = 1u, piece_len = 0xffffffffu;
    if constexpr (POL::SPLIT) { if (prm.n_pieces > 1u && prm.next_chain != nullptr) { n_pieces = prm.n_pieces; piece_len = prm.piece_len; } }
    const uint64_t n_items = C * (uint64_t)n_pieces;
    bool piece_done = false;     // this chain's piece ended with the draw it just finished: it leaves the slot like a finished chain and is published for its next piece

    // (a 32-bit LDS byte address, re-materialised as address-space-3 pointers: through a generic char* the accesses become FLAT instructions,
    //  which queue in order with the wave's global memory operations)
    typedef double __attribute__((address_space(3)))* lds_dptr;
    typedef unsigned long long __attribute__((address_space(3)))* lds_uptr;
    typedef char __attribute__((address_space(3)))* lds_bptr;
    uint32_t lrow = (uint32_t)(uintptr_t)(lds_bptr)lds_rows + (uint32_t)cw * 8u;    // redefined (opaquely) at the top of every tick
#define MI_RD(r) (*(lds_dptr)(uintptr_t)(lrow + (uint32_t)(r) * 512u))
#define MI_RU(r) (*(lds_uptr)(uintptr_t)(lrow + (uint32_t)(r) * 512u))
#define la_(l) MI_RD(R_LA + (l))
#define lp_(l) MI_RU(R_LP + (l))
#define okb_(r) MI_RU(R_OKB + (r))
#define nf_() MI_RD(R_NF)
    // workspace: [wave] blocks of memo_wave_bytes, wave-uniform base + one 32-bit byte offset per access (nuts_async.hpp)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    char* const ws_wave_u = reinterpret_cast<char*>(__builtin_assume_aligned(prm.ws, 256)) + ((size_t)blockIdx.x * nw + wave_u) * memo_wave_bytes(NS);
    // inside a vector: [chain][pair of slices][j4] in 16-byte granules (nuts_async.hpp: why)
    uint32_t lane_b = (uint32_t)(lane & 15) * (uint32_t)(NS * 32) + (uint32_t)j4 * 16u;     // redefined (opaquely) at the top of every tick
    uint32_t lane_sc = (uint32_t)MEMO_NVEC * (uint32_t)(NS * 512) + (uint32_t)lane * 16u;   // this lane's column of the scalar table
    auto wsp = [&](int v, int s) -> double* {                // s even: the pair (s, s + 1) of this lane
        return reinterpret_cast<double*>(ws_wave_u + ((uint32_t)v * (uint32_t)(NS * 512) + lane_b + (uint32_t)(s >> 1) * 64u));
    };
    auto scp = [&](uint32_t n) -> double2* { return reinterpret_cast<double2*>(ws_wave_u + (lane_sc + n * 1024u)); };   // (alpha, U) of point n
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
#ifndef MI_MEMO_NT
#define MI_MEMO_NT 1             // 1: the gradient row of a point's record -- read again only if the point becomes an origin -- is stored NON-TEMPORAL, so that the L2 keeps more of
                                 // the theta / p rows the next ticks' U-turn tests read (65 536 chains: an XCD's 2 048 chains store 4 MB of theta / p per tick, the size of its L2).
                                 // Measured on configs[3], same box, alternating (profiles/r6_nuts_nt_ab.log): 0 (plain stores) 478.7 / 477.1 / 476.6 / 485.1 ms against 1: 474.7 /
                                 // 472.5 / 472.4 / 480.1 (-1.0 %); 2 (all three rows non-temporal): +0.3 %; 3 (1 + the loads of the higher-level tests non-temporal): +3 %; 8 192 chains: no change
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
    // PRE_MOM: the momentum of LOCAL draw k of this lane's chain -- prm.mom [draw][chain] blocks of NS * 32 bytes in the granule order of a workspace row
    // ([pair of slices][j4], so that a lane reads what ld_row would) -- and its scalars prm.msc [draw][chain] = (kinetic energy, log of the slice uniform)
    // (ONE load sequence from a per-lane address -- a workspace row or the table's row -- where a lane-varying choice between the two is made:
    //  as two predicated sequences the choice cost the d = 128 kernel 68 bytes of scratch)
    [[maybe_unused]] auto mom_lane_ptr = [&](uint32_t k) __attribute__((always_inline)) -> const char* {
        if constexpr (POL::PRE_MOM) return reinterpret_cast<const char*>(prm.mom) + (((size_t)k * C + cl) * (size_t)(NS * 32) + (size_t)j4 * 16u);
        else return nullptr;
    };
    [[maybe_unused]] auto ws_lane_ptr = [&](int v) __attribute__((always_inline)) -> const char* {
        return ws_wave_u + ((uint32_t)v * (uint32_t)(NS * 512) + lane_b);
    };
    [[maybe_unused]] auto ld_ptr = [&](const char* lp, auto& dst) __attribute__((always_inline)) {
        constexpr int N = (int)(sizeof(dst) / sizeof(double));
#pragma unroll
        for (int q = 0; q < N; q += 2) {
            const double2 t = *reinterpret_cast<const double2*>(lp + (size_t)(q >> 1) * 64u);
            dst[q] = t.x; dst[q + 1] = t.y;
        }
    };
    [[maybe_unused]] auto ld_ptr_nt = [&](const char* lp, auto& dst) __attribute__((always_inline)) {
        constexpr int N = (int)(sizeof(dst) / sizeof(double));
        typedef double d2v_ __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int q = 0; q < N; q += 2) {
            const d2v_ t = __builtin_nontemporal_load(reinterpret_cast<const d2v_*>(lp + (size_t)(q >> 1) * 64u));
            dst[q] = t.x; dst[q + 1] = t.y;
        }
    };
    auto dim_ok = [&](int s) -> bool { return (uint32_t)(4 * s + j4) < d; };
    constexpr int CHC = (NS < 16) ? NS : 16;          // kept-row stores: slices per chunk
    // the LAST POINT of the chain's trajectory (or the doubling's origin): position, momentum, P * position (MFMA B / D layout).  Loop-carried.
    double th[NS], pm[NS], w[NS];

    // (POL::REPLAY: the policy's arithmetic leaves the reference's in the non-finite regime, so a chain that gets there is flagged, leaves its slot
    //  and is replayed by a kernel that reproduces it; a policy that applies the reference's NaN rules itself needs none of this)
    auto note_nonfinite = [&](bool bad) __attribute__((always_inline)) { if constexpr (POL::REPLAY) { if (__ballot(bad) != 0ull) { if (bad) nf_() = 1.0; } } };
    nf_() = 0.0;
#define eps_() MI_RD(R_CH)
#define prev_U_() MI_RD(R_CH + 1)
#define H0_() MI_RD(R_CH + 2)
#define log_u_() MI_RD(R_CH + 3)
#define esg_() MI_RD(R_CH + 4)
#define next_K_() MI_RD(R_CH + 5)
#define next_lu_() MI_RD(R_CH + 6)
#define n_leap_() MI_RU(R_CT)
#define n_exec_() MI_RU(R_CT + 1)
#define n_acc_() MI_RU(R_CT + 2)
    n_leap_() = 0ull; n_exec_() = 0ull; n_acc_() = 0ull;
    eps_() = 1.0; prev_U_() = 0.0;
    const double log_half = det_log(0.5), neg_log2 = -det_log(2.0);
#define h_val_() MI_RD(R_DA)
#define eps_bar_() MI_RD(R_DA + 1)
#define mu_val_() MI_RD(R_DA + 2)
    const uint32_t n_total = prm.n_burnin + prm.n_keep;
    const uint32_t n_adapt = prm.n_adapt;
    const uint32_t max_depth = prm.max_depth;

    // ---------------------------------------------------------------- per-chain state
    int state = (cl < C) ? NS_INIT : NS_DONE;
    bool s_first = true;         // SEARCH: the leapfrog of nuts.ipp:62-72 (before the loop); vdir carries a of nuts.ipp:75, H0 carries U0 + K0
    uint32_t draw = 0;           // this chain's draw index
    uint32_t jd = 0;             // depth of the doubling in progress
    uint32_t li = 0;             // next leaf of that doubling to be walked
    uint32_t npts = 0;           // points of its trajectory that exist (the registers hold point npts; 0: the origin has to be loaded)
    uint32_t uslot = 0;
    int vdir = 1;
#define prev_K_() MI_RD(R_SC)
#define n_val_() MI_RD(R_SC + 1)
#define alpha_() MI_RD(R_SC + 2)
#define n_alpha_() MI_RD(R_SC + 3)
    int good_round = 0;
    bool org_ok = false;         // the registers already hold the origin of the doubling about to start
    bool cp_pend = false;        // ... and that origin is an accepted proposal that came straight from its point record: prev_draw's copy of it
                                 // (theta, P theta) is stored FROM THE REGISTERS at the top of the next tick -- no load-wait-store in the tick
    double ca_keep = 0.0;        // MI_MEMO_WALK_CAP: alpha of the leaf a walk that was cut short goes on with
    // Draw boundaries without waiting (round 2): what the next draw needs and that does not depend on the chain's state -- momentum
    // (nuts.cpp:200-202), its kinetic energy, the slice uniform -- is generated AHEAD by a phase that serves every chain of the wave that lacks it;
    // prev_draw alternates between two vectors (an accepted proposal goes to the one that did not hold prev_draw when the draw started, so the
    // kept row can be written later from a vector that stays intact), and the edges are the draw's initial vectors until a doubling has written that side
    int mv = MV_MNTM, mvn = MV_MNTM2;
    int pb = 0, pb0 = 0;
    bool mom_ready = false;
    bool row_pend = false, row2_pend = false;
    uint32_t row_draw = 0;
    bool pos_init = true, neg_init = true;
    auto pvec = [](int b) -> int { return b ? MV_PREVB : MV_PREV; };
    auto wvec = [](int b) -> int { return b ? MV_WPREVB : MV_WPREV; };
    // the momentum vector of the running draw (mntm_vec, src/nuts.cpp:200-202) into dst: the table's row of this chain's draw (PRE_MOM), or workspace
    // vector mv (generated inside the tick; INIT's z_init, `ws`, in either mode)
    auto ld_draw_mom = [&](bool ws, auto& dst) __attribute__((always_inline)) {
        if constexpr (POL::PRE_MOM) ld_ptr(ws ? ws_lane_ptr(mv) : mom_lane_ptr(draw), dst);
        else ld_row(mv, 0, dst);
    };
    [[maybe_unused]] bool kl_pend = false;               // PRE_MOM: the scalars of the NEXT draw are on their way from the table (requested when this draw began)
    [[maybe_unused]] double kl_x = 0.0, kl_y = 0.0;

    // The uniforms of a draw are consumed in slot order (direction :233, one per merge nuts.ipp:213, top-level accept :261).  One Philox
    // evaluation per WAVE serves four consecutive slots of every chain: lane class j4 of a chain computes slot ub0 + j4, consumers shuffle.
    // (The walk of the leaves is a chain of merges -- with one evaluation per merge level it was most of a tick.)
    // They travel as the INTEGER they are made from: u = K 2^-53 with K = 2 k + 1 odd, k the 52 random bits (det_math.hpp: u01).  The three
    // places that consume one compare it with a ratio a / b of small integers (u < n'' / (n' + n''), nuts.ipp:212-215; u < n' / n,
    // src/nuts.cpp:263) or with 1 / 2 (:235): u < RN(a / b) is K b < a 2^53 decided in 64-bit integers, except within one grid step of the quotient
    // (|K b - a 2^53| < b), where the rounding of the fp64 division decides and is asked.  Exactly the reference's decisions, without an fp64
    // division per merge in the walk.
    uint32_t ub0 = 0x80000000u;  // first slot in the buffer; valid for slots [ub0, ub0 + 4) of the chain's CURRENT draw
    unsigned long long ubuf = 0ull;
    auto uni = [&](bool need) __attribute__((always_inline)) -> unsigned long long {      // K of slot `uslot` (not advanced here)
        const bool stale = need && (uslot - ub0) >= 4u;
        if (__ballot(stale) != 0ull) {                   // every chain of the wave refills from its own current slot
            ub0 = uslot;
            const u32x4 wv = rng_block(prm.seed, prm.chain0 + cl, draw + prm.draw0, uslot + (uint32_t)j4, STREAM_UNIFORM);
            ubuf = 2ull * ((((unsigned long long)wv.y << 32) | (unsigned long long)wv.x) >> 12) + 1ull;
        }
        const int src = (lane & 15) + 16 * (int)((uslot - ub0) & 3u);
        const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)ubuf, src), hi = (uint32_t)__shfl((int)(uint32_t)(ubuf >> 32), src);
        return ((unsigned long long)hi << 32) | (unsigned long long)lo;
    };
    // u < RN(a / b) for u = K 2^-53 and integers a, b < 2^11 (b = 0: a / b is NaN, the comparison false)
    auto u_below = [&](unsigned long long K, uint32_t a, uint32_t b) __attribute__((always_inline)) -> bool {
        const unsigned long long lhs = K * (unsigned long long)b, rhs = (unsigned long long)a << 53;
        bool r = lhs < rhs;
        const unsigned long long diff = r ? rhs - lhs : lhs - rhs;
        if (diff < (unsigned long long)b) r = ((double)K * 0x1p-53) < ((double)a / (double)b);
        return r;
    };
    // start doubling jd (direction draw, nuts.cpp:233-235) for lanes with `p`
    auto begin_doubling = [&](bool p) __attribute__((always_inline)) {
        const unsigned long long zdir = uni(p);
        if (p) {
            uslot++;
            vdir = (zdir <= (1ull << 52)) ? -1 : 1;       // u <= 0.5
            esg_() = (double)vdir * eps_();
            H0_() = prev_U_() + prev_K_();
            li = 0; npts = 0;
        }
    };
    // end of a draw for lanes with `p` (dual averaging nuts.cpp:294-302; the row store :306-309 is left to the next phase)
    auto end_draw = [&](bool p, uint32_t my_depth) __attribute__((always_inline)) {
        if (p && prm.depth_trace && j4 == 0) prm.depth_trace[(size_t)draw * C + cl] = my_depth;
        if (__ballot(p && draw + prm.draw0 < n_adapt) != 0ull) {
            if (p && draw + prm.draw0 < n_adapt) {
                const double it = (double)(draw + prm.draw0 + 1);
                const double h_new = h_val_() + (1.0 / (it + prm.t0)) * (prm.delta - (alpha_() / n_alpha_()) - h_val_());
                h_val_() = h_new;
                const double e_new = det_exp(mu_val_() - h_new * __builtin_sqrt(it) / prm.gamma);
                eps_() = e_new;
                const double eb = eps_bar_();
                eps_bar_() = eb * det_exp(det_pow(it, -prm.kappa) * (det_log(e_new) - det_log(eb)));
            }
        }
        if (p && !(draw + prm.draw0 < n_adapt)) eps_() = eps_bar_();
        const bool kept = p && draw >= prm.n_burnin;
        if (kept) n_acc_() += (unsigned long long)good_round;
        if (p) {
            row2_pend = kept && prm.draws != nullptr;
            draw++;
        }
        if constexpr (POL::SPLIT) { if (n_pieces > 1u && p && draw < n_total && draw % piece_len == 0u) piece_done = true; }
    };
    // kept row `idx` of lanes with `p` from workspace vector `vec`
    auto store_row = [&](bool p, int vec, uint32_t idx) __attribute__((always_inline)) {
        if (__ballot(p) == 0ull) return;
        if (p) {
            double* out = prm.draws + (size_t)(idx - prm.n_burnin) * d * C;
            const size_t lane_off = (size_t)j4 * C + cl;
#pragma unroll
            for (int c0 = 0; c0 < NS; c0 += CHC) {
                double tmp[CHC];
                ld_row(vec, c0, tmp);
#pragma unroll
                for (int k = 0; k < CHC; ++k)
                    if (dim_ok(c0 + k)) (out + (size_t)(4 * (c0 + k)) * C)[lane_off] = pol.leave(tmp[k], c0 + k);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // lanes with `p` (next momentum ready, no older row pending) enter their next draw (nuts.cpp:200-219)
    auto roll_state = [&](bool p) __attribute__((always_inline)) {
        if (p) {
            if constexpr (!POL::PRE_MOM) { const int t_ = mv; mv = mvn; mvn = t_; }
            const double nk = next_K_();
            prev_K_() = nk;
            log_u_() = next_lu_() - prev_U_() - nk;       // :206
            mom_ready = false;
            if constexpr (POL::PRE_MOM) {                 // the NEXT draw's scalars: requested now, in the LDS rows at the top of the next tick
                if (draw + 1u < n_total) {
                    const double2 v = reinterpret_cast<const double2*>(prm.msc)[(size_t)(draw + 1u) * C + cl];
                    kl_x = v.x; kl_y = v.y; kl_pend = true;
                }
            }
            row_pend = row2_pend; row_draw = draw - 1u; row2_pend = false;
            pb0 = pb; pos_init = true; neg_init = true;
            uslot = 1; ub0 = 0x80000000u;                 // (a new draw: the buffered uniforms are the old draw's)
            jd = 0; n_val_() = 1.0; alpha_() = 0.0; n_alpha_() = 0.0; good_round = 0;
            state = NS_TREE;
        }
    };
    // a chain leaves its slot: final state, counters, step size and dual-averaging state (nuts.cpp:311-330) -- or, flagged, only its flag.
    // SPLIT with pieces: the same record is what the chain's NEXT piece starts from in another slot, so it is written with agent-scope stores, and the
    // chain is published in the next piece's queue once they have completed (vmcnt(0) in front of the publishing store)
    auto retire = [&](bool p) __attribute__((always_inline)) {
        if (__ballot(p) == 0ull) return;
        const bool flagged = p && nf_() != 0.0 && prm.nf_flag != nullptr;
        if (flagged && j4 == 0) { prm.nf_flag[cl] = 1u; prm.nf_flag[C] = 1u; }
        bool pieces = false;
        if constexpr (POL::SPLIT) pieces = n_pieces > 1u;
        if (p && !flagged) {
#pragma unroll 1
            for (int b = 0; b < NS / 2; ++b) {           // (a rolled loop: see INIT)
                const double2 t = *reinterpret_cast<const double2*>(wsp(pvec(pb), 2 * b));
                double* dst = prm.theta + ((size_t)(8u * b + j4) * C + cl);
                if constexpr (POL::SPLIT) {
                    if (pieces) {                        // (a piece's end inside the run: the chain's own -- transformed -- values, which the next piece takes as they are;
                                                         //  through leave() and enter() a bounded dimension would be rounded twice)
                        if (8u * b + j4 < d) coh_st(dst, piece_done ? t.x : pol.leave(t.x, 2 * b));
                        if (8u * b + 4 + j4 < d) coh_st(dst + (size_t)4 * C, piece_done ? t.y : pol.leave(t.y, 2 * b + 1));
                        continue;
                    }
                }
                if (8u * b + j4 < d) dst[0] = pol.leave(t.x, 2 * b);
                if (8u * b + 4 + j4 < d) dst[(size_t)4 * C] = pol.leave(t.y, 2 * b + 1);
            }
            if (j4 == 0) {
                if constexpr (POL::SPLIT) {
                    if (pieces) {        // (the launcher provides every one of these arrays when it cuts the runs into pieces)
                        coh_st(prm.n_accept + cl, (uint64_t)n_acc_()); coh_st(prm.n_leap + cl, (uint64_t)n_leap_()); coh_st(prm.n_exec + cl, (uint64_t)n_exec_());
                        coh_st(prm.step_out + cl, eps_());
                        coh_st(prm.adapt_state + cl, h_val_()); coh_st(prm.adapt_state + C + cl, eps_bar_()); coh_st(prm.adapt_state + 2 * C + cl, mu_val_());
                    }
                }
                if (!pieces) {
                if (prm.n_accept) prm.n_accept[cl] = n_acc_();
                if (prm.n_leap) prm.n_leap[cl] = n_leap_();
                if (prm.n_exec) prm.n_exec[cl] = n_exec_();
                if (prm.step_out) prm.step_out[cl] = eps_();
                if (prm.adapt_state) { prm.adapt_state[cl] = h_val_(); prm.adapt_state[C + cl] = eps_bar_(); prm.adapt_state[2 * C + cl] = mu_val_(); }
                }
            }
        }
        if constexpr (POL::SPLIT) {
            if (pieces) {
                // not flagged: the chain finished a piece (draw is a multiple of piece_len) and is published for piece draw / piece_len -- at the end of the run
                // for none.  Flagged (at any point of piece p): it never continues, so PQ_GONE goes into EVERY later queue -- each queue still receives its C
                // entries and no ticket waits for ever
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (p && j4 == 0) {
                    const uint32_t p_cur = draw / piece_len - (piece_done ? 1u : 0u);
                    const uint32_t first = flagged ? p_cur + 1u : draw / piece_len;
                    const uint32_t last = flagged ? n_pieces - 1u : ((draw < n_total) ? first : 0u);
                    for (uint32_t q = (first < 1u ? 1u : first); q <= last && q < n_pieces; ++q) {
                        const uint32_t t = atomicAdd(prm.piece_tail + q, 1u);
                        coh_st(prm.piece_q + ((size_t)(q - 1u) * C + t), flagged ? (uint32_t)PQ_GONE : (uint32_t)cl);
                    }
                }
            }
        }
        if (p) { state = NS_DONE; piece_done = false; }
    };
    // SEARCH ends (or is skipped by a continuation): the dual-averaging state of nuts.cpp:174-176, then the chain waits for its first phase
    auto start_sampling = [&](bool p) __attribute__((always_inline)) {
        if (__ballot(p) == 0ull) return;
        if (p) {
            mu_val_() = det_log(10 * eps_());                // nuts.cpp:174
            h_val_() = 0.0;
            const uint32_t g0 = prm.draw0 + draw;        // the global index of the chain's next draw (SPLIT: a later piece starts at draw > 0 like a continuation call)
            eps_bar_() = (g0 == 0u) ? prm.eps_bar0 : eps_();
            // a continuation inside the adaptation window -- or (SPLIT) ANY later piece of a run: behind the window the triple is dead weight for the draws, but it is
            // what the call exports at its end (mi_chains.nuts_adapt_state), and that must not depend on the cut
            bool later_piece = false;
            if constexpr (POL::SPLIT) later_piece = draw != 0u;
            if (((g0 > 0u && g0 <= n_adapt) || later_piece) && prm.adapt_state != nullptr) {
                h_val_() = coh_ld(prm.adapt_state + cl); eps_bar_() = coh_ld(prm.adapt_state + C + cl); mu_val_() = coh_ld(prm.adapt_state + 2 * C + cl);
            }
            state = NS_NEED_DRAW;
        }
    };

#ifdef MI_NUTS_REG_PROF   // phase clocks of block 0, wave 0 (tools/nuts_prof.py; a variant build, never the shipped library)
    unsigned long long pc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long n_ticks = 0, n_active = 0, n_walk_it = 0, n_pair_it = 0;
    unsigned long long tmark = clock64();
#define MI_MPROF(k) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long tn_ = clock64(); pc[k] += tn_ - tmark; tmark = tn_; }
#else
#define MI_MPROF(k)
#endif
#pragma unroll 1
    for (;;) {
        asm volatile("" : "+v"(lane_b), "+v"(lane_sc), "+v"(lrow));
        MI_MPROF(7)
        // prev_draw's copy of a proposal accepted in the last tick (requested from its record into the registers as the next doubling's origin):
        // stored before anything of this tick can read prev_draw from memory (the phase's row stores, a later origin load)
        if (__ballot(cp_pend) != 0ull) {
            if (cp_pend) { st_row(pvec(pb), 0, th); st_row(wvec(pb), 0, w); }
            cp_pend = false;
        }
        // the kept row of the draw a chain left in the last tick (src/nuts.cpp:306-309: prev_draw after the draw) IS the origin of its next draw, which
        // the registers hold: stored from there -- no row waits for a phase, so a chain whose next momentum is ready never waits at a draw boundary
        if (__ballot(row_pend) != 0ull) {
            if (row_pend) {
                double* out = prm.draws + (size_t)(row_draw - prm.n_burnin) * d * C;
                const size_t lane_off = (size_t)j4 * C + cl;
#pragma unroll
                for (int k = 0; k < NS; ++k)
                    if (dim_ok(k)) (out + (size_t)(4 * k) * C)[lane_off] = pol.leave(th[k], k);
            }
            row_pend = false;
        }
        if constexpr (POL::PRE_MOM) {
            if (__ballot(kl_pend) != 0ull) {
                if (kl_pend) { next_K_() = kl_x; next_lu_() = kl_y; mom_ready = true; }
                kl_pend = false;
            }
        }
        if constexpr (POL::REPLAY) retire(state != NS_DONE && nf_() != 0.0);   // a flagged chain is replayed from its initial state: nothing of it is kept
        // ------------------------------------------------------------ free slots take the next chains
        {
            const bool want = state == NS_DONE && !exhausted;
            const uint32_t m = (uint32_t)(__ballot(want) & 0xffffull);      // the wave's 16 slots (lanes 0..15; the j4 copies agree)
            if (m != 0u) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(prm.next_chain, (uint32_t)__builtin_popcount(m));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                const uint64_t nid = n_slots + base + (uint32_t)__builtin_popcount(m & ((1u << (lane & 15)) - 1u));
                if (want) {
                    if (nid < n_items) {                 // a new chain in this slot: everything per-chain starts over
                        cl = nid;                        // (nid >= C: the ticket of a later piece -- the chain comes out of the piece queue below)
                        state = (nid < C) ? NS_INIT : NS_WAIT; n_leap_() = 0ull; n_exec_() = 0ull; n_acc_() = 0ull; draw = 0; eps_() = 1.0; nf_() = 0.0;
                        mv = MV_MNTM; mvn = MV_MNTM2; pb = 0; pb0 = 0; mom_ready = false; row_pend = false; row2_pend = false; org_ok = false; cp_pend = false; kl_pend = false;
                        piece_done = false;
                    } else exhausted = true;
                }
            }
        }
        if constexpr (POL::SPLIT) {
            // tickets of later pieces: one look at the queue entry per tick (never a spin: the chain it waits for may run in this very wave).  The chain goes on
            // where its last piece stopped: counters and draw index here, theta / step size / dual-averaging state in INIT, as a continuation call would
            if (__ballot(state == NS_WAIT) != 0ull) {
                uint32_t e = PQ_EMPTY;
                if (state == NS_WAIT) e = coh_ld(prm.piece_q + (size_t)(cl - C));
                if (__ballot(state != NS_WAIT && state != NS_DONE) == 0ull && __ballot(state == NS_WAIT && e != PQ_EMPTY) == 0ull) __builtin_amdgcn_s_sleep(32);   // (only tickets left in this wave, none served: look again in ~2 k cycles)
                if (state == NS_WAIT && e != PQ_EMPTY) {
                    if (e == PQ_GONE) state = NS_DONE;   // flagged before it got here: the slot takes the next item
                    else {
                        draw = (uint32_t)(cl / C) * piece_len;
                        cl = e;
                        n_acc_() = coh_ld(prm.n_accept + cl); n_leap_() = coh_ld(prm.n_leap + cl); n_exec_() = coh_ld(prm.n_exec + cl);
                        state = NS_INIT;
                    }
                }
            }
        }
        if (__ballot(state != NS_DONE) == 0ull) break;
        // ------------------------------------------------------------ A. the phase: rows, momenta ahead, waiting chains start
        if (__ballot(state == NS_NEED_DRAW) != 0ull) {
            store_row(row2_pend, pvec(pb), draw - 1u);       // (a chain that did not go straight on: its last draw's row, from prev_draw in memory)
            row2_pend = false;
            retire(state == NS_NEED_DRAW && (draw >= n_total || piece_done));
            const uint32_t nidx = draw + ((state == NS_TREE) ? 1u : 0u);     // the draw the momentum is for
            if constexpr (POL::PRE_MOM) {
                // (a chain's first draw, or one whose look-ahead found no draw to look at: the scalars straight from the table)
                const bool gen = (state == NS_TREE || state == NS_NEED_DRAW) && !mom_ready && !kl_pend && nidx < n_total;
                if (gen) {
                    const double2 v = reinterpret_cast<const double2*>(prm.msc)[(size_t)nidx * C + cl];
                    next_K_() = v.x; next_lu_() = v.y; mom_ready = true;
                }
            } else {
            const bool gen = (state == NS_TREE || state == NS_NEED_DRAW) && !mom_ready && nidx < n_total;
            double kq = 0.0;
            // (MI_MEMO_RNG_ILP Box-Muller pairs per iteration: a pair is ~250 dependent operations -- Philox rounds, log, sqrt, sincos -- and a wave that is
            //  alone on its SIMD waits out every result latency; independent pairs in one basic block interleave)
            constexpr int RILP = (MI_MEMO_RNG_ILP < NS / 2) ? MI_MEMO_RNG_ILP : NS / 2;
#pragma unroll 1
            for (int b0 = 0; b0 < NS / 2; b0 += RILP) {               // nuts.cpp:200-202, this chain's own draw index
                double zz[2 * RILP];
#pragma unroll
                for (int q = 0; q < RILP; ++q)
                    rng_normal_pair(prm.seed, prm.chain0 + cl, nidx + prm.draw0, (uint32_t)(4 * (b0 + q) + j4), STREAM_NORMAL, zz[2 * q], zz[2 * q + 1]);
#pragma unroll
                for (int q = 0; q < RILP; ++q) {
                    const int b = b0 + q;
                    double pa = (8u * b + j4 < d) ? zz[2 * q] : 0.0;
                    double pb_ = (8u * b + 4 + j4 < d) ? zz[2 * q + 1] : 0.0;
                    pa = pol.msqrt_times(pa, 8 * b + j4); pb_ = pol.msqrt_times(pb_, 8 * b + 4 + j4);       // :202: p = sqrt(M) z
                    kq = dfma(pa, pol.minv_times(pa, 8 * b + j4), kq);                                       // :204: K = p . (Minv p) / 2
                    kq = dfma(pb_, pol.minv_times(pb_, 8 * b + 4 + j4), kq);
                    if (gen) st_pair(mvn, 2 * b, pa, pb_);
                }
            }
            kq = kq + __shfl_xor(kq, 32);
            kq = kq + __shfl_xor(kq, 16);
            const double lu = det_log(rng_uniform(prm.seed, prm.chain0 + cl, nidx + prm.draw0, 0u));
            if (gen) { next_K_() = kq / 2.0; next_lu_() = lu; mom_ready = true; }     // :204
            }
            const bool p = state == NS_NEED_DRAW;             // (all of them have a momentum now and no row pending)
            roll_state(p);
            if (max_depth > 0) begin_doubling(p);
            else { end_draw(p, 0u); if (p) state = NS_NEED_DRAW; }              // while-loop of :227 never entered
        }
        const bool run = state == NS_TREE, init = state == NS_INIT, srch = state == NS_SEARCH;
        MI_MPROF(0)
        if (__ballot(run || init || srch) == 0ull) continue;
#ifdef MI_NUTS_REG_PROF
        n_ticks++; n_active += (unsigned long long)__builtin_popcountll(__ballot(run)) / 4ull;
        pc[10] += (unsigned long long)__builtin_popcountll(__ballot(run && memo_npt(li) > npts)) / 4ull;     // (a count, not cycles: chains that compute a point in this tick)
#endif

        // INIT: first_draw and z_init (nuts.cpp:160-168) are staged in the workspace -- theta in MV_PREV, the momentum in mv -- and enter the
        // registers through the origin load below; the tick is P theta with e = 0 (ROLLED loops, and HERE, where no test operands are live: unrolled they cost the d = 128 kernel 800 bytes of scratch)
        if (__ballot(init) != 0ull) {
#pragma unroll 1
            for (int b = 0; b < NS / 2; ++b) {
                const bool in0 = 8u * b + j4 < d, in1 = 8u * b + 4 + j4 < d;
                const double* src = prm.theta + ((size_t)(in0 ? 8u * b + j4 : 0u) * C + cl);
                double t0 = 0.0, t1 = 0.0;
                if (init) {
                    if constexpr (POL::SPLIT) { t0 = coh_ld(src); t1 = coh_ld(src + (in1 ? (size_t)4 * C : 0)); }     // (a later piece: what another slot stored in this launch)
                    else { t0 = src[0]; t1 = src[in1 ? (size_t)4 * C : 0]; }
                }
                bool raw = false;                        // (SPLIT, a later piece: the values are the chain's own already)
                if constexpr (POL::SPLIT) raw = draw != 0u;
                if (init) st_pair(MV_PREV, 2 * b, in0 ? (raw ? t0 : pol.enter(t0, 8 * b + j4)) : 0.0, in1 ? (raw ? t1 : pol.enter(t1, 8 * b + 4 + j4)) : 0.0);   // nuts.cpp:160-162
            }
            // z_init (nuts.cpp:166-168) feeds K0 of nuts_find_initial_step_size only: a continuation -- a call with draw0 > 0 or (SPLIT) a later piece -- goes
            // straight to its first draw, so its 16 NT Box-Muller normals (a third of a tick of the whole wave) are not made; zeros keep the idle update finite
            const bool any_fresh = __ballot(init && prm.draw0 == 0 && draw == 0u) != 0ull;
#pragma unroll 1
            for (int b = 0; b < NS / 2; ++b) {
                double pa = 0.0, pb_ = 0.0;
                if (any_fresh) {
                    double z0, z1;
                    rng_normal_pair(prm.seed, prm.chain0 + cl, 0u, (uint32_t)(4 * b + j4), STREAM_INIT, z0, z1);
                    pa = (8u * b + j4 < d) ? z0 : 0.0;
                    pb_ = (8u * b + 4 + j4 < d) ? z1 : 0.0;
                    pa = pol.msqrt_times(pa, 8 * b + j4); pb_ = pol.msqrt_times(pb_, 8 * b + 4 + j4);       // nuts.cpp:168
                }
                if (init) st_pair(mv, 2 * b, pa, pb_);
            }
        }
        // ------------------------------------------------------------ B. the next point of every running chain's trajectory
        // (a chain whose walk was cut short by MI_MEMO_WALK_CAP still has leaves on its existing points: it computes no point this tick -- e = 0
        //  leaves its registers as they are, bit for bit -- and goes on walking)
        const bool newpt = run && memo_npt(li) > npts;
        {   // the origin of a doubling (prev_draw, mntm_vec, P prev_draw: src/nuts.cpp:241-256); every later point continues from the registers
            const bool need = (newpt && npts == 0u && !org_ok) || init;      // (!org_ok: not requested at the end of the last tick already)
            if (__ballot(need) != 0ull) {
                const int vt = pvec(pb), vw = init ? pvec(pb) : wvec(pb);      // INIT: first_draw (pb = 0), any finite row as P theta (e = 0)
                if (need) { ld_row(vt, 0, th); ld_draw_mom(init, pm); ld_row(vw, 0, w); }
            }
        }
        if (newpt) org_ok = false;
        const uint32_t mpt = npts + 1u;                  // the point this tick computes (newpt lanes)
        // the tests this point closes: level l against point mpt - l (lds_pm), lowest level first
        // ... and, as "level 0", the TOP-LEVEL test of src/nuts.cpp:286-289 when this point is the doubling's far edge (point 1 + jd): its operands
        // are this point and the OTHER side's edge, which no tick of this doubling writes -- evaluated here, with the point in registers, it costs
        // two row loads that mostly travel under the mat-vec instead of four behind the walk (result: bit 63 of okb_(0), read at the doubling's end)
        uint32_t pmask = newpt ? ((uint32_t)lds_pm[jd * 48u + mpt] | ((mpt == 1u + jd) ? 1u : 0u)) : 0u;
        // the (theta, p) vectors of the other point of a test: level l >= 1: the record of point mpt - l; level 0: the other edge (draw_neg / mntm_neg
        // for a forward doubling, _pos for a backward one) -- the draw's initial vectors until a doubling has written that side
        auto test_vecs = [&](int l, int& vt, int& vp, bool& pmom) __attribute__((always_inline)) {
            vt = MV_PT0 + 3 * ((int)mpt - l - 1); vp = vt + 1; pmom = false;
            if (l == 0) {
                const bool oinit = (vdir > 0) ? neg_init : pos_init;
                vt = oinit ? pvec(pb0) : ((vdir > 0) ? MV_TNEG_T : MV_TPOS_T);
                vp = oinit ? mv : ((vdir > 0) ? MV_TNEG_P : MV_TPOS_P);
                pmom = oinit;                            // (PRE_MOM: the draw's momentum is a row of the table, not a workspace vector)
            }
        };
        auto ld_test_p = [&](int vp, bool pmom, auto& dst) __attribute__((always_inline)) {
            if constexpr (POL::PRE_MOM) ld_ptr(pmom ? mom_lane_ptr(draw) : ws_lane_ptr(vp), dst);
            else ld_row(vp, 0, dst);
        };
        // theta / p of the other point of a test; dd then holds d = theta(mpt) - theta(mpt - l) (by direction).  DEFINED on every lane before
        // the (predicated) loads: left undefined, the values of lanes without a test count as live from the previous tick's loop -- 128 registers
        // held through the walk, 400 bytes of scratch per lane
        double dd[NS], Lp[NS];
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) { dd[s_] = 0.0; Lp[s_] = 0.0; }
        // SEARCH: the step of this leapfrog (nuts.ipp:62, 80-82).  INIT: e = 0, and the state is set after the (idle) updates
        if (srch) {
            if (!s_first) eps_() = eps_() * ((vdir == 1) ? 2.0 : 0.5);
            n_leap_() += 1ull; n_exec_() += 1ull;
        }
        const double e_tick = newpt ? esg_() : (srch ? eps_() : 0.0);
#if MI_MEMO_APF > 0 && MI_MEMO_WALK_CAP > 0
        // alpha of the next MI_MEMO_WALK_CAP leaves of a running chain whose points exist once this tick's point does (the walk below stops at the first
        // one that does not): requested here in one go.  The leaf sequence li, li + 1, .. and their points are known before the point is computed; this
        // tick's own point is taken from the registers in the walk
        double apf[MI_MEMO_WALK_CAP];
#pragma unroll
        for (int k = 0; k < MI_MEMO_WALK_CAP; ++k) apf[k] = 0.0;
        auto request_alphas = [&]() __attribute__((always_inline)) {
            if (__ballot(run) == 0ull) return;
            uint32_t i_ = li, n_ = memo_npt(li);
            const uint32_t np_ = newpt ? mpt : npts;
            bool ok_ = run;
#pragma unroll
            for (int k = 0; k < MI_MEMO_WALK_CAP; ++k) {
                const uint32_t t_ = (uint32_t)__builtin_ctz(~i_);
                const uint32_t nn_ = n_ + (t_ + 1u) - t_ * (t_ + 1u) / 2u;       // the point of leaf i_ + 1
                ok_ = ok_ && nn_ <= np_;
                if (ok_ && !(newpt && nn_ == mpt)) apf[k] = scp(nn_)->x;
                i_ = i_ + 1u; n_ = nn_;
            }
        };
#endif
#if MI_MEMO_APF == 2 && MI_MEMO_WALK_CAP > 0
        request_alphas();
#endif
        // one leapfrog of signed size e (nuts.ipp:132, nuts.cpp:139-154): the policy's half-kick, drift, gradient evaluation, half-kick
        // (act: the lanes that really step.  INIT and cut-short lanes pass e = 0: a policy with REPLAY may let them through -- x + 0 y is x unless y is
        //  non-finite, and then the chain is flagged and replayed anyway -- a policy without must leave their registers alone)
        const bool act = newpt || srch;
        pol.kick(th, pm, w, e_tick, act);
        pol.drift(th, pm, e_tick, act);
        __builtin_amdgcn_sched_barrier(0);
        // the first test's rows are requested here and used AFTER the mat-vec (8.6 us of matrix-pipe time in which the wave has nothing else
        // in flight): their latency costs nothing
        {
            const bool t1 = pmask != 0u;
            if (__ballot(t1) != 0ull) {
                const int l1 = t1 ? __builtin_ctz(pmask) : 1;
                int vq, vqp; bool pmom;
                test_vecs(l1, vq, vqp, pmom);
                if (t1) { ld_row(vq, 0, dd); ld_test_p(vqp, pmom, Lp); }
            }
        }
        MI_MPROF(1)
        double val = 0.0;
        pol.eval(th, w, val);
        pol.kick(th, pm, w, e_tick, act);
        double pU = pol.potential(th, w, val);           // nuts.ipp:134-138 / :50,65
        const double pK = pol.kinetic(pm);               // :140 / :51,66
        MI_MPROF(2)
        // ---- INIT: the chain's first state is on record; SEARCH: one step of nuts_find_initial_step_size
        if (__ballot(init) != 0ull) {
            if (init) {
                st_row(MV_PREV, 0, th); st_row(MV_WPREV, 0, w);
                prev_U_() = pU;                          // nuts.cpp:181 (no finiteness guard there)
                if constexpr (POL::REPLAY) { if (!is_finite(pU)) nf_() = 1.0; }
                H0_() = (is_finite(pU) ? pU : INF) + pK; // U0 + K0 (nuts.ipp:50-52)
                s_first = true;
            }
            const bool cont = prm.draw0 != 0 || draw != 0u;      // a continuation call, or (SPLIT) a later piece of a run: the same thing
            if (init && cont) eps_() = prm.step_out ? coh_ld(prm.step_out + cl) : 1.0;  // the step size comes back in
            start_sampling(init && cont);
            if (init && !cont) state = NS_SEARCH;
        }
        if (!is_finite(pU)) pU = INF;
        if (__ballot(srch) != 0ull) {
            const double dHs = -(pU + pK) + H0_();       // nuts.ipp:68,86
            note_nonfinite(srch && !is_finite(dHs));
            if (srch) { vdir = 2 * (dHs > log_half ? 1 : 0) - 1; s_first = false; }      // :75,88
            start_sampling(srch && !(dHs > neg_log2));   // :78,90: the loop ends
        }
        MI_MPROF(5)
        if (__ballot(run) == 0ull) continue;
        // ---- the point's scalars (nuts.ipp:146-157): n', s' as bits in LDS; alpha and U go to the scalar table with the record, at the END of the tick
        const double dH = -(pU + pK) + H0_();
        note_nonfinite(newpt && !is_finite(dH));         // pU (replaced by +inf above), pK or the draw's H0 non-finite
        const double ca_pt = det_exp((dH < 0.0) ? dH : 0.0);      // :157
        if (newpt) {
            const unsigned long long bit = 1ull << mpt;
            const double lu_ = log_u_();
            const bool cn_b = lu_ <= -pU - pK;           // :146
            const bool cs_b = lu_ < 1000.0 - pU - pK;    // :147
            okb_(0) = (okb_(0) & ~bit) | (cn_b ? bit : 0ull);
            okb_(11) = (okb_(11) & ~bit) | (cs_b ? bit : 0ull);
            n_exec_() += 1ull;
        }
#if MI_MEMO_APF == 1 && MI_MEMO_WALK_CAP > 0
        request_alphas();
#endif
        MI_MPROF(8)
        // ---- the tests whose second point this is: [ d . p(n1) >= 0 ] * [ d . p(mpt) >= 0 ], d = theta(mpt) - theta(n1) by direction (:224-229).
        //      LOADS ONLY between here and the end of the walk: memory operations of a wave complete in order, and a load behind the 3 KB of a
        //      record store waits for all of it (measured: 10 k cycles per test that way)
        {
            bool first = true;
#pragma unroll 1
            while (__ballot(pmask != 0u) != 0ull) {
#ifdef MI_NUTS_REG_PROF
                n_pair_it++;
#endif
                const bool t = pmask != 0u;
                const int l = t ? __builtin_ctz(pmask) : 1;
                const int n1 = (int)mpt - l;
                if (!first) {
                    int vq, vqp; bool pmom;
                    test_vecs(l, vq, vqp, pmom);
#if MI_MEMO_NT == 3
                    if (t) { ld_ptr_nt(ws_lane_ptr(vq), dd); if constexpr (POL::PRE_MOM) ld_ptr_nt(pmom ? mom_lane_ptr(draw) : ws_lane_ptr(vqp), Lp); else ld_ptr_nt(ws_lane_ptr(vqp), Lp); }
#else
                    if (t) { ld_row(vq, 0, dd); ld_test_p(vqp, pmom, Lp); }
#endif
                }
                first = false;
                // (two passes: d . p(n1) with theta, d, p(n1) as operands, then d . p(mpt) with d, p -- four vectors as VALU operands of one loop
                //  are the whole architectural register file)
                double q1 = 0.0, q2 = 0.0;
#pragma unroll
                for (int s_ = 0; s_ < NS; ++s_) {
                    dd[s_] = (vdir > 0) ? (th[s_] - dd[s_]) : (dd[s_] - th[s_]);
                    q1 = dfma(dd[s_], Lp[s_], q1);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s_ = 0; s_ < NS; ++s_) q2 = dfma(dd[s_], pm[s_], q2);
                q1 = q1 + __shfl_xor(q1, 32); q1 = q1 + __shfl_xor(q1, 16);
                q2 = q2 + __shfl_xor(q2, 32); q2 = q2 + __shfl_xor(q2, 16);
                if (t) {
                    const unsigned long long bit = (l == 0) ? (1ull << 63) : (1ull << n1);      // (level 0: q1, q2 are the two products of :286-287, in either order)
                    const bool ok = (q1 >= 0.0) && (q2 >= 0.0);
                    okb_(l) = (okb_(l) & ~bit) | (ok ? bit : 0ull);
                    pmask &= pmask - 1u;
                }
            }
        }
        // the tree's far edge (= the first leaf of its second half, point 1 + jd; the leaf itself at depth 0) is what a successful doubling leaves
        // in draw_pos / draw_neg (src/nuts.cpp:241-256); a doubling that fails before that leaf ends the draw, so writing it early is harmless.
        // Nothing of this doubling reads it (its top-level test was taken above, from the registers): it is stored with the record, at the END of the tick
        const bool st_edge = newpt && mpt == 1u + jd;    // (re-derived here: one lane mask less to carry across the mat-vec)
        if (st_edge) { if (vdir > 0) pos_init = false; else neg_init = false; }
        if (newpt) npts = mpt;
        MI_MPROF(3)
        // ------------------------------------------------------------ C. walk the leaves this point unblocks (nuts.ipp:146-158, 212-239)
        bool wl = run;                                   // (the leaf a chain waits at sits on its newest point)
        bool at_fin = false, complete = false;
        uint32_t cn_i = 0, cna_i = 0, cref = 0;
        double ca = newpt ? ca_pt : ca_keep;             // alpha of the leaf the walk starts at (a chain cut short last tick kept its own)
        double ca_nx = 0.0;
        uint32_t n_walked = 0;                           // leaves walked this tick (= the reference's leapfrogs, nuts.ipp:132)
        uint32_t n = memo_npt(li);                       // the point of leaf li; carried: n(i + 1) = n(i) + (t + 1) - t (t + 1) / 2, t = the trailing ones of i
#if MI_MEMO_WALK_CAP > 0
        int walk_budget = MI_MEMO_WALK_CAP;
#endif
#pragma unroll 1
        while (__ballot(wl) != 0ull) {
#ifdef MI_NUTS_REG_PROF
            n_walk_it++;
#endif
#if MI_MEMO_WALK_CAP > 0
            if (walk_budget-- == 0) break;
#endif
            // the next leaf's alpha, in case the chain gets that far: on record if its point exists -- this tick's own point is still in registers
            const uint32_t t1 = (uint32_t)__builtin_ctz(~li);
            const uint32_t nn = n + (t1 + 1u) - t1 * (t1 + 1u) / 2u;         // the point of leaf li + 1
#if MI_MEMO_APF > 0 && MI_MEMO_WALK_CAP > 0
            ca_nx = (newpt && nn == mpt) ? ca_pt : apf[0];                   // (requested ahead: this lane's k-th iteration is its k-th leaf of the tick)
#pragma unroll
            for (int k = 0; k + 1 < MI_MEMO_WALK_CAP; ++k) apf[k] = apf[k + 1];
#else
            if (wl && nn <= npts) ca_nx = (newpt && nn == mpt) ? ca_pt : scp(nn)->x;
#endif
            const unsigned long long nbit = 1ull << n;
            if (wl) {
                cn_i = (okb_(0) & nbit) ? 1u : 0u;
                cna_i = 1u; cref = n;
                n_walked++;
            }
            bool failed = wl && !(okb_(11) & nbit);
            uint32_t pend_level = jd + 1;
            if constexpr (MI_MEMO_WALK_V2 != 0 && POL::LANE_WALK) {
            // levels 1 .. t1 (the trailing ones of li; li < 2^jd, so t1 <= jd): every one of them merges (a failure on the way does not stop the merges of
            // nuts.ipp:212-222 above it, and their alpha sums reach the dual averaging), in rounds of four
            const bool failed0 = failed;
            const uint32_t tm = wl ? t1 : 0u;
            uint32_t lb = 0;                                                 // levels done
#pragma unroll 1
            while (__ballot(lb < tm) != 0ull) {
                const bool act_r = lb < tm;
                const uint32_t cnt = act_r ? (((tm - lb) < 4u) ? (tm - lb) : 4u) : 0u;
                // the uniforms of slots [uslot, uslot + cnt) (:213, one per merge, in level order): lane class j4 holds slot ub0 + j4
                {
                    const bool stale = act_r && ((uslot - ub0) >= 4u || (uslot - ub0) + cnt > 4u);
                    if (__ballot(stale) != 0ull) {               // every chain of the wave refills from its own current slot
                        ub0 = uslot;
                        const u32x4 wv = rng_block(prm.seed, prm.chain0 + cl, draw + prm.draw0, uslot + (uint32_t)j4, STREAM_UNIFORM);
                        ubuf = 2ull * ((((unsigned long long)wv.y << 32) | (unsigned long long)wv.x) >> 12) + 1ull;
                    }
                }
                const uint32_t off = uslot - ub0;                            // this lane decides merge kj = j4 - off of the round (if there is one)
                const uint32_t kj = (uint32_t)j4 - off;
                const bool valid_j = act_r && (uint32_t)j4 >= off && kj < cnt;
                const uint32_t rb = lrow + lb * 512u;                        // rows of level lb + 1 + k at immediate offsets
                double lav[4]; unsigned long long lpv[4], okv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    lav[k] = *(lds_dptr)(uintptr_t)(rb + (uint32_t)(R_LA + 1 + k) * 512u);
                    lpv[k] = *(lds_uptr)(uintptr_t)(rb + (uint32_t)(R_LP + 1 + k) * 512u);
                    okv[k] = *(lds_uptr)(uintptr_t)(rb + (uint32_t)(R_OKB + 1 + k) * 512u);
                }
                uint32_t pn[4], pna[4], prf[4], pre[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t pk = (uint32_t)lpv[k];
                    pn[k] = pk & 0x7ffu; pna[k] = (pk >> 11) & 0x7ffu; prf[k] = pk >> 22;
                }
                pre[0] = cn_i; pre[1] = pre[0] + pn[0]; pre[2] = pre[1] + pn[1]; pre[3] = pre[2] + pn[2];      // n'' of the merge: the counts below it
                const uint32_t a_j = (kj == 0u) ? pre[0] : (kj == 1u) ? pre[1] : (kj == 2u) ? pre[2] : pre[3];
                const uint32_t p_j = (kj == 0u) ? pn[0] : (kj == 1u) ? pn[1] : (kj == 2u) ? pn[2] : pn[3];
                // :212, :215-217: keep new_draw_p unless u < n'' / (n' + n'') (0 / 0 = NaN keeps); the HIGHEST level that replaces it decides
                const bool rej = valid_j && !u_below(ubuf, a_j, p_j + a_j);
                const unsigned long long mine = (__ballot(rej) >> (lane & 15)) & 0x0001000100010001ull;
                if (act_r && mine != 0ull) {
                    const uint32_t ks = ((63u - (uint32_t)__builtin_clzll(mine)) >> 4) - off;
                    cref = (ks == 0u) ? prf[0] : (ks == 1u) ? prf[1] : (ks == 2u) ? prf[2] : prf[3];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if ((uint32_t)k < cnt) {
                        cn_i = cn_i + pn[k];                                 // :220-222
                        cna_i = cna_i + pna[k];
                        ca = lav[k] + ca;
                        if (!failed0) {                                      // :226-229, evaluated when its second point appeared
                            const uint32_t l = lb + 1u + (uint32_t)k;
                            const uint32_t n1 = n - l * (l + 1u) / 2u;       // the node's first leaf: li with its l low (set) bits cleared
                            if (!((okv[k] >> n1) & 1ull)) failed = true;
                        }
                    }
                }
                if (act_r) { uslot += cnt; lb += cnt; }
            }
            // a leaf that failed: the merges at the set bits of li above its first zero bit (a first half of a failed subtree returns at once, a second half
            // merges).  The draw ends with this doubling, so of those merges only the sums that reach the dual averaging matter (alpha', n_alpha': src/nuts.cpp
            // :246,255,294-302); the proposal, n' and the uniforms of a failed doubling are read by nobody
            {
                uint32_t hb = (wl && failed) ? ((li >> (t1 + 1u)) << (t1 + 1u)) : 0u;
#pragma unroll 1
                while (__ballot(hb != 0u) != 0ull) {
                    if (hb != 0u) {
                        const uint32_t l = (uint32_t)__builtin_ctz(hb) + 1u;
                        const uint32_t ra = lrow + l * 512u;
                        ca = *(lds_dptr)(uintptr_t)(ra + (uint32_t)R_LA * 512u) + ca;
                        cna_i = cna_i + (((uint32_t)*(lds_uptr)(uintptr_t)(ra + (uint32_t)R_LP * 512u) >> 11) & 0x7ffu);
                        hb &= hb - 1u;
                    }
                }
            }
            pend_level = t1 + 1u;
            } else {
            bool walking = wl;
#pragma unroll 1
            for (uint32_t l = 1; l <= (uint32_t)MEMO_MAX_DEPTH; ++l) {
                if (walking && l > jd) walking = false;                      // reached the root of its own tree
                const bool bit = ((li >> (l - 1)) & 1u) != 0u;
                if (walking && !failed && !bit) { pend_level = l; walking = false; }   // first half: wait here
                if (__ballot(walking) == 0ull) break;
                const bool mrg = walking && bit;
                if (__ballot(mrg) == 0ull) continue;
                const unsigned long long z = uni(mrg);                       // :213
                if (mrg) {
                    uslot++;
                    const uint32_t pk = (uint32_t)lp_((int)l);
                    const uint32_t p_n = pk & 0x7ffu, p_na = (pk >> 11) & 0x7ffu, p_ref = pk >> 22;
                    if (!u_below(z, cn_i, p_n + cn_i)) cref = p_ref;             // :212, :215-217: keep new_draw_p unless u < n'' / (n' + n'') (0 / 0 = NaN keeps)
                    cn_i = p_n + cn_i;                                           // :220-222
                    ca = la_((int)l) + ca;
                    cna_i = p_na + cna_i;
                    if (!failed) {                                               // :226-229, evaluated when its second point appeared
                        const uint32_t n1 = n - l * (l + 1u) / 2u;               // the node's first leaf: li with its l low (set) bits cleared
                        if (!((okb_((int)l) >> n1) & 1ull)) failed = true;
                    }
                }
            }
            }
            if (wl) {
                const bool keep = !failed;
                complete = keep && (li == (1u << jd) - 1u);
                if (keep && !complete) {                 // a pending first half: scalars to LDS, the proposal by reference (its point)
                    la_((int)pend_level) = ca;
                    lp_((int)pend_level) = (unsigned long long)(cn_i | (cna_i << 11) | (cref << 22));
                    li = li + 1u; n = nn;
                    ca = ca_nx;
                    wl = nn <= npts;
                } else {
                    at_fin = true; wl = false;
                }
            }
        }
        if (run && !at_fin) ca_keep = ca;                // (only read back by a chain whose walk was cut short: wl still set)
        if (run) n_leap_() += (unsigned long long)n_walked;
        MI_MPROF(4)
        // ------------------------------------------------------------ D. end of a doubling (src/nuts.cpp:260-289)
        if (__ballot(at_fin) != 0ull) {
            bool take = false;
            if (__ballot(complete) != 0ull) {
                const unsigned long long z = uni(complete);                 // :261
                if (complete) {
                    uslot++;
                    take = u_below(z, cn_i, (uint32_t)n_val_());                // :263 (n is a count: 1 + the n' of the doublings before)
                    if (take) { good_round = 1; pb = 1 - pb0; }                 // :264-277
                }
            }
            // the proposal is a point of the trajectory.  Its (theta, P theta) become prev_draw AND the origin of whatever this chain does next:
            //   this tick's own point (whose record is not written, and never will be: the doubling is over): prev_draw stored from the registers,
            //     which stay as they are;
            //   an earlier point: its record is requested INTO THE REGISTERS below (as the next origin) and prev_draw's copy is stored from there at
            //     the top of the next tick (cp_pend) -- no load-wait-store round trip in the tick
            const bool from_regs = take && newpt && cref == mpt;
            const bool from_rec = take && !from_regs;
            if (from_regs) { st_row(pvec(1 - pb0), 0, th); st_row(wvec(1 - pb0), 0, w); prev_U_() = pU; }
            if (__ballot(from_rec) != 0ull) {
                if (from_rec) prev_U_() = scp(cref)->y;
            }
            if (at_fin) { alpha_() = ca; n_alpha_() = (double)cna_i; n_val_() = n_val_() + (double)cn_i; }   // :246,255 ; :283
            // [ (pos - neg) . p_neg >= 0 ] * [ (pos - neg) . p_pos >= 0 ] (:286-289): taken when the far edge was the point in registers
            const bool s_ok = complete && ((okb_(0) >> 63) & 1ull) != 0ull;
            const bool more = at_fin && s_ok && (jd + 1 < max_depth);
            // a doubling that ends ON its far edge (depth 0 and 1: point 1 + jd is its last point) and is followed by another one of this draw:
            // the edge is stored here, before the direction changes and the registers take the next origin
            {
                const bool edge_now = st_edge && more;
                if (__ballot(edge_now) != 0ull) {
                    if (edge_now) { st_row((vdir > 0) ? MV_TPOS_T : MV_TNEG_T, 0, th); st_row((vdir > 0) ? MV_TPOS_P : MV_TNEG_P, 0, pm); }
                }
            }
            if (at_fin) jd = jd + 1;                                     // :284
            const bool ended = at_fin && !more;
            bool roll = false;
            if (__ballot(ended) != 0ull) {
                end_draw(ended, jd);
                roll = ended && draw < n_total && mom_ready && !row_pend && !piece_done;
                if (ended && !roll) state = NS_NEED_DRAW;                // the phase: its row, its next momentum, or the end of its run
                roll_state(roll);
            }
            begin_doubling(more || roll);
            // the origin of the doubling that starts in the next tick (prev_draw, mntm_vec, P prev_draw), requested NOW: its round trip runs under
            // the record stores below, the loop head and the phase instead of in front of the next kick
            const bool go = more || roll;
            if (__ballot(go || from_rec) != 0ull) {
                // an accepted proposal that is an earlier point of the trajectory: its record comes INTO THE REGISTERS -- as the next origin when the
                // chain goes on at once, and in any case as the source of prev_draw's copy at the top of the next tick (cp_pend: stored from there before
                // the phase reads prev_draw for the chain's row or its exit) -- so the tick has no load-wait-store copy at all
                if (from_rec) {
                    const int vq = MV_PT0 + 3 * ((int)cref - 1);
                    ld_row(vq, 0, th); ld_row(vq + 2, 0, w); cp_pend = true;
                }
                if (go) {
                    if (!take) { ld_row(pvec(pb), 0, th); ld_row(wvec(pb), 0, w); }     // (take from the registers: they stay as they are)
                    ld_draw_mom(false, pm);
                    org_ok = true;
                }
            }
        }
        MI_MPROF(9)
        // ------------------------------------------------------------ E. the record of this tick's point, for the chains whose doubling goes on
        {
            const bool rec = newpt && !at_fin;
            if (__ballot(rec) != 0ull) {
                if (rec) {
                    const int vr = MV_PT0 + 3 * ((int)mpt - 1);
#if MI_MEMO_NT == 1 || MI_MEMO_NT == 3
                    st_row(vr, 0, th); st_row(vr + 1, 0, pm); st_row_nt(vr + 2, 0, w);
#elif MI_MEMO_NT == 2
                    st_row_nt(vr, 0, th); st_row_nt(vr + 1, 0, pm); st_row_nt(vr + 2, 0, w);
#else
                    st_row(vr, 0, th); st_row(vr + 1, 0, pm); st_row(vr + 2, 0, w);
#endif
                    *scp(mpt) = double2{ca_pt, pU};
                    if (st_edge) {               // (a chain whose doubling ended in this tick stored it there, if the draw goes on)
                        st_row((vdir > 0) ? MV_TPOS_T : MV_TNEG_T, 0, th); st_row((vdir > 0) ? MV_TPOS_P : MV_TNEG_P, 0, pm);
                    }
                }
            }
        }
        MI_MPROF(6)
    }
#ifdef MI_NUTS_REG_PROF
    if (prm.prof && blockIdx.x == 0 && threadIdx.x == 0) {
        for (int k = 0; k < 12; ++k) prm.prof[k] = pc[k];
        prm.prof[12] = n_ticks; prm.prof[13] = n_active; prm.prof[14] = n_walk_it; prm.prof[15] = n_pair_it;
    }
#endif
#undef MI_MPROF
#undef la_
#undef lp_
#undef okb_
#undef nf_
#undef eps_
#undef prev_U_
#undef H0_
#undef log_u_
#undef esg_
#undef next_K_
#undef next_lu_
#undef n_leap_
#undef n_exec_
#undef n_acc_
#undef h_val_
#undef eps_bar_
#undef mu_val_
#undef prev_K_
#undef n_val_
#undef alpha_
#undef n_alpha_
#undef MI_RD
#undef MI_RU
}


}  // namespace memo
}  // namespace mi
