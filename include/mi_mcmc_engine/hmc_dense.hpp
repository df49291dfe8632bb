// hmc_dense.hpp -- fused many-chain HMC for Gaussian targets whose gradient is a dense
// mat-vec, on the gfx950 fp64 matrix cores.
//
// Replaces, for C independent chains, the sampling loop of mcmc::internal::hmc_impl
// (/root/reference/src/hmc.cpp:155-205) together with the BaseMatrixOps calls inside it
// (rnorm_vec_inplace :156, L*z :158, DOT_PROD :160,184, Minv*p :171, runif :189) for the
// identity preconditioner (precond_mat unset -> BMO_MATOPS_EYE, :57).
//
// Mapping (one wavefront = 16 chains, whole trajectory register-resident):
//   G[d x 16] = P[d x d] * Theta[d x 16] is issued as v_mfma_f64_16x16x4_f64 tiles:
//     A = P fragment   lane l holds P[16t + (l&15)][4s + (l>>4)]      (read from LDS)
//     B = Theta slice  lane l holds theta[4s + (l>>4)] of chain (l&15) (VGPR)
//     D = G tile       lane l holds g[16t + 4r + (l>>4)] of chain (l&15), r = 0..3
//   The D layout of tile t register r IS the B layout of slice s = 4t + r, so the momentum
//   kick / position drift after every gradient are per-lane register updates with no
//   cross-lane traffic, and the 2^k-free leapfrog never leaves the register file.
//   P (d_pad^2 * 8 B, 128 KiB at d = 128) is staged once per workgroup in LDS in fragment
//   order: one conflict-free ds_read_b64 per MFMA.
//
// Arithmetic is specified operation by operation (compile with -ffp-contract=off):
//   each gradient row is one sequential fma chain over k ascending (the MFMA accumulates
//   its four k in order); dot products are 4 strided fma chains (one per lane group)
//   combined as (q0+q2)+(q1+q3); elementwise updates follow the reference expressions
//   p + (eps*g)/2 (:126) and theta + eps*p (:171) with one rounding per operator.
#pragma once

#include "det_math.hpp"

#ifndef MI_HMC_RNG_STAGED
#define MI_HMC_RNG_STAGED 0
#endif

#ifndef MI_HMC_RNG_PAIRS
#define MI_HMC_RNG_PAIRS 1   // (2 measured: no difference, 100.8 ms both) Philox / Box-Muller chains the scheduler may interleave in the plain kernel's momentum draw
#endif

#ifndef MI_HMC_WPB
#define MI_HMC_WPB 8     // waves per workgroup of the plain kernel (two per SIMD)
#endif

namespace mi {

typedef double double4_t __attribute__((ext_vector_type(4)));

struct HmcParams {
    const double* P;        // device, d x d row-major precision
    uint32_t d;
    uint64_t C;             // chains in this launch
    uint64_t chain0;        // global id of local chain 0
    double* theta;          // [d][C] in/out: always the last accepted state
    double* wsave;          // [n_waves][2][NS][64] workspace: last accepted theta and P*theta
    int vals_bound;         // general variant: settings.vals_bound (0: only a diagonal precond_mat)
    uint32_t draw0;         // index of this call's first draw in the chains' random streams (mi_chains.draw0)
    const double* Minv;     // DENSE_M: INV(precond_mat), device: d*d row-major (d <= 64, staged into LDS) / fragment order (d > 64)
    const double* Lchol;    // DENSE_M: CHOL_LOWER(precond_mat), likewise
    double* draws;          // [n_keep][d][C] or nullptr
    uint64_t* n_accept;     // [C] or nullptr
    uint64_t* n_leap;       // [C] or nullptr
    uint64_t seed;
    uint32_t n_burnin, n_keep, n_leap_steps;
    double eps;
    const int* btype;       // bounded runs: per-dimension bounds type 1..4 (determine_bounds_type.hpp:27-57), device
    const double* lb;       // lower / upper bounds, device, d values
    const double* ub;
    const double* m_sqrt;   // general runs: diag of CHOL_LOWER(precond) (hmc.cpp:59), device, d values (ones = identity)
    const double* m_inv;    // diag of INV(precond) (hmc.cpp:58)
    uint32_t ablate;        // profiling only: 1 = skip kick/drift, 2 = skip mat-vec (results meaningless)
    uint32_t stagger;       // start delay of the second wave of each SIMD, in s_sleep(127) units
    uint32_t m_per_chain;   // DIAGM: 0 = one mass for all chains (m_sqrt / m_inv are [d]); 1 = per-chain masses (mi_chains.mass_diag): [d][C]
    uint32_t* nf_flag;      // plain kernels: [C + 1] or nullptr.  A chain whose energies went non-finite is flagged (nf_flag[c] = 1,
                            // nf_flag[C] = 1) and its theta / n_accept / n_leap are left untouched: literal.hpp replays it
    uint32_t sep_target;    // general variant: P is the expanded diagonal of an ISO / DIAG target, whose gradient the reference's target function takes
                            // element-wise (target_times below)
};

template <int NS>
__device__ __forceinline__ double dot4(const double (&x)[NS], const double (&y)[NS])
{
    double q = 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s) q = dfma(x[s], y[s], q);
    q = q + __shfl_xor(q, 32);
    q = q + __shfl_xor(q, 16);
    return q;
}

// The reference multiplies by its (identity / diagonal) matrices as DENSE products (inv_precond_matrix * mntm, hmc.cpp:171;
// jacob_matrix * grad_obj, :122), and the oracle restates them as dense fma chains: one entry x_k = +-inf or NaN makes
// 0 * x_k = NaN in every OTHER row.  The kernels apply the diagonal element-wise (y_i = D_ii x_i) and then call this to
// reproduce that poisoning: y_i = NaN wherever another dimension of the same chain is non-finite.  n_valid = d (padding
// dimensions are never touched).  The common path is one is_finite per slice, two shuffles and a ballot.
template <int NS>
__device__ __forceinline__ void dense_product_poison(const double (&x)[NS], double (&y)[NS], int j, uint32_t d)
{
    int nloc = 0;
#pragma unroll
    for (int s = 0; s < NS; ++s) nloc += is_finite(x[s]) ? 0 : 1;
    int n = nloc + __shfl_xor(nloc, 16);
    n = n + __shfl_xor(n, 32);
    if (__ballot(n != 0) == 0ull) return;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int others = n - (is_finite(x[s]) ? 0 : 1);
        if (others > 0 && (uint32_t)(4 * s + j) < d) y[s] = __builtin_nan("");
    }
}

// w = P * th for the wave's 16 chains. afrag points at this lane's column of the LDS fragments.
// Software pipeline: the NT fragments of slice s+1 are read from LDS while the NT MFMAs of slice
// s issue (NT*64 cycles of matrix pipe cover the LDS latency); sched_barriers pin that order.
template <int NT>
__device__ __forceinline__ void matvec_mfma(const double* __restrict__ afrag, const double (&th)[4 * NT],
                                            double (&w)[4 * NT])
{
    constexpr int NS = 4 * NT;
    double4_t acc[NT];
    double a_cur[NT], a_nxt[NT];
    // ds_read_b64 carries a 16-bit byte offset: fragments (t, s) with t >= 4 of a d = 128 matrix lie 64 KB or more past
    // `afrag`, and the compiler then forms each of those 128 addresses with a VALU add per read (126 v_add_u32 per mat-vec,
    // a quarter of the loop's VALU instructions).  A second, opaque base 64 KB up keeps every read on an immediate offset.
    typedef const double __attribute__((address_space(3)))* lds_cptr;
    lds_cptr lo = (lds_cptr)afrag;
    uint32_t hi_off = (uint32_t)(uintptr_t)lo + (NT > 4 ? 4u * NS * 64u * 8u : 0u);
    if constexpr (NT > 4) asm volatile("" : "+v"(hi_off));
    lds_cptr hi = (lds_cptr)(uintptr_t)hi_off;
    auto frag = [&](int t, int s) __attribute__((always_inline)) -> double {
        return (NT > 4 && t >= 4) ? hi[((t - 4) * NS + s) * 64] : lo[(t * NS + s) * 64];
    };
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
        a_cur[t] = frag(t, 0);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s + 1 < NS) {
#pragma unroll
            for (int t = 0; t < NT; ++t) a_nxt[t] = frag(t, s + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[t], th[s], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) a_cur[t] = a_nxt[t];
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        w[4 * t + 0] = acc[t][0];
        w[4 * t + 1] = acc[t][1];
        w[4 * t + 2] = acc[t][2];
        w[4 * t + 3] = acc[t][3];
    }
}

// Stage P into LDS in MFMA A-fragment order: frag f = t*NS + s, lane l -> P[16t + (l&15)][4s + (l>>4)].
template <int NT>
__device__ __forceinline__ void stage_precision(const double* __restrict__ P, uint32_t d, double* lds)
{
    constexpr int NS = 4 * NT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int f = wave; f < NT * NS; f += nw) {
        const int t = f / NS, s = f % NS;
        const uint32_t row = 16 * t + (lane & 15), col = 4 * s + (lane >> 4);
        lds[f * 64 + lane] = (row < d && col < d) ? P[(size_t)row * d + col] : 0.0;
    }
    __syncthreads();
}


// The same mat-vec with the A fragments read from GLOBAL memory in fragment order (gfrag[(t NS + s) 64 + lane], what
// stage_precision writes to LDS): for the second and third matrix of a run whose d > 64 -- INV(precond_mat), CHOL_LOWER(precond_mat),
// 128 KB each -- which do not fit next to the precision in the 160 KB of LDS.  Every wave of the launch reads the same 128 KB, so
// they live in L2; a wave-load is one coalesced 512-byte segment.  PD slices are in flight ahead of the MFMAs (a slice of NT = 8
// MFMAs lasts 512 cycles, an L2 hit ~ 500-900).  Same fma order as matvec_mfma, hence the same bits.
template <int NT>
__device__ __forceinline__ void matvec_mfma_g(const double* __restrict__ gfrag, const double (&th)[4 * NT], double (&w)[4 * NT])
{
    constexpr int NS = 4 * NT;
    constexpr int PD = 3;
    // Addressing: wave-uniform base (SGPR pair, taken from lane 0: gfrag = base + lane, every lane is active here) + the lane's 8
    // bytes as a 32-bit offset + an immediate.  With the per-lane 64-bit pointer the compiler kept one address pair per fragment
    // row -- 256 per matrix -- as loop invariants of the sampler's loops, spilled them (1 074 spilled VGPRs, 4.3 KB of scratch in the
    // d = 128 dense-precond HMC kernel) and reloaded one in front of every load.  The base is pinned per call (asm) for the same reason.
    uint32_t b_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)gfrag);
    uint32_t b_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uintptr_t)gfrag >> 32));
    asm volatile("" : "+s"(b_lo), "+s"(b_hi));
    const double* const gbase = reinterpret_cast<const double*>(((uintptr_t)b_hi << 32) | (uintptr_t)b_lo);
    const uint32_t lane_ = (uint32_t)(threadIdx.x & 63);
    auto frag = [&](int t, int s_) -> double { return gbase[(uint32_t)((t * NS + s_) * 64) + lane_]; };
    double4_t acc[NT];
    double a[PD + 1][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int p = 0; p < PD && p < NS; ++p)
#pragma unroll
        for (int t = 0; t < NT; ++t) a[p][t] = frag(t, p);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s + PD < NS) {
#pragma unroll
            for (int t = 0; t < NT; ++t) a[(s + PD) % (PD + 1)][t] = frag(t, s + PD);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s % (PD + 1)][t], th[s], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        w[4 * t + 0] = acc[t][0];
        w[4 * t + 1] = acc[t][1];
        w[4 * t + 2] = acc[t][2];
        w[4 * t + 3] = acc[t][3];
    }
}

// second / third matrix of a dense-preconditioner run: LDS fragments (d <= 64) or global fragments (d > 64)
template <int NT>
constexpr bool dense_m_from_global() { return NT > 4; }
template <int NT>
__device__ __forceinline__ void matvec_m2(const double* __restrict__ frag_lane, const double (&x)[4 * NT], double (&y)[4 * NT])
{
    if constexpr (dense_m_from_global<NT>()) matvec_mfma_g<NT>(frag_lane, x, y);
    else matvec_mfma<NT>(frag_lane, x, y);
}

// P x of a built-in Gaussian in the kernels that follow a chain THROUGH the non-finite regime themselves (the general variants: nothing replays their
// chains): a DENSE target's mat-vec -- and for an ISO / DIAG target, whose P here is the expanded diagonal, the element-wise product the reference's
// target function forms (oracle: orc_target_kernel).  The bits of the mat-vec while everything is finite; a +-inf coordinate stays in its own dimension,
// where 0 * inf of the expanded zeros would put NaN into every other one (found by the round-6 fuzz: hmc, DIAG target, dense precond_mat, d = 64, a chain
// started at inf whose proposal is accepted -- draws NaN where the oracle has +-inf).  `sep` is wave-uniform.
template <int NT>
__device__ __forceinline__ void target_times(const double* __restrict__ afrag, const double* __restrict__ P, uint32_t d, bool sep,
                                             const double (&x)[4 * NT], double (&w)[4 * NT])
{
    if (sep) {
        const uint32_t j = (threadIdx.x & 63u) >> 4;
#pragma unroll
        for (int s = 0; s < 4 * NT; ++s) {
            const uint32_t dim = 4u * (uint32_t)s + j;
            w[s] = (dim < d) ? P[(size_t)dim * (d + 1u)] * x[s] : 0.0;
        }
    } else matvec_mfma<NT>(afrag, x, w);
}

// ---- box constraints (element-wise maps of /root/reference/include/misc/transform_vals.hpp:25-119,
//      log_jacobian.hpp:25-58, inv_jacobian_adjust.hpp:25-56), one dimension at a time
constexpr double EPS_DBL = 2.220446049250313e-16;    // mcmc_options.hpp:103

// The four box helpers carry the reference's case analysis with exp / log inside.  Inlined at every slice of every use they made the
// d = 128 general kernels 1 - 1.6 MB of instructions (64 KB instruction cache) and the library 23 MB; as out-of-line leaf functions
// the same kernels are 0.2 - 0.4 MB and faster (hmc 209 -> 154 ms, nuts 501 -> 439 ms, rwmh 22 -> 17.5 ms with a diagonal precond_mat
// on configs[1]'s shape; only bounded MALA, which calls them most, pays: 103 -> 122 ms).
#define MI_BOX_INLINE __attribute__((noinline))
__device__ MI_BOX_INLINE double box_transform(double v, int bt, double lb, double ub)
{
    switch (bt) {
    case 2: return det_log(v - lb + EPS_DBL);
    case 3: return -det_log(ub - v + EPS_DBL);
    case 4: return det_log(v - lb + EPS_DBL) - det_log(ub - v + EPS_DBL);
    default: return v;
    }
}
__device__ MI_BOX_INLINE double box_inv_transform(double v, int bt, double lb, double ub)
{
    switch (bt) {
    case 2: return !is_finite(v) ? lb + EPS_DBL : lb + EPS_DBL + det_exp(v);
    case 3: return !is_finite(v) ? ub - EPS_DBL : ub - EPS_DBL - det_exp(-v);
    case 4: {
        if (!is_finite(v)) {
            if (v != v) return (ub - lb) / 2;
            return (v < 0.0) ? lb + EPS_DBL : ub - EPS_DBL;
        }
        const double e = det_exp(v);
        const double r = (lb - EPS_DBL + (ub + EPS_DBL) * e) / (1.0 + e);
        return is_finite(r) ? r : ub - EPS_DBL;
    }
    default: return v;
    }
}
// diagonal entry of inv_jacobian_adjust
__device__ MI_BOX_INLINE double box_inv_jacobian(double v, int bt, double lb, double ub)
{
    switch (bt) {
    case 2: return det_exp(-v);
    case 3: return det_exp(v);
    case 4: { const double e = det_exp(v); return ((e + 1) * (e + 1)) / (e * (ub - lb)); }
    default: return 1.0;
    }
}
// one summand of log_jacobian (callers skip bt == 1: the reference adds nothing for it)
__device__ MI_BOX_INLINE double box_log_jacobian_term(double v, int bt, double lb, double ub)
{
    switch (bt) {
    case 2: return v;
    case 3: return -v;
    case 4: {
        const double e = det_exp(v);
        if (is_finite(e)) return det_log(ub - lb) + v - 2 * det_log(1 + e);
        return det_log(ub - lb) - v;
    }
    default: return 0.0;
    }
}

// WPB = waves per workgroup: 4 (one wave per SIMD, 512-register budget) or 8 (two waves per SIMD,
// 256 registers each: one wave's VALU phases -- kick/drift, RNG, accept -- hide under the other's MFMAs).
// BOUNDED (the general variant): settings.vals_bound and / or a diagonal precond_mat.
// vals_bound (hmc.cpp:84-95,107-122,134-136,211-218): the chain lives in the transformed
// space; the target is evaluated at x = inv_transform(theta), the kick uses inv_jacobian * grad, the energy
// adds log_jacobian (summed sequentially over dimensions, as the reference's scalar loop does).
// DENSE_M (with BOUNDED): a dense precond_mat (hmc.cpp:57-59,158-160,171,184).  INV(M) and CHOL_LOWER(M) are computed on the
// host with the oracle's Gauss-Jordan / Cholesky and staged into LDS as two more sets of MFMA A-fragments: p = L z,
// theta += eps (Minv p) and K = p.(Minv p)/2 are mat-vecs with the same fma order as the oracle's dense products (so the
// NaN poisoning of section 3 of DESIGN.md happens by itself).  d <= 64: three matrices have to share the LDS.
// DIAGM (without BOUNDED): a DIAGONAL precond_mat and no bounds -- the plain kernel with p = sqrt(m) z, theta += eps (p / m) and
// K = p.(p / m) / 2 applied element-wise from two LDS tables; two waves per SIMD like the plain kernel (the general variant holds a
// fourth vector and runs one).  The NaN poisoning of the reference's dense `inv_precond_matrix * mntm` is handled as in the plain
// kernel: detected through the energies, flagged, replayed by literal.hpp (precond = 1).
// PCM (with DIAGM): PER-CHAIN diagonal masses (mi_chains.mass_diag: chain c runs with precond_mat = diag(mass[:, c])): the two tables are
// [d][C] in global memory, this lane's entry of slice s at [(4 s + j) C + c], read where it is used (a chain's column is 2 KB at d = 128;
// the 16 chains of a wave make 128-byte segments).
template <int NT, int WPB, bool BOUNDED = false, bool DENSE_M = false, bool DIAGM = false, bool PCM = false>
__global__ MI_NO_DS_MERGE __launch_bounds__(64 * WPB, WPB / 4) void hmc_gauss_mfma_kernel(const HmcParams prm)
{
    static_assert(!PCM || DIAGM, "per-chain masses ride the diagonal-mass variant");
    static_assert(!DENSE_M || BOUNDED, "the dense preconditioner rides the general variant");
    static_assert(!DIAGM || !BOUNDED, "DIAGM is the plain kernel with a diagonal mass; bounds take the general variant");
    constexpr int NS = 4 * NT;
    extern __shared__ __attribute__((aligned(16))) double lds_P[];
    stage_precision<NT>(prm.P, prm.d, lds_P);
    double* lds_lb = lds_P + NT * NS * 64;                 // [16*NT] each, BOUNDED only
    double* lds_ub = lds_lb + 16 * NT;
    double* lds_ms = lds_ub + 16 * NT;                     // diag of chol(M) and of inv(M): a DIAGONAL precond_mat
    double* lds_mi = lds_ms + 16 * NT;                     // (hmc.cpp:57-59) is an element-wise scaling
    int* lds_bt = reinterpret_cast<int*>(lds_mi + 16 * NT);
    double* lds_Minv = lds_mi + 16 * NT + 8 * NT;          // after the int table (16*NT ints = 8*NT doubles), DENSE_M only
    double* lds_L = lds_Minv + NT * NS * 64;
    if constexpr (DENSE_M && !dense_m_from_global<NT>()) {
        stage_precision<NT>(prm.Minv, prm.d, lds_Minv);
        stage_precision<NT>(prm.Lchol, prm.d, lds_L);
    }
    if (BOUNDED) {
        for (int i = threadIdx.x; i < 16 * NT; i += blockDim.x) {
            const bool in = (uint32_t)i < prm.d;
            lds_lb[i] = in ? prm.lb[i] : 0.0;
            lds_ub[i] = in ? prm.ub[i] : 0.0;
            lds_bt[i] = in ? prm.btype[i] : 1;
            lds_ms[i] = in ? prm.m_sqrt[i] : 1.0;
            lds_mi[i] = in ? prm.m_inv[i] : 1.0;
        }
        __syncthreads();
    }
    if (DIAGM && !PCM) {
        for (int i = threadIdx.x; i < 16 * NT; i += blockDim.x) {
            const bool in = (uint32_t)i < prm.d;
            lds_ms[i] = in ? prm.m_sqrt[i] : 1.0;
            lds_mi[i] = in ? prm.m_inv[i] : 1.0;
        }
        __syncthreads();
    }

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane >> 4;
    // bit s: slice s (dims 4s .. 4s+3) has a bounded dimension (wave-uniform).  The transforms of a slice without one are the
    // identity (J = 1, no log-Jacobian term), and the branch around them is uniform -- what matters is the code that is NOT
    // fetched: the d = 128 general kernel is 1 MB of instructions (four transcendental-laden cases per slice and use), far beyond
    // the 64 KB instruction cache, and ran 2.9 times slower than the plain kernel with a diagonal precond_mat alone.
    uint32_t bslices = 0;
    if constexpr (BOUNDED) {
        if (prm.vals_bound)
            for (int s_ = 0; s_ < 4 * NT; ++s_) {
                const bool any = lds_bt[4 * s_] != 1 || lds_bt[4 * s_ + 1] != 1 || lds_bt[4 * s_ + 2] != 1 || lds_bt[4 * s_ + 3] != 1;
                bslices |= (any ? 1u : 0u) << s_;
            }
        bslices = (uint32_t)__builtin_amdgcn_readfirstlane((int)bslices);
    }
    auto slice_bounded = [&](int s) -> bool { return ((bslices >> s) & 1u) != 0u; };
    // DIAGM: this lane's column of the 1 / m table, re-derived where it is used -- as a loop invariant the compiler would keep all NS
    // entries in registers across the leapfrog loop (108 more spilled VGPRs at d = 128)
    [[maybe_unused]] auto mi_col = [&]() __attribute__((always_inline)) -> const double* {
        const double* p = lds_mi + (lane >> 4);
        asm volatile("" : "+v"(p));
        return p;
    };
    const uint64_t cl = ((uint64_t)blockIdx.x * WPB + wave) * 16 + (lane & 15);
    const bool live = cl < prm.C;
    const uint64_t cld = live ? cl : prm.C - 1;       // clamped index for loads
    // PCM: 1 / m and sqrt(m) of (slice s, this lane) from the chain's column; padding dimensions (>= d) read entry 0 and multiply zeros
    [[maybe_unused]] auto pcm_at = [&](const double* tab, int s) __attribute__((always_inline)) -> double {
        uint32_t dim = 4u * (uint32_t)s + (uint32_t)(lane >> 4);
        asm volatile("" : "+v"(dim));                   // (opaque: as loop invariants of the leapfrog loop the NS entries were kept in registers
                                                        //  the kernel does not have -- 276 spilled VGPRs, 980 B of scratch, 204 ms on configs[1]'s shape)
        return tab[(size_t)(dim < prm.d ? dim : 0u) * prm.C + cld];
    };
    auto minv_at = [&]([[maybe_unused]] const double* mic, int s) __attribute__((always_inline)) -> double {
        if constexpr (PCM) return pcm_at(prm.m_inv, s); else return mic[4 * s];
    };
    const uint64_t chain = prm.chain0 + cl;           // global chain id (Philox counter)
    const uint32_t d = prm.d;
    const uint64_t C = prm.C;
    const double eps = prm.eps;
    const double* afrag = lds_P + lane;
    // d > 64: prm.Minv / prm.Lchol are already in fragment order in global memory (host: pack_fragments)
    [[maybe_unused]] const double* afrag_minv = (DENSE_M && dense_m_from_global<NT>()) ? prm.Minv + lane : lds_Minv + lane;
    [[maybe_unused]] const double* afrag_l = (DENSE_M && dense_m_from_global<NT>()) ? prm.Lchol + lane : lds_L + lane;

    // Register-resident state of the wave's 16 chains: position, momentum, P*position.
    // The last accepted (theta, P*theta) lives in HBM (prm.theta / prm.wsave): written on accept,
    // re-read on reject, so a rejection costs 2 KiB of traffic per chain instead of 128 VGPRs.
    double th[NS], pm[NS], w[NS];
    double xs[BOUNDED ? NS : 1];   // BOUNDED: x = inv_transform(theta), where the target is evaluated (hmc.cpp:108)
    // addresses = wave-uniform row base (SGPR) + one per-lane element offset (VGPR)
    const size_t lane_off = (size_t)j * C + cld;
    // last accepted (theta, P*theta): wave-local contiguous [wave][2][NS][64 lanes] (512-B coalesced per slice)
    double* const ws_wave = prm.wsave + ((size_t)blockIdx.x * WPB + wave) * ((size_t)3 * NS * 64) + lane;
    // The base of every group of 8 slices is made opaque where it is used: the 2*NS addresses are loop invariants, and the
    // compiler otherwise hoists them out of the draw loop as 64-bit pairs, spills them, and serialises each workspace access
    // behind a scratch reload of its own address.  Inside a group the 512-byte slice stride is the instruction's immediate.
    auto ws_group = [&](int k) -> double* {
        double* b = ws_wave + (size_t)(k & ~7) * 64;
        asm volatile("" : "+v"(b));
        return b + (k & 7) * 64;
    };
    auto th_mem = [&](int s) -> double* { return ws_group(s); };
    auto w_mem = [&](int s) -> double* { return ws_group(NS + s); };
    [[maybe_unused]] auto z_mem = [&](int s) -> double* { return ws_wave + (size_t)(2 * NS + s) * 64; };   // fresh normals, staged

    // w = P * (theta or inv_transform(theta)); BOUNDED also refreshes xs and kw
    auto gradient = [&]() __attribute__((always_inline)) {
        if constexpr (BOUNDED) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int i = 4 * s + j;
                if (slice_bounded(s)) xs[s] = ((uint32_t)i < d) ? box_inv_transform(th[s], lds_bt[i], lds_lb[i], lds_ub[i]) : 0.0;
                else xs[s] = ((uint32_t)i < d) ? th[s] : 0.0;
            }
            target_times<NT>(afrag, prm.P, d, prm.sep_target != 0u, xs, w);
        } else {
            matvec_mfma<NT>(afrag, th, w);
        }
    };
    // BOUNDED: t = (eps * ([J] grad)) / 2 at the current (theta, w), grad = -w (hmc.cpp:122: jacob_matrix * grad_obj, a dense product).
    // A transient of the kick: nothing of it lives across the mat-vec.
    auto kick_terms = [&](double (&t)[BOUNDED ? NS : 1]) __attribute__((always_inline)) {
        if constexpr (BOUNDED) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int i = 4 * s + j;
                if (slice_bounded(s)) t[s] = box_inv_jacobian(th[s], lds_bt[i], lds_lb[i], lds_ub[i]) * w[s];   // J_ii * grad_i (gemv with a diagonal J)
                else t[s] = 1.0 * w[s];
            }
            if (prm.vals_bound) dense_product_poison<NS>(w, t, j, d);
#pragma unroll
            for (int s = 0; s < NS; ++s) t[s] = (eps * t[s]) / 2.0;
        }
    };
    // K = p . (Minv p) / 2 (hmc.cpp:160,184)
    auto kinetic = [&]() __attribute__((always_inline)) -> double {
        if constexpr (BOUNDED) {
            double mp[NS];                               // inv_precond_matrix * mntm, a dense product
            if constexpr (DENSE_M) {
                matvec_m2<NT>(afrag_minv, pm, mp);
            } else {
#pragma unroll
                for (int s = 0; s < NS; ++s) mp[s] = lds_mi[4 * s + j] * pm[s];
                dense_product_poison<NS>(pm, mp, j, d);
            }
            double q = 0.0;
#pragma unroll
            for (int s = 0; s < NS; ++s) q = dfma(pm[s], mp[s], q);
            q = q + __shfl_xor(q, 32);
            q = q + __shfl_xor(q, 16);
            return q / 2.0;
        } else if constexpr (DIAGM) {
            double q = 0.0;
            const double* mic = PCM ? nullptr : mi_col();
#pragma unroll
            for (int s = 0; s < NS; ++s) q = dfma(pm[s], minv_at(mic, s) * pm[s], q);
            q = q + __shfl_xor(q, 32);
            q = q + __shfl_xor(q, 16);
            return q / 2.0;
        } else {
            return dot4<NS>(pm, pm) / 2.0;
        }
    };
    // U = -box_log_kernel(theta) at the state whose w (and xs) are current (hmc.cpp:84-95,140,178)
    auto potential = [&]() __attribute__((always_inline)) -> double {
        if constexpr (BOUNDED) {
            const double kval = -0.5 * dot4<NS>(xs, w);
            double lj = 0.0;                             // log_jacobian.hpp:36-57: scalar loop, i ascending
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (!slice_bounded(s)) continue;
                const int i0 = 4 * s;
                const int bt = lds_bt[4 * s + j];
                const double term = box_log_jacobian_term(th[s], bt, lds_lb[4 * s + j], lds_ub[4 * s + j]);
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

#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const uint32_t dim = 4 * s + j;
        const double v = prm.theta[(size_t)(dim < d ? dim : 0u) * C + cld];   // clamped row: unconditional load
        if constexpr (BOUNDED) th[s] = (dim < d) ? box_transform(v, lds_bt[dim], lds_lb[dim], lds_ub[dim]) : 0.0;   // hmc.cpp:134-136
        else th[s] = (dim < d) ? v : 0.0;
    }
    gradient();
    if (live) {
#pragma unroll
        for (int s = 0; s < NS; ++s) { *th_mem(s) = th[s]; *w_mem(s) = w[s]; }
    }
    double prev_U = potential();                        // -box_log_kernel(first_draw), hmc.cpp:140
    uint64_t n_acc = 0;
    const uint32_t n_total = prm.n_burnin + prm.n_keep;
    // Non-finite regime: the reference's `inv_precond_matrix * mntm` is a dense product, in which one
    // non-finite momentum entry turns every other row into NaN.  The plain kernel applies the identity element-wise; what it
    // does is DETECT the regime -- a non-finite entry of p or theta makes prop_U or prop_K non-finite, and stays -- and hand the
    // chain to the literal replay (literal.hpp) through prm.nf_flag.
    [[maybe_unused]] bool nf_seen = false;

    // Two waves share each SIMD's matrix pipe.  Started together they stay in lock-step and their
    // VALU phases (RNG, accept, kick/drift) coincide; a one-off start offset is self-preserving under
    // round-robin MFMA issue, so one wave's VALU work then always sits under the other's MFMAs.
    if (WPB > 4 && wave >= 4)
        for (uint32_t k = 0; k < prm.stagger; ++k) __builtin_amdgcn_s_sleep(127);

    const unsigned long long t_loop0 = clock64();
#pragma unroll 1
    for (uint32_t draw = 0; draw < n_total; ++draw) {
#if MI_HMC_RNG_STAGED
        // momentum ~ N(0, I): hmc.cpp:156-158.  The 16 Philox blocks + Box-Muller pairs run as a ROLLED loop that
        // stages the normals through the wave-local workspace (16 KiB, cache-resident): few live registers next
        // to the 128 VGPRs of theta / P*theta, no spills, small code; then one burst of 32 coalesced loads.
#pragma unroll 1
        for (int b = 0; b < NS / 2; ++b) {
            double z0, z1;
            if (prm.ablate & 4u) { z0 = 0.25; z1 = -0.5; }               // profiling: no RNG
            else rng_normal_pair_at(prm.seed, chain, draw + prm.draw0, (uint32_t)(4 * b), (uint32_t)j, STREAM_NORMAL, z0, z1);
            *z_mem(2 * b) = (8u * b + j < d) ? z0 : 0.0;
            *z_mem(2 * b + 1) = (8u * b + 4 + j < d) ? z1 : 0.0;
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) pm[s] = *z_mem(s);
        if constexpr (DENSE_M) {                        // p = L z (:158)
            double zz[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) zz[s] = pm[s];
            matvec_m2<NT>(afrag_l, zz, pm);
        } else if constexpr (BOUNDED || DIAGM) {        // p = L z with a diagonal L (:158)
#pragma unroll
            for (int s = 0; s < NS; ++s) pm[s] = lds_ms[4 * s + j] * pm[s];
        }
#else
        // momentum ~ N(0, I): hmc.cpp:156-158, fully unrolled into the momentum registers
#pragma unroll
        for (int b = 0; b < NS / 2; ++b) {
            double z0, z1;
            if (prm.ablate & 4u) { z0 = 0.25; z1 = -0.5; }               // profiling: no RNG
            else rng_normal_pair_at(prm.seed, chain, draw + prm.draw0, (uint32_t)(4 * b), (uint32_t)j, STREAM_NORMAL, z0, z1);
            pm[2 * b] = (8u * b + j < d) ? z0 : 0.0;
            pm[2 * b + 1] = (8u * b + 4 + j < d) ? z1 : 0.0;
            if constexpr (DIAGM && PCM) {               // p = L z, L = diag(sqrt(mass[:, c])) of this chain
                pm[2 * b] = pcm_at(prm.m_sqrt, 2 * b) * pm[2 * b];
                pm[2 * b + 1] = pcm_at(prm.m_sqrt, 2 * b + 1) * pm[2 * b + 1];
            } else if constexpr (DIAGM) {               // p = L z with a diagonal L (:158); the table entry is read where it is used
                const double* msc = lds_ms + (lane >> 4);
                asm volatile("" : "+v"(msc));
                pm[2 * b] = msc[8 * b] * pm[2 * b];
                pm[2 * b + 1] = msc[8 * b + 4] * pm[2 * b + 1];
            } else if constexpr (BOUNDED && !DENSE_M) { // p = L z with a diagonal L (:158)
                pm[2 * b] = lds_ms[8 * b + j] * pm[2 * b];
                pm[2 * b + 1] = lds_ms[8 * b + 4 + j] * pm[2 * b + 1];
            }
            if (BOUNDED || MI_HMC_RNG_PAIRS == 1 || (b & 1)) __builtin_amdgcn_sched_barrier(0);   // plain kernel: two independent Philox / Box-Muller chains interleave
        }
        if constexpr (DENSE_M) {                        // p = L z (:158)
            double zz[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) zz[s] = pm[s];
            matvec_m2<NT>(afrag_l, zz, pm);
        }
#endif
        const double prev_K = kinetic();                // hmc.cpp:160

        if constexpr (!BOUNDED) {
            // hmc.cpp:164-176, grad = -w.  The second half-step of step k and the first half-step of step k+1 use the same
            // gradient, hence the same (eps*w)/2: it is formed once and subtracted twice (two roundings, as the reference's two
            // statements), 6 instead of 8 VALU operations per element and step.
            const uint32_t L = prm.n_leap_steps;
            if (L > 0 && (prm.ablate & 3u) != 1u) {
                [[maybe_unused]] const double* mic = (DIAGM && !PCM) ? mi_col() : nullptr;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    pm[s] = pm[s] - (eps * w[s]) / 2.0;                     // first half-step of step 0 (:167,126)
                    if constexpr (DIAGM) th[s] = th[s] + eps * (minv_at(mic, s) * pm[s]);   // (:171) theta += eps Minv p
                    else th[s] = th[s] + eps * pm[s];                       // (:171)
                }
            }
#pragma unroll 1
            for (uint32_t k = 0; k + 1 < L; ++k) {
                if ((prm.ablate & 3u) != 2u) gradient();
                if ((prm.ablate & 3u) != 1u) {
                    [[maybe_unused]] const double* mic = (DIAGM && !PCM) ? mi_col() : nullptr;
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const double t = (eps * w[s]) / 2.0;
                        pm[s] = pm[s] - t;                                  // second half-step of step k (:175)
                        pm[s] = pm[s] - t;                                  // first half-step of step k+1 (:167)
                        if constexpr (DIAGM) th[s] = th[s] + eps * (minv_at(mic, s) * pm[s]);
                        else th[s] = th[s] + eps * pm[s];                   // (:171)
                    }
                }
            }
            if (L > 0) {
                if ((prm.ablate & 3u) != 2u) gradient();
                if ((prm.ablate & 3u) != 1u) {
#pragma unroll
                    for (int s = 0; s < NS; ++s) pm[s] = pm[s] - (eps * w[s]) / 2.0;       // second half-step of the last step
                }
            }
        } else {
            // The same loop shape for the general variant: the second half-step of step k and the first of step k+1 are at the
            // same (theta, w), so (eps * [J] grad) / 2 is formed once and subtracted twice (the reference's two roundings), and
            // neither it nor Minv p lives across the mat-vec (as loop-carried arrays they spilled: 290 ms for configs[1]'s shape).
            const uint32_t L = prm.n_leap_steps;
            auto drift_step = [&]() __attribute__((always_inline)) {           // theta += eps * Minv p (:171)
                double mp[NS];
                if constexpr (DENSE_M) {
                    matvec_m2<NT>(afrag_minv, pm, mp);                    // inv_precond_matrix * new_mntm
                } else {
#pragma unroll
                    for (int s = 0; s < NS; ++s) mp[s] = lds_mi[4 * s + j] * pm[s];
                    dense_product_poison<NS>(pm, mp, j, d);
                }
#pragma unroll
                for (int s = 0; s < NS; ++s) th[s] = th[s] + eps * mp[s];
            };
            if (L > 0) {
                double t[NS];
                kick_terms(t);
#pragma unroll
                for (int s = 0; s < NS; ++s) pm[s] = pm[s] - t[s];          // first half-step of step 0 (:122,167)
                drift_step();
            }
#pragma unroll 1
            for (uint32_t k = 0; k + 1 < L; ++k) {
                gradient();
                double t[NS];
                kick_terms(t);
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    pm[s] = pm[s] - t[s];                                   // second half-step of step k (:175)
                    pm[s] = pm[s] - t[s];                                   // first half-step of step k+1 (:167)
                }
                drift_step();
            }
            if (L > 0) {
                gradient();
                double t[NS];
                kick_terms(t);
#pragma unroll
                for (int s = 0; s < NS; ++s) pm[s] = pm[s] - t[s];          // second half-step of the last step
            }
        }
        if constexpr (BOUNDED) { if (prm.n_leap_steps == 0) gradient(); }   // xs must match theta for the energy

        double prop_U = potential();                    // -box_log_kernel(new_draw), hmc.cpp:178
        const bool u_nf = !is_finite(prop_U);
        if (u_nf) prop_U = INF;                         // :180-182
        const double prop_K = kinetic();                // :184
        if constexpr (!BOUNDED) nf_seen |= u_nf | !is_finite(prop_K);
        const double x = -(prop_U + prop_K) + (prev_U + prev_K);
        const double comp_val = (x < 0.01) ? x : 0.01;  // std::min(0.01, x), :188
        const double z = rng_uniform(prm.seed, chain, draw + prm.draw0, 0u);   // :189
        const bool accept = z < det_exp(comp_val);      // :191
#ifndef MI_HMC_NO_WSAVE
#define MI_HMC_NO_WSAVE 0      // 1 (plain kernel only): P*theta of the accepted state is not saved; a wave with a rejecting chain recomputes it
#endif
        if constexpr (MI_HMC_NO_WSAVE && !BOUNDED) {
            if (accept) {
                prev_U = prop_U;
                if (live && !(prm.ablate & 8u)) {
#pragma unroll
                    for (int s = 0; s < NS; ++s) *th_mem(s) = th[s];
                }
            }
            if (__ballot(!accept) != 0ull) {            // some chain of the wave keeps prev_draw: its theta comes back, P*theta is recomputed for
                if (!accept) {                          // the whole tile (the accepted chains get the bits they already hold)
#pragma unroll
                    for (int s = 0; s < NS; ++s) th[s] = *th_mem(s);
                }
                gradient();
            }
        } else
        if (accept) {                                   // prev_draw = new_draw (:192-194)
            prev_U = prop_U;
            if (live && !(prm.ablate & 8u)) {
#pragma unroll
                for (int s = 0; s < NS; ++s) { *th_mem(s) = th[s]; *w_mem(s) = w[s]; }
            }
        } else {                                        // keep prev_draw: reload it (padded rows: no predicates)
#pragma unroll
            for (int s = 0; s < NS; ++s) { th[s] = *th_mem(s); w[s] = *w_mem(s); }
        }
        if (draw >= prm.n_burnin) {                     // :196-204
            n_acc += accept ? 1u : 0u;
            if (prm.draws != nullptr && live && !(prm.ablate & 16u)) {
                double* out = prm.draws + (size_t)(draw - prm.n_burnin) * d * C;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const uint32_t dim = 4 * s + j;
                    // bounded runs store inv_transform(row) (the reference does it in the epilogue, :211-218)
                    if (dim < d) (out + (size_t)(4 * s) * C)[lane_off] =
                        (BOUNDED && slice_bounded(s)) ? box_inv_transform(th[s], lds_bt[dim], lds_lb[dim], lds_ub[dim]) : th[s];
                }
            }
        }
    }

    bool replay = false;                                 // the literal kernel owns this chain's outputs
    if constexpr (!BOUNDED) replay = nf_seen && prm.nf_flag != nullptr;
    if (live && replay && j == 0) { prm.nf_flag[cl] = 1u; prm.nf_flag[C] = 1u; }
    if (live && !replay) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const uint32_t dim = 4 * s + j;
            if (dim < d) prm.theta[(size_t)dim * C + cl] =
                (BOUNDED && slice_bounded(s)) ? box_inv_transform(th[s], lds_bt[dim], lds_lb[dim], lds_ub[dim]) : th[s];
        }
    }
    if (live && !replay && j == 0) {
        if (prm.n_accept) prm.n_accept[cl] = n_acc;                            // hmc.cpp:220-222
        if (prm.n_leap) prm.n_leap[cl] = (prm.ablate & 32u) ? (uint64_t)(clock64() - t_loop0)      // profiling: shader cycles of the draw loop
                                                            : (uint64_t)n_total * prm.n_leap_steps;
    }
}

}  // namespace mi
