// gemm_samplers.hip -- hmc / mala / rwmh for dense-gradient Gaussian targets BEYOND d = 512: one fp64 matrix product per gradient, for ALL chains at once.
//
// Replaces the draw loops of mcmc::internal::hmc_impl (/root/reference/src/hmc.cpp:155-205 with the leapfrog :164-176), mala_impl
// (src/mala.cpp:149-186 with mala_mean_fn :97-125, mala_prop_adjustment include/mcmc/mala.ipp:30-70, dmvnorm include/stats/dmvnorm.hpp:28-54) and
// rwmh_impl (src/rwmh.cpp:123-151) for log K(theta) = -1/2 theta' P theta, identity precond_mat / cov_mat, no bounds, where n_vals is past what the
// register- and LDS-resident kernels hold (hmc_dense.hpp: d <= 128; logistic_lds.hpp: d <= 512) and literal.hpp served at ~1 % of the matrix peak.
//
// Why a matrix product.  The samplers above are lock-step in the chain index: every chain of a draw does the same number of gradient evaluations,
// so the gradients of all C chains at one leapfrog step are W = P Theta with Theta the d x C matrix of positions -- a DGEMM of 2 d^2 C flop over
// 3 d C doubles of state: d / 12 flop per byte, compute-bound from d ~ 150 on.  Beyond d = 512 the state of a 16-chain tile no longer fits a
// workgroup's registers, and with 288 GB of HBM it does not have to: Theta, the momenta and the gradients live in HBM as [dimension][chain]
// (chains contiguous -- the layout of mi_chains.theta, and the row-major B operand of the product), and one launch per leapfrog step streams them once.
//
//   gemm_step_kernel<MODE>: 128 x 128 output tile per workgroup of four waves (64 x 64 per wave: 16 accumulators of v_mfma_f64_16x16x4_f64), K in steps
//   of 16 through a double-buffered LDS stage that the direct-to-LDS loads (global_load_lds_dwordx4) fill one step ahead -- P is packed TRANSPOSED
//   and zero-padded once per run so that A and B tiles are both rows of 128 contiguous doubles; the LDS row stride of 144 doubles puts the two
//   16-lane groups of a ds_read_b64 on disjoint bank halves.  Element (i, c) of W is ONE fma chain over k ascending (the k-steps of an MFMA and the
//   K loop both ascend): the order of the oracle's orc_gemv and of every other dense kernel of this engine.  The D layout of the instruction (row 4 r + lane / 16)
//   is the B layout, so the epilogue owns whole (dimension, chain) elements and applies, element-wise and with the reference's roundings,
//       MODE 0 (a leapfrog step that is not the last): p += (eps g) / 2 (:175), p += (eps g) / 2 (:126 of the NEXT step -- same position, same gradient),
//               theta' = theta + eps p (:171) into the OTHER position buffer (other workgroups still read this one as their B operand);
//       MODE 1 (the last step): p += (eps g) / 2, W kept for the accept step and as the next draw's first gradient;
//       MODE 2 (mala / rwmh / the initial evaluation): W only.
//   blockIdx -> tile: XCD-aware -- the row tiles of one chain tile run back to back on ONE XCD, so Theta's tile is read from HBM once and shared in that L2.
//
//   Per draw, next to the n_leap products: gemm_momentum_kernel (Philox + Box-Muller, one slot per thread, canonical slot <-> dimension map of
//   det_math.hpp), gemm_pre_kernel (K = p.p / 2 in the engine's four-strided order, first half-kick, first drift) and gemm_post_kernel (energies,
//   accept / reject :186-204, the accepted state and the kept row) -- element-wise or one fma chain per (chain, dimension class), HBM-bound, ~5 % of a draw at d = 1024.
//   The launches of one draw are captured ONCE into a hipGraph and replayed n_draws times: the draw index lives in device memory (gemm_advance_kernel).
//
// Reduction orders (the oracle: W = 4, one block; literal.hpp: lit_orders beyond d = 512): dot products as four strided fma chains over dimensions
// j, j + 4, ..., combined (q0 + q2) + (q1 + q3).  Non-finite regime: the element-wise kicks are the reference's dense `inv_precond_matrix * mntm`
// only while everything is finite; a chain whose energies (hmc) / proposal densities (mala) go non-finite is flagged and replayed by literal.hpp.

#include "gemm_samplers.hpp"
#include "det_math.hpp"

#include <algorithm>

namespace mi {
namespace gemm {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int TM = 128, TN = 128, TK = 16;
constexpr int LDS_STRIDE = 144;                      // doubles per staged row: 128 + 16 (k rows j and j + 1 of a fragment read sit 32 banks apart)
constexpr int STAGE = 2 * TK * LDS_STRIDE;           // doubles per stage: 16 rows of the A tile, 16 rows of the B tile
constexpr size_t GEMM_LDS_BYTES = (size_t)2 * STAGE * sizeof(double);

struct StepParams {
    const double* Pt;        // [dK][dM]: Pt[k dM + i] = P[i][k], zero padded
    const double* th_in;     // [dK][Cp]
    double* th_out;          // MODE 0
    double* pm;              // MODE 0 / 1
    double* w_out;           // MODE 1 / 2
    uint32_t dK, dM, n_ntiles;
    uint64_t Cp;
    double eps;
};

template <int MODE>
__global__ MI_NO_DS_MERGE __launch_bounds__(256, 2) void gemm_step_kernel(const StepParams prm)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane >> 4, c16 = lane & 15;
    // consecutive workgroup ids go round the 8 XCDs: XCD x takes chain tiles x, x + 8, ... and runs their row tiles back to back
    const uint32_t MT = prm.dM / TM;
    const uint32_t xcd = blockIdx.x & 7u, q = blockIdx.x >> 3;
    const uint32_t nt = xcd + 8u * (q / MT), mt = q % MT;
    if (nt >= prm.n_ntiles) return;
    const size_t m0 = (size_t)mt * TM, n0 = (size_t)nt * TN;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)lds;
    const uint32_t lane16 = (uint32_t)lane * 16u;
    // wave w moves rows w, w + 4, ... of the 32 staged rows (0..15: P^T rows k, columns m0..; 16..31: Theta rows k, chains n0..), 1 KiB each
    auto issue = [&](uint32_t kb, int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = wave + 4 * i;
            const int rr = r & 15;
            const double* src = (i >= 4) ? prm.th_in + ((size_t)(kb * TK + rr) * prm.Cp + n0) : prm.Pt + ((size_t)(kb * TK + rr) * prm.dM + m0);
            const uint32_t dst = lds_base + (uint32_t)((stage * STAGE + r * LDS_STRIDE) * 8);
            uint32_t m0_saved;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(m0_saved) : "v"(lane16), "s"(dst), "s"(src) : "memory");
        }
    };
    double4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};
    const int wm = wave >> 1, wn = wave & 1;             // the wave's 64 x 64 quarter of the tile
    const uint32_t nkb = prm.dK / TK;
    issue(0, 0);
#pragma unroll 1
    for (uint32_t kb = 0; kb < nkb; ++kb) {
        const int stage = (int)(kb & 1u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's rows of step kb landed ...
        __syncthreads();                                      // ... everybody's, and nobody still reads the other stage
        if (kb + 1 < nkb) issue(kb + 1, stage ^ 1);
        const double* As = lds + stage * STAGE + j * LDS_STRIDE + wm * 64 + c16;
        const double* Bs = lds + stage * STAGE + (TK + j) * LDS_STRIDE + wn * 64 + c16;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { a[t] = As[4 * kk * LDS_STRIDE + 16 * t]; b[t] = Bs[4 * kk * LDS_STRIDE + 16 * t]; }
#pragma unroll
            for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[ti][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], b[ni], acc[ti][ni], 0, 0, 0);
        }
    }
    // epilogue: acc[ti][ni][r] is W at dimension m0 + 64 wm + 16 ti + 4 r + j, chain n0 + 64 wn + 16 ni + c16
    const double eps = prm.eps;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t row = m0 + (size_t)(64 * wm + 16 * ti + 4 * r + j);
            if (row < prm.dK) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const size_t idx = row * prm.Cp + n0 + (size_t)(64 * wn + 16 * ni + c16);
                    const double w = acc[ti][ni][r];
                    if constexpr (MODE == 2) prm.w_out[idx] = w;
                    else {
                        const double g = -w;                                   // grad log K = -(P theta)
                        double p = prm.pm[idx];
                        p = p + (eps * g) / 2.0;                               // second half-step of this leapfrog step (hmc.cpp:175)
                        if constexpr (MODE == 1) { prm.pm[idx] = p; prm.w_out[idx] = w; }
                        else {
                            p = p + (eps * g) / 2.0;                           // first half-step of the next one (:126): same position, same gradient
                            prm.pm[idx] = p;
                            prm.th_out[idx] = prm.th_in[idx] + eps * p;        // :171
                        }
                    }
                }
            }
        }
}

// P (d x d row-major) -> Pt[k dM + i] = P[i][k], zeros outside
__global__ void gemm_pack_kernel(const double* __restrict__ P, uint32_t d, uint32_t dK, uint32_t dM, double* __restrict__ Pt)
{
    const size_t n = (size_t)dK * dM;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const uint32_t k = (uint32_t)(e / dM), i = (uint32_t)(e % dM);
        Pt[e] = (k < d && i < d) ? P[(size_t)i * d + k] : 0.0;
    }
}

struct DrawParams {
    int algo;
    uint32_t d, dK;
    uint64_t C, Cp, chain0;
    double* th;              // [dK][Cp] accepted position
    double* wacc;            // P th at the accepted position
    double* thw;             // the proposal (hmc: the leapfrog's end point)
    double* wprop;           // P thw
    double* pm;              // hmc: momentum
    double* prevE;           // [Cp] hmc: prev_U; mala / rwmh: prev_LP
    double* kprev;           // [Cp] hmc: prev_K of the running draw
    uint64_t* nacc;          // [Cp]
    uint32_t* draw_ctr;      // LOCAL index of the running draw
    const double* theta_in;  // [d][C]
    double* theta_out;
    double* draws;           // [n_keep][d][C]
    uint64_t* n_accept;
    uint32_t* nf_flag;
    uint64_t seed;
    uint32_t n_burnin, draw0;
    double eps, s2, rs, log_det, cons_term;
};

// theta ([d][C]) into the padded state, zeros elsewhere
__global__ void gemm_load_kernel(const DrawParams prm)
{
    const size_t n = (size_t)prm.dK * prm.Cp;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t i = e / prm.Cp, c = e % prm.Cp;
        prm.th[e] = (i < prm.d && c < prm.C) ? prm.theta_in[i * prm.C + c] : 0.0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *prm.draw_ctr = 0u;
}

// the normals of a draw (hmc.cpp:156, mala.cpp:150, rwmh.cpp:124): one Philox slot -- two dimensions, i = 8 b + 4 h + j <-> slot 4 b + j, component h -- per thread.
// hmc: p = z (:158, identity).  mala: proposal = mala_mean_fn(prev) + eps z (mala.cpp:123,159).  rwmh: proposal = prev + par_scale z (rwmh.cpp:126).
__global__ __launch_bounds__(256) void gemm_normals_kernel(const DrawParams prm)
{
    const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= prm.Cp) return;
    const uint32_t slot = blockIdx.y;
    const uint32_t da = 8u * (slot >> 2) + (slot & 3u), db = da + 4u;
    const uint32_t draw = *prm.draw_ctr;
    double z0 = 0.0, z1 = 0.0;
    if (c < prm.C && da < prm.d) rng_normal_pair(prm.seed, prm.chain0 + c, draw + prm.draw0, slot, STREAM_NORMAL, z0, z1);
    if (db >= prm.d) z1 = 0.0;
    const size_t ia = (size_t)da * prm.Cp + c, ib = (size_t)db * prm.Cp + c;
    if (prm.algo == GEMM_HMC) { prm.pm[ia] = z0; prm.pm[ib] = z1; }
    else if (prm.algo == GEMM_MALA) {
        prm.thw[ia] = (prm.th[ia] + (prm.s2 * -prm.wacc[ia]) / 2.0) + prm.eps * z0;
        prm.thw[ib] = (prm.th[ib] + (prm.s2 * -prm.wacc[ib]) / 2.0) + prm.eps * z1;
    } else {
        prm.thw[ia] = prm.th[ia] + prm.eps * z0;
        prm.thw[ib] = prm.th[ib] + prm.eps * z1;
    }
}

// One thread per (chain, dimension class j = dim mod 4): the engine's dot products are four strided fma chains, combined (q0 + q2) + (q1 + q3); a wave
// holds 16 chains x 4 classes (the MFMA B layout: its loads are the epilogue's 128-byte segments).
__device__ __forceinline__ double class_sum(double q)
{
    q = q + __shfl_xor(q, 32);
    q = q + __shfl_xor(q, 16);
    return q;
}

// hmc: prev_K = p.p / 2 (hmc.cpp:160), the first half-step (:126) and the first drift (:171) of the draw; new_draw = prev_draw (:162)
__global__ __launch_bounds__(256) void gemm_pre_kernel(const DrawParams prm)
{
    const int lane = threadIdx.x & 63, j = lane >> 4;
    const uint64_t c = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (lane & 15);
    double q = 0.0;
#pragma unroll 4
    for (uint32_t i = (uint32_t)j; i < prm.dK; i += 4u) {
        const size_t e = (size_t)i * prm.Cp + c;
        double p = prm.pm[e];
        q = dfma(p, p, q);
        const double g = -prm.wacc[e];
        p = p + (prm.eps * g) / 2.0;
        prm.pm[e] = p;
        prm.thw[e] = prm.th[e] + prm.eps * p;
    }
    q = class_sum(q);
    if (j == 0) prm.kprev[c] = q / 2.0;
}

// the value at the initial state (hmc.cpp:140, mala.cpp:138, rwmh.cpp:113): log K = -1/2 theta . (P theta)
__global__ __launch_bounds__(256) void gemm_first_kernel(const DrawParams prm)
{
    const int lane = threadIdx.x & 63, j = lane >> 4;
    const uint64_t c = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (lane & 15);
    double q = 0.0;
#pragma unroll 4
    for (uint32_t i = (uint32_t)j; i < prm.dK; i += 4u) {
        const size_t e = (size_t)i * prm.Cp + c;
        q = dfma(prm.th[e], prm.wacc[e], q);
    }
    q = class_sum(q);
    const double first_lp = -0.5 * q;
    if (j == 0) { prm.prevE[c] = (prm.algo == GEMM_HMC) ? -first_lp : first_lp; prm.nacc[c] = 0ull; }
}

// the accept step (hmc.cpp:178-204; mala.cpp:162-184 with mala.ipp:59-64 and dmvnorm.hpp:37-41; rwmh.cpp:128-149), the accepted state and the kept row
template <int ALGO>
__global__ __launch_bounds__(256) void gemm_post_kernel(const DrawParams prm)
{
    const int lane = threadIdx.x & 63, j = lane >> 4;
    const uint64_t c = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (lane & 15);
    const bool live = c < prm.C;
    const uint32_t draw = *prm.draw_ctr;
    double qv = 0.0, qk = 0.0, qa = 0.0, qb = 0.0;
#pragma unroll 4
    for (uint32_t i = (uint32_t)j; i < prm.dK; i += 4u) {
        const size_t e = (size_t)i * prm.Cp + c;
        const double x = prm.thw[e], w = prm.wprop[e];
        qv = dfma(x, w, qv);
        if constexpr (ALGO == GEMM_HMC) { const double p = prm.pm[e]; qk = dfma(p, p, qk); }
        if constexpr (ALGO == GEMM_MALA) {
            const double be = prm.th[e], gr = -prm.wacc[e], gp = -w;
            const double mean_prop = x + (prm.s2 * gp) / 2.0;
            const double xa = be - mean_prop;                      // dmvnorm.hpp:37
            qa = dfma(xa, prm.rs * xa, qa);
            const double mean_prev = be + (prm.s2 * gr) / 2.0;
            const double xb = x - mean_prev;
            qb = dfma(xb, prm.rs * xb, qb);
        }
    }
    qv = class_sum(qv);
    const double lp = -0.5 * qv;
    const double prevE = prm.prevE[c];
    const double z = rng_uniform(prm.seed, prm.chain0 + (live ? c : 0), draw + prm.draw0, 0u);
    bool accept, flag = false;
    double newE;
    if constexpr (ALGO == GEMM_HMC) {
        qk = class_sum(qk);
        const double prop_K = qk / 2.0;                            // :184
        double prop_U = -lp;                                       // :178
        const bool u_nf = !is_finite(prop_U);
        if (u_nf) prop_U = INF;                                    // :180-182
        flag = u_nf || !is_finite(prop_K);
        const double x = -(prop_U + prop_K) + (prevE + prm.kprev[c]);
        const double comp_val = (x < 0.01) ? x : 0.01;             // :188
        accept = z < det_exp(comp_val);                            // :191
        newE = prop_U;
    } else if constexpr (ALGO == GEMM_MALA) {
        qa = class_sum(qa); qb = class_sum(qb);
        double pl = lp;
        if (!is_finite(pl)) pl = -INF;                             // mala.cpp:164-166
        const double da = prm.cons_term - 0.5 * (prm.log_det + qa);    // dmvnorm.hpp:41
        const double db = prm.cons_term - 0.5 * (prm.log_det + qb);
        flag = !is_finite(da) || !is_finite(db);
        const double x = pl - prevE + (da - db);
        const double comp_val = (x < 0.01) ? x : 0.01;             // mala.cpp:170
        accept = z < det_exp(comp_val);                            // :173
        newE = pl;
    } else {
        double pl = lp;
        if (!is_finite(pl)) pl = -INF;                             // rwmh.cpp:130-132
        const double x = pl - prevE;
        const double comp_val = (x < 0.0) ? x : 0.0;               // :136
        accept = z < det_exp(comp_val);                            // :139
        newE = pl;
    }
    const bool kept = draw >= prm.n_burnin;
    if (j == 0) {
        if (accept) prm.prevE[c] = newE;
        if (accept && kept) prm.nacc[c] += 1ull;
        if (flag && live && prm.nf_flag) { prm.nf_flag[c] = 1u; prm.nf_flag[prm.C] = 1u; }
    }
    double* out = (kept && prm.draws != nullptr && live) ? prm.draws + (size_t)(draw - prm.n_burnin) * prm.d * prm.C + c : nullptr;
#pragma unroll 4
    for (uint32_t i = (uint32_t)j; i < prm.dK; i += 4u) {
        const size_t e = (size_t)i * prm.Cp + c;
        double v;
        if (accept) { v = prm.thw[e]; prm.th[e] = v; prm.wacc[e] = prm.wprop[e]; }
        else v = prm.th[e];
        if (out != nullptr && i < prm.d) out[(size_t)i * prm.C] = v;
    }
}

__global__ void gemm_advance_kernel(uint32_t* draw_ctr) { *draw_ctr += 1u; }

// final state and accept counts of the chains that were not flagged (a flagged chain is replayed from theta, which must stay its initial state)
__global__ void gemm_store_kernel(const DrawParams prm)
{
    const size_t n = (size_t)prm.d * prm.C;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t i = e / prm.C, c = e % prm.C;
        const bool flagged = prm.nf_flag != nullptr && prm.nf_flag[c] != 0u;
        if (!flagged) {
            prm.theta_out[e] = prm.th[i * prm.Cp + c];
            if (i == 0 && prm.n_accept) prm.n_accept[c] = prm.nacc[c];
        }
    }
}

static inline uint32_t round_up(uint32_t x, uint32_t m) { return (x + m - 1) / m * m; }

struct Layout {
    uint32_t dK, dM;
    uint64_t Cp;
    size_t vec;            // doubles per state array
    size_t n_doubles;
};
static Layout layout_of(uint32_t d, uint64_t C)
{
    Layout l;
    l.dK = round_up(d, TK); l.dM = round_up(d, TM);
    l.Cp = (C + TN - 1) / TN * TN;
    l.vec = (size_t)l.dK * l.Cp;
    // Pt | th, wacc, thw0, thw1, wprop, pm | prevE, kprev, nacc | draw counter
    l.n_doubles = (size_t)l.dK * l.dM + 6 * l.vec + 3 * l.Cp + 32;
    return l;
}
size_t gemm_ws_bytes(uint32_t d, uint64_t C) { return layout_of(d, C).n_doubles * sizeof(double); }

#define GEMM_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

template <int MODE>
static int step_attr()      // (73 728 bytes of dynamic LDS: above the 64 KiB default)
{
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_step_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_LDS_BYTES);
}
template <int MODE>
static int launch_step(const StepParams& sp, hipStream_t st)
{
    const uint32_t MT = sp.dM / TM;
    const uint32_t grid = 8u * MT * ((sp.n_ntiles + 7u) / 8u);
    hipLaunchKernelGGL(gemm_step_kernel<MODE>, dim3(grid), dim3(256), GEMM_LDS_BYTES, st, sp);
    return (int)hipGetLastError();
}

int gemm_run(const GemmRun& r, hipStream_t st, const char** kernel_name)
{
    const Layout l = layout_of(r.d, r.C);
    double* base = static_cast<double*>(r.ws);
    double* Pt = base;
    double* th = Pt + (size_t)l.dK * l.dM;
    double* wacc = th + l.vec;
    double* thw[2] = {wacc + l.vec, wacc + 2 * l.vec};
    double* wprop = wacc + 3 * l.vec;
    double* pm = wacc + 4 * l.vec;
    double* prevE = pm + l.vec;
    double* kprev = prevE + l.Cp;
    uint64_t* nacc = reinterpret_cast<uint64_t*>(kprev + l.Cp);
    uint32_t* draw_ctr = reinterpret_cast<uint32_t*>(kprev + 2 * l.Cp);

    DrawParams dp{};
    dp.algo = r.algo; dp.d = r.d; dp.dK = l.dK; dp.C = r.C; dp.Cp = l.Cp; dp.chain0 = r.chain0;
    dp.th = th; dp.wacc = wacc; dp.thw = thw[0]; dp.wprop = wprop; dp.pm = pm; dp.prevE = prevE; dp.kprev = kprev; dp.nacc = nacc; dp.draw_ctr = draw_ctr;
    dp.theta_in = r.theta; dp.theta_out = r.theta; dp.draws = r.draws; dp.n_accept = r.n_accept; dp.nf_flag = r.nf_flag;
    dp.seed = r.seed; dp.n_burnin = r.n_burnin; dp.draw0 = r.draw0;
    dp.eps = r.eps; dp.s2 = r.s2; dp.rs = r.rs; dp.log_det = r.log_det; dp.cons_term = r.cons_term;

    StepParams sp{};
    sp.Pt = Pt; sp.dK = l.dK; sp.dM = l.dM; sp.n_ntiles = (uint32_t)(l.Cp / TN); sp.Cp = l.Cp; sp.eps = r.eps; sp.pm = pm;

    static const int attr_rc = [] { int e = step_attr<0>(); if (!e) e = step_attr<1>(); if (!e) e = step_attr<2>(); return e; }();
    if (attr_rc) return attr_rc;
    const unsigned ew_grid = (unsigned)std::min<size_t>((l.vec + 255) / 256, 65535);
    const unsigned cls_grid = (unsigned)(l.Cp / 64);                 // 4 waves x 16 chains per workgroup
    hipLaunchKernelGGL(gemm_pack_kernel, dim3(std::min<unsigned>((unsigned)(((size_t)l.dK * l.dM + 255) / 256), 65535u)), dim3(256), 0, st, r.P, r.d, l.dK, l.dM, Pt);
    hipLaunchKernelGGL(gemm_load_kernel, dim3(ew_grid), dim3(256), 0, st, dp);
    {   // the evaluation at the initial values
        StepParams s0 = sp; s0.th_in = th; s0.w_out = wacc;
        if (int e = launch_step<2>(s0, st)) return e;
    }
    hipLaunchKernelGGL(gemm_first_kernel, dim3(cls_grid), dim3(256), 0, st, dp);
    GEMM_TRY(hipGetLastError());

    const uint32_t n_total = r.n_burnin + r.n_keep;
    const uint32_t L = r.n_leap;
    // the launches of ONE draw
    auto enqueue_draw = [&](hipStream_t s) -> int {
        hipLaunchKernelGGL(gemm_normals_kernel, dim3((unsigned)(l.Cp / 256 + (l.Cp % 256 ? 1 : 0)), l.dK / 2), dim3(256), 0, s, dp);
        DrawParams pp = dp;
        if (r.algo == GEMM_HMC) {
            hipLaunchKernelGGL(gemm_pre_kernel, dim3(cls_grid), dim3(256), 0, s, dp);
            for (uint32_t k = 0; k < L; ++k) {
                StepParams sk = sp;
                sk.th_in = thw[k & 1u]; sk.th_out = thw[(k + 1u) & 1u]; sk.w_out = wprop;
                const int e = (k + 1 < L) ? launch_step<0>(sk, s) : launch_step<1>(sk, s);
                if (e) return e;
            }
            pp.thw = thw[(L - 1u) & 1u];
            hipLaunchKernelGGL(gemm_post_kernel<GEMM_HMC>, dim3(cls_grid), dim3(256), 0, s, pp);
        } else {
            StepParams sk = sp; sk.th_in = thw[0]; sk.w_out = wprop;
            if (int e = launch_step<2>(sk, s)) return e;
            if (r.algo == GEMM_MALA) hipLaunchKernelGGL(gemm_post_kernel<GEMM_MALA>, dim3(cls_grid), dim3(256), 0, s, pp);
            else hipLaunchKernelGGL(gemm_post_kernel<GEMM_RWMH>, dim3(cls_grid), dim3(256), 0, s, pp);
        }
        hipLaunchKernelGGL(gemm_advance_kernel, dim3(1), dim3(1), 0, s, draw_ctr);
        return (int)hipGetLastError();
    };

    bool graphed = false;
    if (r.use_graph && n_total > 1) {
        // one draw's launches captured once, replayed n_total times (the draw index is device memory; every pointer is the same in every draw)
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        // (captured on a stream of our own: the caller's may be the legacy default stream, which cannot be captured)
        static hipStream_t cap_st = [] { hipStream_t s = nullptr; if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) s = nullptr; return s; }();
        if (cap_st != nullptr && hipStreamBeginCapture(cap_st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            const int e = enqueue_draw(cap_st);
            const hipError_t ec = hipStreamEndCapture(cap_st, &graph);
            if (e == 0 && ec == hipSuccess && graph != nullptr && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                int rc = 0;
                for (uint32_t t = 0; t < n_total && rc == 0; ++t) rc = (int)hipGraphLaunch(exec, st);
                if (rc == 0) rc = (int)hipStreamSynchronize(st);      // the executable graph is ours: it must outlive its launches
                (void)hipGraphExecDestroy(exec);
                (void)hipGraphDestroy(graph);
                if (rc) return rc;
                graphed = true;
            } else {
                if (exec) (void)hipGraphExecDestroy(exec);
                if (graph) (void)hipGraphDestroy(graph);
                (void)hipGetLastError();
            }
        } else (void)hipGetLastError();
    }
    if (!graphed)
        for (uint32_t t = 0; t < n_total; ++t) { if (int e = enqueue_draw(st)) return e; }

    hipLaunchKernelGGL(gemm_store_kernel, dim3((unsigned)std::min<size_t>(((size_t)r.d * r.C + 255) / 256, 65535)), dim3(256), 0, st, dp);
    GEMM_TRY(hipGetLastError());
    if (kernel_name) *kernel_name = r.algo == GEMM_HMC ? (L > 1 ? "gemm_step_kernel<0> (hmc)" : "gemm_step_kernel<1> (hmc)")
                                    : r.algo == GEMM_MALA ? "gemm_step_kernel<2> (mala)" : "gemm_step_kernel<2> (rwmh)";
    return 0;
}

}  // namespace gemm
}  // namespace mi
