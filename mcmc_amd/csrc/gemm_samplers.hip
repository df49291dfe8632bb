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
#include <cstdio>
#include <cstring>

namespace mi {
namespace gemm {

typedef double double4_t __attribute__((ext_vector_type(4)));

#ifndef MI_GEMM_ABLATE
#define MI_GEMM_ABLATE 0          // timing experiments only (results meaningless): 1 no epilogue memory traffic, 2 no tile loads, 4 no barriers
#endif

constexpr int TM = 128, TN = 128, TK = 16;
constexpr int LDS_STRIDE = 144;                      // doubles per staged row: 128 + 16 (k rows j and j + 1 of a fragment read sit 32 banks apart)
constexpr int STAGE = 2 * TK * LDS_STRIDE;           // doubles per stage: 16 rows of the A tile, 16 rows of the B tile
constexpr size_t GEMM_LDS_BYTES = (size_t)2 * STAGE * sizeof(double);

enum : int { TGT_DENSE = 0, TGT_LOGISTIC = 1 };

// One product D = A B over K: A^T as [Kp][ldA] (k-major: a row holds the 128 output rows of a tile contiguously), B as [Kp][Cp]
struct StepParams {
    const double* At;        // dense: P^T; logistic: X^T (MODE 3, eta = X Theta) or X itself (X^T r: K runs over the data rows)
    const double* Bm;        // dense: the positions; logistic: the positions (MODE 3) or the row terms y - sigmoid(eta)
    uint32_t Kp, ldA, M_store, n_ntiles;      // K extent (multiple of 16), padded output rows (multiple of 128), output rows that exist in memory
    uint64_t Cp;
    double eps;
    const double* pos;       // MODE 0..2: the positions the gradient is taken at (the logistic gradient subtracts them; MODE 0 drifts them)
    double* pos_out;         // MODE 0
    double* pm;              // MODE 0 / 1
    double* g_out;           // MODE 1 / 2: grad log K
    const double* m_inv;     // MODE 0: diagonal of INV(precond_mat) per dimension (ones: the identity -- 1.0 * p is p, bit for bit)
    double* term_out;        // MODE 3: eta = X Theta [rows padded to 16][Cp] (gemm_rowterm_kernel turns it into the row terms)
};

template <int MODE, int TGT>
__global__ MI_NO_DS_MERGE __launch_bounds__(256, 2) void gemm_step_kernel(const StepParams prm)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane >> 4, c16 = lane & 15;
    // consecutive workgroup ids go round the 8 XCDs: XCD x takes chain tiles x, x + 8, ... and runs their row tiles back to back
    const uint32_t MT = prm.ldA / TM;
    const uint32_t xcd = blockIdx.x & 7u, q = blockIdx.x >> 3;
    const uint32_t nt = xcd + 8u * (q / MT), mt = q % MT;
    if (nt >= prm.n_ntiles) return;
    const size_t m0 = (size_t)mt * TM, n0 = (size_t)nt * TN;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)lds;
    const uint32_t lane16 = (uint32_t)lane * 16u;
    // wave w moves rows w, w + 4, ... of the 32 staged rows (0..15: A^T rows k, columns m0..; 16..31: B rows k, chains n0..), 1 KiB each
    auto issue = [&](uint32_t kb, int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = wave + 4 * i;
            const int rr = r & 15;
            const double* src = (i >= 4) ? prm.Bm + ((size_t)(kb * TK + rr) * prm.Cp + n0) : prm.At + ((size_t)(kb * TK + rr) * prm.ldA + m0);
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
    const uint32_t nkb = prm.Kp / TK;
    issue(0, 0);
#pragma unroll 1
    for (uint32_t kb = 0; kb < nkb; ++kb) {
        const int stage = (int)(kb & 1u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's rows of step kb landed ...
        if (!(MI_GEMM_ABLATE & 4)) __syncthreads();           // ... everybody's, and nobody still reads the other stage
        if (kb + 1 < nkb && !(MI_GEMM_ABLATE & 2)) issue(kb + 1, stage ^ 1);
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
    // epilogue: acc[ti][ni][r] is D at row m0 + 64 wm + 16 ti + 4 r + j, chain n0 + 64 wn + 16 ni + c16.  One batch per 16-row block ti: its 32 loads
    // (momentum and position of 16 elements per lane) are issued together, then the arithmetic, then the stores -- element by element the epilogue was a
    // chain of 64 dependent round trips to memory (66 us per tile with the matrix pipe idle: 0.70 instead of 0.89 of the peak for the whole call)
    const double eps = prm.eps;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
        const size_t row0 = m0 + (size_t)(64 * wm + 16 * ti);
        if (row0 >= prm.M_store || (MI_GEMM_ABLATE & 1)) continue;          // (M_store is a multiple of 16: the block exists or it does not)
        const size_t base = (row0 + (size_t)j) * prm.Cp + n0 + (size_t)(64 * wn + c16);
        auto at = [&](int r, int ni) -> size_t { return base + (size_t)(4 * r) * prm.Cp + (size_t)(16 * ni); };
        if constexpr (MODE == 3) {                                         // eta = X Theta as it is: gemm_rowterm_kernel makes the row terms of it, at full occupancy
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) prm.term_out[at(r, ni)] = acc[ti][ni][r];
        } else {
            [[maybe_unused]] double pv[4][4], xv[4][4], mi[4];
            if constexpr (MODE == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) mi[r] = prm.m_inv[row0 + (size_t)(4 * r + j)];
            }
            if constexpr (MODE != 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) pv[r][ni] = prm.pm[at(r, ni)];
            }
            if constexpr (MODE == 0 || TGT == TGT_LOGISTIC) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) xv[r][ni] = prm.pos[at(r, ni)];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const size_t idx = at(r, ni);
                    const double v = acc[ti][ni][r];
                    double g;
                    if constexpr (TGT == TGT_DENSE) g = -v;                // grad log K = -(P theta)
                    else g = v - xv[r][ni];                                // X^T (y - sigmoid(eta)) - beta
                    if constexpr (MODE == 2) prm.g_out[idx] = g;
                    else {
                        double p = pv[r][ni];
                        p = p + (eps * g) / 2.0;                           // second half-step of this leapfrog step (hmc.cpp:175)
                        if constexpr (MODE == 1) { prm.pm[idx] = p; prm.g_out[idx] = g; }
                        else {
                            p = p + (eps * g) / 2.0;                       // first half-step of the next one (:126): same position, same gradient
                            prm.pm[idx] = p;
                            prm.pos_out[idx] = xv[r][ni] + eps * (mi[r] * p);   // :171: theta += eps (inv_precond_matrix p), the matrix diagonal
                        }
                    }
                }
        }
    }
}

// M (rows x cols row-major) -> out[k ld + i] = TRANSPOSE ? M[i][k] : M[k][i] for k < Kp, i < ld, zeros outside the matrix
template <bool TRANSPOSE>
__global__ void gemm_pack_kernel(const double* __restrict__ M, uint32_t rows, uint32_t cols, uint32_t Kp, uint32_t ld, double* __restrict__ out)
{
    const size_t n = (size_t)Kp * ld;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const uint32_t k = (uint32_t)(e / ld), i = (uint32_t)(e % ld);
        if (TRANSPOSE) out[e] = (i < rows && k < cols) ? M[(size_t)i * cols + k] : 0.0;
        else out[e] = (k < rows && i < cols) ? M[(size_t)k * cols + i] : 0.0;
    }
}

// the row terms of the logistic target (the oracle's ORC_TARGET_LOGISTIC; softplus / sigmoid of det_math.hpp), in place over eta [nK][Cp]:
// res = y - sigmoid(eta) (what X^T multiplies), term = y eta - log(1 + e^eta) (what the log-likelihood sums); zeros in the padding rows
__global__ __launch_bounds__(256) void gemm_rowterm_kernel(const double* __restrict__ y, uint32_t n_rows, uint32_t nK, uint64_t Cp, double* __restrict__ res, double* __restrict__ term)
{
    const size_t n = (size_t)nK * Cp;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const uint32_t row = (uint32_t)(e / Cp);
        const bool valid = row < n_rows;
        const double yv = valid ? y[row] : 0.0;
        const double eta = valid ? term[e] : 0.0;
        // softplus / sigmoid (det_math.hpp) share e = exp(-|eta|): each of them evaluates it once, on the same argument -- the same bits (logistic_lds.hpp does the same)
        const double ex = det_exp(eta > 0.0 ? -eta : eta);
        const double l1p = det_log(1.0 + ex);
        const double sp = (eta > 0.0) ? (eta + l1p) : l1p;
        const double sg = (eta >= 0.0) ? (1.0 / (1.0 + ex)) : (ex / (1.0 + ex));
        res[e] = valid ? (yv - sg) : 0.0;
        term[e] = valid ? (yv * eta - sp) : 0.0;
    }
}

struct DrawParams {
    int algo, tgt;
    uint32_t d, dK, nK;      // nK: the data rows padded to 16 (logistic)
    uint64_t C, Cp, chain0;
    double* th;              // [dK][Cp] accepted position
    double* gacc;            // grad log K at the accepted position
    double* thw;             // the proposal (hmc: the leapfrog's end point)
    double* gprop;           // grad log K there
    double* pm;              // hmc: momentum
    const double* term;      // logistic: [nK][Cp] y eta - log(1 + e^eta) of the LAST evaluation
    const double* m;         // [dK] the diagonal of precond_mat, its CHOL_LOWER (sqrt) and INV (reciprocal), and of INV(eps^2 M) (mala); ones / 1 / eps^2 for the
    const double* m_sqrt;    //      identity (1.0 * x is x bit for bit, so the identity runs the same statements)
    const double* m_inv;
    const double* s_inv;
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
    if (prm.algo == GEMM_HMC) { prm.pm[ia] = prm.m_sqrt[da] * z0; prm.pm[ib] = prm.m_sqrt[db] * z1; }      // p = sqrt_precond_matrix z (:158), the matrix diagonal
    else if (prm.algo == GEMM_MALA) {                    // mean = x + eps^2 (M grad) / 2 (mala.cpp:123), proposal = mean + eps (sqrt(M) z) (:159)
        prm.thw[ia] = (prm.th[ia] + (prm.s2 * (prm.m[da] * prm.gacc[ia])) / 2.0) + prm.eps * (prm.m_sqrt[da] * z0);
        prm.thw[ib] = (prm.th[ib] + (prm.s2 * (prm.m[db] * prm.gacc[ib])) / 2.0) + prm.eps * (prm.m_sqrt[db] * z1);
    } else {
        prm.thw[ia] = prm.th[ia] + prm.eps * z0;
        prm.thw[ib] = prm.th[ib] + prm.eps * z1;
    }
}

// One thread per (chain, class j = index mod 4): the engine's dot products and row sums are four strided chains, combined (q0 + q2) + (q1 + q3); a wave
// holds 16 chains x 4 classes (the MFMA B layout: its loads are the epilogue's 128-byte segments).
__device__ __forceinline__ double class_sum(double q)
{
    q = q + __shfl_xor(q, 32);
    q = q + __shfl_xor(q, 16);
    return q;
}
// log K at x, given what the evaluation left in memory.  dense: -1/2 x . (P x) with P x = -g; logistic: sum_r [y_r eta_r - log(1 + e^eta_r)] - 1/2 |x|^2
// (the oracle's ORC_TARGET_LOGISTIC: orc_sum over the rows, orc_dot over the dimensions, both four-strided)
template <int TGT>
__device__ __forceinline__ double log_kernel_value(const DrawParams& prm, const double* x, const double* g, uint64_t c, int j)
{
    double q = 0.0;
#pragma unroll 4
    for (uint32_t i = (uint32_t)j; i < prm.dK; i += 4u) {
        const size_t e = (size_t)i * prm.Cp + c;
        const double xv = x[e];
        if constexpr (TGT == TGT_DENSE) q = dfma(xv, -g[e], q); else q = dfma(xv, xv, q);
    }
    q = class_sum(q);
    if constexpr (TGT == TGT_DENSE) return -0.5 * q;
    else {
        double ll = 0.0;
#pragma unroll 4
        for (uint32_t r = (uint32_t)j; r < prm.nK; r += 4u) ll = ll + prm.term[(size_t)r * prm.Cp + c];
        ll = class_sum(ll);
        return ll - 0.5 * q;
    }
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
        const double mi = prm.m_inv[i];
        q = dfma(p, mi * p, q);                              // K = p . (Minv p) / 2 (:160)
        p = p + (prm.eps * prm.gacc[e]) / 2.0;
        prm.pm[e] = p;
        prm.thw[e] = prm.th[e] + prm.eps * (mi * p);         // :171
    }
    q = class_sum(q);
    if (j == 0) prm.kprev[c] = q / 2.0;
}

// the value at the initial state (hmc.cpp:140, mala.cpp:138, rwmh.cpp:113)
template <int TGT>
__global__ __launch_bounds__(256) void gemm_first_kernel(const DrawParams prm)
{
    const int lane = threadIdx.x & 63, j = lane >> 4;
    const uint64_t c = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (lane & 15);
    const double first_lp = log_kernel_value<TGT>(prm, prm.th, prm.gacc, c, j);
    if (j == 0) { prm.prevE[c] = (prm.algo == GEMM_HMC) ? -first_lp : first_lp; prm.nacc[c] = 0ull; }
}

// the accept step (hmc.cpp:178-204; mala.cpp:162-184 with mala.ipp:59-64 and dmvnorm.hpp:37-41; rwmh.cpp:128-149), the accepted state and the kept row
template <int ALGO, int TGT>
__global__ __launch_bounds__(256) void gemm_post_kernel(const DrawParams prm)
{
    const int lane = threadIdx.x & 63, j = lane >> 4;
    const uint64_t c = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (lane & 15);
    const bool live = c < prm.C;
    const uint32_t draw = *prm.draw_ctr;
    const double lp = log_kernel_value<TGT>(prm, prm.thw, prm.gprop, c, j);
    double qk = 0.0, qa = 0.0, qb = 0.0;
    if constexpr (ALGO != GEMM_RWMH) {
#pragma unroll 4
        for (uint32_t i = (uint32_t)j; i < prm.dK; i += 4u) {
            const size_t e = (size_t)i * prm.Cp + c;
            if constexpr (ALGO == GEMM_HMC) { const double p = prm.pm[e]; qk = dfma(p, prm.m_inv[i] * p, qk); }      // :184
            else {                                                     // Sigma = eps^2 M: INV(Sigma)_ii from the host (s_inv), the means with M grad
                const double x = prm.thw[e], be = prm.th[e], gr = prm.gacc[e], gp = prm.gprop[e];
                const double mm = prm.m[i], si = prm.s_inv[i];
                const double mean_prop = x + (prm.s2 * (mm * gp)) / 2.0;
                const double xa = be - mean_prop;                      // dmvnorm.hpp:37
                qa = dfma(xa, si * xa, qa);
                const double mean_prev = be + (prm.s2 * (mm * gr)) / 2.0;
                const double xb = x - mean_prev;
                qb = dfma(xb, si * xb, qb);
            }
        }
    }
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
        if (accept) { v = prm.thw[e]; prm.th[e] = v; prm.gacc[e] = prm.gprop[e]; }
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
    uint32_t dK, dM, nK, nM;
    uint64_t Cp;
    size_t vec, rvec;      // doubles per state array / per row-term array
    size_t mat;            // doubles of the packed matrices
    size_t n_doubles;
};
static Layout layout_of(uint32_t d, uint32_t n_rows, uint64_t C)
{
    Layout l;
    l.dK = round_up(d, TK); l.dM = round_up(d, TM);
    l.nK = n_rows ? round_up(n_rows, TK) : 0; l.nM = n_rows ? round_up(n_rows, TM) : 0;
    l.Cp = (C + TN - 1) / TN * TN;
    l.vec = (size_t)l.dK * l.Cp;
    l.rvec = (size_t)l.nK * l.Cp;
    // dense: P^T [dK][dM]; logistic: X^T [dK][nM] and X [nK][dM]
    l.mat = n_rows ? (size_t)l.dK * l.nM + (size_t)l.nK * l.dM : (size_t)l.dK * l.dM;
    // matrices | th, gacc, thw0, thw1, gprop, pm | res, term | prevE, kprev, nacc | draw counter
    l.n_doubles = l.mat + 6 * l.vec + 2 * l.rvec + 3 * l.Cp + 32;
    return l;
}
uint32_t gemm_padded_d(uint32_t d) { return round_up(d, TK); }
size_t gemm_ws_bytes(uint32_t d, uint32_t n_rows, uint64_t C) { return layout_of(d, n_rows, C).n_doubles * sizeof(double); }

#define GEMM_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

template <int MODE, int TGT>
static int step_attr()      // (73 728 bytes of dynamic LDS: above the 64 KiB default)
{
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_step_kernel<MODE, TGT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_LDS_BYTES);
}
template <int MODE, int TGT>
static int launch_step(const StepParams& sp, hipStream_t st)
{
    const uint32_t MT = sp.ldA / TM;
    const uint32_t grid = 8u * MT * ((sp.n_ntiles + 7u) / 8u);
    hipLaunchKernelGGL((gemm_step_kernel<MODE, TGT>), dim3(grid), dim3(256), GEMM_LDS_BYTES, st, sp);
    return (int)hipGetLastError();
}

template <int TGT>
static int gemm_run_t(const GemmRun& r, hipStream_t st, const char** kernel_name)
{
    constexpr bool LOGIT = TGT == TGT_LOGISTIC;
    const Layout l = layout_of(r.d, LOGIT ? r.n_rows : 0u, r.C);
    double* base = static_cast<double*>(r.ws);
    double* A1 = base;                                        // dense: P^T; logistic: X^T [dK][nM]
    double* A2 = LOGIT ? A1 + (size_t)l.dK * l.nM : nullptr;  // logistic: X [nK][dM]
    double* th = base + l.mat;
    double* gacc = th + l.vec;
    double* thw[2] = {gacc + l.vec, gacc + 2 * l.vec};
    double* gprop = gacc + 3 * l.vec;
    double* pm = gacc + 4 * l.vec;
    double* res = pm + l.vec;
    double* term = res + l.rvec;
    double* prevE = term + l.rvec;
    double* kprev = prevE + l.Cp;
    uint64_t* nacc = reinterpret_cast<uint64_t*>(kprev + l.Cp);
    uint32_t* draw_ctr = reinterpret_cast<uint32_t*>(kprev + 2 * l.Cp);

    DrawParams dp{};
    dp.algo = r.algo; dp.tgt = TGT; dp.d = r.d; dp.dK = l.dK; dp.nK = l.nK; dp.C = r.C; dp.Cp = l.Cp; dp.chain0 = r.chain0;
    dp.th = th; dp.gacc = gacc; dp.thw = thw[0]; dp.gprop = gprop; dp.pm = pm; dp.term = term; dp.prevE = prevE; dp.kprev = kprev; dp.nacc = nacc; dp.draw_ctr = draw_ctr;
    dp.theta_in = r.theta; dp.theta_out = r.theta; dp.draws = r.draws; dp.n_accept = r.n_accept; dp.nf_flag = r.nf_flag;
    dp.seed = r.seed; dp.n_burnin = r.n_burnin; dp.draw0 = r.draw0;
    dp.eps = r.eps; dp.s2 = r.s2; dp.rs = r.rs; dp.log_det = r.log_det; dp.cons_term = r.cons_term;
    dp.m = r.mass_tables; dp.m_sqrt = r.mass_tables + l.dK; dp.m_inv = r.mass_tables + 2 * (size_t)l.dK; dp.s_inv = r.mass_tables + 3 * (size_t)l.dK;

    static const int attr_rc = [] { int e = step_attr<0, TGT>(); if (!e) e = step_attr<1, TGT>(); if (!e) e = step_attr<2, TGT>(); if constexpr (LOGIT) { if (!e) e = step_attr<3, TGT>(); } return e; }();
    if (attr_rc) return attr_rc;
    const uint32_t n_ntiles = (uint32_t)(l.Cp / TN);
    // grad log K (and, logistic, the row terms) at `pos`; mode 0: a leapfrog step that is not the last (pos_out: the next position), 1: the last, 2: the gradient alone
    auto evaluate = [&](const double* pos, int mode, double* pos_out, double* g_out, hipStream_t s) -> int {
        StepParams sp{};
        sp.n_ntiles = n_ntiles; sp.Cp = l.Cp; sp.eps = r.eps; sp.pm = pm; sp.pos = pos; sp.pos_out = pos_out; sp.g_out = g_out; sp.m_inv = dp.m_inv;
        if constexpr (LOGIT) {
            StepParams se = sp;                               // eta = X Theta and the row terms
            se.At = A1; se.Bm = pos; se.Kp = l.dK; se.ldA = l.nM; se.M_store = l.nK; se.term_out = term;
            if (int e = launch_step<3, TGT>(se, s)) return e;
            hipLaunchKernelGGL(gemm_rowterm_kernel, dim3((unsigned)std::min<size_t>((l.rvec + 255) / 256, 1u << 20)), dim3(256), 0, s, r.y, r.n_rows, l.nK, l.Cp, res, term);
            sp.At = A2; sp.Bm = res; sp.Kp = l.nK; sp.ldA = l.dM; sp.M_store = l.dK;       // X^T (y - sigmoid(eta)), rows ascending
        } else {
            sp.At = A1; sp.Bm = pos; sp.Kp = l.dK; sp.ldA = l.dM; sp.M_store = l.dK;
        }
        return mode == 0 ? launch_step<0, TGT>(sp, s) : mode == 1 ? launch_step<1, TGT>(sp, s) : launch_step<2, TGT>(sp, s);
    };

    const unsigned ew_grid = (unsigned)std::min<size_t>((l.vec + 255) / 256, 65535);
    const unsigned cls_grid = (unsigned)(l.Cp / 64);                 // 4 waves x 16 chains per workgroup
    auto pack_grid = [](size_t n) { return dim3((unsigned)std::min<size_t>((n + 255) / 256, 65535)); };
    if constexpr (LOGIT) {
        hipLaunchKernelGGL(gemm_pack_kernel<true>, pack_grid((size_t)l.dK * l.nM), dim3(256), 0, st, r.X, r.n_rows, r.d, l.dK, l.nM, A1);
        hipLaunchKernelGGL(gemm_pack_kernel<false>, pack_grid((size_t)l.nK * l.dM), dim3(256), 0, st, r.X, r.n_rows, r.d, l.nK, l.dM, A2);
    } else {
        hipLaunchKernelGGL(gemm_pack_kernel<true>, pack_grid((size_t)l.dK * l.dM), dim3(256), 0, st, r.P, r.d, r.d, l.dK, l.dM, A1);
    }
    hipLaunchKernelGGL(gemm_load_kernel, dim3(ew_grid), dim3(256), 0, st, dp);
    if (int e = evaluate(th, 2, nullptr, gacc, st)) return e;          // the evaluation at the initial values
    hipLaunchKernelGGL(gemm_first_kernel<TGT>, dim3(cls_grid), dim3(256), 0, st, dp);
    GEMM_TRY(hipGetLastError());

    const uint32_t n_total = r.n_burnin + r.n_keep;
    const uint32_t L = r.n_leap;
    // the launches of ONE draw
    auto enqueue_draw = [&](hipStream_t s) -> int {
        hipLaunchKernelGGL(gemm_normals_kernel, dim3((unsigned)(l.Cp / 256 + (l.Cp % 256 ? 1 : 0)), l.dK / 2), dim3(256), 0, s, dp);
        DrawParams pp = dp;
        if (r.algo == GEMM_HMC) {
            hipLaunchKernelGGL(gemm_pre_kernel, dim3(cls_grid), dim3(256), 0, s, dp);
            for (uint32_t k = 0; k < L; ++k)
                if (int e = evaluate(thw[k & 1u], (k + 1 < L) ? 0 : 1, thw[(k + 1u) & 1u], gprop, s)) return e;
            pp.thw = thw[(L - 1u) & 1u];
            hipLaunchKernelGGL((gemm_post_kernel<GEMM_HMC, TGT>), dim3(cls_grid), dim3(256), 0, s, pp);
        } else {
            if (int e = evaluate(thw[0], 2, nullptr, gprop, s)) return e;
            if (r.algo == GEMM_MALA) hipLaunchKernelGGL((gemm_post_kernel<GEMM_MALA, TGT>), dim3(cls_grid), dim3(256), 0, s, pp);
            else hipLaunchKernelGGL((gemm_post_kernel<GEMM_RWMH, TGT>), dim3(cls_grid), dim3(256), 0, s, pp);
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
    if (kernel_name) {
        static thread_local char name[96];
        snprintf(name, sizeof(name), "gemm_step_kernel<%d, %d> (%s%s)", r.algo == GEMM_HMC ? (L > 1 ? 0 : 1) : 2, TGT,
                 r.algo == GEMM_HMC ? "hmc" : r.algo == GEMM_MALA ? "mala" : "rwmh", graphed ? ", graph" : "");
        if (r.diag_mass) { const size_t n = strlen(name); snprintf(name + n - 1, sizeof(name) - n + 1, ", diagonal precond_mat)"); }
        *kernel_name = name;
    }
    return 0;
}

int gemm_run(const GemmRun& r, hipStream_t st, const char** kernel_name)
{
    return r.X != nullptr ? gemm_run_t<TGT_LOGISTIC>(r, st, kernel_name) : gemm_run_t<TGT_DENSE>(r, st, kernel_name);
}

}  // namespace gemm
}  // namespace mi
