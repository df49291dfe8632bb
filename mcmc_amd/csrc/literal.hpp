// literal.hpp -- mcmc::hmc / mcmc::mala stated literally, one workgroup per chain: the replay path of the non-finite regime.
//
// The reference forms `inv_precond_matrix * mntm` (/root/reference/src/hmc.cpp:160,171,184), `jacob_matrix * grad_obj` (:122),
// `precond_matrix * grad_obj` (src/mala.cpp:123), `inv_jacob * precond_matrix` (:115), CHOL_LOWER(J) * sqrt_precond (:155-157),
// INV / LOG_DET of eps^2 J M (include/stats/dmvnorm.hpp:39-41 through include/mcmc/mala.ipp:52-64) as DENSE operations, also when
// the matrices are the identity or diagonal.  As long as every value is finite that is the element-wise arithmetic the
// throughput kernels (hmc_dense.hpp, hmc_diag.hpp, hmc_split.hpp, mala_dense.hpp, logistic_lds.hpp) run.  Once ONE entry of a
// vector is +-inf or NaN, 0 * inf = NaN reaches every other row of a dense product, Gauss-Jordan pivots on NaN, and the
// reference's chain becomes something no element-wise kernel computes.  Those kernels therefore only DETECT the regime (a
// non-finite energy / proposal density is its necessary consequence), flag the chain, leave its outputs untouched, and the
// kernels below replay the flagged chains from their initial values with the reference's operations as written: every
// matrix product an fma chain over all columns (k ascending), BMO_MATOPS_INV as Gauss-Jordan with partial pivoting,
// BMO_MATOPS_CHOL_LOWER as column Cholesky, in the operation order DESIGN.md section 3 states -- so the replayed chain is the
// reference's chain bit for bit, NaN patterns included.  A chain costs O(d^2) per leapfrog step (hmc) or O(d^3) per draw
// (bounded mala) here; that is what the reference itself spends.
//
// The same kernel, run on ALL chains, is the device path of the one configuration the MFMA kernels do not implement: bounded
// mala with a dense precond_mat (INV(eps^2 J(theta') M) per draw, mala.ipp:52-53).
//
// Everything below is __host__ __device__: tests/lit_host.hip compiles the host instantiation, and the CPU test suite holds it
// against the oracle on non-finite cases -- the GPU then only has to agree with itself.
#pragma once

#include "det_math.hpp"

namespace mi {
namespace lit {

constexpr double LIT_EPS_DBL = 2.220446049250313e-16;    // mcmc_options.hpp:103
constexpr double LIT_LOG_2PI = 1.83787706640934548356;   // stats/mcmc_stats.hpp:28-30

enum { LIT_ISO = 0, LIT_DIAG = 1, LIT_DENSE = 2, LIT_LOGISTIC = 3,
       LIT_CALLBACK = 4 };   // the user's HOST callbacks (the reference's std::function contract): see LitMailbox

// LIT_CALLBACK.  The target (and, for rmhmc, the metric tensor) is a function that can only run on the host, while the sampler's own
// arithmetic runs here.  The kernel therefore ASKS: the workgroup writes the evaluation point into a mailbox in host-pinned, fine-grained
// memory, thread 0 publishes a request number (release, system scope) and waits for the host's acknowledgement (acquire, system scope),
// and the workgroup reads the value / gradient / tensor back.  The host side is a loop that serves requests while the kernel runs
// (mi_mcmc.hip: serve_callbacks).  One chain, one workgroup.  A wait is bounded (timeout_ticks of the 100 MHz wall clock): after a
// timeout the abort word is set, nothing is waited for any more, values become NaN, the kernel runs out and the host reports the error --
// a host that died cannot hang the GPU.  In the HOST instantiation of this file (tests/lit_host.hip) the callbacks are simply called.
typedef double (*lit_kernel_cb)(const double* vals, double* grad_out /* nullptr: value only */, void* data);
typedef void (*lit_tensor_cb)(const double* vals, double* tensor_out /* d*d row-major */, double* deriv_out /* d*d*d: dG/dvals_i at + i*d*d, or nullptr */, void* data);
enum { LIT_MB_REQ = 0, LIT_MB_ACK = 1, LIT_MB_KIND = 2, LIT_MB_WANT = 3, LIT_MB_ABORT = 4, LIT_MB_WORDS = 16 };
enum { LIT_REQ_KERNEL = 1, LIT_REQ_TENSOR = 2 };
struct LitMailbox {
    uint32_t* ctl;           // [LIT_MB_WORDS] pinned: request number (device), acknowledged number (host), kind, want (gradient / derivative), abort
    double* value;           // [1] pinned: the log kernel's value
    double* x;               // [d] pinned: the evaluation point
    double* out;             // [d] gradient, or [d*d] tensor followed by [d*d*d] derivative
    uint64_t timeout_ticks;  // of the 100 MHz wall clock
    lit_kernel_cb kernel;    // host instantiation only
    lit_tensor_cb tensor;
    void* kernel_data;
    void* tensor_data;
};

struct LitTarget {
    int kind;
    uint32_t d, n_rows;
    const double* prec;      // LIT_DIAG: d precisions, prec[i * prec_stride]; LIT_DENSE: the precision TRANSPOSED, prec[k * d + i] = P[i][k]
                             // (a workgroup forms row i in thread i: consecutive threads then read consecutive addresses; the fma
                             // chain of a row still runs over k ascending)
    uint32_t prec_stride;    // LIT_DIAG: 1, or d + 1 when prec is the diagonal of a dense d*d matrix
    const double* X;         // LIT_LOGISTIC: n_rows*d row-major (X^T r: thread j walks the rows)
    const double* Xt;        // LIT_LOGISTIC: the same matrix transposed, Xt[j * n_rows + r] (eta = X beta: thread r walks the columns)
    const double* y;
    int W;                   // strided fma chains of a dot product (4: the layout of the MFMA kernels)
    int nblk;                // > 1: dimension-blocked reductions, block size bs (the logistic kernels: 4 blocks of 16 NTQ)
    uint32_t bs;
    int eta_chains;          // LIT_LOGISTIC: sub-chains of eta inside a dimension block
    LitMailbox mb;           // LIT_CALLBACK
};

struct LitParams {
    LitTarget t;
    uint64_t C, chain0;
    double* theta;           // [d][C] in/out
    double* draws;           // [n_keep][d][C] or nullptr
    uint64_t* n_accept;
    uint64_t* n_leap;
    uint64_t seed;
    uint32_t n_burnin, n_keep, n_leap_steps, draw0;
    double eps;              // step_size
    int vals_bound;
    const int* btype;        // [d] 1 none, 2 lower, 3 upper, 4 both (determine_bounds_type.hpp:27-57); vals_bound only
    const double* lb;
    const double* ub;
    int precond;             // 0 identity; 1 diagonal: m / m_sqrt / m_inv [d]; 2 dense: Mfull / Lchol / Minv, d*d TRANSPOSED (M_t[k * d + i] = M[i][k])
    const double* m;
    const double* m_sqrt;
    const double* m_inv;
    uint64_t m_chain_stride; // precond 1, hmc: 0 = one mass for all chains; C = per-chain masses (mi_chains.mass_diag): element i of chain c at [i * C + c]
    const double* Mfull;
    const double* Lchol;
    const double* Minv;
    // mala, unbounded: INV / LOG_DET of Sigma = eps^2 M are constants of the run, from the host (the oracle's operation order)
    const double* sinv_diag; // precond 0 / 1: [d] 1 / (eps^2 m_i) (nullptr: all equal rs)
    const double* Sinv;      // precond 2: d*d TRANSPOSED
    double rs, log_det, cons_term;
    // nuts (nuts_settings_t, mcmc_structs.hpp:89-97): step_size in `eps` is the initial epsilon_bar
    uint32_t n_adapt, max_depth;
    double delta, gamma, t0, kappa;
    uint32_t n_fp_steps;     // rmhmc (rmhmc_settings_t::n_fp_steps, mcmc_structs.hpp:105-119)
    double* step_out;        // [C] or nullptr: in (draw0 > 0: the adapted step sizes of the call before) / out (final step size)
    double* adapt_state;     // [3][C] or nullptr: nuts dual-averaging state (h, epsilon_bar, mu), out always, in when 0 < draw0 <= n_adapt
    uint32_t* depth_trace;   // [n_total][C] or nullptr
    const uint32_t* flag;    // [C]: replay only the chains whose entry is non-zero; nullptr: every chain
    const uint32_t* any;     // nullptr, or one word: zero = nothing was flagged, the launch returns at once
    double* work;            // workspace, work_stride doubles per workgroup
    size_t work_stride;
};

// workspace doubles per workgroup
constexpr int LIT_NUTS_MAX_DEPTH = 30;   // the reference leaves max_tree_depth a free size_t; a tree of depth 30 is 2^30 leapfrog steps per draw
constexpr int LIT_NUTS_FRAME_VECS = 6;    // new_draw_p, new_draw_pp, dummy_draw, dummy_mntm, edge_draw, edge_mntm (nuts.ipp:160-208)
constexpr int LIT_NUTS_TOP_VECS = 12;
constexpr int LIT_RMHMC_MAX_D = 64;      // two d x d x d derivative cubes per workgroup
constexpr int LIT_RMHMC_VECS = 10;
constexpr int LIT_RMHMC_MATS = 10;
MI_HD size_t lit_work_doubles(uint32_t d, uint32_t n_rows, bool mala_bounded, uint32_t nuts_depth = 0, bool nuts = false, bool rmhmc = false)
{
    const size_t dv = (size_t)d + 8;
    return 19 * dv + 2 * ((size_t)n_rows + 8) + (mala_bounded ? 10 * (size_t)d * d : 0)
         + (nuts ? ((size_t)LIT_NUTS_TOP_VECS + (size_t)LIT_NUTS_FRAME_VECS * (nuts_depth + 1)) * dv : 0)
         + (rmhmc ? (size_t)LIT_RMHMC_VECS * dv + 3 * ((size_t)n_rows + 8) + (size_t)LIT_RMHMC_MATS * d * d + 2 * (size_t)d * d * d : 0);
}

struct Par {
    int tid, nth;
    MI_HD void sync() const
    {
#if defined(__HIP_DEVICE_COMPILE__)
        __syncthreads();
#endif
    }
};

#define LIT_PFOR(i, n) for (uint32_t i = (uint32_t)par.tid; i < (uint32_t)(n); i += (uint32_t)par.nth)

MI_HD double lit_nan() { return __builtin_nan(""); }

// ---- BMO shim, stated.  Scalar reductions are computed redundantly by every thread: same bits everywhere.
MI_HD double dot_w(const double* x, const double* y, uint32_t n, int W)      // BMO_MATOPS_DOT_PROD
{
    if (W <= 1) {
        double q = 0.0;
        for (uint32_t i = 0; i < n; ++i) q = dfma(x[i], y[i], q);
        return q;
    }
    double q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (W > 8) W = 8;
    for (uint32_t i = 0; i < n; ++i) q[i % (uint32_t)W] = dfma(x[i], y[i], q[i % (uint32_t)W]);
    for (int h = W / 2; h >= 1; h /= 2)
        for (int c = 0; c < h; ++c) q[c] = q[c] + q[c + h];
    return q[0];
}
MI_HD double dot_b(const LitTarget& t, const double* x, const double* y)
{
    const uint32_t d = t.d;
    if (t.nblk <= 1 || t.bs == 0) return dot_w(x, y, d, t.W);
    double r = 0.0;
    for (int k = 0; k < t.nblk; ++k) {
        const uint32_t lo = (uint32_t)k * t.bs;
        const uint32_t len = (lo < d) ? ((d - lo < t.bs) ? d - lo : t.bs) : 0u;
        const double bk = dot_w(x + (len ? lo : 0u), y + (len ? lo : 0u), len, t.W);
        r = (k == 0) ? bk : r + bk;
    }
    return r;
}
MI_HD double sum_w(const double* x, uint32_t n, int W)
{
    if (W <= 1) {
        double q = 0.0;
        for (uint32_t i = 0; i < n; ++i) q = q + x[i];
        return q;
    }
    double q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (W > 8) W = 8;
    for (uint32_t i = 0; i < n; ++i) q[i % (uint32_t)W] = q[i % (uint32_t)W] + x[i];
    for (int h = W / 2; h >= 1; h /= 2)
        for (int c = 0; c < h; ++c) q[c] = q[c] + q[c + h];
    return q[0];
}
MI_HD uint32_t count_nonfinite(const double* x, uint32_t n)
{
    uint32_t c = 0;
    for (uint32_t i = 0; i < n; ++i) c += is_finite(x[i]) ? 0u : 1u;
    return c;
}
// A state that is NaN in EVERY dimension is absorbing for hmc and mala: every entry of the next proposal has prev_i as a summand, so the
// proposal is all NaN again; its energies / densities are NaN, the acceptance exponent is std::min(0.01, NaN) = 0.01 (hmc.cpp:188,
// mala.cpp:170: the comparison with NaN is false), exp(0.01) > 1 > u, so it is accepted -- whatever the random numbers.  The literal kernels
// use that to fast-forward such a chain (a chain that starts outside its bounds is NaN from the transform on: without this, each of its
// draws would still pay the O(d^3) of bounded mala).  Rows are stored through the same store_row as always.
MI_HD bool all_nan(const double* x, uint32_t n)
{
    for (uint32_t i = 0; i < n; ++i) if (x[i] == x[i]) return false;
    return true;
}
// y = A x, A dense row-major: every row one fma chain, k ascending.  y must not alias x.
MI_HD void gemv(const Par& par, const double* A, const double* x, uint32_t d, double* y)
{
    LIT_PFOR(i, d) {
        double acc = 0.0;
        const double* a = A + (size_t)i * d;
        for (uint32_t k = 0; k < d; ++k) acc = dfma(a[k], x[k], acc);
        y[i] = acc;
    }
    par.sync();
}
// the same product from the TRANSPOSED matrix At[k * d + i] = A[i][k]: per row the identical k-ascending fma chain, coalesced reads
MI_HD void gemv_t(const Par& par, const double* At, const double* x, uint32_t d, double* y)
{
    LIT_PFOR(i, d) {
        double acc = 0.0;
        for (uint32_t k = 0; k < d; ++k) acc = dfma(At[(size_t)k * d + i], x[k], acc);
        y[i] = acc;
    }
    par.sync();
}
// y = D x for a matrix D whose off-diagonal entries are exact zeros (the identity when dg == nullptr), as the dense fma chain
// it is in the reference: acc = +0; fma(0, x_k, acc) leaves acc alone for a finite x_k and makes it NaN otherwise; the
// diagonal term is fma(D_ii, x_i, +0).  y must not alias x.
MI_HD void diag_gemv(const Par& par, const double* dg, const double* x, uint32_t d, double* y)
{
    const uint32_t n = count_nonfinite(x, d);
    LIT_PFOR(i, d) {
        const uint32_t others = n - (is_finite(x[i]) ? 0u : 1u);
        y[i] = (others > 0u) ? lit_nan() : dfma(dg ? dg[i] : 1.0, x[i], 0.0);
    }
    par.sync();
}
// C = A B (d x d): every entry one fma chain, k ascending
MI_HD void matmul(const Par& par, const double* A, const double* B, uint32_t d, double* Cm)
{
    LIT_PFOR(e, d * d) {
        const uint32_t i = e / d, j = e % d;
        double acc = 0.0;
        for (uint32_t k = 0; k < d; ++k) acc = dfma(A[(size_t)i * d + k], B[(size_t)k * d + j], acc);
        Cm[e] = acc;
    }
    par.sync();
}
MI_HD void scale_mat(const Par& par, double s, double* A, uint32_t d)
{
    LIT_PFOR(e, d * d) A[e] = s * A[e];
    par.sync();
}
// BMO_MATOPS_INV: Gauss-Jordan with partial pivoting (first strictly larger |a|, NaN never larger).  a: d*d scratch.
MI_HD void inverse(const Par& par, const double* A, uint32_t d, double* a, double* Ainv)
{
    LIT_PFOR(e, d * d) { a[e] = A[e]; Ainv[e] = (e / d == e % d) ? 1.0 : 0.0; }
    par.sync();
    for (uint32_t c = 0; c < d; ++c) {
        uint32_t piv = c;                                   // every thread runs the same scan
        double best = __builtin_fabs(a[(size_t)c * d + c]);
        for (uint32_t r = c + 1; r < d; ++r) {
            const double v = __builtin_fabs(a[(size_t)r * d + c]);
            if (v > best) { best = v; piv = r; }
        }
        par.sync();                                         // the scan has read column c of every row
        if (piv != c) {
            LIT_PFOR(j, d) {
                double t = a[(size_t)c * d + j]; a[(size_t)c * d + j] = a[(size_t)piv * d + j]; a[(size_t)piv * d + j] = t;
                t = Ainv[(size_t)c * d + j]; Ainv[(size_t)c * d + j] = Ainv[(size_t)piv * d + j]; Ainv[(size_t)piv * d + j] = t;
            }
            par.sync();
        }
        const double pv = a[(size_t)c * d + c];
        par.sync();                                         // everybody holds pv before the row changes
        LIT_PFOR(j, d) { a[(size_t)c * d + j] = a[(size_t)c * d + j] / pv; Ainv[(size_t)c * d + j] = Ainv[(size_t)c * d + j] / pv; }
        par.sync();
        // rows r != c: a[r][:] -= f a[c][:], Ainv[r][:] -= f Ainv[c][:] with f = a[r][c] as it is BEFORE the row changes: column c
        // (which holds every row's f) is updated in a second phase, behind a barrier
        LIT_PFOR(e, d * d) {
            const uint32_t r = e / d, j = e % d;
            if (r == c || j == c) continue;
            const double f = a[(size_t)r * d + c];
            if (f == 0.0) continue;
            a[e] = a[e] - f * a[(size_t)c * d + j];
        }
        LIT_PFOR(e, d * d) {
            const uint32_t r = e / d, j = e % d;
            if (r == c) continue;
            const double f = a[(size_t)r * d + c];
            if (f == 0.0) continue;
            Ainv[e] = Ainv[e] - f * Ainv[(size_t)c * d + j];
        }
        par.sync();
        LIT_PFOR(r, d) {                                    // column c itself last: a[r][c] -= f a[c][c]
            if (r == c) continue;
            const double f = a[(size_t)r * d + c];
            if (f == 0.0) continue;
            a[(size_t)r * d + c] = f - f * a[(size_t)c * d + c];
        }
        par.sync();
    }
}
// BMO_MATOPS_CHOL_LOWER: column Cholesky
MI_HD void chol_lower(const Par& par, const double* A, uint32_t d, double* L)
{
    LIT_PFOR(e, d * d) L[e] = 0.0;
    par.sync();
    for (uint32_t j = 0; j < d; ++j) {
        double sum = A[(size_t)j * d + j];
        for (uint32_t k = 0; k < j; ++k) sum = sum - L[(size_t)j * d + k] * L[(size_t)j * d + k];
        const double ljj = __builtin_sqrt(sum);
        par.sync();                                         // row j (columns < j) has been read by everybody
        LIT_PFOR(i, d) {
            if (i < j) continue;
            if (i == j) { L[(size_t)j * d + j] = ljj; continue; }
            double t = A[(size_t)i * d + j];
            for (uint32_t k = 0; k < j; ++k) t = t - L[(size_t)i * d + k] * L[(size_t)j * d + k];
            L[(size_t)i * d + j] = t / ljj;
        }
        par.sync();
    }
}
MI_HD double log_det_from_chol(const double* L, uint32_t d)
{
    double ld = 0.0;
    for (uint32_t i = 0; i < d; ++i) ld = ld + 2.0 * det_log(L[(size_t)i * d + i]);
    return ld;
}

// ---- box constraints (transform_vals.hpp:25-119, log_jacobian.hpp:25-58, inv_jacobian_adjust.hpp:25-56)
MI_HD double lit_transform(double v, int bt, double lb, double ub)
{
    switch (bt) {
    case 2: return det_log(v - lb + LIT_EPS_DBL);
    case 3: return -det_log(ub - v + LIT_EPS_DBL);
    case 4: return det_log(v - lb + LIT_EPS_DBL) - det_log(ub - v + LIT_EPS_DBL);
    default: return v;
    }
}
MI_HD double lit_inv_transform(double v, int bt, double lb, double ub)
{
    switch (bt) {
    case 2: return !is_finite(v) ? lb + LIT_EPS_DBL : lb + LIT_EPS_DBL + det_exp(v);
    case 3: return !is_finite(v) ? ub - LIT_EPS_DBL : ub - LIT_EPS_DBL - det_exp(-v);
    case 4: {
        if (!is_finite(v)) {
            if (v != v) return (ub - lb) / 2;
            return (v < 0.0) ? lb + LIT_EPS_DBL : ub - LIT_EPS_DBL;
        }
        const double e = det_exp(v);
        const double r = (lb - LIT_EPS_DBL + (ub + LIT_EPS_DBL) * e) / (1.0 + e);
        return is_finite(r) ? r : ub - LIT_EPS_DBL;
    }
    default: return v;
    }
}
MI_HD double lit_inv_jacobian(double v, int bt, double lb, double ub)
{
    switch (bt) {
    case 2: return det_exp(-v);
    case 3: return det_exp(v);
    case 4: { const double e = det_exp(v); return ((e + 1) * (e + 1)) / (e * (ub - lb)); }
    default: return 1.0;
    }
}
MI_HD double lit_log_jacobian(const LitParams& p, const double* v)
{
    double ret = 0.0;
    for (uint32_t i = 0; i < p.t.d; ++i) {
        switch (p.btype[i]) {
        case 2: ret += v[i]; break;
        case 3: ret += -v[i]; break;
        case 4: {
            const double e = det_exp(v[i]);
            if (is_finite(e)) ret += det_log(p.ub[i] - p.lb[i]) + v[i] - 2 * det_log(1 + e);
            else ret += det_log(p.ub[i] - p.lb[i]) - v[i];
            break; }
        default: break;
        }
    }
    return ret;
}

// ---- the target: value (returned, same bits in every thread) and, when grad != nullptr, the gradient of the log kernel.
// w: d doubles of scratch; rows: 2 n_rows doubles of scratch (logistic).  x / grad / w / rows distinct.
// one request to the host (LIT_CALLBACK): the point goes out, the answer comes back; see LitMailbox.  Returns false after an abort.
MI_HD bool mailbox_call(const Par& par, const LitTarget& t, const double* x, uint32_t kind, bool want)
{
    const uint32_t d = t.d;
#if defined(__HIP_DEVICE_COMPILE__)
    LIT_PFOR(i, d) t.mb.x[i] = x[i];
    __threadfence_system();
    par.sync();
    if (par.tid == 0) {
        const uint32_t seq = t.mb.ctl[LIT_MB_REQ] + 1u;      // (only this thread ever writes the request word)
        t.mb.ctl[LIT_MB_KIND] = kind; t.mb.ctl[LIT_MB_WANT] = want ? 1u : 0u;
        if (t.mb.ctl[LIT_MB_ABORT] == 0u) {
            __hip_atomic_store(&t.mb.ctl[LIT_MB_REQ], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            const uint64_t t0 = wall_clock64();
            bool ok = false;
            while (true) {
                if (__hip_atomic_load(&t.mb.ctl[LIT_MB_ACK], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == seq) { ok = true; break; }
                if (wall_clock64() - t0 > t.mb.timeout_ticks) break;
                __builtin_amdgcn_s_sleep(64);
            }
            if (!ok) __hip_atomic_store(&t.mb.ctl[LIT_MB_ABORT], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    par.sync();
    return __hip_atomic_load(&t.mb.ctl[LIT_MB_ABORT], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == 0u;
#else
    for (uint32_t i = 0; i < d; ++i) t.mb.x[i] = x[i];
    if (kind == LIT_REQ_KERNEL) *t.mb.value = t.mb.kernel(t.mb.x, want ? t.mb.out : nullptr, t.mb.kernel_data);
    else t.mb.tensor(t.mb.x, t.mb.out, want ? t.mb.out + (size_t)d * d : nullptr, t.mb.tensor_data);
    (void)par;
    return true;
#endif
}

MI_HD double target_eval(const Par& par, const LitTarget& t, const double* x, double* grad, double* w, double* rows)
{
    const uint32_t d = t.d;
    switch (t.kind) {
    case LIT_CALLBACK: {                                    // the reference's target_log_kernel(vals_inp, grad_out, target_data), on the host
        const bool ok = mailbox_call(par, t, x, LIT_REQ_KERNEL, grad != nullptr);
        if (grad) { LIT_PFOR(i, d) grad[i] = ok ? t.mb.out[i] : lit_nan(); }
        const double r = ok ? *t.mb.value : lit_nan();
        par.sync();                                         // the mailbox may be reused by the caller's next request
        return r;
    }
    case LIT_ISO: {
        if (grad) { LIT_PFOR(i, d) grad[i] = -x[i]; par.sync(); }
        const double r = -0.5 * dot_b(t, x, x);
        par.sync();                                         // x may be overwritten by the caller's next phase
        return r;
    }
    case LIT_DIAG: {
        LIT_PFOR(i, d) w[i] = t.prec[(size_t)i * t.prec_stride] * x[i];
        par.sync();
        if (grad) { LIT_PFOR(i, d) grad[i] = -w[i]; par.sync(); }
        const double r = -0.5 * dot_b(t, x, w);
        par.sync();                                         // w may be overwritten by the next call
        return r;
    }
    case LIT_DENSE: {
        gemv_t(par, t.prec, x, d, w);
        if (grad) { LIT_PFOR(i, d) grad[i] = -w[i]; par.sync(); }
        const double r = -0.5 * dot_b(t, x, w);
        par.sync();
        return r;
    }
    default: {   // LIT_LOGISTIC: log K = sum_r [y_r eta_r - log(1 + e^eta_r)] - |beta|^2 / 2, grad = X^T (y - sigmoid(eta)) - beta
        const uint32_t n = t.n_rows;
        double* eta = rows;
        double* term = rows + n;
        LIT_PFOR(r, n) {
            double acc = 0.0;
            const double* xr = t.Xt + r;                 // X[r][j] = xr[j * n]
            if (t.nblk <= 1 || t.bs == 0) {
                for (uint32_t j = 0; j < d; ++j) acc = dfma(xr[(size_t)j * n], x[j], acc);
            } else {
                const int nch = t.eta_chains > 1 ? t.eta_chains : 1;
                const uint32_t sub = t.bs / (uint32_t)nch;
                for (int k = 0; k < t.nblk; ++k) {
                    double e = 0.0;
                    for (int c = 0; c < nch; ++c) {
                        const uint32_t lo = (uint32_t)k * t.bs + (uint32_t)c * sub;
                        const uint32_t hi = (c == nch - 1) ? (uint32_t)(k + 1) * t.bs : lo + sub;
                        double h = 0.0;
                        for (uint32_t j = lo; j < d && j < hi; ++j) h = dfma(xr[(size_t)j * n], x[j], h);
                        e = (c == 0) ? h : e + h;
                    }
                    acc = (k == 0) ? e : acc + e;
                }
            }
            eta[r] = acc;
            term[r] = t.y[r] * acc - softplus(acc);
        }
        par.sync();
        const double ll = sum_w(term, n, t.W);
        const double ret = ll - 0.5 * dot_b(t, x, x);
        par.sync();
        if (grad) {
            LIT_PFOR(r, n) term[r] = t.y[r] - sigmoid(eta[r]);
            par.sync();
            LIT_PFOR(j, d) {
                double acc = 0.0;
                for (uint32_t r = 0; r < n; ++r) acc = dfma(t.X[(size_t)r * d + j], term[r], acc);
                grad[j] = acc - x[j];
            }
            par.sync();
        }
        return ret;
    }
    }
}

// vectors of one chain
struct Vecs {
    double *prev, *cur, *mntm, *z, *mp, *grad, *vi, *jd, *jg, *w, *mean, *t, *pmean, *qmean, *xc, *tt, *rows;
    double *mc, *msc, *mic;                                          // this chain's column of per-chain mass tables (m, sqrt, inverse)
    double *J, *JM, *CJ, *T, *Sigma, *ga, *Sinv, *L, *Pm, *SPm;      // d*d each (bounded mala)
};
MI_HD Vecs carve(double* wk, uint32_t d, uint32_t n_rows, bool mats)
{
    Vecs v;
    const size_t dv = (size_t)d + 8;
    double* p = wk;
    v.prev = p; p += dv; v.cur = p; p += dv; v.mntm = p; p += dv; v.z = p; p += dv; v.mp = p; p += dv; v.grad = p; p += dv;
    v.vi = p; p += dv; v.jd = p; p += dv; v.jg = p; p += dv; v.w = p; p += dv; v.mean = p; p += dv; v.t = p; p += dv;
    v.pmean = p; p += dv; v.qmean = p; p += dv; v.xc = p; p += dv; v.tt = p; p += dv;
    v.mc = p; p += dv; v.msc = p; p += dv; v.mic = p; p += dv;
    v.rows = p; p += 2 * ((size_t)n_rows + 8);
    const size_t dd = (size_t)d * d;
    v.J = v.JM = v.CJ = v.T = v.Sigma = v.ga = v.Sinv = v.L = v.Pm = v.SPm = nullptr;
    if (mats) { v.J = p; p += dd; v.JM = p; p += dd; v.CJ = p; p += dd; v.T = p; p += dd; v.Sigma = p; p += dd; v.ga = p; p += dd;
                v.Sinv = p; p += dd; v.L = p; p += dd; v.Pm = p; p += dd; v.SPm = p; p += dd; }
    return v;
}

// rnorm_vec_inplace (hmc.cpp:156, mala.cpp:150): dimension i = 8b + 4h + j takes component h of Philox slot 4b + j
MI_HD void normal_vec(const Par& par, const LitParams& p, uint64_t chain, uint32_t draw, double* z)
{
    const uint32_t d = p.t.d, n_slots = (d + 7) / 8 * 4;
    LIT_PFOR(s, n_slots) {
        double z0, z1;
        rng_normal_pair(p.seed, chain, draw, s, STREAM_NORMAL, z0, z1);
        const uint32_t i0 = 8 * (s / 4) + (s % 4), i1 = i0 + 4;
        if (i0 < d) z[i0] = z0;
        if (i1 < d) z[i1] = z1;
    }
    par.sync();
}

// box_log_kernel (hmc.cpp:84-95, mala.cpp:84-95)
MI_HD double box_log_kernel(const Par& par, const LitParams& p, const Vecs& v, const double* vals)
{
    const uint32_t d = p.t.d;
    if (p.vals_bound) {
        LIT_PFOR(i, d) v.vi[i] = lit_inv_transform(vals[i], p.btype[i], p.lb[i], p.ub[i]);
        par.sync();
        const double k = target_eval(par, p.t, v.vi, nullptr, v.w, v.rows);
        const double r = k + lit_log_jacobian(p, vals);
        par.sync();
        return r;
    }
    return target_eval(par, p.t, vals, nullptr, v.w, v.rows);
}

// y = INV(precond) x, y = CHOL_LOWER(precond) x, y = precond x
MI_HD void times_minv(const Par& par, const LitParams& p, const double* x, double* y)
{
    if (p.precond == 2) gemv_t(par, p.Minv, x, p.t.d, y); else diag_gemv(par, p.precond == 1 ? p.m_inv : nullptr, x, p.t.d, y);
}
MI_HD void times_lchol(const Par& par, const LitParams& p, const double* x, double* y)
{
    if (p.precond == 2) gemv_t(par, p.Lchol, x, p.t.d, y); else diag_gemv(par, p.precond == 1 ? p.m_sqrt : nullptr, x, p.t.d, y);
}
MI_HD void times_m(const Par& par, const LitParams& p, const double* x, double* y)
{
    if (p.precond == 2) gemv_t(par, p.Mfull, x, p.t.d, y); else diag_gemv(par, p.precond == 1 ? p.m : nullptr, x, p.t.d, y);
}

MI_HD void copy_vec(const Par& par, const double* a, double* b, uint32_t d) { LIT_PFOR(i, d) b[i] = a[i]; par.sync(); }

MI_HD void store_outputs(const Par& par, const LitParams& p, uint64_t c, const Vecs& v, uint64_t n_acc, uint64_t n_leap)
{
    const uint32_t d = p.t.d;
    LIT_PFOR(i, d)
        p.theta[(size_t)i * p.C + c] = p.vals_bound ? lit_inv_transform(v.prev[i], p.btype[i], p.lb[i], p.ub[i]) : v.prev[i];
    if (par.tid == 0) {
        if (p.n_accept) p.n_accept[c] = n_acc;
        if (p.n_leap) p.n_leap[c] = n_leap;
    }
    par.sync();
}
MI_HD void store_row(const Par& par, const LitParams& p, uint64_t c, uint32_t row, const double* x)
{
    if (!p.draws) return;
    const uint32_t d = p.t.d;
    double* out = p.draws + (size_t)row * d * p.C + c;      // hmc.cpp:211-218 / mala.cpp:193-200: rows reported through inv_transform
    LIT_PFOR(i, d) out[(size_t)i * p.C] = p.vals_bound ? lit_inv_transform(x[i], p.btype[i], p.lb[i], p.ub[i]) : x[i];
    par.sync();
}

// ---- mcmc::internal::hmc_impl (hmc.cpp:30-227) for local chain c
MI_HD void hmc_chain(const Par& par, const LitParams& p_in, uint64_t c, double* wk)
{
    const uint32_t d = p_in.t.d;
    const Vecs v = carve(wk, d, p_in.t.n_rows, false);
    LitParams p = p_in;
    if (p.precond == 1 && p.m_chain_stride != 0) {          // per-chain diagonal mass: this chain's column becomes the [d] tables
        LIT_PFOR(i, d) {
            const size_t e = (size_t)i * p_in.m_chain_stride + c;
            v.msc[i] = p_in.m_sqrt[e]; v.mic[i] = p_in.m_inv[e];
            if (p_in.m) v.mc[i] = p_in.m[e];
        }
        par.sync();
        p.m_sqrt = v.msc; p.m_inv = v.mic; p.m = p_in.m ? v.mc : nullptr;
    }
    const uint64_t chain = p.chain0 + c;
    const double step = p.eps;
    LIT_PFOR(i, d) {
        const double x = p.theta[(size_t)i * p.C + c];
        v.prev[i] = p.vals_bound ? lit_transform(x, p.btype[i], p.lb[i], p.ub[i]) : x;       // :134-136
    }
    par.sync();
    // mntm_update_fn (:99-128): mntm += step [J] grad / 2 at pos
    auto mntm_update = [&](const double* pos) {
        if (p.vals_bound) {
            LIT_PFOR(i, d) {
                v.vi[i] = lit_inv_transform(pos[i], p.btype[i], p.lb[i], p.ub[i]);           // :108
                v.jd[i] = lit_inv_jacobian(pos[i], p.btype[i], p.lb[i], p.ub[i]);            // :114
            }
            par.sync();
            (void)target_eval(par, p.t, v.vi, v.grad, v.w, v.rows);                          // :110
            diag_gemv(par, v.jd, v.grad, d, v.jg);                                           // jacob_matrix * grad_obj (:122)
            LIT_PFOR(i, d) v.mntm[i] = v.mntm[i] + (step * v.jg[i]) / 2.0;
        } else {
            (void)target_eval(par, p.t, pos, v.grad, v.w, v.rows);                           // :124
            LIT_PFOR(i, d) v.mntm[i] = v.mntm[i] + (step * v.grad[i]) / 2.0;                 // :126
        }
        par.sync();
    };
    auto kinetic = [&]() -> double {                        // p . (Minv p) / 2 (:160,184)
        times_minv(par, p, v.mntm, v.mp);
        const double k = dot_b(p.t, v.mntm, v.mp) / 2.0;
        par.sync();
        return k;
    };
    double prev_U = -box_log_kernel(par, p, v, v.prev);     // :140
    uint64_t n_acc = 0;
    const uint32_t n_total = p.n_burnin + p.n_keep;
    for (uint32_t draw = 0; draw < n_total; ++draw) {
        if (all_nan(v.prev, d)) {                           // absorbing (see all_nan): every remaining draw is this state, accepted
            for (uint32_t r = draw; r < n_total; ++r)
                if (r >= p.n_burnin) { n_acc += 1u; store_row(par, p, c, r - p.n_burnin, v.prev); }
            break;
        }
        normal_vec(par, p, chain, draw + p.draw0, v.z);     // :156
        times_lchol(par, p, v.z, v.mntm);                   // :158
        const double prev_K = kinetic();                    // :160
        copy_vec(par, v.prev, v.cur, d);                    // :162
        for (uint32_t k = 0; k < p.n_leap_steps; ++k) {     // :164-176
            mntm_update(v.cur);
            times_minv(par, p, v.mntm, v.mp);
            LIT_PFOR(i, d) v.cur[i] = v.cur[i] + step * v.mp[i];                             // :171
            par.sync();
            mntm_update(v.cur);
        }
        double prop_U = -box_log_kernel(par, p, v, v.cur);  // :178
        if (!is_finite(prop_U)) prop_U = INF;               // :180-182
        const double prop_K = kinetic();                    // :184
        const double x = -(prop_U + prop_K) + (prev_U + prev_K);
        const double comp_val = (x < 0.01) ? x : 0.01;      // std::min(0.01, x), :188
        const double u = rng_uniform(p.seed, chain, draw + p.draw0, 0u);                     // :189
        const bool accept = u < det_exp(comp_val);          // :191
        if (accept) {
            copy_vec(par, v.cur, v.prev, d);
            prev_U = prop_U;
        }
        if (draw >= p.n_burnin) {
            n_acc += accept ? 1u : 0u;
            store_row(par, p, c, draw - p.n_burnin, v.prev);
        }
    }
    store_outputs(par, p, c, v, n_acc, (uint64_t)n_total * p.n_leap_steps);
}

// ---- mcmc::internal::mala_impl (mala.cpp:30-208) with mala_prop_adjustment (mala.ipp:30-70) and dmvnorm (dmvnorm.hpp:28-54)
MI_HD void mala_chain(const Par& par, const LitParams& p, uint64_t c, double* wk)
{
    const uint32_t d = p.t.d;
    const bool vb = p.vals_bound != 0;
    const Vecs v = carve(wk, d, p.t.n_rows, vb);
    const uint64_t chain = p.chain0 + c;
    const double step = p.eps, s2 = step * step;            // mala.ipp:41
    const size_t dd = (size_t)d * d;
    if (vb) {                                               // precond_matrix / sqrt_precond_matrix as the dense matrices they are (mala.cpp:57-58)
        LIT_PFOR(e, dd) {
            const uint32_t i = (uint32_t)(e / d), j = (uint32_t)(e % d);
            v.Pm[e] = p.precond == 2 ? p.Mfull[(size_t)j * d + i] : (i == j ? (p.precond == 1 ? p.m[i] : 1.0) : 0.0);
            v.SPm[e] = p.precond == 2 ? p.Lchol[(size_t)j * d + i] : (i == j ? (p.precond == 1 ? p.m_sqrt[i] : 1.0) : 0.0);
        }
        par.sync();
    }
    LIT_PFOR(i, d) {
        const double x = p.theta[(size_t)i * p.C + c];
        v.prev[i] = vb ? lit_transform(x, p.btype[i], p.lb[i], p.ub[i]) : x;                 // :132-134
    }
    par.sync();
    // mala_mean_fn (mala.cpp:97-125): out = vals + step^2 [J] M grad / 2; J_out (d*d) receives inv_jacobian_adjust(vals) when bounded
    auto mean_fn = [&](const double* vals, double* J_out, double* out) {
        if (vb) {
            LIT_PFOR(i, d) v.vi[i] = lit_inv_transform(vals[i], p.btype[i], p.lb[i], p.ub[i]);
            par.sync();
            (void)target_eval(par, p.t, v.vi, v.grad, v.w, v.rows);                          // :109
            LIT_PFOR(e, dd) {
                const uint32_t i = (uint32_t)(e / d), j = (uint32_t)(e % d);
                J_out[e] = (i == j) ? lit_inv_jacobian(vals[i], p.btype[i], p.lb[i], p.ub[i]) : 0.0;   // :113
            }
            par.sync();
            matmul(par, J_out, v.Pm, d, v.JM);
            scale_mat(par, s2, v.JM, d);
            gemv(par, v.JM, v.grad, d, v.t);
            LIT_PFOR(i, d) out[i] = vals[i] + v.t[i] / 2.0;                                  // :121
        } else {
            (void)target_eval(par, p.t, vals, v.grad, v.w, v.rows);                          // :123
            times_m(par, p, v.grad, v.t);
            LIT_PFOR(i, d) out[i] = vals[i] + (s2 * v.t[i]) / 2.0;
        }
        par.sync();
    };
    // dmvnorm(x | mu, Sigma) with INV(Sigma) as `sinv` (dense, or nullptr: the diagonal p.sinv_diag / p.rs) and LOG_DET(Sigma)
    auto dmvnorm = [&](const double* x, const double* mu, const double* sinv, double log_det) -> double {
        LIT_PFOR(i, d) v.xc[i] = x[i] - mu[i];                                               // dmvnorm.hpp:37
        par.sync();
        if (sinv == v.Sinv && vb) gemv(par, sinv, v.xc, d, v.tt);                            // built this draw: row-major
        else if (sinv) gemv_t(par, sinv, v.xc, d, v.tt);                                     // the host's constant INV(eps^2 M), transposed
        else if (p.sinv_diag) diag_gemv(par, p.sinv_diag, v.xc, d, v.tt);
        else {                                              // INV(eps^2 I) = diag(rs)
            const uint32_t n = count_nonfinite(v.xc, d);
            LIT_PFOR(i, d) v.tt[i] = (n - (is_finite(v.xc[i]) ? 0u : 1u) > 0u) ? lit_nan() : dfma(p.rs, v.xc[i], 0.0);
            par.sync();
        }
        const double quad = dot_b(p.t, v.xc, v.tt);                                          // :39
        par.sync();
        return p.cons_term - 0.5 * (log_det + quad);                                         // :41
    };
    double prev_LP = box_log_kernel(par, p, v, v.prev);     // :138
    uint64_t n_acc = 0;
    const uint32_t n_total = p.n_burnin + p.n_keep;
    for (uint32_t draw = 0; draw < n_total; ++draw) {
        if (all_nan(v.prev, d)) {                           // absorbing (see all_nan): every remaining draw is this state, accepted
            for (uint32_t r = draw; r < n_total; ++r)
                if (r >= p.n_burnin) { n_acc += 1u; store_row(par, p, c, r - p.n_burnin, v.prev); }
            break;
        }
        normal_vec(par, p, chain, draw + p.draw0, v.z);     // :150
        if (vb) {                                           // :152-157
            mean_fn(v.prev, v.J, v.mean);
            chol_lower(par, v.J, d, v.CJ);
            matmul(par, v.CJ, v.SPm, d, v.T);
            scale_mat(par, step, v.T, d);
            gemv(par, v.T, v.z, d, v.tt);
            LIT_PFOR(i, d) v.cur[i] = v.mean[i] + v.tt[i];
        } else {                                            // :159
            mean_fn(v.prev, nullptr, v.mean);
            times_lchol(par, p, v.z, v.tt);
            LIT_PFOR(i, d) v.cur[i] = v.mean[i] + step * v.tt[i];
        }
        par.sync();
        double prop_LP = box_log_kernel(par, p, v, v.cur);  // :162
        if (!is_finite(prop_LP)) prop_LP = -INF;            // :164-166
        double adj;                                         // mala_prop_adjustment
        if (vb) {
            mean_fn(v.cur, v.CJ, v.pmean);                  // :49  (CJ now holds prop_inv_jacob)
            mean_fn(v.prev, v.J, v.qmean);                  // :50
            matmul(par, v.CJ, v.Pm, d, v.Sigma);            // :52-53: prop_inv_jacob in BOTH terms
            scale_mat(par, s2, v.Sigma, d);
            inverse(par, v.Sigma, d, v.ga, v.Sinv);
            chol_lower(par, v.Sigma, d, v.L);
            const double ld = log_det_from_chol(v.L, d);
            const double a = dmvnorm(v.prev, v.pmean, v.Sinv, ld);
            const double b = dmvnorm(v.cur, v.qmean, v.Sinv, ld);
            adj = a - b;
        } else {
            mean_fn(v.cur, nullptr, v.pmean);               // :60
            mean_fn(v.prev, nullptr, v.qmean);              // :61
            const double a = dmvnorm(v.prev, v.pmean, p.precond == 2 ? p.Sinv : nullptr, p.log_det);
            const double b = dmvnorm(v.cur, v.qmean, p.precond == 2 ? p.Sinv : nullptr, p.log_det);
            adj = a - b;
        }
        const double x = prop_LP - prev_LP + adj;
        const double comp_val = (x < 0.01) ? x : 0.01;      // mala.cpp:170
        const double u = rng_uniform(p.seed, chain, draw + p.draw0, 0u);                     // :171
        const bool accept = u < det_exp(comp_val);          // :173
        if (accept) {
            copy_vec(par, v.cur, v.prev, d);
            prev_LP = prop_LP;
        }
        if (draw >= p.n_burnin) {
            n_acc += accept ? 1u : 0u;
            store_row(par, p, c, draw - p.n_burnin, v.prev);
        }
    }
    store_outputs(par, p, c, v, n_acc, 0);
}

// ---- the dynamics nuts shares with hmc (nuts.cpp:84-154 = hmc.cpp:84-128,164-176) on caller-chosen buffers
// mntm_update_fn: mntm += step [J] grad / 2 at pos
MI_HD void dyn_mntm_update(const Par& par, const LitParams& p, const Vecs& v, double step, const double* pos, double* mntm)
{
    const uint32_t d = p.t.d;
    if (p.vals_bound) {
        LIT_PFOR(i, d) {
            v.vi[i] = lit_inv_transform(pos[i], p.btype[i], p.lb[i], p.ub[i]);
            v.jd[i] = lit_inv_jacobian(pos[i], p.btype[i], p.lb[i], p.ub[i]);
        }
        par.sync();
        (void)target_eval(par, p.t, v.vi, v.grad, v.w, v.rows);
        diag_gemv(par, v.jd, v.grad, d, v.jg);
        LIT_PFOR(i, d) mntm[i] = mntm[i] + (step * v.jg[i]) / 2.0;
    } else {
        (void)target_eval(par, p.t, pos, v.grad, v.w, v.rows);
        LIT_PFOR(i, d) mntm[i] = mntm[i] + (step * v.grad[i]) / 2.0;
    }
    par.sync();
}
// leap_frog_fn (nuts.cpp:139-154), one step of signed size `step`
MI_HD void dyn_leap_frog(const Par& par, const LitParams& p, const Vecs& v, double step, double* draw, double* mntm)
{
    const uint32_t d = p.t.d;
    dyn_mntm_update(par, p, v, step, draw, mntm);
    times_minv(par, p, mntm, v.mp);
    LIT_PFOR(i, d) draw[i] = draw[i] + step * v.mp[i];
    par.sync();
    dyn_mntm_update(par, p, v, step, draw, mntm);
}
MI_HD double dyn_kinetic(const Par& par, const LitParams& p, const Vecs& v, const double* mntm)
{
    times_minv(par, p, mntm, v.mp);
    const double k = dot_b(p.t, mntm, v.mp) / 2.0;
    par.sync();
    return k;
}

// ---- mcmc::internal::rwmh_impl (rwmh.cpp:30-175): eps carries par_scale, the preconditioner slots carry cov_mat
MI_HD void rwmh_chain(const Par& par, const LitParams& p, uint64_t c, double* wk)
{
    const uint32_t d = p.t.d;
    const Vecs v = carve(wk, d, p.t.n_rows, false);
    const uint64_t chain = p.chain0 + c;
    const double par_scale = p.eps;
    LIT_PFOR(i, d) {
        const double x = p.theta[(size_t)i * p.C + c];
        v.prev[i] = p.vals_bound ? lit_transform(x, p.btype[i], p.lb[i], p.ub[i]) : x;       // :105-107
    }
    par.sync();
    double prev_LP = box_log_kernel(par, p, v, v.prev);     // :113
    uint64_t n_acc = 0;
    const uint32_t n_total = p.n_burnin + p.n_keep;
    for (uint32_t draw = 0; draw < n_total; ++draw) {
        normal_vec(par, p, chain, draw + p.draw0, v.z);     // :124
        // (par_scale * CHOL_LOWER(cov)) * z (:119,126): the scaled matrix times the vector, entry by entry
        if (p.precond == 2) {
            LIT_PFOR(i, d) {
                double acc = 0.0;
                for (uint32_t k = 0; k < d; ++k) acc = dfma(par_scale * p.Lchol[(size_t)k * d + i], v.z[k], acc);
                v.tt[i] = acc;
            }
            par.sync();
        } else {
            const uint32_t nnf = count_nonfinite(v.z, d);
            LIT_PFOR(i, d) {
                const double c_i = par_scale * (p.precond == 1 ? p.m_sqrt[i] : 1.0);
                v.tt[i] = (nnf - (is_finite(v.z[i]) ? 0u : 1u) > 0u) ? lit_nan() : dfma(c_i, v.z[i], 0.0);
            }
            par.sync();
        }
        LIT_PFOR(i, d) v.cur[i] = v.prev[i] + v.tt[i];
        par.sync();
        double prop_LP = box_log_kernel(par, p, v, v.cur);  // :128
        if (!is_finite(prop_LP)) prop_LP = -INF;            // :130-132
        const double x = prop_LP - prev_LP;
        const double comp_val = (x < 0.0) ? x : 0.0;        // std::min(0.0, x): NaN -> 0 (:136)
        const double u = rng_uniform(p.seed, chain, draw + p.draw0, 0u);                     // :137
        const bool accept = u < det_exp(comp_val);          // :139
        if (accept) { copy_vec(par, v.cur, v.prev, d); prev_LP = prop_LP; }
        if (draw >= p.n_burnin) {
            n_acc += accept ? 1u : 0u;
            store_row(par, p, c, draw - p.n_burnin, v.prev);
        }
    }
    store_outputs(par, p, c, v, n_acc, 0);
}

// ---- mcmc::internal::nuts_impl (nuts.cpp:30-332) with nuts_find_initial_step_size (nuts.ipp:30-93) and the RECURSIVE
// nuts_build_tree (nuts.ipp:97-241) run as an explicit call / return machine: one frame per tree level, the reference's argument
// plumbing kept literally (every doubling restarts from (prev_draw, mntm_vec); the second-half calls get the crossed edge outputs
// of :195 / :207; one runif per completed second half, in post-order).
struct NutsFrame {
    int state;                       // 0: call the first half; 1: it returned; 2: the second half returned
    uint32_t depth;
    const double *draw_vec, *mntm_vec;
    double *new_draw, *pos, *neg, *mpos, *mneg;
    uint64_t n_p, s_p, n_alpha_p;
    double alpha_p;
};
MI_HD void nuts_chain(const Par& par, const LitParams& p, uint64_t c, double* wk)
{
    const uint32_t d = p.t.d;
    const Vecs v = carve(wk, d, p.t.n_rows, false);
    const size_t dv = (size_t)d + 8;
    double* extra = v.rows + 2 * ((size_t)p.t.n_rows + 8);
    double* const new_draw = extra + 0 * dv; double* const draw_pos = extra + 1 * dv; double* const draw_neg = extra + 2 * dv;
    double* const mntm_pos = extra + 3 * dv; double* const mntm_neg = extra + 4 * dv; double* const dummy_draw = extra + 5 * dv;
    double* const dummy_mntm = extra + 6 * dv; double* const start_draw = extra + 7 * dv; double* const mntm_vec = extra + 8 * dv;
    double* const leaf_start = extra + 9 * dv; double* const leaf_mntm = extra + 10 * dv; double* const diff = extra + 11 * dv;
    double* const frames = extra + (size_t)LIT_NUTS_TOP_VECS * dv;
    auto fvec = [&](uint32_t level, int k) -> double* { return frames + ((size_t)level * LIT_NUTS_FRAME_VECS + k) * dv; };
    const uint64_t chain = p.chain0 + c;
    const uint32_t n_total = p.n_burnin + p.n_keep;
    const uint32_t n_adapt = p.n_adapt;                     // the RUN's adaptation window in global draw indices (the clamp of :54 is immaterial)
    uint64_t n_leap = 0;

    LIT_PFOR(i, d) {
        const double x = p.theta[(size_t)i * p.C + c];
        v.prev[i] = p.vals_bound ? lit_transform(x, p.btype[i], p.lb[i], p.ub[i]) : x;       // :160-162
    }
    par.sync();
    double step_size;
    if (p.draw0 == 0) {
        // rand_vec from the INIT stream, mntm_vec = sqrt_precond * rand_vec (:166-168); nuts_find_initial_step_size (:172)
        {
            const uint32_t n_slots = (d + 7) / 8 * 4;
            LIT_PFOR(sl, n_slots) {
                double z0, z1;
                rng_normal_pair(p.seed, chain, 0u, sl, STREAM_INIT, z0, z1);
                const uint32_t i0 = 8 * (sl / 4) + (sl % 4), i1 = i0 + 4;
                if (i0 < d) v.z[i0] = z0;
                if (i1 < d) v.z[i1] = z1;
            }
            par.sync();
        }
        times_lchol(par, p, v.z, mntm_vec);
        step_size = 1.0;                                     // nuts.ipp:40
        double pU = -box_log_kernel(par, p, v, v.prev);      // :44
        if (!is_finite(pU)) pU = INF;
        const double pK = dyn_kinetic(par, p, v, mntm_vec);  // :51
        copy_vec(par, v.prev, v.cur, d);
        copy_vec(par, mntm_vec, v.mntm, d);
        dyn_leap_frog(par, p, v, step_size, v.cur, v.mntm); n_leap++;                        // :58
        double qU = -box_log_kernel(par, p, v, v.cur);
        if (!is_finite(qU)) qU = INF;
        double qK = dyn_kinetic(par, p, v, v.mntm);          // :66
        const double log_half = det_log(0.5), neg_log2 = -det_log(2.0);
        int a_val = 2 * ((-(qU + qK) + (pU + pK)) > log_half ? 1 : 0) - 1;                   // :70
        bool cond = (-(qU + qK) + (pU + pK)) > neg_log2;     // :71
        while (cond) {
            step_size *= (a_val == 1) ? 2.0 : 0.5;           // :74
            dyn_leap_frog(par, p, v, step_size, v.cur, v.mntm); n_leap++;                    // :76: continues from the moved state
            qU = -box_log_kernel(par, p, v, v.cur);
            if (!is_finite(qU)) qU = INF;
            qK = dyn_kinetic(par, p, v, v.mntm);
            a_val = 2 * ((-(qU + qK) + (pU + pK)) > log_half ? 1 : 0) - 1;                   // :88
            cond = (-(qU + qK) + (pU + pK)) > neg_log2;      // :89
        }
    } else {
        step_size = p.step_out ? p.step_out[c] : 1.0;        // continuation after the adaptation window
    }
    double mu_val = det_log(10 * step_size);                 // nuts.cpp:174
    double h_val = 0.0;
    double epsilon_bar = (p.draw0 == 0) ? p.eps : step_size; // :59
    if (p.draw0 > 0 && p.draw0 <= n_adapt && p.adapt_state != nullptr) {                     // a continuation inside the adaptation window
        h_val = p.adapt_state[c]; epsilon_bar = p.adapt_state[p.C + c]; mu_val = p.adapt_state[2 * p.C + c];
    }
    double prev_U = -box_log_kernel(par, p, v, v.prev);      // :181 (no finiteness guard there)
    uint64_t n_acc = 0;

    NutsFrame st[LIT_NUTS_MAX_DEPTH + 2];
    for (uint32_t draw = 0; draw < n_total; ++draw) {
        const uint32_t dabs = draw + p.draw0;
        uint32_t uslot = 0;
        normal_vec(par, p, chain, dabs, v.z);                // :200
        times_lchol(par, p, v.z, mntm_vec);                  // :202
        const double prev_K = dyn_kinetic(par, p, v, mntm_vec);                              // :204
        const double log_rand = det_log(rng_uniform(p.seed, chain, dabs, uslot++)) - prev_U - prev_K;   // :206
        copy_vec(par, v.prev, new_draw, d); copy_vec(par, v.prev, draw_pos, d); copy_vec(par, v.prev, draw_neg, d);   // :210-215
        copy_vec(par, mntm_vec, mntm_pos, d); copy_vec(par, mntm_vec, mntm_neg, d);
        uint32_t tree_depth = 0;
        uint64_t n_val = 1, s_val = 1, n_alpha_val = 0;
        double alpha_val = 0.0;
        int good_round = 0;
        while (s_val == 1 && tree_depth < p.max_depth) {     // :227
            const double z = rng_uniform(p.seed, chain, dabs, uslot++);                      // :233
            const int dir = (z <= 0.5) ? -1 : 1;             // :235
            copy_vec(par, v.prev, start_draw, d);
            // the reference's top-level call (:241-246 / :251-256): the far side's edges go into dummies
            if (dir == -1) { copy_vec(par, draw_pos, dummy_draw, d); copy_vec(par, mntm_pos, dummy_mntm, d); }
            else { copy_vec(par, draw_neg, dummy_draw, d); copy_vec(par, mntm_neg, dummy_mntm, d); }
            int sp = 0;
            st[0].state = 0; st[0].depth = tree_depth; st[0].draw_vec = start_draw; st[0].mntm_vec = mntm_vec;
            st[0].new_draw = new_draw;
            st[0].pos = (dir == -1) ? dummy_draw : draw_pos; st[0].neg = (dir == -1) ? draw_neg : dummy_draw;
            st[0].mpos = (dir == -1) ? dummy_mntm : mntm_pos; st[0].mneg = (dir == -1) ? mntm_neg : dummy_mntm;
            uint64_t r_n = 0, r_s = 0, r_na = 0; double r_a = 0.0;       // return registers of the callee
            while (sp >= 0) {
                NutsFrame& f = st[sp];
                if (f.depth == 0) {                          // nuts.ipp:126-158
                    copy_vec(par, f.draw_vec, leaf_start, d);
                    copy_vec(par, f.mntm_vec, leaf_mntm, d); // :128
                    copy_vec(par, leaf_start, f.new_draw, d);   // :127
                    dyn_leap_frog(par, p, v, (double)dir * step_size, f.new_draw, leaf_mntm); n_leap++;   // :132
                    double qU = -box_log_kernel(par, p, v, f.new_draw);   // :134
                    if (!is_finite(qU)) qU = INF;
                    const double qK = dyn_kinetic(par, p, v, leaf_mntm);  // :140
                    r_n = (log_rand <= -qU - qK) ? 1u : 0u;  // :146
                    r_s = (log_rand < 1000.0 - qU - qK) ? 1u : 0u;       // :147
                    copy_vec(par, f.new_draw, f.pos, d); copy_vec(par, f.new_draw, f.neg, d);          // :151-155
                    copy_vec(par, leaf_mntm, f.mpos, d); copy_vec(par, leaf_mntm, f.mneg, d);
                    const double dd = -(qU + qK) + (prev_U + prev_K);
                    r_a = det_exp((dd < 0.0) ? dd : 0.0);    // :157
                    r_na = 1;
                    --sp;
                    continue;
                }
                double* const draw_p = fvec((uint32_t)sp, 0); double* const draw_pp = fvec((uint32_t)sp, 1);
                double* const f_dummy_d = fvec((uint32_t)sp, 2); double* const f_dummy_m = fvec((uint32_t)sp, 3);
                double* const edge_d = fvec((uint32_t)sp, 4); double* const edge_m = fvec((uint32_t)sp, 5);
                if (f.state == 0) {                          // :166-171
                    f.state = 1;
                    NutsFrame& g = st[sp + 1];
                    g.state = 0; g.depth = f.depth - 1; g.draw_vec = f.draw_vec; g.mntm_vec = f.mntm_vec;
                    g.new_draw = draw_p; g.pos = f.pos; g.neg = f.neg; g.mpos = f.mpos; g.mneg = f.mneg;
                    ++sp;
                    continue;
                }
                if (f.state == 1) {
                    f.n_p = r_n; f.s_p = r_s; f.alpha_p = r_a; f.n_alpha_p = r_na;
                    if (f.s_p == 1) {
                        f.state = 2;
                        NutsFrame& g = st[sp + 1];
                        g.state = 0; g.depth = f.depth - 1; g.draw_vec = edge_d; g.mntm_vec = edge_m; g.new_draw = draw_pp;
                        if (dir == -1) {                     // :186-196
                            copy_vec(par, f.pos, f_dummy_d, d); copy_vec(par, f.mpos, f_dummy_m, d);
                            copy_vec(par, f.neg, edge_d, d); copy_vec(par, f.mneg, edge_m, d);
                            g.pos = f.neg; g.neg = f_dummy_d; g.mpos = f.mneg; g.mneg = f_dummy_m;     // crossed (:195)
                        } else {                             // :198-208
                            copy_vec(par, f.neg, f_dummy_d, d); copy_vec(par, f.mneg, f_dummy_m, d);
                            copy_vec(par, f.pos, edge_d, d); copy_vec(par, f.mpos, edge_m, d);
                            g.pos = f_dummy_d; g.neg = f.pos; g.mpos = f_dummy_m; g.mneg = f.mpos;     // crossed (:207)
                        }
                        ++sp;
                        continue;
                    }
                } else {                                     // state 2: :212-229
                    const uint64_t n_pp = r_n, s_pp = r_s, n_alpha_pp = r_na;
                    const double alpha_pp = r_a;
                    const double prob = (double)n_pp / (double)(f.n_p + n_pp);               // :212
                    const double zz = rng_uniform(p.seed, chain, dabs, uslot++);             // :213
                    if (zz < prob) copy_vec(par, draw_pp, draw_p, d);                        // :215-217
                    f.n_p += n_pp; f.alpha_p += alpha_pp; f.n_alpha_p += n_alpha_pp;         // :220-222
                    LIT_PFOR(i, d) diff[i] = f.pos[i] - f.neg[i];
                    par.sync();
                    const bool c1 = dot_b(p.t, diff, f.mneg) >= 0.0;                         // :226
                    const bool c2 = dot_b(p.t, diff, f.mpos) >= 0.0;                         // :227
                    par.sync();
                    f.s_p = s_pp * (c1 ? 1u : 0u) * (c2 ? 1u : 0u);                          // :229
                }
                r_n = f.n_p; r_s = f.s_p; r_a = f.alpha_p; r_na = f.n_alpha_p;               // :234-239
                copy_vec(par, draw_p, f.new_draw, d);
                --sp;
            }
            const uint64_t n_p_val = r_n, s_p_val = r_s;
            alpha_val = r_a; n_alpha_val = r_na;             // the top-level call writes alpha_val / n_alpha_val directly (:246,255)
            if (s_p_val == 1) {                              // :260
                const double z2 = rng_uniform(p.seed, chain, dabs, uslot++);                 // :261
                if (z2 < (double)n_p_val / (double)n_val) {  // :263
                    double qU = -box_log_kernel(par, p, v, new_draw);                        // :264
                    if (!is_finite(qU)) qU = INF;
                    copy_vec(par, new_draw, v.prev, d);      // :272-273
                    prev_U = qU;
                    good_round = 1;                          // :277
                }
            }
            n_val += n_p_val;                                // :283
            tree_depth += 1;
            LIT_PFOR(i, d) diff[i] = draw_pos[i] - draw_neg[i];
            par.sync();
            const bool c1 = dot_b(p.t, diff, mntm_neg) >= 0.0;                               // :286
            const bool c2 = dot_b(p.t, diff, mntm_pos) >= 0.0;                               // :287
            par.sync();
            s_val = s_p_val * (c1 ? 1u : 0u) * (c2 ? 1u : 0u);                               // :289
        }
        if (dabs < n_adapt) {                                // :294-302 (the chain's draw index in the RUN)
            const double it = (double)(dabs + 1);
            h_val += (1 / (it + p.t0)) * (p.delta - (alpha_val / (double)n_alpha_val) - h_val);
            step_size = det_exp(mu_val - h_val * __builtin_sqrt(it) / p.gamma);
            epsilon_bar *= det_exp(det_pow(it, -p.kappa) * (det_log(step_size) - det_log(epsilon_bar)));
        } else {
            step_size = epsilon_bar;
        }
        if (p.depth_trace && par.tid == 0) p.depth_trace[(size_t)draw * p.C + c] = tree_depth;
        if (draw >= p.n_burnin) {                            // :306-309
            n_acc += (uint64_t)good_round;
            store_row(par, p, c, draw - p.n_burnin, v.prev);
        }
    }
    if (p.step_out && par.tid == 0) p.step_out[c] = step_size;
    if (p.adapt_state && par.tid == 0) { p.adapt_state[c] = h_val; p.adapt_state[p.C + c] = epsilon_bar; p.adapt_state[2 * p.C + c] = mu_val; }
    store_outputs(par, p, c, v, n_acc, n_leap);
}

// ---- mcmc::internal::rmhmc_impl (rmhmc.cpp:30-287) with the built-in metric tensors (the oracle's orc_target_tensor): the constant
// precision for the Gaussian kinds (zero derivative), the Fisher information X' Lambda X + I for the logistic target.
// G: d*d row-major; dG (may be nullptr): d matrices dG/dvals_i.  lam: 3 n_rows doubles of scratch.
MI_HD void target_tensor(const Par& par, const LitTarget& t, const double* x, double* G, double* dG, double* lam)
{
    const uint32_t d = t.d;
    const size_t dd = (size_t)d * d;
    if (t.kind == LIT_CALLBACK) {                           // the reference's tensor_fn(vals_inp, tensor_deriv_out, tensor_data), on the host
        const bool ok = mailbox_call(par, t, x, LIT_REQ_TENSOR, dG != nullptr);
        for (size_t e = (size_t)par.tid; e < dd; e += (size_t)par.nth) G[e] = ok ? t.mb.out[e] : lit_nan();
        if (dG) { for (size_t e = (size_t)par.tid; e < dd * d; e += (size_t)par.nth) dG[e] = ok ? t.mb.out[dd + e] : lit_nan(); }
        par.sync();
        return;
    }
    if (t.kind != LIT_LOGISTIC) {
        LIT_PFOR(e, dd) {
            const uint32_t i = (uint32_t)(e / d), j = (uint32_t)(e % d);
            G[e] = t.kind == LIT_DENSE ? t.prec[(size_t)j * d + i] : (i == j ? (t.kind == LIT_DIAG ? t.prec[(size_t)i * t.prec_stride] : 1.0) : 0.0);
        }
        if (dG) { for (size_t e = (size_t)par.tid; e < dd * d; e += (size_t)par.nth) dG[e] = 0.0; }
        par.sync();
        return;
    }
    // rows k ascending, one fma per row and entry; eta as ONE sequential chain over the dimensions (orc_target_tensor)
    const uint32_t n = t.n_rows;
    double* lm = lam; double* dl = lam + n;
    LIT_PFOR(k, n) {
        double eta = 0.0;
        for (uint32_t j = 0; j < d; ++j) eta = dfma(t.Xt[(size_t)j * n + k], x[j], eta);
        const double sg = sigmoid(eta);
        const double l_ = sg * (1.0 - sg);
        lm[k] = l_;
        dl[k] = l_ * (1.0 - 2.0 * sg);
    }
    par.sync();
    LIT_PFOR(e, dd) {
        const uint32_t r = (uint32_t)(e / d), c = (uint32_t)(e % d);
        double acc = 0.0;
        for (uint32_t k = 0; k < n; ++k) {
            const double xx = t.X[(size_t)k * d + r] * t.X[(size_t)k * d + c];
            acc = dfma(xx, lm[k], acc);
        }
        G[e] = (r == c) ? acc + 1.0 : acc;
    }
    if (dG) {
        for (size_t e = (size_t)par.tid; e < dd * d; e += (size_t)par.nth) {
            const uint32_t i = (uint32_t)(e / dd), r = (uint32_t)((e % dd) / d), c = (uint32_t)(e % d);
            double acc = 0.0;
            for (uint32_t k = 0; k < n; ++k) {
                const double xx = t.X[(size_t)k * d + r] * t.X[(size_t)k * d + c];
                acc = dfma(xx, dl[k] * t.X[(size_t)k * d + i], acc);
            }
            dG[e] = acc;
        }
    }
    par.sync();
}

MI_HD void rmhmc_chain(const Par& par, const LitParams& p, uint64_t c, double* wk)
{
    const uint32_t d = p.t.d;
    const size_t dd = (size_t)d * d, dv = (size_t)d + 8;
    const Vecs v = carve(wk, d, p.t.n_rows, false);
    double* q = v.rows + 2 * ((size_t)p.t.n_rows + 8);
    double* const new_mntm = q; q += dv; double* const prop_mntm = q; q += dv; double* const prop_draw = q; q += dv; double* const incr = q; q += dv;
    double* const tmpv = q; q += dv; double* const gobj = q; q += dv; double* const av = q; q += dv; double* const bv = q; q += dv;
    double* const jd = q; q += dv; double* const jg = q; q += dv;
    double* const lam = q; q += 3 * ((size_t)p.t.n_rows + 8);
    double* const new_tensor = q; q += dd; double* const prev_tensor = q; q += dd; double* const inv_new = q; q += dd; double* const inv_prev = q; q += dd;
    double* const L = q; q += dd; double* const S = q; q += dd; double* const Tn = q; q += dd; double* const T = q; q += dd; double* const ga = q; q += dd;
    q += dd;
    double* const new_deriv = q; q += dd * d; double* const prev_deriv = q;
    const uint64_t chain = p.chain0 + c;
    const double step = p.eps;
    const bool vb = p.vals_bound != 0;
    auto copy_n = [&](const double* a, double* b, size_t n) { for (size_t e = (size_t)par.tid; e < n; e += (size_t)par.nth) b[e] = a[e]; par.sync(); };
    // box_tensor_fn (rmhmc.cpp:152-164)
    auto box_tensor = [&](const double* vals, double* G, double* dG) {
        if (vb) {
            LIT_PFOR(i, d) v.vi[i] = lit_inv_transform(vals[i], p.btype[i], p.lb[i], p.ub[i]);
            par.sync();
            target_tensor(par, p.t, v.vi, G, dG, lam);
        } else target_tensor(par, p.t, vals, G, dG, lam);
    };
    // mntm_update_fn (rmhmc.cpp:99-150): out = step [J] grad_obj / 2,
    //   grad_obj(i) = -grad(i) + 0.5 (trace(T_i) - dot(T_i' p, Ginv p)),  T_i = Ginv dG_i
    auto mntm_increment = [&](const double* pos, const double* mntm, const double* Ginv, const double* dG, double* out) {
        if (vb) {
            LIT_PFOR(i, d) v.vi[i] = lit_inv_transform(pos[i], p.btype[i], p.lb[i], p.ub[i]);      // :107
            par.sync();
            (void)target_eval(par, p.t, v.vi, v.grad, v.w, v.rows);                                // :108
        } else (void)target_eval(par, p.t, pos, v.grad, v.w, v.rows);                               // :131
        gemv(par, Ginv, mntm, d, bv);                                                              // b = Ginv p
        for (uint32_t i = 0; i < d; ++i) {
            matmul(par, Ginv, dG + (size_t)i * dd, d, T);
            LIT_PFOR(j, d) {                                                                       // a = T' p
                double acc = 0.0;
                for (uint32_t k = 0; k < d; ++k) acc = dfma(T[(size_t)k * d + j], mntm[k], acc);
                av[j] = acc;
            }
            par.sync();
            double tr = 0.0, dp = 0.0;
            for (uint32_t j = 0; j < d; ++j) tr = tr + T[(size_t)j * d + j];
            for (uint32_t j = 0; j < d; ++j) dp = dfma(av[j], bv[j], dp);
            par.sync();
            if (par.tid == 0) gobj[i] = -v.grad[i] + 0.5 * (tr - dp);
        }
        par.sync();
        if (vb) {
            LIT_PFOR(i, d) jd[i] = lit_inv_jacobian(pos[i], p.btype[i], p.lb[i], p.ub[i]);         // :121
            par.sync();
            diag_gemv(par, jd, gobj, d, jg);
            LIT_PFOR(i, d) out[i] = (step * jg[i]) / 2.0;                                          // :129
        } else {
            LIT_PFOR(i, d) out[i] = (step * gobj[i]) / 2.0;                                        // :145
        }
        par.sync();
    };
    auto quad_half = [&](const double* Ginv, const double* mntm) -> double {                       // p' Ginv p / 2 (:211,253)
        gemv(par, Ginv, mntm, d, tmpv);
        double k = 0.0;
        for (uint32_t i = 0; i < d; ++i) k = dfma(mntm[i], tmpv[i], k);
        par.sync();
        return k / 2.0;
    };
    LIT_PFOR(i, d) {
        const double x = p.theta[(size_t)i * p.C + c];
        v.prev[i] = vb ? lit_transform(x, p.btype[i], p.lb[i], p.ub[i]) : x;                      // :170-172
    }
    par.sync();
    copy_vec(par, v.prev, v.cur, d);
    box_tensor(v.cur, new_tensor, new_deriv);                                                      // :187
    copy_n(new_tensor, prev_tensor, dd);
    inverse(par, new_tensor, d, ga, inv_new);                                                      // :190
    copy_n(inv_new, inv_prev, dd);
    copy_n(new_deriv, prev_deriv, dd * d);
    const double cons_term = 0.5 * (double)d * LIT_LOG_2PI;                                        // :195
    chol_lower(par, new_tensor, d, L);
    double prev_U = cons_term - box_log_kernel(par, p, v, v.prev) + 0.5 * log_det_from_chol(L, d); // :197
    par.sync();
    uint64_t n_acc = 0, n_leap = 0;
    const uint32_t n_total = p.n_burnin + p.n_keep;
    for (uint32_t draw = 0; draw < n_total; ++draw) {
        normal_vec(par, p, chain, draw + p.draw0, v.z);                                            // :207
        chol_lower(par, prev_tensor, d, L);
        gemv(par, L, v.z, d, new_mntm);                                                            // :209
        const double prev_K = quad_half(inv_prev, new_mntm);                                       // :211
        copy_vec(par, v.prev, v.cur, d);                                                           // :213
        for (uint32_t k = 0; k < p.n_leap_steps; ++k) {
            copy_vec(par, new_mntm, prop_mntm, d);
            for (uint32_t kk = 0; kk < p.n_fp_steps; ++kk) {                                       // :220-222
                mntm_increment(v.cur, prop_mntm, inv_prev, prev_deriv, incr);
                LIT_PFOR(i, d) prop_mntm[i] = new_mntm[i] + incr[i];
                par.sync();
            }
            copy_vec(par, prop_mntm, new_mntm, d);                                                 // :224
            copy_vec(par, v.cur, prop_draw, d);                                                    // :228
            for (uint32_t kk = 0; kk < p.n_fp_steps; ++kk) {                                       // :231-235
                box_tensor(prop_draw, Tn, nullptr);
                inverse(par, Tn, d, ga, inv_new);
                LIT_PFOR(e, dd) S[e] = inv_prev[e] + inv_new[e];
                par.sync();
                gemv(par, S, new_mntm, d, tmpv);
                LIT_PFOR(i, d) prop_draw[i] = v.cur[i] + (0.5 * step) * tmpv[i];
                par.sync();
            }
            copy_vec(par, prop_draw, v.cur, d);                                                    // :237
            box_tensor(v.cur, new_tensor, new_deriv);                                              // :239
            inverse(par, new_tensor, d, ga, inv_new);                                              // :240
            mntm_increment(v.cur, new_mntm, inv_new, new_deriv, incr);                             // :244
            LIT_PFOR(i, d) new_mntm[i] = new_mntm[i] + incr[i];
            par.sync();
            n_leap++;
        }
        chol_lower(par, new_tensor, d, L);
        double prop_U = cons_term - box_log_kernel(par, p, v, v.cur) + 0.5 * log_det_from_chol(L, d);   // :247
        par.sync();
        if (!is_finite(prop_U)) prop_U = INF;                                                      // :249-251
        const double prop_K = quad_half(inv_new, new_mntm);                                        // :253
        const double x = -(prop_U + prop_K) + (prev_U + prev_K);
        const double comp_val = (x < 0.01) ? x : 0.01;                                             // :257
        const double u = rng_uniform(p.seed, chain, draw + p.draw0, 0u);                           // :258
        const bool accept = u < det_exp(comp_val);                                                 // :260
        if (accept) {
            copy_vec(par, v.cur, v.prev, d);
            prev_U = prop_U;
            copy_n(new_tensor, prev_tensor, dd);
            copy_n(inv_new, inv_prev, dd);
            copy_n(new_deriv, prev_deriv, dd * d);
        }
        if (draw >= p.n_burnin) {
            n_acc += accept ? 1u : 0u;
            store_row(par, p, c, draw - p.n_burnin, v.prev);
        }
    }
    store_outputs(par, p, c, v, n_acc, n_leap);
}

#if defined(__HIPCC__)
template <int ALGO>     // 0 hmc, 1 mala, 2 nuts, 3 rwmh, 4 rmhmc
__global__ __launch_bounds__(256) void literal_kernel(const LitParams prm)
{
    if (prm.any != nullptr && *prm.any == 0u) return;
    const Par par{(int)threadIdx.x, (int)blockDim.x};
    double* wk = prm.work + (size_t)blockIdx.x * prm.work_stride;
    for (uint64_t c = blockIdx.x; c < prm.C; c += gridDim.x) {
        if (prm.flag != nullptr && prm.flag[c] == 0u) continue;
        if (ALGO == 0) hmc_chain(par, prm, c, wk); else if (ALGO == 1) mala_chain(par, prm, c, wk);
        else if (ALGO == 2) nuts_chain(par, prm, c, wk); else if (ALGO == 3) rwmh_chain(par, prm, c, wk); else rmhmc_chain(par, prm, c, wk);
        __syncthreads();
    }
}
#endif

}  // namespace lit
}  // namespace mi
