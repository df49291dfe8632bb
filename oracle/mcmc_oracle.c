/*
 * oracle/mcmc_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See mcmc_oracle.h.
 *
 * Every function cites the reference lines it restates ("ref:" = /root/reference/).
 * Arithmetic conventions (stated here because Eigen / BaseMatrixOps are absent):
 *   - expressions are evaluated exactly as written in the reference source, one IEEE
 *     operation per C++ operator, NO implicit contraction (build: -ffp-contract=off);
 *   - Eigen's "scalar * Matrix * vector" products evaluate as alpha * (A x) (the scalar is
 *     pulled out of the product), which is how they are written below;
 *   - matrix-vector products accumulate each row as a sequential fma chain, j ascending
 *     (this is also the accumulation order of v_mfma_f64_16x16x4_f64 on gfx950);
 *   - BMO_MATOPS_DOT_PROD is orc_dot: W strided fma chains + butterfly (W from settings);
 *   - exp/log/pow are the deterministic versions of orc_math.h.
 */
#include "mcmc_oracle.h"
#include "orc_math.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_LOG_2PI 1.83787706640934548356  /* ref: include/stats/mcmc_stats.hpp:28-30 */

/* ------------------------------------------------------------------ BMO shim restated */

/* BMO_MATOPS_DOT_PROD (ref call sites: src/hmc.cpp:160,184; nuts.ipp:51,66,84,140,226-227) */
double orc_dot(const double* x, const double* y, size_t d, int W)
{
    if (W <= 1) {
        double q = 0.0;
        for (size_t i = 0; i < d; ++i) q = fma(x[i], y[i], q);
        return q;
    }
    double q[64];
    if (W > 64) W = 64;
    for (int c = 0; c < W; ++c) q[c] = 0.0;
    for (size_t i = 0; i < d; ++i) q[i % (size_t)W] = fma(x[i], y[i], q[i % (size_t)W]);
    for (int h = W / 2; h >= 1; h /= 2)
        for (int c = 0; c < h; ++c) q[c] = q[c] + q[c + h];
    return q[0];
}

/* dimension-blocked dot: ((B0 + B1) + B2) + ... with B_k = orc_dot over block k */
double orc_dot_b(const double* x, const double* y, size_t d, int W, int nblk, size_t bs)
{
    if (nblk <= 1 || bs == 0) return orc_dot(x, y, d, W);
    if ((size_t)nblk * bs < d) return NAN;      /* a blocking that does not cover every dimension is a mis-configuration, never a
                                                   silent truncation: poison the result (tests/orc.py refuses it up front) */
    double r = 0.0;
    for (int k = 0; k < nblk; ++k) {
        const size_t lo = (size_t)k * bs;
        const size_t len = (lo < d) ? ((d - lo < bs) ? d - lo : bs) : 0;
        const double bk = orc_dot(x + (len ? lo : 0), y + (len ? lo : 0), len, W);
        r = (k == 0) ? bk : r + bk;
    }
    return r;
}

static double orc_sum(const double* x, size_t n, int W)
{
    if (W <= 1) {
        double q = 0.0;
        for (size_t i = 0; i < n; ++i) q = q + x[i];
        return q;
    }
    double q[64];
    if (W > 64) W = 64;
    for (int c = 0; c < W; ++c) q[c] = 0.0;
    for (size_t i = 0; i < n; ++i) q[i % (size_t)W] = q[i % (size_t)W] + x[i];
    for (int h = W / 2; h >= 1; h /= 2)
        for (int c = 0; c < h; ++c) q[c] = q[c] + q[c + h];
    return q[0];
}

/* Mat * vec (ref: src/hmc.cpp:158,160,171) */
void orc_gemv(const double* A, const double* x, size_t d, double* y)
{
    for (size_t i = 0; i < d; ++i) {
        double acc = 0.0;
        const double* a = A + i * d;
        for (size_t j = 0; j < d; ++j) acc = fma(a[j], x[j], acc);
        y[i] = acc;
    }
}

/* the same mat-vec from the transposed matrix, in axpy form: per output row the identical k-ascending fma chain, but the
 * inner loop runs over rows and vectorises (Mode B of the CPU baseline; At[k*d + i] = A[i*d + k]) */
static void orc_gemv_t(const double* At, const double* x, size_t d, double* y)
{
    for (size_t i = 0; i < d; ++i) y[i] = 0.0;
    for (size_t k = 0; k < d; ++k) {
        const double xk = x[k];
        const double* a = At + k * d;
        for (size_t i = 0; i < d; ++i) y[i] = fma(a[i], xk, y[i]);
    }
}

static void orc_matmul(const double* A, const double* B, size_t d, double* C)
{
    for (size_t i = 0; i < d; ++i)
        for (size_t j = 0; j < d; ++j) {
            double acc = 0.0;
            for (size_t k = 0; k < d; ++k) acc = fma(A[i * d + k], B[k * d + j], acc);
            C[i * d + j] = acc;
        }
}

/* BMO_MATOPS_INV (ref: src/hmc.cpp:58): Gauss-Jordan, partial pivoting. */
int orc_inv(const double* A, size_t d, double* Ainv)
{
    double* a = (double*)malloc(d * d * sizeof(double));
    memcpy(a, A, d * d * sizeof(double));
    for (size_t i = 0; i < d; ++i)
        for (size_t j = 0; j < d; ++j) Ainv[i * d + j] = (i == j) ? 1.0 : 0.0;
    for (size_t c = 0; c < d; ++c) {
        size_t piv = c;
        double best = fabs(a[c * d + c]);
        for (size_t r = c + 1; r < d; ++r)
            if (fabs(a[r * d + c]) > best) { best = fabs(a[r * d + c]); piv = r; }
        if (piv != c)
            for (size_t j = 0; j < d; ++j) {
                double t = a[c * d + j]; a[c * d + j] = a[piv * d + j]; a[piv * d + j] = t;
                t = Ainv[c * d + j]; Ainv[c * d + j] = Ainv[piv * d + j]; Ainv[piv * d + j] = t;
            }
        const double pv = a[c * d + c];
        for (size_t j = 0; j < d; ++j) { a[c * d + j] = a[c * d + j] / pv; Ainv[c * d + j] = Ainv[c * d + j] / pv; }
        for (size_t r = 0; r < d; ++r) {
            if (r == c) continue;
            const double f = a[r * d + c];
            if (f == 0.0) continue;
            for (size_t j = 0; j < d; ++j) {
                a[r * d + j] = a[r * d + j] - f * a[c * d + j];
                Ainv[r * d + j] = Ainv[r * d + j] - f * Ainv[c * d + j];
            }
        }
    }
    free(a);
    return 0;
}

/* BMO_MATOPS_CHOL_LOWER (ref: src/hmc.cpp:59, src/mala.cpp:58,157) */
int orc_chol_lower(const double* A, size_t d, double* L)
{
    memset(L, 0, d * d * sizeof(double));
    for (size_t j = 0; j < d; ++j) {
        double sum = A[j * d + j];
        for (size_t k = 0; k < j; ++k) sum = sum - L[j * d + k] * L[j * d + k];
        const double ljj = sqrt(sum);
        L[j * d + j] = ljj;
        for (size_t i = j + 1; i < d; ++i) {
            double t = A[i * d + j];
            for (size_t k = 0; k < j; ++k) t = t - L[i * d + k] * L[j * d + k];
            L[i * d + j] = t / ljj;
        }
    }
    return 0;
}

/* BMO_MATOPS_LOG_DET (ref: include/stats/dmvnorm.hpp:41).  SEMANTICS CHOICE:
 * the mathematically intended log-determinant via Cholesky, sum_i 2 log L_ii, which cannot
 * underflow at d=512 the way log(det) would. */
static double orc_log_det_from_chol(const double* L, size_t d)
{
    double ld = 0.0;
    for (size_t i = 0; i < d; ++i) ld = ld + 2.0 * orc_log(L[i * d + i]);
    return ld;
}

/* stats_mcmc::dmvnorm(X, mu, Sigma, true)  (ref: include/stats/dmvnorm.hpp:28-54)
 * QUAD_FORM_INV(x,S) restated as dot(x, INV(S) x). */
static double orc_dmvnorm_core_b(const double* x, const double* mu, size_t d, const double* Sinv,
                                 double log_det, int W, int nblk, size_t bs, double* xc, double* t)
{
    const double cons_term = -0.5 * (double)d * ORC_LOG_2PI;             /* dmvnorm.hpp:36 */
    for (size_t i = 0; i < d; ++i) xc[i] = x[i] - mu[i];                 /* :37 */
    orc_gemv(Sinv, xc, d, t);
    const double quad_term = orc_dot_b(xc, t, d, W, nblk, bs);           /* :39 */
    return cons_term - 0.5 * (log_det + quad_term);                      /* :41 */
}

static double orc_dmvnorm_core(const double* x, const double* mu, size_t d, const double* Sinv,
                               double log_det, int W, double* xc, double* t)
{
    const double cons_term = -0.5 * (double)d * ORC_LOG_2PI;             /* dmvnorm.hpp:36 */
    for (size_t i = 0; i < d; ++i) xc[i] = x[i] - mu[i];                 /* :37 */
    orc_gemv(Sinv, xc, d, t);
    const double quad_term = orc_dot(xc, t, d, W);                       /* :39 */
    return cons_term - 0.5 * (log_det + quad_term);                      /* :41 */
}

double orc_dmvnorm_log(const double* x, const double* mu, const double* Sigma, size_t d, int W)
{
    double* Sinv = (double*)malloc(d * d * sizeof(double));
    double* L = (double*)malloc(d * d * sizeof(double));
    double* xc = (double*)malloc(2 * d * sizeof(double));
    orc_inv(Sigma, d, Sinv);
    orc_chol_lower(Sigma, d, L);
    const double ld = orc_log_det_from_chol(L, d);
    const double r = orc_dmvnorm_core(x, mu, d, Sinv, ld, W, xc, xc + d);
    free(Sinv); free(L); free(xc);
    return r;
}

/* ------------------------------------------------------------------ box constraints */

/* ref: include/misc/determine_bounds_type.hpp:27-57 */
void orc_determine_bounds_type(int vals_bound, size_t d, const double* lb, const double* ub, int* out)
{
    for (size_t i = 0; i < d; ++i) out[i] = 1;
    if (!vals_bound) return;
    for (size_t i = 0; i < d; ++i) {
        const int fl = isfinite(lb[i]), fu = isfinite(ub[i]);
        if (fl && fu) out[i] = 4;
        else if (fl && !fu) out[i] = 2;
        else if (!fl && fu) out[i] = 3;
    }
}

#define ORC_EPS_DBL 2.220446049250313e-16  /* ref: include/misc/mcmc_options.hpp:103 */

/* ref: include/misc/transform_vals.hpp:25-59 */
void orc_transform(const double* v, const int* bt, const double* lb, const double* ub, size_t d, double* out)
{
    for (size_t i = 0; i < d; ++i) {
        switch (bt[i]) {
        case 1: out[i] = v[i]; break;
        case 2: out[i] = orc_log(v[i] - lb[i] + ORC_EPS_DBL); break;
        case 3: out[i] = -orc_log(ub[i] - v[i] + ORC_EPS_DBL); break;
        case 4: out[i] = orc_log(v[i] - lb[i] + ORC_EPS_DBL) - orc_log(ub[i] - v[i] + ORC_EPS_DBL); break;
        }
    }
}

/* ref: include/misc/transform_vals.hpp:61-119 */
void orc_inv_transform(const double* v, const int* bt, const double* lb, const double* ub, size_t d, double* out)
{
    for (size_t i = 0; i < d; ++i) {
        switch (bt[i]) {
        case 1: out[i] = v[i]; break;
        case 2:
            if (!isfinite(v[i])) out[i] = lb[i] + ORC_EPS_DBL;
            else out[i] = lb[i] + ORC_EPS_DBL + orc_exp(v[i]);
            break;
        case 3:
            if (!isfinite(v[i])) out[i] = ub[i] - ORC_EPS_DBL;
            else out[i] = ub[i] - ORC_EPS_DBL - orc_exp(-v[i]);
            break;
        case 4:
            if (!isfinite(v[i])) {
                if (isnan(v[i])) out[i] = (ub[i] - lb[i]) / 2;
                else if (v[i] < 0.0) out[i] = lb[i] + ORC_EPS_DBL;
                else out[i] = ub[i] - ORC_EPS_DBL;
            } else {
                const double e = orc_exp(v[i]);
                out[i] = (lb[i] - ORC_EPS_DBL + (ub[i] + ORC_EPS_DBL) * e) / (1.0 + e);
                if (!isfinite(out[i])) out[i] = ub[i] - ORC_EPS_DBL;
            }
            break;
        }
    }
}

/* ref: include/misc/log_jacobian.hpp:25-58 */
double orc_log_jacobian(const double* v, const int* bt, const double* lb, const double* ub, size_t d)
{
    double ret = 0.0;
    for (size_t i = 0; i < d; ++i) {
        switch (bt[i]) {
        case 2: ret += v[i]; break;
        case 3: ret += -v[i]; break;
        case 4: {
            const double e = orc_exp(v[i]);
            if (isfinite(e)) ret += orc_log(ub[i] - lb[i]) + v[i] - 2 * orc_log(1 + e);
            else ret += orc_log(ub[i] - lb[i]) - v[i];
            break; }
        default: break;
        }
    }
    return ret;
}

/* ref: include/misc/inv_jacobian_adjust.hpp:25-56 (dense d x d with the diagonal filled) */
static void orc_inv_jacobian_adjust(const double* v, const int* bt, const double* lb, const double* ub,
                                    size_t d, double* J)
{
    for (size_t i = 0; i < d; ++i)
        for (size_t j = 0; j < d; ++j) J[i * d + j] = (i == j) ? 1.0 : 0.0;
    for (size_t i = 0; i < d; ++i) {
        switch (bt[i]) {
        case 2: J[i * d + i] = orc_exp(-v[i]); break;
        case 3: J[i * d + i] = orc_exp(v[i]); break;
        case 4: {
            const double e = orc_exp(v[i]);
            J[i * d + i] = ((e + 1) * (e + 1)) / (e * (ub[i] - lb[i]));
            break; }
        default: break;
        }
    }
}

/* ------------------------------------------------------------------ built-in targets (ours) */

static inline double orc_softplus(double eta)   /* log(1 + e^eta) */
{
    if (eta > 0.0) return eta + orc_log(1.0 + orc_exp(-eta));
    return orc_log(1.0 + orc_exp(eta));
}
static inline double orc_sigmoid(double eta)
{
    if (eta >= 0.0) return 1.0 / (1.0 + orc_exp(-eta));
    const double e = orc_exp(eta);
    return e / (1.0 + e);
}

/* one row of a blocked mat-vec: ((e0 + e1) + e2) + ... with e_k the dot over dimension block k (block size bs), inside a block nch
 * contiguous fma sub-chains summed left to right (the reduction order of logit_lds_kernel's eta, logistic_lds.hpp) */
static double orc_row_dot_blocked(const double* x, const double* th, size_t d, int nblk, size_t bs, int eta_chains)
{
    const int nch = eta_chains > 1 ? eta_chains : 1;
    const size_t sub = bs / (size_t)nch;
    if ((size_t)nblk * bs < d) return NAN;      /* as orc_dot_b: never truncate */
    double acc = 0.0;
    for (int k = 0; k < nblk; ++k) {
        double e = 0.0;
        for (int c = 0; c < nch; ++c) {
            const size_t lo = (size_t)k * bs + (size_t)c * sub;
            const size_t hi = (c == nch - 1) ? (size_t)(k + 1) * bs : lo + sub;
            double h = 0.0;
            for (size_t j = lo; j < d && j < hi; ++j) h = fma(x[j], th[j], h);
            e = (c == 0) ? h : e + h;
        }
        acc = (k == 0) ? e : acc + e;
    }
    return acc;
}

double orc_target_kernel(const double* th, double* grad_out, void* data)
{
    orc_target* t = (orc_target*)data;
    const size_t d = t->d;
    const int W = t->reduce_width;
    const int nblk = t->reduce_blocks;
    const size_t bs = t->reduce_block_size;
    if (grad_out) t->n_grad_calls++; else t->n_value_calls++;
    switch (t->kind) {
    case ORC_TARGET_GAUSS_ISO: {
        if (grad_out) for (size_t i = 0; i < d; ++i) grad_out[i] = -th[i];
        return -0.5 * orc_dot_b(th, th, d, W, nblk, bs);
    }
    case ORC_TARGET_GAUSS_DIAG: {
        double* w = (double*)malloc(d * sizeof(double));
        for (size_t i = 0; i < d; ++i) w[i] = t->prec[i] * th[i];
        if (grad_out) for (size_t i = 0; i < d; ++i) grad_out[i] = -w[i];
        const double r = -0.5 * orc_dot_b(th, w, d, W, nblk, bs);
        free(w);
        return r;
    }
    case ORC_TARGET_GAUSS_DENSE: {
        double* w = (double*)malloc(d * sizeof(double));
        if (t->prec_t) orc_gemv_t(t->prec_t, th, d, w); else orc_gemv(t->prec, th, d, w);
        if (grad_out) for (size_t i = 0; i < d; ++i) grad_out[i] = -w[i];
        const double r = -0.5 * orc_dot_b(th, w, d, W, nblk, bs);
        free(w);
        return r;
    }
    case ORC_TARGET_LOGISTIC: {
        /* log K = sum_r [y_r eta_r - log(1+e^eta_r)] - 0.5 |beta|^2,  eta = X beta
         * grad  = X^T (y - sigmoid(eta)) - beta */
        const size_t n = t->n_rows;
        double* eta = (double*)malloc(2 * n * sizeof(double));
        double* term = eta + n;
        for (size_t r = 0; r < n; ++r) {
            double acc = 0.0;
            const double* x = t->X + r * d;
            if (nblk <= 1 || bs == 0) {
                for (size_t j = 0; j < d; ++j) acc = fma(x[j], th[j], acc);
            } else acc = orc_row_dot_blocked(x, th, d, nblk, bs, t->eta_chains);
            eta[r] = acc;
            term[r] = t->y[r] * acc - orc_softplus(acc);
        }
        const double ll = orc_sum(term, n, W);
        const double ret = ll - 0.5 * orc_dot_b(th, th, d, W, nblk, bs);
        if (grad_out) {
            for (size_t r = 0; r < n; ++r) term[r] = t->y[r] - orc_sigmoid(eta[r]);
            for (size_t j = 0; j < d; ++j) {
                double acc = 0.0;
                for (size_t r = 0; r < n; ++r) acc = fma(t->X[r * d + j], term[r], acc);
                grad_out[j] = acc - th[j];
            }
        }
        free(eta);
        return ret;
    }
    case ORC_TARGET_NORMAL_MODEL: {
        /* the user code of the reference's examples (ref: examples/eigen/rmhmc_normal.cpp:44-73), restated with plain
         * sequential sums over the data:  m1 = sum (x - mu),  m2 = sum (x - mu)^2 */
        const double mu = th[0], sigma = th[1];
        const double n = (double)t->n_rows;
        double m1 = 0.0, m2 = 0.0;
        for (size_t r = 0; r < t->n_rows; ++r) {
            const double e = t->y[r] - mu;
            m1 = m1 + e;
            m2 = fma(e, e, m2);
        }
        const double s2 = sigma * sigma;
        const double ret = -(n * (0.5 * ORC_LOG_2PI + orc_log(sigma))) - m2 / (2.0 * s2);
        if (grad_out) {
            grad_out[0] = m1 / s2;
            grad_out[1] = m2 / (s2 * sigma) - n / sigma;
        }
        return ret;
    }
    default: return NAN;
    }
}

void orc_target_tensor(const double* th, double* G, double* dG, void* data)
{
    const orc_target* t = (const orc_target*)data;
    const size_t d = t->d;
    for (size_t i = 0; i < d * d; ++i) G[i] = 0.0;
    if (dG) for (size_t i = 0; i < d * d * d; ++i) dG[i] = 0.0;
    switch (t->kind) {
    case ORC_TARGET_GAUSS_ISO:  for (size_t i = 0; i < d; ++i) G[i * d + i] = 1.0; break;
    case ORC_TARGET_GAUSS_DIAG: for (size_t i = 0; i < d; ++i) G[i * d + i] = t->prec[i]; break;
    case ORC_TARGET_GAUSS_DENSE: for (size_t i = 0; i < d * d; ++i) G[i] = t->prec[i]; break;
    case ORC_TARGET_NORMAL_MODEL: {                       /* ref: examples/eigen/rmhmc_normal.cpp:75-106 */
        const double sigma = th[1];
        const double n = (double)t->n_rows;
        const double s2 = sigma * sigma;
        G[0] = n / s2;
        G[3] = (2.0 * n) / s2;
        if (dG) for (size_t i = 0; i < 4; ++i) dG[4 + i] = (-2.0 * G[i]) / sigma;    /* mat(1) = -2 G / sigma; mat(0) = 0 */
        break;
    }
    case ORC_TARGET_LOGISTIC: {
        /* Fisher information of the Bayesian logistic regression plus the prior precision (the metric of Girolami & Calderhead's
         * RM-HMC logistic example; ours -- the reference ships no tensor for it):  G = X^T diag(lam) X + I,  lam_k = s_k (1 - s_k),
         * s_k = sigmoid(eta_k);  dG/dbeta_i = X^T diag(lam_k (1 - 2 s_k) X_ki) X.  Rows k ascending, one fma per row and entry. */
        for (size_t k = 0; k < t->n_rows; ++k) {
            const double* x = t->X + k * d;
            double eta = 0.0;
            for (size_t j = 0; j < d; ++j) eta = fma(x[j], th[j], eta);
            const double sg = orc_sigmoid(eta);
            const double lam = sg * (1.0 - sg);
            const double dl = lam * (1.0 - 2.0 * sg);
            for (size_t r = 0; r < d; ++r)
                for (size_t c = 0; c < d; ++c) {
                    const double xx = x[r] * x[c];
                    G[r * d + c] = fma(xx, lam, G[r * d + c]);
                    if (dG) for (size_t i = 0; i < d; ++i) dG[i * d * d + r * d + c] = fma(xx, dl * x[i], dG[i * d * d + r * d + c]);
                }
        }
        for (size_t r = 0; r < d; ++r) G[r * d + r] = G[r * d + r] + 1.0;
        break;
    }
    default: for (size_t i = 0; i < d * d; ++i) G[i] = NAN;
    }
}

/* ------------------------------------------------------------------ shared sampler context */

typedef struct orc_ctx {
    size_t d;
    orc_kernel_fn kernel;
    void* data;
    int vals_bound;
    int* btype;
    const double* lb;
    const double* ub;
    double* precond;       /* d*d */
    double* inv_precond;
    double* sqrt_precond;
    int W;
    int nblk;
    size_t bs;
    uint64_t seed, chain;
    uint64_t n_leap;
} orc_ctx;

static double* dvec(size_t n) { return (double*)malloc((n ? n : 1) * sizeof(double)); }

static void ctx_init(orc_ctx* c, size_t d, orc_kernel_fn kernel, void* data, const orc_settings* s, int need_inv)
{
    memset(c, 0, sizeof(*c));
    c->d = d; c->kernel = kernel; c->data = data;
    c->vals_bound = s->vals_bound; c->lb = s->lower_bounds; c->ub = s->upper_bounds;
    c->W = s->reduce_width > 0 ? s->reduce_width : 1;
    c->nblk = s->reduce_blocks; c->bs = s->reduce_block_size;
    c->seed = s->rng_seed_value; c->chain = s->chain_id;
    /* ref: src/hmc.cpp:57-59 -- user matrix if it has d*d elements, else identity; dense either way */
    c->precond = dvec(d * d);
    if (s->precond_mat) memcpy(c->precond, s->precond_mat, d * d * sizeof(double));
    else for (size_t i = 0; i < d; ++i) for (size_t j = 0; j < d; ++j) c->precond[i * d + j] = (i == j) ? 1.0 : 0.0;
    if (need_inv) { c->inv_precond = dvec(d * d); orc_inv(c->precond, d, c->inv_precond); }
    c->sqrt_precond = dvec(d * d);
    orc_chol_lower(c->precond, d, c->sqrt_precond);
    c->btype = (int*)malloc((d ? d : 1) * sizeof(int));
    orc_determine_bounds_type(c->vals_bound, d, c->lb, c->ub, c->btype);   /* ref: src/hmc.cpp:66 */
}

static void ctx_free(orc_ctx* c)
{
    free(c->precond); free(c->inv_precond); free(c->sqrt_precond); free(c->btype);
}

/* box_log_kernel lambda (ref: src/hmc.cpp:84-95, src/mala.cpp:84-95, src/nuts.cpp:93-104):
 * always passes grad_out = nullptr */
static double box_log_kernel(orc_ctx* c, const double* vals)
{
    if (c->vals_bound) {
        double* vi = dvec(c->d);
        orc_inv_transform(vals, c->btype, c->lb, c->ub, c->d, vi);
        const double r = c->kernel(vi, NULL, c->data) + orc_log_jacobian(vals, c->btype, c->lb, c->ub, c->d);
        free(vi);
        return r;
    }
    return c->kernel(vals, NULL, c->data);
}

/* mntm_update_fn lambda (ref: src/hmc.cpp:99-128, src/nuts.cpp:108-135):
 * returns mntm + step * [J] grad / 2, written into out (may alias mntm) */
static void mntm_update(orc_ctx* c, const double* pos, const double* mntm, double step, double* out)
{
    const size_t d = c->d;
    double* grad = dvec(d);                                            /* hmc.cpp:105 */
    if (c->vals_bound) {
        double* pi = dvec(d);
        double* J = dvec(d * d);
        double* jg = dvec(d);
        orc_inv_transform(pos, c->btype, c->lb, c->ub, d, pi);         /* :108 */
        c->kernel(pi, grad, c->data);                                  /* :110 */
        orc_inv_jacobian_adjust(pos, c->btype, c->lb, c->ub, d, J);    /* :114 */
        orc_gemv(J, grad, d, jg);
        for (size_t i = 0; i < d; ++i) out[i] = mntm[i] + (step * jg[i]) / 2.0;   /* :122 */
        free(pi); free(J); free(jg);
    } else {
        c->kernel(pos, grad, c->data);                                 /* :124 */
        for (size_t i = 0; i < d; ++i) out[i] = mntm[i] + (step * grad[i]) / 2.0; /* :126 */
    }
    free(grad);
}

/* leap_frog_fn (ref: src/nuts.cpp:139-154; identical body inline at src/hmc.cpp:164-176) */
static void leap_frog(orc_ctx* c, double step, size_t n_steps, double* draw, double* mntm)
{
    const size_t d = c->d;
    double* mp = dvec(d);
    for (size_t k = 0; k < n_steps; ++k) {
        mntm_update(c, draw, mntm, step, mntm);                        /* first half-step  */
        orc_gemv(c->inv_precond, mntm, d, mp);
        for (size_t i = 0; i < d; ++i) draw[i] = draw[i] + step * mp[i];   /* hmc.cpp:171 */
        mntm_update(c, draw, mntm, step, mntm);                        /* second half-step */
        c->n_leap++;
    }
    free(mp);
}

/* K = p . (Minv p) / 2 (ref: src/hmc.cpp:160,184) */
static double kinetic(orc_ctx* c, const double* mntm)
{
    double* mp = dvec(c->d);
    orc_gemv(c->inv_precond, mntm, c->d, mp);
    const double k = orc_dot_b(mntm, mp, c->d, c->W, c->nblk, c->bs) / 2.0;
    free(mp);
    return k;
}

static void store_row(double* draws_out, size_t row, size_t d, const double* v)
{
    memcpy(draws_out + row * d, v, d * sizeof(double));
}

static void epilogue_inv_transform(orc_ctx* c, double* draws_out, size_t n_keep)
{
    /* ref: src/hmc.cpp:211-218 */
    if (!c->vals_bound) return;
    double* t = dvec(c->d);
    for (size_t r = 0; r < n_keep; ++r) {
        orc_inv_transform(draws_out + r * c->d, c->btype, c->lb, c->ub, c->d, t);
        memcpy(draws_out + r * c->d, t, c->d * sizeof(double));
    }
    free(t);
}

/* ------------------------------------------------------------------ HMC */


/* Mode B of the CPU baseline (orc_settings.work_mode = 1): hmc_impl with the work a tuned CPU
 * implementation would do -- same arithmetic on the same values in the same order, so the draws are the bits of orc_hmc
 * (asserted by tests/test_oracle_samplers.py).  Unbounded runs. */
static int orc_hmc_mode_b(const double* initial_vals, size_t d, orc_kernel_fn kernel, void* data,
                          const orc_settings* s, double* draws_out, orc_stats* st)
{
    const size_t n_burnin = s->n_burnin_draws, n_keep = s->n_keep_draws, n_total = n_burnin + n_keep;
    const double eps = s->step_size;
    const size_t L = s->n_leap_steps;
    const int W = s->reduce_width > 0 ? s->reduce_width : 1;
    const int identity = (s->precond_mat == NULL);
    double *Minv = NULL, *Lc = NULL;
    if (!identity) {
        Minv = dvec(d * d); Lc = dvec(d * d);
        orc_inv(s->precond_mat, d, Minv);
        orc_chol_lower(s->precond_mat, d, Lc);
    }
    double* buf = dvec(7 * d);
    double *prev_draw = buf, *new_draw = buf + d, *mntm = buf + 2 * d, *z = buf + 3 * d, *mp = buf + 4 * d,
           *g_prev = buf + 5 * d, *g = buf + 6 * d;
    memcpy(prev_draw, initial_vals, d * sizeof(double));
    double prev_U = -kernel(prev_draw, g_prev, data);       /* value and gradient at the current state, kept until it changes */
    size_t n_accept = 0;
    uint64_t n_leap = 0;
    for (size_t draw_ind = 0; draw_ind < n_total; ++draw_ind) {
        orc_rng_normal_vec(s->rng_seed_value, s->chain_id, (uint32_t)draw_ind, ORC_STREAM_NORMAL, d, z);
        if (identity) memcpy(mntm, z, d * sizeof(double)); else orc_gemv(Lc, z, d, mntm);
        const double* pm = mntm;
        if (!identity) { orc_gemv(Minv, mntm, d, mp); pm = mp; }
        const double prev_K = orc_dot_b(mntm, pm, d, W, s->reduce_blocks, s->reduce_block_size) / 2.0;
        memcpy(new_draw, prev_draw, d * sizeof(double));
        memcpy(g, g_prev, d * sizeof(double));
        double val = -prev_U;
        for (size_t k = 0; k < L; ++k) {
            for (size_t i = 0; i < d; ++i) mntm[i] = mntm[i] + (eps * g[i]) / 2.0;
            if (!identity) orc_gemv(Minv, mntm, d, mp);
            for (size_t i = 0; i < d; ++i) new_draw[i] = new_draw[i] + eps * pm[i];
            val = kernel(new_draw, g, data);
            for (size_t i = 0; i < d; ++i) mntm[i] = mntm[i] + (eps * g[i]) / 2.0;
            n_leap++;
        }
        double prop_U = -val;
        if (!isfinite(prop_U)) prop_U = INFINITY;
        if (!identity) orc_gemv(Minv, mntm, d, mp);
        const double prop_K = orc_dot_b(mntm, pm, d, W, s->reduce_blocks, s->reduce_block_size) / 2.0;
        const double x = -(prop_U + prop_K) + (prev_U + prev_K);
        const double comp_val = (x < 0.01) ? x : 0.01;
        const double u = orc_rng_uniform(s->rng_seed_value, s->chain_id, (uint32_t)draw_ind, 0);
        int acc = 0;
        if (u < orc_exp(comp_val)) {
            memcpy(prev_draw, new_draw, d * sizeof(double));
            memcpy(g_prev, g, d * sizeof(double));
            prev_U = prop_U;
            acc = 1;
            if (draw_ind >= n_burnin) n_accept++;
        }
        if (draw_ind >= n_burnin) store_row(draws_out, draw_ind - n_burnin, d, prev_draw);
        if (st && st->accept_trace) st->accept_trace[draw_ind] = (uint8_t)acc;
    }
    if (st) { st->n_accept_draws = n_accept; st->n_leapfrogs = n_leap; st->final_step_size = eps; }
    free(buf); free(Minv); free(Lc);
    return 0;
}

/* ref: src/hmc.cpp:30-227 */
int orc_hmc(const double* initial_vals, size_t d, orc_kernel_fn kernel, void* data,
            const orc_settings* s, double* draws_out, orc_stats* st)
{
    if (s->work_mode == 1 && !s->vals_bound) return orc_hmc_mode_b(initial_vals, d, kernel, data, s, draws_out, st);
    orc_ctx c;
    ctx_init(&c, d, kernel, data, s, 1);
    const size_t n_burnin = s->n_burnin_draws, n_keep = s->n_keep_draws, n_total = n_burnin + n_keep;
    const double step_size = s->step_size;
    const size_t n_leap_steps = s->n_leap_steps;

    double* first_draw = dvec(d);
    memcpy(first_draw, initial_vals, d * sizeof(double));
    if (c.vals_bound) orc_transform(initial_vals, c.btype, c.lb, c.ub, d, first_draw);   /* :134-136 */

    double prev_U = -box_log_kernel(&c, first_draw);                    /* :140 */
    double prop_U = prev_U, prop_K, prev_K;
    double* prev_draw = dvec(d); memcpy(prev_draw, first_draw, d * sizeof(double));
    double* new_draw = dvec(d);  memcpy(new_draw, first_draw, d * sizeof(double));
    double* new_mntm = dvec(d);
    double* rand_vec = dvec(d);
    size_t n_accept = 0;

    for (size_t draw_ind = 0; draw_ind < n_total; ++draw_ind) {         /* :155 */
        orc_rng_normal_vec(c.seed, c.chain, (uint32_t)draw_ind, ORC_STREAM_NORMAL, d, rand_vec);   /* :156 */
        orc_gemv(c.sqrt_precond, rand_vec, d, new_mntm);                /* :158 */
        prev_K = kinetic(&c, new_mntm);                                 /* :160 */
        memcpy(new_draw, prev_draw, d * sizeof(double));                /* :162 */
        leap_frog(&c, step_size, n_leap_steps, new_draw, new_mntm);     /* :164-176 */
        prop_U = -box_log_kernel(&c, new_draw);                         /* :178 */
        if (!isfinite(prop_U)) prop_U = INFINITY;                       /* :180-182 */
        prop_K = kinetic(&c, new_mntm);                                 /* :184 */
        const double x = -(prop_U + prop_K) + (prev_U + prev_K);
        const double comp_val = (x < 0.01) ? x : 0.01;                  /* std::min(0.01, x) :188 */
        const double z = orc_rng_uniform(c.seed, c.chain, (uint32_t)draw_ind, 0);   /* :189 */
        int acc = 0;
        if (z < orc_exp(comp_val)) {                                    /* :191 */
            memcpy(prev_draw, new_draw, d * sizeof(double));
            prev_U = prop_U;
            prev_K = prop_K;
            acc = 1;
            if (draw_ind >= n_burnin) { store_row(draws_out, draw_ind - n_burnin, d, new_draw); n_accept++; }
        } else {
            if (draw_ind >= n_burnin) store_row(draws_out, draw_ind - n_burnin, d, prev_draw);
        }
        if (st && st->accept_trace) st->accept_trace[draw_ind] = (uint8_t)acc;
    }
    (void)prev_K;
    epilogue_inv_transform(&c, draws_out, n_keep);
    if (st) { st->n_accept_draws = n_accept; st->n_leapfrogs = c.n_leap; st->final_step_size = step_size; }
    free(first_draw); free(prev_draw); free(new_draw); free(new_mntm); free(rand_vec);
    ctx_free(&c);
    return 0;
}

/* ------------------------------------------------------------------ MALA */

typedef struct mala_fact {     /* factorisation of Sigma = eps^2 [J] M */
    double* Sinv;
    double  log_det;
} mala_fact;

/* mala_mean_fn lambda (ref: src/mala.cpp:97-125).  J_out (d*d) filled when bounded. */
static void mala_mean(orc_ctx* c, const double* vals, double step, double* J_out, double* out)
{
    const size_t d = c->d;
    double* grad = dvec(d);
    double* t = dvec(d);
    const double s2 = step * step;
    if (c->vals_bound) {
        double* vi = dvec(d);
        double* J = J_out ? J_out : dvec(d * d);
        double* JM = dvec(d * d);
        orc_inv_transform(vals, c->btype, c->lb, c->ub, d, vi);
        c->kernel(vi, grad, c->data);                                   /* :109 */
        orc_inv_jacobian_adjust(vals, c->btype, c->lb, c->ub, d, J);    /* :113 */
        orc_matmul(J, c->precond, d, JM);
        for (size_t i = 0; i < d * d; ++i) JM[i] = s2 * JM[i];
        orc_gemv(JM, grad, d, t);
        for (size_t i = 0; i < d; ++i) out[i] = vals[i] + t[i] / 2.0;   /* :121 */
        free(vi); free(JM);
        if (!J_out) free(J);
    } else {
        c->kernel(vals, grad, c->data);                                 /* :123 */
        orc_gemv(c->precond, grad, d, t);
        for (size_t i = 0; i < d; ++i) out[i] = vals[i] + (s2 * t[i]) / 2.0;      /* :123 */
    }
    free(grad); free(t);
}

static void mala_factorise(const double* Sigma, size_t d, mala_fact* f)
{
    double* L = dvec(d * d);
    f->Sinv = dvec(d * d);
    orc_inv(Sigma, d, f->Sinv);
    orc_chol_lower(Sigma, d, L);
    f->log_det = orc_log_det_from_chol(L, d);
    free(L);
}

/* mala_prop_adjustment (ref: include/mcmc/mala.ipp:30-70) */
static double mala_prop_adjustment(orc_ctx* c, const double* prop_vals, const double* prev_vals, double step,
                                   const mala_fact* hoisted)
{
    const size_t d = c->d;
    const double s2 = step * step;                                      /* mala.ipp:41 */
    double* prop_mean = dvec(d);
    double* prev_mean = dvec(d);
    double* Sigma = dvec(d * d);
    double* scratch = dvec(2 * d);
    double ret;
    if (c->vals_bound) {
        double* Jprop = dvec(d * d);
        double* Jprev = dvec(d * d);
        mala_mean(c, prop_vals, step, Jprop, prop_mean);                /* :49 */
        mala_mean(c, prev_vals, step, Jprev, prev_mean);                /* :50 */
        orc_matmul(Jprop, c->precond, d, Sigma);                        /* :52-53: prop_inv_jacob in BOTH terms */
        for (size_t i = 0; i < d * d; ++i) Sigma[i] = s2 * Sigma[i];
        /* the reference factorises inside each dmvnorm call; bits are the same */
        mala_fact f; mala_factorise(Sigma, d, &f);
        ret = orc_dmvnorm_core_b(prev_vals, prop_mean, d, f.Sinv, f.log_det, c->W, c->nblk, c->bs, scratch, scratch + d)
            - orc_dmvnorm_core_b(prop_vals, prev_mean, d, f.Sinv, f.log_det, c->W, c->nblk, c->bs, scratch, scratch + d);
        free(f.Sinv); free(Jprop); free(Jprev);
    } else {
        mala_mean(c, prop_vals, step, NULL, prop_mean);                 /* :60 */
        mala_mean(c, prev_vals, step, NULL, prev_mean);                 /* :61 */
        if (hoisted) {
            ret = orc_dmvnorm_core_b(prev_vals, prop_mean, d, hoisted->Sinv, hoisted->log_det, c->W, c->nblk, c->bs, scratch, scratch + d)
                - orc_dmvnorm_core_b(prop_vals, prev_mean, d, hoisted->Sinv, hoisted->log_det, c->W, c->nblk, c->bs, scratch, scratch + d);
        } else {
            for (size_t i = 0; i < d * d; ++i) Sigma[i] = s2 * c->precond[i];
            mala_fact f1; mala_factorise(Sigma, d, &f1);                /* :63 (inside dmvnorm) */
            const double a = orc_dmvnorm_core_b(prev_vals, prop_mean, d, f1.Sinv, f1.log_det, c->W, c->nblk, c->bs, scratch, scratch + d);
            free(f1.Sinv);
            mala_fact f2; mala_factorise(Sigma, d, &f2);                /* :64 (inside dmvnorm) */
            const double b = orc_dmvnorm_core_b(prop_vals, prev_mean, d, f2.Sinv, f2.log_det, c->W, c->nblk, c->bs, scratch, scratch + d);
            free(f2.Sinv);
            ret = a - b;
        }
    }
    free(prop_mean); free(prev_mean); free(Sigma); free(scratch);
    return ret;
}

/* test hook (tests/test_oracle_structure.py): mala_prop_adjustment(prop, prev) of a built-in target under settings s */
double orc_mala_prop_adjustment_eval(orc_target* t, const orc_settings* s, const double* prop_vals, const double* prev_vals)
{
    orc_ctx c;
    ctx_init(&c, t->d, orc_target_kernel, t, s, 0);
    mala_fact hoisted; hoisted.Sinv = NULL; hoisted.log_det = 0.0;
    if (s->hoist_factorizations && !c.vals_bound) {
        double* Sigma = dvec(t->d * t->d);
        const double s2 = s->step_size * s->step_size;
        for (size_t i = 0; i < t->d * t->d; ++i) Sigma[i] = s2 * c.precond[i];
        mala_factorise(Sigma, t->d, &hoisted);
        free(Sigma);
    }
    const double r = mala_prop_adjustment(&c, prop_vals, prev_vals, s->step_size, hoisted.Sinv ? &hoisted : NULL);
    free(hoisted.Sinv);
    ctx_free(&c);
    return r;
}

/* ref: src/mala.cpp:30-208 */
int orc_mala(const double* initial_vals, size_t d, orc_kernel_fn kernel, void* data,
             const orc_settings* s, double* draws_out, orc_stats* st)
{
    orc_ctx c;
    ctx_init(&c, d, kernel, data, s, 0);
    const size_t n_burnin = s->n_burnin_draws, n_keep = s->n_keep_draws, n_total = n_burnin + n_keep;
    const double step_size = s->step_size;

    mala_fact hoisted; hoisted.Sinv = NULL; hoisted.log_det = 0.0;
    if (s->hoist_factorizations && !c.vals_bound) {
        double* Sigma = dvec(d * d);
        const double s2 = step_size * step_size;
        for (size_t i = 0; i < d * d; ++i) Sigma[i] = s2 * c.precond[i];
        mala_factorise(Sigma, d, &hoisted);
        free(Sigma);
    }

    double* first_draw = dvec(d);
    memcpy(first_draw, initial_vals, d * sizeof(double));
    if (c.vals_bound) orc_transform(initial_vals, c.btype, c.lb, c.ub, d, first_draw);   /* :132-134 */

    double prev_LP = box_log_kernel(&c, first_draw);                    /* :138 */
    double prop_LP = prev_LP;
    double* prev_draw = dvec(d); memcpy(prev_draw, first_draw, d * sizeof(double));
    double* new_draw = dvec(d);  memcpy(new_draw, first_draw, d * sizeof(double));
    double* rand_vec = dvec(d);
    double* mean_vec = dvec(d);
    double* t = dvec(d);
    size_t n_accept = 0;

    for (size_t draw_ind = 0; draw_ind < n_total; ++draw_ind) {         /* :149 */
        orc_rng_normal_vec(c.seed, c.chain, (uint32_t)draw_ind, ORC_STREAM_NORMAL, d, rand_vec);   /* :150 */
        if (c.vals_bound) {                                             /* :152-157 */
            double* J = dvec(d * d);
            double* CJ = dvec(d * d);
            double* T = dvec(d * d);
            mala_mean(&c, prev_draw, step_size, J, mean_vec);
            orc_chol_lower(J, d, CJ);
            orc_matmul(CJ, c.sqrt_precond, d, T);
            for (size_t i = 0; i < d * d; ++i) T[i] = step_size * T[i];
            orc_gemv(T, rand_vec, d, t);
            for (size_t i = 0; i < d; ++i) new_draw[i] = mean_vec[i] + t[i];
            free(J); free(CJ); free(T);
        } else {                                                        /* :159 */
            mala_mean(&c, prev_draw, step_size, NULL, mean_vec);
            orc_gemv(c.sqrt_precond, rand_vec, d, t);
            for (size_t i = 0; i < d; ++i) new_draw[i] = mean_vec[i] + step_size * t[i];
        }
        prop_LP = box_log_kernel(&c, new_draw);                         /* :162 */
        if (!isfinite(prop_LP)) prop_LP = -INFINITY;                    /* :164-166 */
        const double adj = mala_prop_adjustment(&c, new_draw, prev_draw, step_size,
                                                hoisted.Sinv ? &hoisted : NULL);
        const double x = prop_LP - prev_LP + adj;
        const double comp_val = (x < 0.01) ? x : 0.01;                  /* :170 */
        const double z = orc_rng_uniform(c.seed, c.chain, (uint32_t)draw_ind, 0);   /* :171 */
        int acc = 0;
        if (z < orc_exp(comp_val)) {                                    /* :173 */
            memcpy(prev_draw, new_draw, d * sizeof(double));
            prev_LP = prop_LP;
            acc = 1;
            if (draw_ind >= n_burnin) { store_row(draws_out, draw_ind - n_burnin, d, new_draw); n_accept++; }
        } else {
            if (draw_ind >= n_burnin) store_row(draws_out, draw_ind - n_burnin, d, prev_draw);
        }
        if (st && st->accept_trace) st->accept_trace[draw_ind] = (uint8_t)acc;
    }
    epilogue_inv_transform(&c, draws_out, n_keep);
    if (st) { st->n_accept_draws = n_accept; st->n_leapfrogs = 0; st->final_step_size = step_size; }
    free(hoisted.Sinv);
    free(first_draw); free(prev_draw); free(new_draw); free(rand_vec); free(mean_vec); free(t);
    ctx_free(&c);
    return 0;
}

/* ------------------------------------------------------------------ RWMH */

/* ref: src/rwmh.cpp:30-175.  settings: step_size carries rwmh_settings.par_scale (mcmc_structs.hpp:145), precond_mat carries
 * rwmh_settings.cov_mat (:146, identity when absent, rwmh.cpp:58). */
int orc_rwmh(const double* initial_vals, size_t d, orc_kernel_fn kernel, void* data,
             const orc_settings* s, double* draws_out, orc_stats* st)
{
    orc_ctx c;
    ctx_init(&c, d, kernel, data, s, 0);
    const size_t n_burnin = s->n_burnin_draws, n_keep = s->n_keep_draws, n_total = n_burnin + n_keep;
    const double par_scale = s->step_size;

    double* first_draw = dvec(d);
    memcpy(first_draw, initial_vals, d * sizeof(double));
    if (c.vals_bound) orc_transform(initial_vals, c.btype, c.lb, c.ub, d, first_draw);   /* :105-107 */
    double prev_LP = box_log_kernel(&c, first_draw);                    /* :113 */
    double prop_LP = prev_LP;
    double* prev_draw = dvec(d); memcpy(prev_draw, first_draw, d * sizeof(double));
    double* new_draw = dvec(d);  memcpy(new_draw, first_draw, d * sizeof(double));
    double* cov_chol = dvec(d * d);                                     /* par_scale * CHOL_LOWER(cov) :119 */
    for (size_t i = 0; i < d * d; ++i) cov_chol[i] = par_scale * c.sqrt_precond[i];
    double* rand_vec = dvec(d);
    double* t = dvec(d);
    size_t n_accept = 0;

    for (size_t draw_ind = 0; draw_ind < n_total; ++draw_ind) {         /* :123 */
        orc_rng_normal_vec(c.seed, c.chain, (uint32_t)draw_ind, ORC_STREAM_NORMAL, d, rand_vec);   /* :124 */
        orc_gemv(cov_chol, rand_vec, d, t);
        for (size_t i = 0; i < d; ++i) new_draw[i] = prev_draw[i] + t[i];            /* :126 */
        prop_LP = box_log_kernel(&c, new_draw);                         /* :128 */
        if (!isfinite(prop_LP)) prop_LP = -INFINITY;                    /* :130-132 */
        const double x = prop_LP - prev_LP;
        const double comp_val = (x < 0.0) ? x : 0.0;                    /* std::min(0.0, x): NaN -> 0 :136 */
        const double z = orc_rng_uniform(c.seed, c.chain, (uint32_t)draw_ind, 0);    /* :137 */
        int acc = 0;
        if (z < orc_exp(comp_val)) {                                    /* :139 */
            memcpy(prev_draw, new_draw, d * sizeof(double));
            prev_LP = prop_LP;
            acc = 1;
            if (draw_ind >= n_burnin) n_accept++;
        }
        if (draw_ind >= n_burnin) store_row(draws_out, draw_ind - n_burnin, d, prev_draw);   /* :148-150 */
        if (st && st->accept_trace) st->accept_trace[draw_ind] = (uint8_t)acc;
    }
    epilogue_inv_transform(&c, draws_out, n_keep);                      /* :157-166 */
    if (st) { st->n_accept_draws = n_accept; st->n_leapfrogs = 0; st->final_step_size = par_scale; }
    free(first_draw); free(prev_draw); free(new_draw); free(cov_chol); free(rand_vec); free(t);
    ctx_free(&c);
    return 0;
}

/* ------------------------------------------------------------------ NUTS */

/* ref: include/mcmc/nuts.ipp:30-93 */
static double nuts_find_initial_step_size(orc_ctx* c, const double* draw_vec, const double* mntm_vec)
{
    const size_t d = c->d;
    double step_size = 1.0;                                             /* :40 */
    double prev_U = -box_log_kernel(c, draw_vec);                       /* :44 */
    if (!isfinite(prev_U)) prev_U = INFINITY;
    const double prev_K = kinetic(c, mntm_vec);                         /* :51 */
    double* new_draw = dvec(d); memcpy(new_draw, draw_vec, d * sizeof(double));
    double* new_mntm = dvec(d); memcpy(new_mntm, mntm_vec, d * sizeof(double));
    leap_frog(c, step_size, 1, new_draw, new_mntm);                     /* :58 */
    double prop_U = -box_log_kernel(c, new_draw);
    if (!isfinite(prop_U)) prop_U = INFINITY;
    double prop_K = kinetic(c, new_mntm);                               /* :66 */
    const double log_half = orc_log(0.5), neg_log2 = -orc_log(2.0);
    int a_val = 2 * (-(prop_U + prop_K) + (prev_U + prev_K) > log_half) - 1;    /* :70 */
    int check_cond = (-(prop_U + prop_K) + (prev_U + prev_K)) > neg_log2;       /* :71 */
    while (check_cond) {
        step_size *= (a_val == 1) ? 2.0 : 0.5;                          /* std::pow(2, a_val) :74 */
        leap_frog(c, step_size, 1, new_draw, new_mntm);                 /* :76 continues from moved state */
        prop_U = -box_log_kernel(c, new_draw);
        if (!isfinite(prop_U)) prop_U = INFINITY;
        prop_K = kinetic(c, new_mntm);
        a_val = 2 * ((-(prop_U + prop_K) + (prev_U + prev_K)) > log_half) - 1;  /* :88 */
        check_cond = (-(prop_U + prop_K) + (prev_U + prev_K)) > neg_log2;       /* :89 */
    }
    free(new_draw); free(new_mntm);
    return step_size;
}

/* ref: include/mcmc/nuts.ipp:97-241 -- recursive, argument plumbing kept literally
 * (including the crossed edge outputs of the second-half calls, :195 and :207). */
static void nuts_build_tree(orc_ctx* c, int direction_val, double step_size, double log_rand_val,
                            double prev_U, double prev_K, const double* draw_vec, const double* mntm_vec,
                            size_t tree_depth,
                            double* new_draw, double* new_draw_pos, double* new_draw_neg,
                            double* new_mntm_pos, double* new_mntm_neg,
                            size_t* n_val, size_t* s_val, double* alpha_val, size_t* n_alpha_val,
                            uint32_t draw_ind, uint32_t* uslot)
{
    const size_t d = c->d;
    const size_t nb = d * sizeof(double);
    const double max_tuning_par = 1000;                                 /* :124 */
    if (tree_depth == 0) {
        double* new_mntm = dvec(d);
        /* draw_vec may alias an output buffer of the caller's caller: copy first */
        double* start = dvec(d); memcpy(start, draw_vec, nb);
        memcpy(new_mntm, mntm_vec, nb);                                 /* :128 */
        memcpy(new_draw, start, nb);                                    /* :127 */
        free(start);
        leap_frog(c, direction_val * step_size, 1, new_draw, new_mntm); /* :132 */
        double prop_U = -box_log_kernel(c, new_draw);                   /* :134 */
        if (!isfinite(prop_U)) prop_U = INFINITY;
        const double prop_K = kinetic(c, new_mntm);                     /* :140 */
        *n_val = (log_rand_val <= -prop_U - prop_K);                    /* :146 */
        *s_val = (log_rand_val < max_tuning_par - prop_U - prop_K);     /* :147 */
        memcpy(new_draw_pos, new_draw, nb);                             /* :151-155 */
        memcpy(new_draw_neg, new_draw, nb);
        memcpy(new_mntm_pos, new_mntm, nb);
        memcpy(new_mntm_neg, new_mntm, nb);
        const double dd = -(prop_U + prop_K) + (prev_U + prev_K);
        *alpha_val = orc_exp((dd < 0.0) ? dd : 0.0);                    /* std::min(0, dd) = (dd < 0) ? dd : 0, so NaN -> 0 :157 */
        *n_alpha_val = 1;
        free(new_mntm);
    } else {
        size_t n_p_val, s_p_val, n_alpha_p_val;
        double alpha_p_val;
        double* new_draw_p = dvec(d);
        nuts_build_tree(c, direction_val, step_size, log_rand_val, prev_U, prev_K, draw_vec, mntm_vec,
                        tree_depth - 1, new_draw_p, new_draw_pos, new_draw_neg, new_mntm_pos, new_mntm_neg,
                        &n_p_val, &s_p_val, &alpha_p_val, &n_alpha_p_val, draw_ind, uslot);   /* :166-171 */
        if (s_p_val == 1) {
            size_t n_pp_val, s_pp_val, n_alpha_pp_val;
            double alpha_pp_val;
            double* new_draw_pp = dvec(d);
            double* dummy_draw = dvec(d);
            double* dummy_mntm = dvec(d);
            double* edge_draw = dvec(d);
            double* edge_mntm = dvec(d);
            if (direction_val == -1) {
                memcpy(dummy_draw, new_draw_pos, nb);                   /* :186-189 */
                memcpy(dummy_mntm, new_mntm_pos, nb);
                memcpy(edge_draw, new_draw_neg, nb);
                memcpy(edge_mntm, new_mntm_neg, nb);
                nuts_build_tree(c, direction_val, step_size, log_rand_val, prev_U, prev_K, edge_draw, edge_mntm,
                                tree_depth - 1, new_draw_pp, new_draw_neg, dummy_draw, new_mntm_neg, dummy_mntm,
                                &n_pp_val, &s_pp_val, &alpha_pp_val, &n_alpha_pp_val, draw_ind, uslot);   /* :191-196 */
            } else {
                memcpy(dummy_draw, new_draw_neg, nb);                   /* :198-201 */
                memcpy(dummy_mntm, new_mntm_neg, nb);
                memcpy(edge_draw, new_draw_pos, nb);
                memcpy(edge_mntm, new_mntm_pos, nb);
                nuts_build_tree(c, direction_val, step_size, log_rand_val, prev_U, prev_K, edge_draw, edge_mntm,
                                tree_depth - 1, new_draw_pp, dummy_draw, new_draw_pos, dummy_mntm, new_mntm_pos,
                                &n_pp_val, &s_pp_val, &alpha_pp_val, &n_alpha_pp_val, draw_ind, uslot);   /* :203-208 */
            }
            const double prob_val = (double)n_pp_val / (double)(n_p_val + n_pp_val);        /* :212 */
            const double z = orc_rng_uniform(c->seed, c->chain, draw_ind, (*uslot)++);      /* :213 */
            if (z < prob_val) memcpy(new_draw_p, new_draw_pp, nb);      /* :215-217 (0/0 = NaN keeps) */
            n_p_val += n_pp_val;                                        /* :220-222 */
            alpha_p_val += alpha_pp_val;
            n_alpha_p_val += n_alpha_pp_val;
            double* diff = dvec(d);
            for (size_t i = 0; i < d; ++i) diff[i] = new_draw_pos[i] - new_draw_neg[i];
            const int check_val_1 = orc_dot_b(diff, new_mntm_neg, d, c->W, c->nblk, c->bs) >= 0.0;            /* :226 */
            const int check_val_2 = orc_dot_b(diff, new_mntm_pos, d, c->W, c->nblk, c->bs) >= 0.0;            /* :227 */
            s_p_val = s_pp_val * (size_t)check_val_1 * (size_t)check_val_2;                 /* :229 */
            free(diff); free(new_draw_pp); free(dummy_draw); free(dummy_mntm); free(edge_draw); free(edge_mntm);
        }
        *n_val = n_p_val;                                               /* :234-239 */
        *s_val = s_p_val;
        *alpha_val = alpha_p_val;
        *n_alpha_val = n_alpha_p_val;
        memcpy(new_draw, new_draw_p, nb);
        free(new_draw_p);
    }
}


/* ------------------------------------------------------------------ NUTS, one doubling evaluated with a MEMOISED trajectory
 *
 * Not a second algorithm: the SAME doubling as nuts_build_tree above, evaluated in another order.  From the argument plumbing of
 * ref: include/mcmc/nuts.ipp:166-209 (the crossed edge outputs of the second-half calls, :195 and :207) leaf i of a doubling (i > 0,
 * c = ctz(i)) starts from the result of leaf i - 1 (c <= 1) or of leaf i - 2^(c-1) (c >= 2), and every doubling starts from
 * (prev_draw, mntm_vec) (ref: src/nuts.cpp:241-256).  So the state after leaf i is
 *        s(i) = LF^{n(i)}(prev_draw, mntm_vec),     n(i) = 1 + sum over the set bits k of i of (k + 1):
 * all 2^j leaves of a doubling lie on ONE trajectory and visit only 1 + j (j + 1) / 2 distinct points of it (56 of 1024 at j = 10), and
 * the U-turn test of a level-l node whose first leaf sits at point n1 (ref: nuts.ipp:224-229) compares the points n1 and n1 + l.  A leaf's
 * n', s', alpha (ref: :146-157) depend on its point only.  Here each point is computed ONCE (one leap_frog + one kernel evaluation), each
 * distinct test once (when its second point appears), and the tree is then walked leaf by leaf exactly as the recursion walks it -- the
 * same merges in the same order, the same uniforms from the same slots, the same early exit -- on the memoised scalars.  Same bits as the
 * recursion by construction; tests/test_oracle_memo.py checks that on random cases (including the non-finite regime and bounds).
 * This is what mcmc_amd/csrc/nuts_memo.hpp runs on the device; *n_exec counts the leap_frog calls it really makes, c->n_leap keeps the
 * reference's count (one per leaf walked). */
#define ORC_MEMO_MAXPTS 57        /* 1 + 10 * 11 / 2 = 56 points at max_tree_depth 10 (deeper trees: the recursion) */
static int memo_npt(uint32_t i) { int n = 1; for (int k = 0; i >> k; ++k) if ((i >> k) & 1u) n += k + 1; return n; }
/* is there a level-l node (l >= 1) in a doubling of depth j whose first leaf sits at point n1?  First leaves of level-l nodes are the
 * multiples of 2^l below 2^j: n1 - 1 must be a sum of distinct integers of {l + 1, ..., j}; the sums of t of these consecutive integers
 * are exactly the integers between the t smallest and the t largest. */
static int memo_pair_used(int l, int n1, int j)
{
    const int m = n1 - 1;
    for (int t = 0; t <= j - l; ++t) {
        const int lo = t * (l + 1) + t * (t - 1) / 2, hi = t * j - t * (t - 1) / 2;
        if (m >= lo && m <= hi) return 1;
    }
    return 0;
}

/* ACROSS doublings (round 6, orc_nuts_memo_xd): consecutive doublings of a draw in the SAME direction start from the same (prev_draw, mntm_vec) with the
 * same step as long as no proposal was accepted in between (src/nuts.cpp:241-256, :272), so they walk the same trajectory: a doubling of depth j re-uses the
 * points 1 .. n the last doubling of its direction left (depth j' < j) and computes only the points behind them.  memo_dir is what survives a doubling:
 * the points, their scalars and the test results of one direction.  A DEEPER tree tests pairs among points it did not compute itself (the pairs of a
 * shallower tree are pairs of every deeper one: memo_pair_used's sums nest), so when a point is computed every test it closes in the DEEPEST doubling the run
 * can make (depth max_tree_depth - 1) is evaluated, whatever the depth of the doubling that computes it. */
typedef struct memo_dir {
    int n_valid, je;                 /* points 1 .. n_valid exist (0: none); je: the depth whose tests are evaluated when a point appears */
    double* pt_th; double* pt_p;     /* point n at [n * d] */
    double pt_a[ORC_MEMO_MAXPTS];
    uint64_t cnb, csb, okb[12], okc[12];
} memo_dir;
static void memo_dir_reset(memo_dir* m)
{
    m->n_valid = 0; m->cnb = 0; m->csb = 0;
    memset(m->okb, 0, sizeof(m->okb)); memset(m->okc, 0, sizeof(m->okc));
}

static void nuts_doubling_memo(orc_ctx* c, int direction_val, double step_size, double log_rand_val, double prev_U, double prev_K,
                               const double* draw_vec, const double* mntm_vec, size_t tree_depth,
                               double* new_draw, double* edge_draw, double* edge_mntm,
                               size_t* n_val, size_t* s_val, double* alpha_val, size_t* n_alpha_val,
                               uint32_t draw_ind, uint32_t* uslot, uint64_t* n_exec, memo_dir* xd)
{
    const size_t d = c->d, nb = d * sizeof(double);
    const int jd = (int)tree_depth;
    const int je = xd ? xd->je : jd;                            /* the depth whose tests are evaluated when a point appears */
    const double max_tuning_par = 1000;
    memo_dir local;
    memo_dir* const md = xd ? xd : &local;
    if (!xd) { memo_dir_reset(md); md->pt_th = dvec((size_t)ORC_MEMO_MAXPTS * d); md->pt_p = dvec((size_t)ORC_MEMO_MAXPTS * d); }
    double* const pt_th = md->pt_th;                            /* point n at [n * d] (n = 1 ..) */
    double* const pt_p = md->pt_p;
    double* const pt_a = md->pt_a;
#define cnb (md->cnb)
#define csb (md->csb)
#define okb (md->okb)
#define okc (md->okc)
    int npts = md->n_valid;
    double* cur_th = dvec(d); memcpy(cur_th, npts ? pt_th + (size_t)npts * d : draw_vec, nb);
    double* cur_p = dvec(d);  memcpy(cur_p, npts ? pt_p + (size_t)npts * d : mntm_vec, nb);
    double* diff = dvec(d);
    if (npts >= 1 + jd) { memcpy(edge_draw, pt_th + (size_t)(1 + jd) * d, nb); memcpy(edge_mntm, pt_p + (size_t)(1 + jd) * d, nb); }   /* the far edge is a point this doubling does not compute */
    const uint64_t leap_before = c->n_leap;
    uint64_t leaves = 0;
    /* pending first halves per level (ref: the frames of the recursion) */
    double p_n[12], p_a[12], p_na[12]; int p_ref[12];
    double cn = 0, ca = 0, cna = 0; int cref = 0, complete = 0;
    for (uint32_t li = 0; li < (1u << jd); ++li) {
        const int n = memo_npt(li);
        while (npts < n) {                                          /* a point of the trajectory that no leaf has visited yet */
            const int m = npts + 1;
            leap_frog(c, direction_val * step_size, 1, cur_th, cur_p);      /* ref: nuts.ipp:132 */
            double prop_U = -box_log_kernel(c, cur_th);             /* :134 */
            if (!isfinite(prop_U)) prop_U = INFINITY;
            const double prop_K = kinetic(c, cur_p);                /* :140 */
            memcpy(pt_th + (size_t)m * d, cur_th, nb); memcpy(pt_p + (size_t)m * d, cur_p, nb);
            cnb &= ~(1ull << m); csb &= ~(1ull << m);
            if (log_rand_val <= -prop_U - prop_K) cnb |= 1ull << m;                       /* :146 */
            if (log_rand_val < max_tuning_par - prop_U - prop_K) csb |= 1ull << m;        /* :147 */
            const double dd = -(prop_U + prop_K) + (prev_U + prev_K);
            pt_a[m] = orc_exp((dd < 0.0) ? dd : 0.0);               /* :157 */
            for (int l = 1; l <= je; ++l) {                         /* the tests whose second point this is (:224-229) */
                const int n1 = m - l;
                if (n1 < 1 || !memo_pair_used(l, n1, je)) continue;
                okb[l] &= ~(1ull << n1);
                const double* ta = pt_th + (size_t)n1 * d; const double* pa = pt_p + (size_t)n1 * d;
                for (size_t i = 0; i < d; ++i) diff[i] = (direction_val > 0) ? cur_th[i] - ta[i] : ta[i] - cur_th[i];   /* pos - neg */
                const int c1 = orc_dot_b(diff, pa, d, c->W, c->nblk, c->bs) >= 0.0;
                const int c2 = orc_dot_b(diff, cur_p, d, c->W, c->nblk, c->bs) >= 0.0;
                if (c1 && c2) okb[l] |= 1ull << n1;
                okc[l] |= 1ull << n1;
            }
            if (m == 1 + jd) { memcpy(edge_draw, cur_th, nb); memcpy(edge_mntm, cur_p, nb); }   /* the far edge: first leaf of the second half */
            npts = m;
        }
        /* the leaf (:146-158) on the memoised scalars */
        leaves++;
        cn = (double)((cnb >> n) & 1ull); ca = pt_a[n]; cna = 1.0; cref = n;
        int failed = !((csb >> n) & 1ull);
        int pend_level = jd + 1;
        for (int l = 1; l <= jd; ++l) {                             /* unwind: the returns of the recursion (:212-239) */
            const int bit = (int)((li >> (l - 1)) & 1u);
            if (!failed && !bit) { pend_level = l; break; }         /* a first half with s' = 1: its second half is built next */
            if (!bit) continue;                                     /* a first half with s' = 0 returns through its parent (:234-239) */
            const double z = orc_rng_uniform(c->seed, c->chain, draw_ind, (*uslot)++);    /* :213 */
            const double prob = cn / (p_n[l] + cn);                 /* :212 */
            if (!(z < prob)) cref = p_ref[l];                       /* :215-217 (0/0 = NaN keeps) */
            cn = p_n[l] + cn; ca = p_a[l] + ca; cna = p_na[l] + cna;    /* :220-222 */
            if (!failed) {
                const int n1 = memo_npt(li + 1u - (1u << l));        /* the node's first leaf */
                if (!((okc[l] >> n1) & 1ull)) { fprintf(stderr, "orc memo: test (%d, %d) not evaluated\n", l, n1); abort(); }
                if (!((okb[l] >> n1) & 1ull)) failed = 1;           /* :226-229 */
            }
        }
        if (failed) break;
        if (li == (1u << jd) - 1u) { complete = 1; break; }
        p_n[pend_level] = cn; p_a[pend_level] = ca; p_na[pend_level] = cna; p_ref[pend_level] = cref;
    }
    *n_val = (size_t)cn; *s_val = (size_t)complete; *alpha_val = ca; *n_alpha_val = (size_t)cna;
    if (complete) memcpy(new_draw, pt_th + (size_t)cref * d, nb);
    *n_exec += c->n_leap - leap_before;
    c->n_leap = leap_before + leaves;                               /* the reference's count: one leapfrog per leaf */
    md->n_valid = npts;
#undef cnb
#undef csb
#undef okb
#undef okc
    if (!xd) { free(md->pt_th); free(md->pt_p); }
    free(cur_th); free(cur_p); free(diff);
}

/* ref: src/nuts.cpp:30-332; memo != 0: every doubling through nuts_doubling_memo (same bits, fewer leap_frog calls) */
static int orc_nuts_impl(const double* initial_vals, size_t d, orc_kernel_fn kernel, void* data,
             const orc_settings* s, double* draws_out, orc_stats* st, int memo, uint64_t* n_exec_out);
int orc_nuts(const double* initial_vals, size_t d, orc_kernel_fn kernel, void* data,
             const orc_settings* s, double* draws_out, orc_stats* st)
{
    return orc_nuts_impl(initial_vals, d, kernel, data, s, draws_out, st, 0, NULL);
}
int orc_nuts_memo(const double* initial_vals, size_t d, orc_kernel_fn kernel, void* data,
                  const orc_settings* s, double* draws_out, orc_stats* st, uint64_t* n_exec_out)
{
    if (s->max_tree_depth > 10) return orc_nuts_impl(initial_vals, d, kernel, data, s, draws_out, st, 0, n_exec_out);
    return orc_nuts_impl(initial_vals, d, kernel, data, s, draws_out, st, 1, n_exec_out);
}
/* ... and the trajectory memoised ACROSS the doublings of a draw as well (memo_dir above): what nuts_gauss_memo_kernel runs since round 6 */
int orc_nuts_memo_xd(const double* initial_vals, size_t d, orc_kernel_fn kernel, void* data,
                     const orc_settings* s, double* draws_out, orc_stats* st, uint64_t* n_exec_out)
{
    if (s->max_tree_depth > 10) return orc_nuts_impl(initial_vals, d, kernel, data, s, draws_out, st, 0, n_exec_out);
    return orc_nuts_impl(initial_vals, d, kernel, data, s, draws_out, st, 2, n_exec_out);
}
static int orc_nuts_impl(const double* initial_vals, size_t d, orc_kernel_fn kernel, void* data,
             const orc_settings* s, double* draws_out, orc_stats* st, int memo, uint64_t* n_exec_out)
{
    uint64_t n_exec = 0;
    orc_ctx c;
    ctx_init(&c, d, kernel, data, s, 1);
    const size_t nb = d * sizeof(double);
    const size_t n_burnin = s->n_burnin_draws, n_keep = s->n_keep_draws, n_total = n_burnin + n_keep;
    const size_t n_adapt_draws = (s->n_adapt_draws <= n_total) ? s->n_adapt_draws : n_total;   /* :54 */
    const double target_accept_rate = s->target_accept_rate;
    const size_t max_tree_depth = s->max_tree_depth;
    double epsilon_bar = s->step_size;                                  /* :59 */
    const double gamma_val = s->gamma_val, t0_val = s->t0_val, kappa_val = s->kappa_val;

    double* first_draw = dvec(d);
    memcpy(first_draw, initial_vals, nb);
    if (c.vals_bound) orc_transform(initial_vals, c.btype, c.lb, c.ub, d, first_draw);   /* :160-162 */

    double* rand_vec = dvec(d);
    double* mntm_vec = dvec(d);
    orc_rng_normal_vec(c.seed, c.chain, 0u, ORC_STREAM_INIT, d, rand_vec);   /* :166 */
    orc_gemv(c.sqrt_precond, rand_vec, d, mntm_vec);                    /* :168 */

    double step_size = nuts_find_initial_step_size(&c, first_draw, mntm_vec);   /* :172 */
    const uint64_t n_search_leaps = c.n_leap;                           /* (the search's leapfrogs are executed as they are) */
    const double mu_val = orc_log(10 * step_size);                      /* :174 */
    double h_val = 0;

    double prev_U = -box_log_kernel(&c, first_draw);                    /* :181 */
    double prop_U = prev_U;
    double prev_K, log_rand_val;
    double* prev_draw = dvec(d); memcpy(prev_draw, first_draw, nb);
    double* new_draw = dvec(d);  memcpy(new_draw, first_draw, nb);
    double* draw_pos = dvec(d);  memcpy(draw_pos, first_draw, nb);
    double* draw_neg = dvec(d);  memcpy(draw_neg, first_draw, nb);
    double* mntm_pos = dvec(d);  memcpy(mntm_pos, mntm_vec, nb);
    double* mntm_neg = dvec(d);  memcpy(mntm_neg, mntm_vec, nb);
    double* dummy_draw = dvec(d);
    double* dummy_mntm = dvec(d);
    double* start_draw = dvec(d);
    double* diff = dvec(d);
    size_t n_accept = 0;
    (void)prop_U;
    memo_dir xdir[2];                                                   /* memo == 2: what the doublings of a draw share, by direction (0: backward) */
    for (int k = 0; k < 2; ++k) {
        xdir[k].pt_th = (memo == 2) ? dvec((size_t)ORC_MEMO_MAXPTS * d) : NULL; xdir[k].pt_p = (memo == 2) ? dvec((size_t)ORC_MEMO_MAXPTS * d) : NULL;
        xdir[k].je = (max_tree_depth >= 1) ? (int)max_tree_depth - 1 : 0;
        memo_dir_reset(&xdir[k]);
    }

    for (size_t draw_ind = 0; draw_ind < n_total; ++draw_ind) {         /* :199 */
        const uint64_t leap0 = c.n_leap;
        memo_dir_reset(&xdir[0]); memo_dir_reset(&xdir[1]);             /* a new momentum: new trajectories */
        const double eps_used = step_size;
        uint32_t uslot = 0;
        orc_rng_normal_vec(c.seed, c.chain, (uint32_t)draw_ind, ORC_STREAM_NORMAL, d, rand_vec);  /* :200 */
        orc_gemv(c.sqrt_precond, rand_vec, d, mntm_vec);                /* :202 */
        prev_K = kinetic(&c, mntm_vec);                                 /* :204 */
        log_rand_val = orc_log(orc_rng_uniform(c.seed, c.chain, (uint32_t)draw_ind, uslot++)) - prev_U - prev_K;  /* :206 */
        memcpy(new_draw, prev_draw, nb);                                /* :210-215 */
        memcpy(draw_pos, prev_draw, nb);
        memcpy(draw_neg, prev_draw, nb);
        memcpy(mntm_pos, mntm_vec, nb);
        memcpy(mntm_neg, mntm_vec, nb);
        size_t tree_depth = 0, n_val = 1, s_val = 1;
        double alpha_val = 0;
        size_t n_alpha_val = 0;
        int good_round = 0;

        while (s_val == 1 && tree_depth < max_tree_depth) {             /* :227 */
            size_t n_p_val, s_p_val;
            double z = orc_rng_uniform(c.seed, c.chain, (uint32_t)draw_ind, uslot++);   /* :233 */
            const int direction_val = (z <= 0.5) ? -1 : 1;              /* :235 */
            memcpy(start_draw, prev_draw, nb);   /* prev_draw is passed by const ref; it is not modified in the call */
            memo_dir* const xd = (memo == 2) ? &xdir[direction_val > 0] : NULL;
            if (direction_val == -1) {
                memcpy(dummy_draw, draw_pos, nb);                       /* :238-239 */
                memcpy(dummy_mntm, mntm_pos, nb);
                if (memo) nuts_doubling_memo(&c, direction_val, step_size, log_rand_val, prev_U, prev_K, start_draw, mntm_vec, tree_depth,
                                             new_draw, draw_neg, mntm_neg, &n_p_val, &s_p_val, &alpha_val, &n_alpha_val, (uint32_t)draw_ind, &uslot, &n_exec, xd);
                else
                nuts_build_tree(&c, direction_val, step_size, log_rand_val, prev_U, prev_K, start_draw, mntm_vec,
                                tree_depth, new_draw, dummy_draw, draw_neg, dummy_mntm, mntm_neg,
                                &n_p_val, &s_p_val, &alpha_val, &n_alpha_val, (uint32_t)draw_ind, &uslot);   /* :241-246 */
            } else {
                memcpy(dummy_draw, draw_neg, nb);                       /* :248-249 */
                memcpy(dummy_mntm, mntm_neg, nb);
                if (memo) nuts_doubling_memo(&c, direction_val, step_size, log_rand_val, prev_U, prev_K, start_draw, mntm_vec, tree_depth,
                                             new_draw, draw_pos, mntm_pos, &n_p_val, &s_p_val, &alpha_val, &n_alpha_val, (uint32_t)draw_ind, &uslot, &n_exec, xd);
                else
                nuts_build_tree(&c, direction_val, step_size, log_rand_val, prev_U, prev_K, start_draw, mntm_vec,
                                tree_depth, new_draw, draw_pos, dummy_draw, mntm_pos, dummy_mntm,
                                &n_p_val, &s_p_val, &alpha_val, &n_alpha_val, (uint32_t)draw_ind, &uslot);   /* :251-256 */
            }
            if (s_p_val == 1) {                                         /* :260 */
                z = orc_rng_uniform(c.seed, c.chain, (uint32_t)draw_ind, uslot++);      /* :261 */
                if (z < (double)n_p_val / (double)n_val) {              /* :263 */
                    prop_U = -box_log_kernel(&c, new_draw);             /* :264 */
                    if (!isfinite(prop_U)) prop_U = INFINITY;
                    memcpy(prev_draw, new_draw, nb);                    /* :272-273 */
                    prev_U = prop_U;
                    memo_dir_reset(&xdir[0]); memo_dir_reset(&xdir[1]);   /* the next doubling starts from another state */
                    good_round = 1;                                     /* :277 */
                }
            }
            n_val += n_p_val;                                           /* :283 */
            tree_depth += 1;
            for (size_t i = 0; i < d; ++i) diff[i] = draw_pos[i] - draw_neg[i];
            const int check_val_1 = orc_dot_b(diff, mntm_neg, d, c.W, c.nblk, c.bs) >= 0.0;             /* :286 */
            const int check_val_2 = orc_dot_b(diff, mntm_pos, d, c.W, c.nblk, c.bs) >= 0.0;             /* :287 */
            s_val = s_p_val * (size_t)check_val_1 * (size_t)check_val_2;                /* :289 */
        }

        if (draw_ind < n_adapt_draws) {                                 /* :294-302 */
            h_val += (1 / ((double)(draw_ind + 1) + t0_val)) * (target_accept_rate - (alpha_val / (double)n_alpha_val) - h_val);
            step_size = orc_exp(mu_val - h_val * sqrt((double)(draw_ind + 1)) / gamma_val);
            epsilon_bar *= orc_exp(orc_pow((double)(draw_ind + 1), -kappa_val) * (orc_log(step_size) - orc_log(epsilon_bar)));
        } else {
            step_size = epsilon_bar;
        }

        if (draw_ind >= n_burnin) {                                     /* :306-309 */
            store_row(draws_out, draw_ind - n_burnin, d, prev_draw);
            n_accept += (size_t)good_round;
        }
        if (st) {
            if (st->accept_trace) st->accept_trace[draw_ind] = (uint8_t)good_round;
            if (st->depth_trace) st->depth_trace[draw_ind] = (uint32_t)tree_depth;
            if (st->leap_trace) st->leap_trace[draw_ind] = (uint32_t)(c.n_leap - leap0);
            if (st->eps_trace) st->eps_trace[draw_ind] = eps_used;
        }
    }
    epilogue_inv_transform(&c, draws_out, n_keep);
    if (st) { st->n_accept_draws = n_accept; st->n_leapfrogs = c.n_leap; st->final_step_size = step_size; }
    if (n_exec_out) *n_exec_out = memo ? n_exec + n_search_leaps : c.n_leap;
    free(first_draw); free(rand_vec); free(mntm_vec); free(prev_draw); free(new_draw);
    free(draw_pos); free(draw_neg); free(mntm_pos); free(mntm_neg);
    free(dummy_draw); free(dummy_mntm); free(start_draw); free(diff);
    for (int k = 0; k < 2; ++k) { free(xdir[k].pt_th); free(xdir[k].pt_p); }
    ctx_free(&c);
    return 0;
}

/* ------------------------------------------------------------------ RM-HMC */

typedef struct rm_ctx {
    orc_ctx* c;
    orc_tensor_fn tensor;
    void* tensor_data;
} rm_ctx;

/* box_tensor_fn lambda (ref: src/rmhmc.cpp:152-164) */
static void box_tensor(rm_ctx* r, const double* vals, double* G, double* dG)
{
    orc_ctx* c = r->c;
    if (c->vals_bound) {
        double* vi = dvec(c->d);
        orc_inv_transform(vals, c->btype, c->lb, c->ub, c->d, vi);
        r->tensor(vi, G, dG, r->tensor_data);
        free(vi);
    } else r->tensor(vals, G, dG, r->tensor_data);
}

/* mntm_update_fn lambda (ref: src/rmhmc.cpp:99-150): returns the INCREMENT step * [J] grad_obj / 2 in out.
 *   grad_obj(i) = -grad(i) + 0.5 * ( trace(T_i) - dot(T_i^T p, Ginv p) ),  T_i = Ginv * dG_i       (:112-116, :135-139)
 * all sums sequential, index ascending; products with fma as in orc_gemv / orc_matmul. */
static void rm_mntm_update(rm_ctx* r, const double* pos, const double* mntm, double step, const double* Ginv,
                           const double* dG, double* out)
{
    orc_ctx* c = r->c;
    const size_t d = c->d;
    double* grad = dvec(d);
    double* gobj = dvec(d);
    double* T = dvec(d * d);
    double* a = dvec(d);
    double* b = dvec(d);
    if (c->vals_bound) {
        double* pi = dvec(d);
        orc_inv_transform(pos, c->btype, c->lb, c->ub, d, pi);         /* :107 */
        c->kernel(pi, grad, c->data);                                  /* :108 */
        free(pi);
    } else c->kernel(pos, grad, c->data);                              /* :131 */
    for (size_t i = 0; i < d; ++i) {
        orc_matmul(Ginv, dG + i * d * d, d, T);                        /* tmp_mat = inv_tensor_mat * tensor_deriv.mat(i) */
        double tr = 0.0;
        for (size_t j = 0; j < d; ++j) tr = tr + T[j * d + j];
        for (size_t j = 0; j < d; ++j) {                               /* a = T^T p */
            double acc = 0.0;
            for (size_t k = 0; k < d; ++k) acc = fma(T[k * d + j], mntm[k], acc);
            a[j] = acc;
        }
        orc_gemv(Ginv, mntm, d, b);                                    /* b = Ginv p */
        double dp = 0.0;
        for (size_t j = 0; j < d; ++j) dp = fma(a[j], b[j], dp);
        gobj[i] = -grad[i] + 0.5 * (tr - dp);
    }
    if (c->vals_bound) {
        double* J = dvec(d * d);
        double* jg = dvec(d);
        orc_inv_jacobian_adjust(pos, c->btype, c->lb, c->ub, d, J);    /* :121 */
        orc_gemv(J, gobj, d, jg);
        for (size_t i = 0; i < d; ++i) out[i] = (step * jg[i]) / 2.0;  /* :129 */
        free(J); free(jg);
    } else {
        for (size_t i = 0; i < d; ++i) out[i] = (step * gobj[i]) / 2.0;   /* :145 */
    }
    free(grad); free(gobj); free(T); free(a); free(b);
}

/* ref: src/rmhmc.cpp:30-287.  The reference draws one extra normal vector before the loop (:184) whose values are never
 * used; with the counter-based generator nothing needs to be consumed for it. */
int orc_rmhmc(const double* initial_vals, size_t d, orc_kernel_fn kernel, orc_tensor_fn tensor, void* data, void* tensor_data,
              const orc_settings* s, double* draws_out, orc_stats* st)
{
    orc_ctx c;
    orc_settings s0 = *s;
    s0.precond_mat = NULL;                                              /* rmhmc has no precond_mat */
    ctx_init(&c, d, kernel, data, &s0, 0);
    rm_ctx r = { &c, tensor, tensor_data };
    const size_t n_burnin = s->n_burnin_draws, n_keep = s->n_keep_draws, n_total = n_burnin + n_keep;
    const double step_size = s->step_size;
    const size_t n_leap_steps = s->n_leap_steps, n_fp_steps = s->n_fp_steps;
    const size_t dd = d * d;

    double* first_draw = dvec(d);
    memcpy(first_draw, initial_vals, d * sizeof(double));
    if (c.vals_bound) orc_transform(initial_vals, c.btype, c.lb, c.ub, d, first_draw);   /* :170-172 */

    double* prev_draw = dvec(d); memcpy(prev_draw, first_draw, d * sizeof(double));
    double* new_draw = dvec(d);  memcpy(new_draw, first_draw, d * sizeof(double));
    double* prop_draw = dvec(d);
    double* new_mntm = dvec(d);
    double* prop_mntm = dvec(d);
    double* incr = dvec(d);
    double* rand_vec = dvec(d);
    double* tmpv = dvec(d);
    double* new_tensor = dvec(dd); double* prev_tensor = dvec(dd);
    double* inv_new = dvec(dd);    double* inv_prev = dvec(dd);
    double* new_deriv = dvec(dd * d); double* prev_deriv = dvec(dd * d);
    double* L = dvec(dd); double* S = dvec(dd); double* Tn = dvec(dd);

    box_tensor(&r, new_draw, new_tensor, new_deriv);                    /* :187 */
    memcpy(prev_tensor, new_tensor, dd * sizeof(double));
    orc_inv(new_tensor, d, inv_new);                                    /* :190 */
    memcpy(inv_prev, inv_new, dd * sizeof(double));
    memcpy(prev_deriv, new_deriv, dd * d * sizeof(double));

    const double cons_term = 0.5 * (double)d * ORC_LOG_2PI;             /* :195 */
    orc_chol_lower(new_tensor, d, L);
    double prev_U = cons_term - box_log_kernel(&c, first_draw) + 0.5 * orc_log_det_from_chol(L, d);   /* :197 */
    double prop_U = prev_U, prop_K, prev_K;
    size_t n_accept = 0;

    for (size_t draw_ind = 0; draw_ind < n_total; ++draw_ind) {         /* :206 */
        orc_rng_normal_vec(c.seed, c.chain, (uint32_t)draw_ind, ORC_STREAM_NORMAL, d, rand_vec);   /* :207 */
        orc_chol_lower(prev_tensor, d, L);
        orc_gemv(L, rand_vec, d, new_mntm);                             /* :209 */
        orc_gemv(inv_prev, new_mntm, d, tmpv);
        prev_K = 0.0;
        for (size_t i = 0; i < d; ++i) prev_K = fma(new_mntm[i], tmpv[i], prev_K);
        prev_K = prev_K / 2.0;                                          /* :211 */
        memcpy(new_draw, prev_draw, d * sizeof(double));                /* :213 */

        for (size_t k = 0; k < n_leap_steps; ++k) {                     /* :215 */
            memcpy(prop_mntm, new_mntm, d * sizeof(double));
            for (size_t kk = 0; kk < n_fp_steps; ++kk) {                /* :220-222 */
                rm_mntm_update(&r, new_draw, prop_mntm, step_size, inv_prev, prev_deriv, incr);
                for (size_t i = 0; i < d; ++i) prop_mntm[i] = new_mntm[i] + incr[i];
            }
            memcpy(new_mntm, prop_mntm, d * sizeof(double));            /* :224 */

            memcpy(prop_draw, new_draw, d * sizeof(double));            /* :228 */
            for (size_t kk = 0; kk < n_fp_steps; ++kk) {                /* :231-235 */
                box_tensor(&r, prop_draw, Tn, NULL);
                orc_inv(Tn, d, inv_new);
                for (size_t i = 0; i < dd; ++i) S[i] = inv_prev[i] + inv_new[i];
                orc_gemv(S, new_mntm, d, tmpv);
                for (size_t i = 0; i < d; ++i) prop_draw[i] = new_draw[i] + (0.5 * step_size) * tmpv[i];
            }
            memcpy(new_draw, prop_draw, d * sizeof(double));            /* :237 */

            box_tensor(&r, new_draw, new_tensor, new_deriv);            /* :239 */
            orc_inv(new_tensor, d, inv_new);                            /* :240 */
            rm_mntm_update(&r, new_draw, new_mntm, step_size, inv_new, new_deriv, incr);   /* :244 */
            for (size_t i = 0; i < d; ++i) new_mntm[i] = new_mntm[i] + incr[i];
            c.n_leap++;
        }

        orc_chol_lower(new_tensor, d, L);
        prop_U = cons_term - box_log_kernel(&c, new_draw) + 0.5 * orc_log_det_from_chol(L, d);   /* :247 */
        if (!isfinite(prop_U)) prop_U = INFINITY;                       /* :249-251 */
        orc_gemv(inv_new, new_mntm, d, tmpv);
        prop_K = 0.0;
        for (size_t i = 0; i < d; ++i) prop_K = fma(new_mntm[i], tmpv[i], prop_K);
        prop_K = prop_K / 2.0;                                          /* :253 */

        const double x = -(prop_U + prop_K) + (prev_U + prev_K);
        const double comp_val = (x < 0.01) ? x : 0.01;                  /* std::min(0.01, x) :257 */
        const double z = orc_rng_uniform(c.seed, c.chain, (uint32_t)draw_ind, 0);   /* :258 */
        int acc = 0;
        if (z < orc_exp(comp_val)) {                                    /* :260 */
            memcpy(prev_draw, new_draw, d * sizeof(double));
            prev_U = prop_U;
            prev_K = prop_K;
            memcpy(prev_tensor, new_tensor, dd * sizeof(double));
            memcpy(inv_prev, inv_new, dd * sizeof(double));
            memcpy(prev_deriv, new_deriv, dd * d * sizeof(double));
            acc = 1;
            if (draw_ind >= n_burnin) { store_row(draws_out, draw_ind - n_burnin, d, new_draw); n_accept++; }
        } else {
            if (draw_ind >= n_burnin) store_row(draws_out, draw_ind - n_burnin, d, prev_draw);
        }
        if (st && st->accept_trace) st->accept_trace[draw_ind] = (uint8_t)acc;
    }
    (void)prev_K;
    epilogue_inv_transform(&c, draws_out, n_keep);                      /* :277-284 */
    if (st) { st->n_accept_draws = n_accept; st->n_leapfrogs = c.n_leap; st->final_step_size = step_size; }
    free(first_draw); free(prev_draw); free(new_draw); free(prop_draw); free(new_mntm); free(prop_mntm); free(incr);
    free(rand_vec); free(tmpv); free(new_tensor); free(prev_tensor); free(inv_new); free(inv_prev);
    free(new_deriv); free(prev_deriv); free(L); free(S); free(Tn);
    ctx_free(&c);
    return 0;
}

/* ------------------------------------------------------------------ many chains (CPU baseline harness) */

int orc_run_many(int algo, const orc_target* tgt, const orc_settings* s, size_t n_chains,
                 uint64_t chain0, const double* init, double* draws_out,
                 uint64_t* n_accept_out, uint64_t* n_leap_out, double* eps_out, int n_threads)
{
    const size_t d = tgt->d, n_keep = s->n_keep_draws;
    int rc = 0;
    double* prec_t = NULL;          /* Mode B: transposed precision, shared by all chains (read-only) */
    if (s->work_mode == 1 && tgt->kind == ORC_TARGET_GAUSS_DENSE && tgt->prec && !tgt->prec_t) {
        prec_t = (double*)malloc(d * d * sizeof(double));
        for (size_t i = 0; i < d; ++i) for (size_t k = 0; k < d; ++k) prec_t[k * d + i] = tgt->prec[i * d + k];
    }
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
    #pragma omp parallel for schedule(dynamic) num_threads(n_threads)
    for (long long ci = 0; ci < (long long)n_chains; ++ci) {
        orc_target t = *tgt;            /* private call counters */
        if (prec_t) t.prec_t = prec_t;
        orc_settings sc = *s;
        sc.chain_id = chain0 + (uint64_t)ci;
        orc_stats st; memset(&st, 0, sizeof(st));
        double* local = (double*)malloc((n_keep ? n_keep : 1) * d * sizeof(double));
        int r;
        if (algo == 0) r = orc_hmc(init + (size_t)ci * d, d, orc_target_kernel, &t, &sc, local, &st);
        else if (algo == 1) r = orc_mala(init + (size_t)ci * d, d, orc_target_kernel, &t, &sc, local, &st);
        else if (algo == 3) r = orc_rwmh(init + (size_t)ci * d, d, orc_target_kernel, &t, &sc, local, &st);
        else if (algo == 4) r = orc_rmhmc(init + (size_t)ci * d, d, orc_target_kernel, orc_target_tensor, &t, &t, &sc, local, &st);
        else if (algo == 5) r = orc_nuts_memo(init + (size_t)ci * d, d, orc_target_kernel, &t, &sc, local, &st, NULL);
        else r = orc_nuts(init + (size_t)ci * d, d, orc_target_kernel, &t, &sc, local, &st);
        if (r) rc = r;
        if (draws_out)
            for (size_t k = 0; k < n_keep; ++k)
                for (size_t j = 0; j < d; ++j)
                    draws_out[(k * d + j) * n_chains + (size_t)ci] = local[k * d + j];
        if (n_accept_out) n_accept_out[ci] = st.n_accept_draws;
        if (n_leap_out) n_leap_out[ci] = st.n_leapfrogs;
        if (eps_out) eps_out[ci] = st.final_step_size;
        free(local);
    }
    free(prec_t);
    return rc;
}

/* ------------------------------------------------------------------ unit-test exports */

void orc_math_eval(int fn, const double* x, size_t n, double* out, double* out2)
{
    for (size_t i = 0; i < n; ++i) {
        switch (fn) {
        case 0: out[i] = orc_exp(x[i]); break;
        case 1: out[i] = orc_log(x[i]); break;
        case 2: orc_sincos2pi(x[i], &out[i], &out2[i]); break;
        case 3: out[i] = orc_softplus(x[i]); break;
        case 4: out[i] = orc_sigmoid(x[i]); break;
        default: out[i] = NAN;
        }
    }
}
void orc_philox_eval(const uint32_t* ctr, const uint32_t* key, uint32_t* out) { orc_philox4x32(ctr, key, out); }
void orc_normal_vec(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t stream, size_t d, double* out)
{ orc_rng_normal_vec(seed, chain, draw, stream, d, out); }
double orc_uniform(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t slot)
{ return orc_rng_uniform(seed, chain, draw, slot); }
