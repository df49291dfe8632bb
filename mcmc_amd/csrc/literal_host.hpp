// literal_host.hpp -- host-side preparation of a literal replay (literal.hpp): bounds types, the preconditioner in the form the
// literal kernels read (identity / diagonal vectors / dense matrices with INV and CHOL_LOWER from the host), and, for unbounded
// mala, INV / LOG_DET of the constant Sigma = eps^2 M -- all in the operation order the oracle states (host_linalg.hpp).
// Used by the C ABI (mi_mcmc.hip) and by the host test shim (tests/lit_host.hip), so that the CPU tests exercise this code too.
#pragma once

#include <cmath>
#include <cstdint>
#include <vector>

#include "host_linalg.hpp"
#include "literal.hpp"

namespace mi {
namespace lit {

struct LitPrep {
    int precond = 0;                          // 0 identity, 1 diagonal, 2 dense
    std::vector<int> bt;
    std::vector<double> lb, ub;
    std::vector<double> m, m_sqrt, m_inv;     // precond 1
    std::vector<double> Mfull, Lchol, Minv;   // precond 2: d*d TRANSPOSED (what literal.hpp's gemv_t reads)
    std::vector<double> sinv_diag, Sinv;      // mala, unbounded
    double rs = 0.0, log_det = 0.0, cons_term = 0.0;
};

// out[c * rows + r] = in[r * cols + c]
inline void lit_transpose(const double* in, size_t rows, size_t cols, std::vector<double>& out)
{
    out.resize(rows * cols);
    for (size_t r = 0; r < rows; ++r)
        for (size_t c = 0; c < cols; ++c) out[c * rows + r] = in[r * cols + c];
}

// algo: 0 hmc, 1 mala.  precond_mat: d*d row-major or nullptr.
// Returns 0, or the status of the (device) factorisation of a dense precond_mat (host_linalg.hpp).
inline int lit_prepare(int algo, uint32_t d, double eps, int vals_bound, const double* lower, const double* upper,
                       const double* precond_mat, LitPrep& o)
{
    o.bt.assign(d, 1); o.lb.assign(d, 0.0); o.ub.assign(d, 0.0);
    if (vals_bound)
        for (uint32_t i = 0; i < d; ++i) {       // determine_bounds_type.hpp:27-57
            o.lb[i] = lower[i]; o.ub[i] = upper[i];
            const bool fl = std::isfinite(lower[i]), fu = std::isfinite(upper[i]);
            o.bt[i] = (fl && fu) ? 4 : (fl && !fu) ? 2 : (!fl && fu) ? 3 : 1;
        }
    o.precond = 0;
    if (precond_mat) {
        o.precond = 1;
        for (uint32_t i = 0; i < d && o.precond == 1; ++i)
            for (uint32_t k = 0; k < d; ++k)
                if (i != k && precond_mat[(size_t)i * d + k] != 0.0) { o.precond = 2; break; }
        if (o.precond == 1) {
            o.m.resize(d); o.m_sqrt.resize(d); o.m_inv.resize(d);
            for (uint32_t i = 0; i < d; ++i) {
                const double v = precond_mat[(size_t)i * d + i];
                o.m[i] = v; o.m_sqrt[i] = __builtin_sqrt(v); o.m_inv[i] = 1.0 / v;
            }
        } else {
            std::vector<double> Minv, Lc;
            if (int rc = host_inverse(precond_mat, d, Minv)) return rc;
            if (int rc = host_cholesky_lower(precond_mat, d, Lc)) return rc;
            lit_transpose(precond_mat, d, d, o.Mfull);
            lit_transpose(Minv.data(), d, d, o.Minv);
            lit_transpose(Lc.data(), d, d, o.Lchol);
        }
    }
    if (algo == 1) {
        const double s2 = eps * eps;
        o.rs = 1.0 / s2;
        o.cons_term = -0.5 * (double)d * LIT_LOG_2PI;
        double ld = 0.0;
        if (o.precond == 0) {
            const double lii = __builtin_sqrt(s2);
            for (uint32_t i = 0; i < d; ++i) ld = ld + 2.0 * det_log(lii);
        } else if (o.precond == 1) {
            o.sinv_diag.resize(d);
            for (uint32_t i = 0; i < d; ++i) {
                const double sig = s2 * o.m[i];
                o.sinv_diag[i] = 1.0 / sig;
                ld = ld + 2.0 * det_log(__builtin_sqrt(sig));
            }
        } else {
            std::vector<double> Sigma((size_t)d * d), Ls;
            for (size_t i = 0; i < (size_t)d * d; ++i) Sigma[i] = s2 * precond_mat[i];
            std::vector<double> Sinv;
            if (int rc = host_inverse(Sigma.data(), d, Sinv)) return rc;
            lit_transpose(Sinv.data(), d, d, o.Sinv);
            if (int rc = host_cholesky_lower(Sigma.data(), d, Ls)) return rc;
            for (uint32_t i = 0; i < d; ++i) ld = ld + 2.0 * det_log(Ls[(size_t)i * d + i]);
        }
        o.log_det = ld;
    }
    return 0;
}

// the reduction orders of the throughput kernels: four strided fma chains per dot product; the logistic
// kernels additionally sum over four dimension blocks of 16 NTQ dims with two eta sub-chains per block (logistic_lds.hpp)
inline void lit_orders(LitTarget& t)
{
    t.W = 4; t.nblk = 1; t.bs = 0; t.eta_chains = 1;
    if (t.kind == LIT_LOGISTIC && t.d <= 512) {          // beyond d = 512 there is no LDS kernel to agree with: plain orders (one block, one
        t.nblk = 4; t.eta_chains = 2;                    // eta chain) -- four blocks of 128 would DROP the dimensions from 512 on
        t.bs = (t.d <= 64) ? 16u : (t.d <= 128) ? 32u : (t.d <= 256) ? 64u : 128u;
    }
    // dense Gaussians with 128 < d <= 512 run on the LDS-streamed kernel too (logistic_lds.hpp, LOGIT_TARGET_DENSE): rows of P theta as
    // one ascending fma chain (what every dense kernel does), dot products over its four dimension quarters
    if (t.kind == LIT_DENSE && t.d > 128 && t.d <= 512) {
        t.nblk = 4;
        t.bs = (t.d <= 192) ? 48u : (t.d <= 256) ? 64u : (t.d <= 384) ? 96u : 128u;     // 16 NTQ of the instantiation (logistic_lds.hip)
    }
}
// every blocked reduction must cover every dimension (dot_b and the blocked eta loop of literal.hpp stop at nblk * bs)
inline bool lit_orders_cover(const LitTarget& t) { return t.nblk <= 1 || t.bs == 0 || (uint64_t)t.nblk * t.bs >= t.d; }

}  // namespace lit
}  // namespace mi
