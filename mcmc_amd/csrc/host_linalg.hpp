// host_linalg.hpp -- INV / CHOL_LOWER of a dense precond_mat in the operation order the oracle states for the reference's BMO_MATOPS_INV /
// BMO_MATOPS_CHOL_LOWER, shared by the C ABI (mi_mcmc.hip), the literal replay preparation (literal_host.hpp) and its host test shim: the host
// loops, and the switch to their device form (linalg_device.hip) that the product installs.
#pragma once

#include <cmath>
#include <cstddef>
#include <cstring>
#include <utility>
#include <vector>

namespace {

// row <- row - f * pivot_row, element by element (no contraction: -ffp-contract=off): the inner loop of the Gauss-Jordan elimination.  Out of
// line with restrict-qualified rows so that it vectorises, and cloned for AVX2 where the host has it (the x86-64 baseline is two doubles per
// instruction) -- the same IEEE operations per element either way, so the same bits: d = 512 152 -> 89 ms, d = 256 18 -> 7 ms per inversion.
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target_clones("avx2", "default")))
#endif
inline void host_row_axpy(double* __restrict row, const double* __restrict pivot_row, double f, size_t d)
{
    for (size_t j = 0; j < d; ++j) row[j] = row[j] - f * pivot_row[j];
}

// INV and CHOL_LOWER of a dense precond_mat on the host, with the operation order the oracle states for the reference's
// BMO_MATOPS_INV / BMO_MATOPS_CHOL_LOWER (Gauss-Jordan with partial pivoting; column Cholesky).  Compiled with
// -ffp-contract=off like everything else, so the bits are the oracle's.
inline void host_inverse_compute(const double* A, size_t d, std::vector<double>& Ainv)
{
    std::vector<double> a(A, A + d * d);
    Ainv.assign(d * d, 0.0);
    for (size_t i = 0; i < d; ++i) Ainv[i * d + i] = 1.0;
    for (size_t c = 0; c < d; ++c) {
        size_t piv = c;
        double best = std::fabs(a[c * d + c]);
        for (size_t r = c + 1; r < d; ++r)
            if (std::fabs(a[r * d + c]) > best) { best = std::fabs(a[r * d + c]); piv = r; }
        if (piv != c)
            for (size_t j = 0; j < d; ++j) { std::swap(a[c * d + j], a[piv * d + j]); std::swap(Ainv[c * d + j], Ainv[piv * d + j]); }
        const double pv = a[c * d + c];
        for (size_t j = 0; j < d; ++j) { a[c * d + j] = a[c * d + j] / pv; Ainv[c * d + j] = Ainv[c * d + j] / pv; }
        for (size_t r = 0; r < d; ++r) {
            if (r == c) continue;
            const double f = a[r * d + c];
            if (f == 0.0) continue;
            host_row_axpy(&a[r * d], &a[c * d], f, d);
            host_row_axpy(&Ainv[r * d], &Ainv[c * d], f, d);
        }
    }
}

// Where the factorisations run.  The product (mi_mcmc.hip) installs the device implementations (linalg_device.hip: the same IEEE operations per
// element, hence the same bits, as one cooperative launch instead of 89 ms / 40 ms of one host core at d = 512) for d >= min_d; a translation
// unit that installs nothing (the host test shim tests/lit_host.hip) runs the loops above.  A device failure is an ERROR (the status is returned
// and mi_mcmc_last_error says why), never a silent switch to the host loops.
struct LinalgAccel {
    int (*inverse)(const double* A, size_t d, double* Ainv) = nullptr;
    int (*cholesky_lower)(const double* A, size_t d, double* L) = nullptr;
    size_t min_d = 64;           // below: the host loops (microseconds, and no launch)
};
inline LinalgAccel& linalg_accel() { static LinalgAccel a; return a; }

// INV behind a two-entry memo keyed by the matrix itself: one sampler call asks for the same inverse more than once (its own and the literal
// replay's preparation; mala: INV(M) and INV(eps^2 M)), as do the calls of a run cut into pieces (checkpoint / resume) -- a deterministic function
// of its input, so the copy has the bits of a recomputation.  Per thread; small matrices are not kept; mi_mcmc_release_workspace empties it.
struct LinalgMemoEntry { std::vector<double> key, val; };
struct LinalgMemo { LinalgMemoEntry e[2]; int last = 0; };
inline LinalgMemo& linalg_memo(int which) { static thread_local LinalgMemo m[2]; return m[which]; }      // 0: INV, 1: CHOL_LOWER
inline void linalg_memo_clear()
{
    for (int w = 0; w < 2; ++w)
        for (auto& e : linalg_memo(w).e) { std::vector<double>().swap(e.key); std::vector<double>().swap(e.val); }
}
inline bool linalg_memo_find(int which, const double* A, size_t n, std::vector<double>& out)
{
    LinalgMemo& m = linalg_memo(which);
    for (int e = 0; e < 2; ++e)
        if (m.e[e].key.size() == n && std::memcmp(m.e[e].key.data(), A, n * sizeof(double)) == 0) { out = m.e[e].val; m.last = e; return true; }
    return false;
}
inline void linalg_memo_keep(int which, const double* A, size_t n, const std::vector<double>& val)
{
    LinalgMemo& m = linalg_memo(which);
    LinalgMemoEntry& slot = m.e[1 - m.last];             // the entry not used last
    slot.key.assign(A, A + n); slot.val = val;
    m.last = 1 - m.last;
}

// returns 0, or the device implementation's status
inline int host_inverse(const double* A, size_t d, std::vector<double>& Ainv)
{
    const size_t n = d * d;
    if (d >= 64 && linalg_memo_find(0, A, n, Ainv)) return 0;
    const LinalgAccel& acc = linalg_accel();
    if (acc.inverse != nullptr && d >= acc.min_d) {
        Ainv.assign(n, 0.0);
        const int rc = acc.inverse(A, d, Ainv.data());
        if (rc != 0) return rc;
    } else {
        host_inverse_compute(A, d, Ainv);
    }
    if (d >= 64) linalg_memo_keep(0, A, n, Ainv);
    return 0;
}

inline int host_cholesky_lower(const double* A, size_t d, std::vector<double>& L)
{
    if (d >= 64 && linalg_memo_find(1, A, d * d, L)) return 0;
    L.assign(d * d, 0.0);
    const LinalgAccel& acc = linalg_accel();
    if (acc.cholesky_lower != nullptr && d >= acc.min_d) {
        const int rc = acc.cholesky_lower(A, d, L.data());
        if (rc == 0) linalg_memo_keep(1, A, d * d, L);
        return rc;
    }
    for (size_t j = 0; j < d; ++j) {
        double sum = A[j * d + j];
        for (size_t k = 0; k < j; ++k) sum = sum - L[j * d + k] * L[j * d + k];
        const double ljj = std::sqrt(sum);
        L[j * d + j] = ljj;
        for (size_t i = j + 1; i < d; ++i) {
            double t = A[i * d + j];
            for (size_t k = 0; k < j; ++k) t = t - L[i * d + k] * L[j * d + k];
            L[i * d + j] = t / ljj;
        }
    }
    if (d >= 64) linalg_memo_keep(1, A, d * d, L);
    return 0;
}

}  // namespace
