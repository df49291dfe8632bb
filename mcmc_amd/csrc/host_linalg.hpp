// host_linalg.hpp -- host-side INV / CHOL_LOWER in the operation order the oracle states for the reference's BMO_MATOPS_INV /
// BMO_MATOPS_CHOL_LOWER, shared by the C ABI (mi_mcmc.hip), the literal replay preparation (literal_host.hpp) and its host test shim.
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

// host_inverse_compute behind a two-entry memo keyed by the matrix itself.  The elimination is O(d^3) scalar-order work (~90 ms at d = 512), and
// one sampler call asks for the same inverse more than once (its own and the literal replay's preparation; mala: INV(M) and INV(eps^2 M)), as do
// the calls of a run cut into pieces (checkpoint / resume) -- a deterministic function of its input, so the copy has the bits of a recomputation.
// Per thread; small matrices are not kept.
inline void host_inverse(const double* A, size_t d, std::vector<double>& Ainv)
{
    struct Entry { std::vector<double> key, val; };
    static thread_local Entry memo[2];
    static thread_local int last = 0;
    const size_t n = d * d;
    if (d >= 64) {
        for (int e = 0; e < 2; ++e)
            if (memo[e].key.size() == n && std::memcmp(memo[e].key.data(), A, n * sizeof(double)) == 0) { Ainv = memo[e].val; last = e; return; }
    }
    host_inverse_compute(A, d, Ainv);
    if (d >= 64) {
        Entry& slot = memo[1 - last];                    // the entry not used last
        slot.key.assign(A, A + n); slot.val = Ainv;
        last = 1 - last;
    }
}

inline void host_cholesky_lower(const double* A, size_t d, std::vector<double>& L)
{
    L.assign(d * d, 0.0);
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
}

}  // namespace
