// gemm_samplers.hpp -- interface of the lock-step samplers for dense-gradient Gaussian targets BEYOND d = 512 (gemm_samplers.hip): the state of every
// chain lives in HBM ([dimension][chain], chains contiguous) and one leapfrog step of ALL chains is one fp64 matrix product W = P Theta on the matrix
// cores with the half-kicks and the drift fused into its epilogue.
#pragma once

#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace mi {
namespace gemm {

enum : int { GEMM_HMC = 0, GEMM_MALA = 1, GEMM_RWMH = 3 };     // (the C ABI's algo numbers)

struct GemmRun {
    int algo = GEMM_HMC;
    uint32_t d = 0;
    uint64_t C = 0, chain0 = 0;
    const double* P = nullptr;        // dense Gaussian: d x d row-major precision, device
    const double* X = nullptr;        // logistic regression (when set): n_rows x d row-major design matrix and the labels, device
    const double* y = nullptr;
    uint32_t n_rows = 0;
    double* theta = nullptr;          // [d][C] in: initial values, out: final state (left alone for flagged chains)
    double* draws = nullptr;          // [n_keep][d][C] or nullptr
    uint64_t* n_accept = nullptr;     // [C] or nullptr (left alone for flagged chains)
    uint32_t* nf_flag = nullptr;      // [C + 1], zeroed by the caller: chains whose energies / proposal densities went non-finite (literal.hpp replays them)
    uint64_t seed = 0;
    uint32_t n_burnin = 0, n_keep = 0, n_leap = 0, draw0 = 0;
    double eps = 0.0;                 // step_size (hmc, mala) / par_scale (rwmh)
    double s2 = 0.0, rs = 0.0, log_det = 0.0, cons_term = 0.0;     // mala: dmvnorm's constants for Sigma = eps^2 I (mi_mcmc.hip: as the oracle states them)
    const double* mass_tables = nullptr;   // device, 4 tables of gemm_padded_d(d) doubles: diag(M), its sqrt, its reciprocal, diag(INV(eps^2 M)) (mala) -- ones / 1 / eps^2
                                           // for the identity, ones in the padding
    bool diag_mass = false;                // (for the kernel's name only: the tables decide)
    void* ws = nullptr;               // gemm_ws_bytes(d, n_rows, C) bytes of device memory
    bool use_graph = true;            // replay the launches of one draw from a captured hipGraph (the draw index lives in device memory)
};

uint32_t gemm_padded_d(uint32_t d);
size_t gemm_ws_bytes(uint32_t d, uint32_t n_rows, uint64_t C);      // n_rows = 0: the dense Gaussian
// enqueues the whole run on `st`; returns a hipError_t as int (0 = enqueued).  *kernel_name: what ran, for mi_mcmc_last_kernel()
int gemm_run(const GemmRun& r, hipStream_t st, const char** kernel_name);

}  // namespace gemm
}  // namespace mi
