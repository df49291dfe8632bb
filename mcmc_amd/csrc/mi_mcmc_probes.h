/* mi_mcmc_probes.h -- TEST / MEASUREMENT infrastructure, not part of the product: the diagnostics of libmi_mcmc_probes.so (probes.hip;
 * built next to libmi_mcmc.so and linked against it): one MFMA tile, the deterministic math functions, the per-chain RNG, fp64
 * throughput ceilings.  Host pointers, blocking; status codes and mi_mcmc_last_error() as in mi_mcmc.h. */
#pragma once
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
int mi_probe_mfma_f64(const double* A16x4, const double* B4x16, const double* C16x16, double* D16x16);
int mi_probe_math(int fn, const double* x, uint64_t n, double* out, double* out2);
int mi_probe_normals(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t stream, uint64_t d, double* out);
int mi_probe_uniform(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t slot, double* out);
int mi_probe_fp64_peak(int use_mfma, int iters, double* tflops_out);
int mi_probe_mfma_cycles(int waves_per_simd, int use_lds, int iters, double* cycles_per_mfma, double* tflops_out);
/* Defined in libmi_mcmc.so itself (a test hook, not a product entry point: it is declared here, not in mi_mcmc.h): limits the PERSISTENT grids of
 * the NUTS kernels with dynamic chain hand-out (nuts_memo.hpp, nuts_lds.hpp) to max_workgroups (0 = no limit), so that a test with a
 * few hundred chains runs the global counter, slot re-use and retire-on-leave.  Process-wide.  Results do not depend on it. */
void mi_mcmc_test_set_grid_cap(uint32_t max_workgroups);
#ifdef __cplusplus
}
#endif
