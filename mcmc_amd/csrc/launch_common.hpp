// launch_common.hpp -- shared by the *_launch.hip translation units
#pragma once
#include <hip/hip_runtime.h>

namespace mi {
// the kernel a call's time goes to, by the name rocprofv3 prints: read back through mi_mcmc_last_kernel() (thread-local;
// defined in mi_mcmc.hip).  Set by the launchers, so bench.py labels a measurement with what actually ran.
void note_kernel(const char* fmt, ...);
// TEST HOOK (mi_mcmc_test_set_grid_cap, mi_mcmc_probes.h): an upper limit on the workgroups of the PERSISTENT grids (nuts_memo.hpp,
// nuts_lds.hpp), 0 = none.  With a small cap a handful of chains exercises what only > 16 384 chains reach otherwise: the
// global counter, a slot taking its second and third chain, retire-on-leave.  Process-wide; results never depend on it.
uint64_t test_grid_cap();
inline uint64_t cap_grid(uint64_t n_wg) { const uint64_t c = test_grid_cap(); return (c != 0 && c < n_wg) ? c : n_wg; }
}  // namespace mi

#define MI_LAUNCH_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

// dispatch on the tile count: F<NT>() for NT in {1, 2, 4, 8}
#define MI_DISPATCH_NT(nt, CALL1, CALL2, CALL4, CALL8) \
    ((nt) <= 1 ? (CALL1) : (nt) == 2 ? (CALL2) : (nt) <= 4 ? (CALL4) : (CALL8))
