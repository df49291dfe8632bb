// launch_common.hpp -- shared by the *_launch.hip translation units
#pragma once
#include <hip/hip_runtime.h>

namespace mi {
// the kernel a call's time goes to, by the name rocprofv3 prints: read back through mi_mcmc_last_kernel() (thread-local;
// defined in mi_mcmc.hip).  Set by the launchers, so bench.py labels a measurement with what actually ran.
void note_kernel(const char* fmt, ...);
}  // namespace mi

#define MI_LAUNCH_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

// dispatch on the tile count: F<NT>() for NT in {1, 2, 4, 8}
#define MI_DISPATCH_NT(nt, CALL1, CALL2, CALL4, CALL8) \
    ((nt) <= 1 ? (CALL1) : (nt) == 2 ? (CALL2) : (nt) <= 4 ? (CALL4) : (CALL8))
