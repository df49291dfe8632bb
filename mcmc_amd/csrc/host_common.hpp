// host_common.hpp -- shared by the host-side translation units of the C ABI (mi_mcmc.hip: the samplers; callback_host.hip: the
// host-callback routes; stats_collate.hip: reducers, converters, multi-GPU helpers; probes.hip: diagnostics): the error channel
// (status code + thread-local message behind mi_mcmc_last_error) and the RAII device buffer.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/mi_mcmc.h"

namespace mi {
namespace host {

std::string& last_error();                         // thread-local, defined in mi_mcmc.hip
std::string& last_kernel();                        // thread-local: what mi_mcmc_last_kernel() returns

inline int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}

// RAII device buffer
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// One chain of `algo` (0 hmc, 1 mala, 2 nuts, 3 rwmh, 4 rmhmc) on the literal kernel with HOST callbacks as target (mi_mcmc.hip; the
// kernel asks the host through a pinned mailbox): bounds, precond_mat / cov_mat, any max_tree_depth <= 30.  tensor_fn: rmhmc only.
int literal_run_callback(const char* who, int algo, const double* initial_vals, uint64_t d, mi_log_kernel_cb target_log_kernel, void* target_data,
                         mi_tensor_cb tensor_fn, void* tensor_data, const mi_settings* settings, double* draws_out,
                         uint64_t* n_accept_draws, double* step_size_out);

}  // namespace host
}  // namespace mi

#define HIP_TRY(expr)                                                                                    \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return ::mi::host::fail(e_ == hipErrorOutOfMemory ? MI_ERR_OOM : MI_ERR_HIP, "%s failed: %s", \
                                    #expr, hipGetErrorString(e_));                                       \
    } while (0)
