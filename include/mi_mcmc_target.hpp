// mi_mcmc_target.hpp -- user-defined DEVICE targets for the many-chain samplers.
//
// The reference takes the target as a host callback,
//     std::function<fp_t (const ColVec_t& vals_inp, ColVec_t* grad_out, void* target_data)>      (ref: include/mcmc/hmc.hpp:42-48)
// which cannot be called from a GPU kernel.  Its device form is a plain struct -- data members + a __device__ member function
// with the same contract -- compiled by hipcc into a small side library next to libmi_mcmc.so:
//
//     struct MyTarget {
//         static constexpr int D = 3;                              // n_vals, compile time, 1 <= D <= 8
//         // static constexpr int W = 1;                           // optional: order of the samplers' dot products (1 | 4)
//         double a, b;  const double* data;                        // anything trivially copyable; pointers must be DEVICE pointers
//         __host__ __device__ double kernel(const double (&vals)[D], double (&grad)[D], bool want_grad) const;
//         // mcmc::rmhmc only -- the metric tensor callback (ref: include/mcmc/rmhmc.hpp), G and, if dG != nullptr, dG[i] = dG/dvals_i:
//         // __device__ void tensor(const double (&vals)[D], double (&G)[D][D], double (*dG)[D][D]) const;
//     };
//     MI_MCMC_DEFINE_TARGET(my_target, MyTarget)
//
//     hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I<repo>/include -shared my_target.hip \
//           -L<repo>/mcmc_amd -lmi_mcmc -o libmy_target.so
//
// The macro defines   extern "C" int my_target_run(int algo, const MyTarget* target, const mi_settings*, mi_chains*, void* stream)
// with algo 0 = mcmc::hmc, 1 = mala, 2 = nuts, 3 = rwmh, 4 = rmhmc (only if the struct has tensor()), the settings / chains
// contract of include/mi_mcmc.h (host or device memory, global chain ids, draw0 continuation) and every feature of the
// one-chain-per-lane engine: any dense precond_mat / cov_mat, any box constraints, dual averaging.  Each chain is one lane and
// keeps its whole state in registers, so there is no PCIe round trip per gradient as on the host-callback route
// (mi_mcmc_hmc_run_callback): 10^5 chains run at once.  Arithmetic: the oracle's (oracle/mcmc_oracle.c with reduce_width W); a
// kernel() written with IEEE + - * / fma and the mi:: det_exp / det_log functions gives the same bits on host and device, so
// the same member function can serve as the host callback of a CPU check (examples/user_target.hip does exactly that).
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "mi_mcmc.h"
#include "mi_mcmc_engine/small_samplers.hpp"

namespace mi {

template <class T, class = void> struct target_has_tensor : std::false_type {};
template <class T>
struct target_has_tensor<T, std::void_t<decltype(std::declval<const T&>().tensor(std::declval<const double (&)[T::D]>(),
                                                                                 std::declval<double (&)[T::D][T::D]>(),
                                                                                 (double (*)[T::D][T::D]) nullptr))>> : std::true_type {};

// launch function of a user target: instantiates the engine's kernels for T (what mi_mcmc_run_user_target calls back)
template <class T>
int user_target_launch(int algo, const void* small_params, const void* target_pod, void* stream)
{
    static_assert(std::is_trivially_copyable<T>::value, "a device target is passed to the kernels by value");
    static_assert(T::D >= 1 && T::D <= SMALL_MAX_D, "1 <= D <= 8");
    const SmallParams& prm = *static_cast<const SmallParams*>(small_params);
    const T& tgt = *static_cast<const T*>(target_pod);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const unsigned block = 64;        // one wave per workgroup: the chains spread over as many CUs as possible
    const dim3 grid((unsigned)((prm.C + block - 1) / block));
    switch (algo) {
    case 0: hipLaunchKernelGGL(hmc_small_kernel<T>, grid, dim3(block), 0, st, prm, tgt); break;
    case 1: hipLaunchKernelGGL(mala_small_kernel<T>, grid, dim3(block), 0, st, prm, tgt); break;
    case 2: hipLaunchKernelGGL(nuts_small_kernel<T>, grid, dim3(block), 0, st, prm, tgt); break;
    case 3: hipLaunchKernelGGL(rwmh_small_kernel<T>, grid, dim3(block), 0, st, prm, tgt); break;
    case 4:
        if constexpr (target_has_tensor<T>::value) { hipLaunchKernelGGL(rmhmc_small_kernel<T>, grid, dim3(block), 0, st, prm, tgt); break; }
        else return (int)hipErrorInvalidValue;       // mcmc::rmhmc needs the metric tensor
    default: return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}

}  // namespace mi

#define MI_MCMC_DEFINE_TARGET(NAME, TARGET_T)                                                                             \
    extern "C" int NAME##_run(int algo, const TARGET_T* target, const mi_settings* settings, mi_chains* chains, void* stream) \
    {                                                                                                                     \
        return mi_mcmc_run_user_target_v(algo, (uint64_t)TARGET_T::D, &mi::user_target_launch<TARGET_T>, target,          \
                                         (uint64_t)sizeof(mi::SmallParams), MI_MCMC_VERSION, settings, chains, stream);   \
    }
