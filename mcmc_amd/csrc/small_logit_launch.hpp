// small_logit_launch.hpp -- one translation unit per pair of dimensions of LogisticSmallModel (small_logit_d*.hip), so that the
// kernel instantiations (d = 1..8 x hmc / mala / nuts / rwmh, d = 1..4 x rmhmc) compile in parallel
#pragma once
#include "small_samplers.hpp"
#include "small_targets.hpp"
#include "launch_common.hpp"

namespace mi {

template <int D>
int launch_small_logistic_d(int algo, const SmallParams& prm, const double* X_dev, const double* y_dev, uint32_t n_rows, hipStream_t st)
{
    using T = LogisticSmallModel<D>;
    const T tgt{X_dev, y_dev, n_rows};
    const unsigned block = 64;
    const dim3 grid((unsigned)((prm.C + block - 1) / block));
    switch (algo) {
    case 0: hipLaunchKernelGGL(hmc_small_kernel<T>, grid, dim3(block), 0, st, prm, tgt); break;
    case 1: hipLaunchKernelGGL(mala_small_kernel<T>, grid, dim3(block), 0, st, prm, tgt); break;
    case 2: hipLaunchKernelGGL(nuts_small_kernel<T>, grid, dim3(block), 0, st, prm, tgt); break;
    case 3: hipLaunchKernelGGL(rwmh_small_kernel<T>, grid, dim3(block), 0, st, prm, tgt); break;
    case 4:                 // mcmc::rmhmc with the Fisher metric: d x d x d derivative cubes per lane, d <= 4
        if constexpr (D <= 4) { hipLaunchKernelGGL(rmhmc_small_kernel<T>, grid, dim3(block), 0, st, prm, tgt); break; }
        else return (int)hipErrorInvalidValue;
    default: return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}

int launch_small_logistic_d12(int algo, int d, const SmallParams&, const double*, const double*, uint32_t, hipStream_t);
int launch_small_logistic_d34(int algo, int d, const SmallParams&, const double*, const double*, uint32_t, hipStream_t);
int launch_small_logistic_d56(int algo, int d, const SmallParams&, const double*, const double*, uint32_t, hipStream_t);
int launch_small_logistic_d78(int algo, int d, const SmallParams&, const double*, const double*, uint32_t, hipStream_t);

}  // namespace mi
