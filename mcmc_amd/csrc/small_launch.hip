// small_launch.hip -- translation unit of the one-lane-per-chain kernels for small-dimensional targets
// (rmhmc_small.hpp, small_samplers.hpp)
#include "small_samplers.hpp"
#include "small_logit_launch.hpp"
#include "launchers.hpp"
#include "launch_common.hpp"

namespace mi {

int launch_small_normal_model(int algo, const SmallParams& prm, hipStream_t st)
{
    NormalModel tgt{prm.data, prm.n_rows};
    const unsigned block = 64;        // one wave per workgroup: C chains spread over as many CUs as possible
    const dim3 grid((unsigned)((prm.C + block - 1) / block));
    switch (algo) {
    case 0: hipLaunchKernelGGL(hmc_small_kernel<NormalModel>, grid, dim3(block), 0, st, prm, tgt); break;
    case 1: hipLaunchKernelGGL(mala_small_kernel<NormalModel>, grid, dim3(block), 0, st, prm, tgt); break;
    case 2: hipLaunchKernelGGL(nuts_small_kernel<NormalModel>, grid, dim3(block), 0, st, prm, tgt); break;
    case 3: hipLaunchKernelGGL(rwmh_small_kernel<NormalModel>, grid, dim3(block), 0, st, prm, tgt); break;
    case 4: hipLaunchKernelGGL(rmhmc_small_kernel<NormalModel>, grid, dim3(block), 0, st, prm, tgt); break;
    default: return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}

int launch_small_logistic(int algo, int d, const SmallParams& prm, const double* X_dev, const double* y_dev, uint32_t n_rows, hipStream_t st)
{
    if (d < 1 || d > SMALL_MAX_D) return (int)hipErrorInvalidValue;
    return d <= 2 ? launch_small_logistic_d12(algo, d, prm, X_dev, y_dev, n_rows, st)
         : d <= 4 ? launch_small_logistic_d34(algo, d, prm, X_dev, y_dev, n_rows, st)
         : d <= 6 ? launch_small_logistic_d56(algo, d, prm, X_dev, y_dev, n_rows, st)
                  : launch_small_logistic_d78(algo, d, prm, X_dev, y_dev, n_rows, st);
}

}  // namespace mi
