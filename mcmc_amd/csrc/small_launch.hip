// small_launch.hip -- translation unit of the one-lane-per-chain kernels for small-dimensional targets (rmhmc_small.hpp)
#include "rmhmc_small.hpp"
#include "launchers.hpp"
#include "launch_common.hpp"

namespace mi {

int launch_rmhmc_normal_model(const SmallParams& prm, hipStream_t st)
{
    NormalModel tgt{prm.data, prm.n_rows};
    const unsigned block = 64;        // one wave per workgroup: C chains spread over as many CUs as possible
    hipLaunchKernelGGL(rmhmc_small_kernel<NormalModel>, dim3((unsigned)((prm.C + block - 1) / block)), dim3(block), 0, st, prm, tgt);
    return (int)hipGetLastError();
}

}  // namespace mi
