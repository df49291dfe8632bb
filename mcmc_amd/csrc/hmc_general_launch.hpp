// hmc_general_launch.hpp -- launcher template of the general HMC variant (bounds / precond_mat), shared by hmc_general_launch.hip
// (diagonal) and hmc_dense_launch.hip (dense precond_mat)
#pragma once
#include "hmc_dense.hpp"
#include "launch_common.hpp"

namespace mi {
namespace {

template <int NT, bool DENSE_M>
int general(const HmcParams& prm, hipStream_t st)
{
    constexpr int WPB = 4;      // one wave per SIMD: the general variant holds more register-resident vectors
    const size_t mat = (size_t)NT * 4 * NT * 64 * sizeof(double);
    const size_t lds = mat * ((DENSE_M && NT <= 4) ? 3 : 1) + (size_t)16 * NT * (4 * sizeof(double) + sizeof(int));   // d > 64: Minv, L in L2
    auto kern = hmc_gauss_mfma_kernel<NT, WPB, true, DENSE_M>;
    MI_LAUNCH_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned grid = (unsigned)((prm.C + 16 * WPB - 1) / (16 * WPB));
    note_kernel("hmc_gauss_mfma_kernel<%d, %d, true, %s, false>", NT, WPB, DENSE_M ? "true" : "false");
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WPB), lds, st, prm);
    return (int)hipGetLastError();
}

}  // namespace
}  // namespace mi
