// nuts_async_launch.hpp -- launcher template of nuts_gauss_async_kernel, shared by the translation units that instantiate its variants
// (nuts_launch.hip: plain; nuts_general_launch.hip: bounds / diagonal precond_mat; nuts_dense_launch.hip: dense precond_mat)
#pragma once
#include "nuts_async.hpp"
#include "launch_common.hpp"

namespace mi {
namespace {

template <int NT, bool GENERAL, bool DENSE_M>
int async(const NutsParams& prm, uint32_t batch, hipStream_t st)
{
    const size_t lds = ((size_t)NT * 4 * NT * 64 * ((DENSE_M && NT <= 4) ? 3 : 1) + (size_t)NUTS_LVLS * 4 * 64 + 3 * 64) * sizeof(double)
                     + (GENERAL ? (size_t)16 * NT * (4 * sizeof(double) + sizeof(int)) : 0);
    auto kern = nuts_gauss_async_kernel<NT, GENERAL, DENSE_M>;
    note_kernel("nuts_gauss_async_kernel<%d, %s, %s>", NT, GENERAL ? "true" : "false", DENSE_M ? "true" : "false");
    MI_LAUNCH_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)((prm.C + 63) / 64)), dim3(256), lds, st, prm, batch);
    return (int)hipGetLastError();
}

}  // namespace
}  // namespace mi
