// mala_launch.hip -- translation unit of the MALA MFMA kernels (mala_dense.hpp)
#include "mala_dense.hpp"
#include "launchers.hpp"
#include "launch_common.hpp"

namespace mi {
namespace {

template <int NT, bool GENERAL>
int gauss(const MalaParams& prm, hipStream_t st)
{
    const size_t lds = (size_t)NT * 4 * NT * 64 * sizeof(double) + (GENERAL ? (size_t)16 * NT * (4 * sizeof(double) + sizeof(int)) : 0);
    auto kern = mala_gauss_mfma_kernel<NT, GENERAL>;
    note_kernel("mala_gauss_mfma_kernel<%d, %s>", NT, GENERAL ? "true" : "false");
    MI_LAUNCH_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)((prm.C + 63) / 64)), dim3(256), lds, st, prm);
    return (int)hipGetLastError();
}

template <int NT>
int dense_m(const MalaParams& prm, hipStream_t st)
{
    const size_t lds = (size_t)(NT <= 4 ? 4 : 1) * NT * 4 * NT * 64 * sizeof(double);
    auto kern = mala_gauss_dense_m_kernel<NT>;
    note_kernel("mala_gauss_dense_m_kernel<%d>", NT);
    MI_LAUNCH_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)((prm.C + 63) / 64)), dim3(256), lds, st, prm);
    return (int)hipGetLastError();
}

}  // namespace

int launch_mala_gauss(const MalaParams& prm, int nt, int variant, hipStream_t st)
{
    if (variant == 2) {
        return MI_DISPATCH_NT(nt, dense_m<1>(prm, st), dense_m<2>(prm, st), dense_m<4>(prm, st), dense_m<8>(prm, st));
    }
    if (variant == 1) return MI_DISPATCH_NT(nt, (gauss<1, true>(prm, st)), (gauss<2, true>(prm, st)), (gauss<4, true>(prm, st)), (gauss<8, true>(prm, st)));
    return MI_DISPATCH_NT(nt, (gauss<1, false>(prm, st)), (gauss<2, false>(prm, st)), (gauss<4, false>(prm, st)), (gauss<8, false>(prm, st)));
}

}  // namespace mi
