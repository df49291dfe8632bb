// rwmh_launch.hip -- translation unit of the RWMH MFMA kernels (rwmh_dense.hpp)
#include "rwmh_dense.hpp"
#include "launchers.hpp"
#include "launch_common.hpp"

namespace mi {
namespace {

template <int NT, bool GENERAL, bool DENSE_C>
int gauss(const RwmhParams& prm, hipStream_t st)
{
    const size_t mat = (size_t)NT * 4 * NT * 64 * sizeof(double);
    const size_t lds = mat * ((DENSE_C && NT <= 4) ? 2 : 1) + (GENERAL ? (size_t)16 * NT * (3 * sizeof(double) + sizeof(int)) : 0);
    auto kern = rwmh_gauss_mfma_kernel<NT, GENERAL, DENSE_C>;
    note_kernel("rwmh_gauss_mfma_kernel<%d, %s, %s>", NT, GENERAL ? "true" : "false", DENSE_C ? "true" : "false");
    MI_LAUNCH_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)((prm.C + 63) / 64)), dim3(256), lds, st, prm);
    return (int)hipGetLastError();
}

}  // namespace

int launch_rwmh_gauss(const RwmhParams& prm, int nt, bool gen, bool dense_c, hipStream_t st)
{
    if (dense_c) return MI_DISPATCH_NT(nt, (gauss<1, true, true>(prm, st)), (gauss<2, true, true>(prm, st)), (gauss<4, true, true>(prm, st)), (gauss<8, true, true>(prm, st)));
    if (gen) return MI_DISPATCH_NT(nt, (gauss<1, true, false>(prm, st)), (gauss<2, true, false>(prm, st)), (gauss<4, true, false>(prm, st)), (gauss<8, true, false>(prm, st)));
    return MI_DISPATCH_NT(nt, (gauss<1, false, false>(prm, st)), (gauss<2, false, false>(prm, st)), (gauss<4, false, false>(prm, st)), (gauss<8, false, false>(prm, st)));
}

}  // namespace mi
