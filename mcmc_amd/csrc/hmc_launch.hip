// hmc_launch.hip -- translation unit of the HMC MFMA kernels (hmc_dense.hpp)
#include "hmc_dense.hpp"
#include "launchers.hpp"
#include "launch_common.hpp"

namespace mi {
namespace {

template <int NT>
int plain(const HmcParams& prm, hipStream_t st)
{
    constexpr int WPB = MI_HMC_WPB;
    const size_t lds = (size_t)NT * 4 * NT * 64 * sizeof(double);
    auto kern = hmc_gauss_mfma_kernel<NT, WPB>;
    MI_LAUNCH_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned grid = (unsigned)((prm.C + 16 * WPB - 1) / (16 * WPB));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WPB), lds, st, prm);
    return (int)hipGetLastError();
}

template <int NT, bool DENSE_M>
int general(const HmcParams& prm, hipStream_t st)
{
    constexpr int WPB = 4;      // one wave per SIMD: the general variant holds more register-resident vectors
    const size_t mat = (size_t)NT * 4 * NT * 64 * sizeof(double);
    const size_t lds = mat * (DENSE_M ? 3 : 1) + (size_t)16 * NT * (4 * sizeof(double) + sizeof(int));
    auto kern = hmc_gauss_mfma_kernel<NT, WPB, true, DENSE_M>;
    MI_LAUNCH_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned grid = (unsigned)((prm.C + 16 * WPB - 1) / (16 * WPB));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WPB), lds, st, prm);
    return (int)hipGetLastError();
}

}  // namespace

int launch_hmc_gauss(const HmcParams& prm, int nt, bool gen, bool dense_m, hipStream_t st)
{
    if (dense_m) {
        if (nt > 4) return (int)hipErrorInvalidValue;
        return nt <= 1 ? general<1, true>(prm, st) : nt == 2 ? general<2, true>(prm, st) : general<4, true>(prm, st);
    }
    if (gen) return MI_DISPATCH_NT(nt, (general<1, false>(prm, st)), (general<2, false>(prm, st)), (general<4, false>(prm, st)), (general<8, false>(prm, st)));
    return MI_DISPATCH_NT(nt, plain<1>(prm, st), plain<2>(prm, st), plain<4>(prm, st), plain<8>(prm, st));
}

}  // namespace mi
