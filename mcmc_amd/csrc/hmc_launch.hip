// hmc_launch.hip -- translation unit of the HMC MFMA kernels (hmc_dense.hpp)
#include "hmc_dense.hpp"
#include "hmc_split.hpp"
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
    note_kernel("hmc_gauss_mfma_kernel<%d, %d, false, false, false>", NT, WPB);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WPB), lds, st, prm);
    return (int)hipGetLastError();
}

// diagonal precond_mat, no bounds: the plain kernel's shape with the two mass tables next to P in LDS
template <int NT, bool PCM = false>
int plain_diagm(const HmcParams& prm, hipStream_t st)
{
    constexpr int WPB = MI_HMC_WPB;
    const size_t lds = (size_t)NT * 4 * NT * 64 * sizeof(double) + (size_t)16 * NT * (4 * sizeof(double) + sizeof(int));
    auto kern = hmc_gauss_mfma_kernel<NT, WPB, false, false, true, PCM>;
    MI_LAUNCH_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned grid = (unsigned)((prm.C + 16 * WPB - 1) / (16 * WPB));
    note_kernel("hmc_gauss_mfma_kernel<%d, %d, false, false, true%s>", NT, WPB, PCM ? ", true" : "");
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WPB), lds, st, prm);
    return (int)hipGetLastError();
}

// one wave per SIMD (WPB = 4): when the chains do not fill the chip at two waves per SIMD
template <int NT>
int plain4(const HmcParams& prm, hipStream_t st)
{
    const size_t lds = (size_t)NT * 4 * NT * 64 * sizeof(double);
    auto kern = hmc_gauss_mfma_kernel<NT, 4>;
    MI_LAUNCH_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    note_kernel("hmc_gauss_mfma_kernel<%d, 4, false, false, false>", NT);
    hipLaunchKernelGGL(kern, dim3((unsigned)((prm.C + 63) / 64)), dim3(256), lds, st, prm);
    return (int)hipGetLastError();
}

// SPLIT waves per chain tile (hmc_split.hpp): fewer tiles than SIMDs
template <int NT, int SPLIT, int WPB>
int split(const HmcParams& prm, hipStream_t st)
{
    const size_t lds = hmc_split_lds_bytes<NT, SPLIT, WPB>();
    int dev = 0, lds_max = 0;
    MI_LAUNCH_TRY(hipGetDevice(&dev));
    MI_LAUNCH_TRY(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
    if ((size_t)lds_max < lds) return (int)hipErrorInvalidValue;
    auto kern = hmc_gauss_split_kernel<NT, SPLIT, WPB>;
    MI_LAUNCH_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned chains_per_wg = 16 * (WPB / SPLIT);
    note_kernel("hmc_gauss_split_kernel<%d, %d, %d>", NT, SPLIT, WPB);
    hipLaunchKernelGGL(kern, dim3((unsigned)((prm.C + chains_per_wg - 1) / chains_per_wg)), dim3(64 * WPB), lds, st, prm);
    return (int)hipGetLastError();
}

}  // namespace

int launch_hmc_gauss(const HmcParams& prm, int nt, bool gen, bool dense_m, hipStream_t st)
{
    if (dense_m) return launch_hmc_gauss_dense_m(prm, nt, st);      // hmc_dense_launch.hip
    if (gen) return launch_hmc_gauss_general(prm, nt, st);         // hmc_general_launch.hip
    return MI_DISPATCH_NT(nt, plain<1>(prm, st), plain<2>(prm, st), plain<4>(prm, st), plain<8>(prm, st));
}

int launch_hmc_gauss_diagm(const HmcParams& prm, int nt, hipStream_t st)
{
    if (prm.m_per_chain)      // per-chain masses (mi_chains.mass_diag): the tables are [d][C] in global memory
        return MI_DISPATCH_NT(nt, (plain_diagm<1, true>(prm, st)), (plain_diagm<2, true>(prm, st)), (plain_diagm<4, true>(prm, st)), (plain_diagm<8, true>(prm, st)));
    return MI_DISPATCH_NT(nt, plain_diagm<1>(prm, st), plain_diagm<2>(prm, st), plain_diagm<4>(prm, st), plain_diagm<8>(prm, st));
}

// plain case, 64 < d <= 128 (nt = 8), few chains: shape 1 = one wave per SIMD, one tile per wave; 2 = two waves per tile, one
// wave per SIMD; 3 = four waves per tile, two waves per SIMD (32 chains per workgroup like shape 2); 4 = four waves per tile,
// one wave per SIMD (16 chains per workgroup)
int launch_hmc_gauss_few_chains(const HmcParams& prm, int shape, hipStream_t st)
{
    return shape == 1 ? plain4<8>(prm, st) : shape == 2 ? split<8, 2, 4>(prm, st) : shape == 3 ? split<8, 4, 8>(prm, st) : split<8, 4, 4>(prm, st);
}

}  // namespace mi
