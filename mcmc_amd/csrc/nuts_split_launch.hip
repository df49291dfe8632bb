// nuts_split_launch.hip -- translation unit of nuts_gauss_split_kernel (nuts_split.hpp): the plain NUTS case at d in (64, 128]
#include "nuts_split.hpp"
#include "launchers.hpp"
#include "launch_common.hpp"

namespace mi {

namespace {
template <int NT, int TPW>
int split(const NutsParams& prm, hipStream_t st)
{
    const size_t lds = nuts_split_lds_bytes<NT, TPW>();
    auto kern = nuts_gauss_split_kernel<NT, TPW>;
    note_kernel("nuts_gauss_split_kernel<%d, %d>", NT, TPW);
    MI_LAUNCH_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)((prm.C + 16 * TPW - 1) / (16 * TPW))), dim3(128 * TPW), lds, st, prm);
    return (int)hipGetLastError();
}
}  // namespace

int launch_nuts_gauss_split(const NutsParams& prm_in, int nt, int tiles_per_wg, double* pfrag, hipStream_t st)
{
    if (nt <= 4) return (int)hipErrorInvalidValue;
    constexpr int NT = 8;
    NutsParams prm = prm_in;
    hipLaunchKernelGGL(pack_precision_fragments_kernel<NT>, dim3(NT * 4 * NT), dim3(64), 0, st, prm.P, prm.d, pfrag);
    MI_LAUNCH_TRY(hipGetLastError());
    prm.Pfrag = pfrag;
    return tiles_per_wg >= 4 ? split<NT, 4>(prm, st) : tiles_per_wg == 2 ? split<NT, 2>(prm, st) : split<NT, 1>(prm, st);
}

}  // namespace mi
