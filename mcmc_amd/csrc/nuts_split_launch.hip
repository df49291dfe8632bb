// nuts_split_launch.hip -- translation unit of nuts_gauss_split_kernel (nuts_split.hpp): the plain NUTS case at d in (64, 128]
#include "nuts_split.hpp"
#include "launchers.hpp"
#include "launch_common.hpp"

namespace mi {

int launch_nuts_gauss_split(const NutsParams& prm_in, int nt, double* pfrag, hipStream_t st)
{
    if (nt <= 4) return (int)hipErrorInvalidValue;
    constexpr int NT = 8;
    NutsParams prm = prm_in;
    hipLaunchKernelGGL(pack_precision_fragments_kernel<NT>, dim3(NT * 4 * NT), dim3(64), 0, st, prm.P, prm.d, pfrag);
    MI_LAUNCH_TRY(hipGetLastError());
    prm.Pfrag = pfrag;
    const size_t lds = nuts_split_lds_bytes<NT>();
    auto kern = nuts_gauss_split_kernel<NT>;
    note_kernel("nuts_gauss_split_kernel<%d>", NT);
    MI_LAUNCH_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)((prm.C + 63) / 64)), dim3(512), lds, st, prm);
    return (int)hipGetLastError();
}

}  // namespace mi
