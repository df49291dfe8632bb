// nuts_general_launch.hip -- nuts_gauss_async_kernel<NT, GENERAL = true>: vals_bound and / or a diagonal precond_mat
#include "nuts_async_launch.hpp"
#include "launchers.hpp"

namespace mi {

int launch_nuts_gauss_general(const NutsParams& prm, int nt, uint32_t batch, hipStream_t st)
{
    return MI_DISPATCH_NT(nt, (async<1, true, false>(prm, batch, st)), (async<2, true, false>(prm, batch, st)), (async<4, true, false>(prm, batch, st)), (async<8, true, false>(prm, batch, st)));
}

}  // namespace mi
