// nuts_dense_launch.hip -- nuts_gauss_async_kernel<NT, true, DENSE_M = true>: a dense precond_mat (d <= 64 in LDS, beyond from L2)
#include "nuts_async_launch.hpp"
#include "launchers.hpp"

namespace mi {

int launch_nuts_gauss_dense_m(const NutsParams& prm, int nt, uint32_t batch, hipStream_t st)
{
    return MI_DISPATCH_NT(nt, (async<1, true, true>(prm, batch, st)), (async<2, true, true>(prm, batch, st)), (async<4, true, true>(prm, batch, st)), (async<8, true, true>(prm, batch, st)));
}

}  // namespace mi
