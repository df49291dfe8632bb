// hmc_general_launch.hip -- hmc_gauss_mfma_kernel<NT, 4, BOUNDED = true>: vals_bound and / or a diagonal precond_mat
#include "hmc_general_launch.hpp"
#include "launchers.hpp"

namespace mi {

int launch_hmc_gauss_general(const HmcParams& prm, int nt, hipStream_t st)
{
    return MI_DISPATCH_NT(nt, (general<1, false>(prm, st)), (general<2, false>(prm, st)), (general<4, false>(prm, st)), (general<8, false>(prm, st)));
}

}  // namespace mi
