// hmc_dense_launch.hip -- hmc_gauss_mfma_kernel<NT, 4, true, DENSE_M = true>: a dense precond_mat (d <= 64 in LDS, beyond from L2)
#include "hmc_general_launch.hpp"
#include "launchers.hpp"

namespace mi {

int launch_hmc_gauss_dense_m(const HmcParams& prm, int nt, hipStream_t st)
{
    return MI_DISPATCH_NT(nt, (general<1, true>(prm, st)), (general<2, true>(prm, st)), (general<4, true>(prm, st)), (general<8, true>(prm, st)));
}

}  // namespace mi
