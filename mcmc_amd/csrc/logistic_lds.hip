// logistic_lds.hip -- translation unit of the logistic-regression kernels (see logistic_launch.hpp for why it is separate).
#define MI_KC_MODE 2
#define MI_RNG_NOINLINE 1      // the Philox + Box-Muller pair as an out-of-line leaf function: smaller per-draw code, fewer spills (config 3: 61.2 -> 56.2 ms per 20 draws)
#include "logistic_lds_impl.hpp"

namespace mi {


size_t logit_lds_workspace_bytes(uint32_t d, uint32_t NB, uint64_t C, int target, int algo)
{
    const size_t n = (target == LOGIT_TARGET_DENSE)
                         ? ((d <= 192) ? ws_doubles<3>(NB, C, target, algo) : (d <= 256) ? ws_doubles<4>(NB, C, target, algo)
                            : (d <= 384) ? ws_doubles<6>(NB, C, target, algo) : ws_doubles<8>(NB, C, target, algo))
                   : (d <= 64) ? ws_doubles<1>(NB, C, target, algo) : (d <= 128) ? ws_doubles<2>(NB, C, target, algo)
                   : (d <= 256) ? ws_doubles<4>(NB, C, target, algo) : ws_doubles<8>(NB, C, target, algo);
    return n * sizeof(double);
}

int logit_lds_launch(int algo, LogitParams prm, const double* X_dev, const double* y_dev, void* workspace, hipStream_t st, int target)
{
    if (algo == LOGIT_NUTS) return logit_lds_launch_nuts(prm, X_dev, y_dev, workspace, st, target);     // logistic_nuts.hip
    if (algo == LOGIT_HMC && prm.btype != nullptr) return logit_lds_launch_hmc_box(prm, X_dev, y_dev, workspace, st, target);     // logistic_hmc_box.hip
    return algo == LOGIT_HMC  ? launch_any<LOGIT_HMC>(prm, X_dev, y_dev, workspace, st, target)
         : algo == LOGIT_RWMH ? launch_any<LOGIT_RWMH>(prm, X_dev, y_dev, workspace, st, target)
                              : launch_any<LOGIT_MALA>(prm, X_dev, y_dev, workspace, st, target);
}

}  // namespace mi
