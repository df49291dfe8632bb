// logistic_nuts.hip -- translation unit of the nuts instantiations of the LDS-streamed kernel (logistic_lds.hpp + nuts_lds.hpp); same
// compile modes as logistic_lds.hip (see logistic_launch.hpp for why these kernels are apart from mi_mcmc.hip).
#define MI_KC_MODE 2
#define MI_RNG_NOINLINE 1
#include "logistic_nuts_impl.hpp"

namespace mi {


uint64_t logit_lds_nuts_workgroups(uint32_t d, uint64_t C, int target)
{
    if (target == LOGIT_TARGET_DENSE)
        return d <= 192 ? grid_of<3, LOGIT_TARGET_DENSE>(C) : d <= 256 ? grid_of<4, LOGIT_TARGET_DENSE>(C)
             : d <= 384 ? grid_of<6, LOGIT_TARGET_DENSE>(C) : grid_of<8, LOGIT_TARGET_DENSE>(C);
    return d <= 64 ? grid_of<1, LOGIT_TARGET_LOGISTIC>(C) : d <= 128 ? grid_of<2, LOGIT_TARGET_LOGISTIC>(C)
         : d <= 256 ? grid_of<4, LOGIT_TARGET_LOGISTIC>(C) : grid_of<8, LOGIT_TARGET_LOGISTIC>(C);
}

size_t logit_lds_nuts_split_bytes(uint64_t C, uint32_t d)
{
    // tails | queues | stand-ins for 3 counters, the step size and the dual-averaging state (7 C words) | the copy of the initial values
    return lds_nuts_split_bytes(C, d);
}

int logit_lds_launch_nuts(LogitParams prm, const double* X_dev, const double* y_dev, void* workspace, hipStream_t st, int target)
{
    if (prm.btype != nullptr) return logit_lds_launch_nuts_box(prm, X_dev, y_dev, workspace, st, target);       // logistic_nuts_box.hip
    return prm.m_sqrt != nullptr ? dispatch_nuts<true, false>(prm, X_dev, y_dev, workspace, st, target)
                                 : dispatch_nuts<false, false>(prm, X_dev, y_dev, workspace, st, target);
}

}  // namespace mi
