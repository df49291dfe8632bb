// logistic_hmc_box.hip -- translation unit of the hmc instantiations with settings.vals_bound of the LDS-streamed kernel (logistic_lds.hpp,
// lds_box.hpp); same compile modes as logistic_lds.hip.
#define MI_KC_MODE 2
#define MI_RNG_NOINLINE 1
#include "logistic_lds_impl.hpp"

namespace mi {

int logit_lds_launch_hmc_box(LogitParams prm, const double* X_dev, const double* y_dev, void* workspace, hipStream_t st, int target)
{
    return launch_any<LOGIT_HMC, true>(prm, X_dev, y_dev, workspace, st, target);
}

}  // namespace mi
