// logistic_nuts_box.hip -- translation unit of the nuts instantiations with settings.vals_bound of the LDS-streamed kernel (nuts_lds.hpp,
// lds_box.hpp); same compile modes as logistic_lds.hip.
#define MI_KC_MODE 2
#define MI_RNG_NOINLINE 1
#include "logistic_nuts_impl.hpp"

namespace mi {

int logit_lds_launch_nuts_box(LogitParams prm, const double* X_dev, const double* y_dev, void* workspace, hipStream_t st, int target)
{
    return dispatch_nuts<true, true>(prm, X_dev, y_dev, workspace, st, target);
}

}  // namespace mi
