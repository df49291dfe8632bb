// logistic_mala_dense_m.hip -- translation unit of the mala instantiations with a DENSE precond_mat of the LDS-streamed kernel
// (logistic_lds.hpp: DENSEM); same compile modes as logistic_lds.hip.
#define MI_KC_MODE 2
#define MI_RNG_NOINLINE 1
#include "logistic_dense_m_impl.hpp"

namespace mi {

int logit_lds_launch_mala_dense_m(LogitParams prm, const double* X_dev, const double* y_dev, void* workspace, void* mws, hipStream_t st, int target)
{
    return launch_dense_m_any<LOGIT_MALA>(prm, X_dev, y_dev, workspace, mws, st, target);
}

}  // namespace mi
