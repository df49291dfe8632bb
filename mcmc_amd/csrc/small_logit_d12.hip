// small_logit_d12.hip -- LogisticSmallModel<1>, LogisticSmallModel<2> on the one-lane-per-chain engine (small_logit_launch.hpp)
#include "small_logit_launch.hpp"

namespace mi {

int launch_small_logistic_d12(int algo, int d, const SmallParams& prm, const double* X_dev, const double* y_dev, uint32_t n_rows, hipStream_t st)
{
    return d == 1 ? launch_small_logistic_d<1>(algo, prm, X_dev, y_dev, n_rows, st) : launch_small_logistic_d<2>(algo, prm, X_dev, y_dev, n_rows, st);
}

}  // namespace mi
