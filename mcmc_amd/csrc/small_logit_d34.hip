// small_logit_d34.hip -- LogisticSmallModel<3>, LogisticSmallModel<4> on the one-lane-per-chain engine (small_logit_launch.hpp)
#include "small_logit_launch.hpp"

namespace mi {

int launch_small_logistic_d34(int algo, int d, const SmallParams& prm, const double* X_dev, const double* y_dev, uint32_t n_rows, hipStream_t st)
{
    return d == 3 ? launch_small_logistic_d<3>(algo, prm, X_dev, y_dev, n_rows, st) : launch_small_logistic_d<4>(algo, prm, X_dev, y_dev, n_rows, st);
}

}  // namespace mi
