// small_logit_d56.hip -- LogisticSmallModel<5>, LogisticSmallModel<6> on the one-lane-per-chain engine (small_logit_launch.hpp)
#include "small_logit_launch.hpp"

namespace mi {

int launch_small_logistic_d56(int algo, int d, const SmallParams& prm, const double* X_dev, const double* y_dev, uint32_t n_rows, hipStream_t st)
{
    return d == 5 ? launch_small_logistic_d<5>(algo, prm, X_dev, y_dev, n_rows, st) : launch_small_logistic_d<6>(algo, prm, X_dev, y_dev, n_rows, st);
}

}  // namespace mi
