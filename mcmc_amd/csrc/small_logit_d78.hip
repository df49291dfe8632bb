// small_logit_d78.hip -- LogisticSmallModel<7>, LogisticSmallModel<8> on the one-lane-per-chain engine (small_logit_launch.hpp)
#include "small_logit_launch.hpp"

namespace mi {

int launch_small_logistic_d78(int algo, int d, const SmallParams& prm, const double* X_dev, const double* y_dev, uint32_t n_rows, hipStream_t st)
{
    return d == 7 ? launch_small_logistic_d<7>(algo, prm, X_dev, y_dev, n_rows, st) : launch_small_logistic_d<8>(algo, prm, X_dev, y_dev, n_rows, st);
}

}  // namespace mi
