// logistic_hmc_dense_m.hip -- translation unit of the hmc instantiations with a DENSE precond_mat of the LDS-streamed kernel
// (logistic_lds.hpp: DENSEM); same compile modes as logistic_lds.hip.  Also the sizes and the dispatch of both DENSEM samplers.
#define MI_KC_MODE 2
#define MI_RNG_NOINLINE 1
#include "logistic_dense_m_impl.hpp"

namespace mi {

size_t logit_lds_dense_m_bytes(uint32_t d, uint64_t C, int target, int algo)
{
    const size_t n = (target == LOGIT_TARGET_DENSE)
                         ? ((d <= 192) ? dense_m_doubles<3, LOGIT_TARGET_DENSE>(d, C, algo) : (d <= 256) ? dense_m_doubles<4, LOGIT_TARGET_DENSE>(d, C, algo)
                            : (d <= 384) ? dense_m_doubles<6, LOGIT_TARGET_DENSE>(d, C, algo) : dense_m_doubles<8, LOGIT_TARGET_DENSE>(d, C, algo))
                   : (d <= 64) ? dense_m_doubles<1, LOGIT_TARGET_LOGISTIC>(d, C, algo) : (d <= 128) ? dense_m_doubles<2, LOGIT_TARGET_LOGISTIC>(d, C, algo)
                   : (d <= 256) ? dense_m_doubles<4, LOGIT_TARGET_LOGISTIC>(d, C, algo) : dense_m_doubles<8, LOGIT_TARGET_LOGISTIC>(d, C, algo);
    return n * sizeof(double);
}

int logit_lds_launch_hmc_dense_m(LogitParams prm, const double* X_dev, const double* y_dev, void* workspace, void* mws, hipStream_t st, int target)
{
    return launch_dense_m_any<LOGIT_HMC>(prm, X_dev, y_dev, workspace, mws, st, target);
}

int logit_lds_launch_dense_m(int algo, LogitParams prm, const double* X_dev, const double* y_dev, void* workspace, void* mws, hipStream_t st, int target)
{
    return algo == LOGIT_MALA ? logit_lds_launch_mala_dense_m(prm, X_dev, y_dev, workspace, mws, st, target)
                              : logit_lds_launch_hmc_dense_m(prm, X_dev, y_dev, workspace, mws, st, target);
}

}  // namespace mi
