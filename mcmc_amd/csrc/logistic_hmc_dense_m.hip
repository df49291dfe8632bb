// logistic_hmc_dense_m.hip -- translation unit of the hmc instantiations with a DENSE precond_mat of the LDS-streamed kernel
// (logistic_lds.hpp: DENSEM); same compile modes as logistic_lds.hip.
#define MI_KC_MODE 2
#define MI_RNG_NOINLINE 1
#include "logistic_lds_impl.hpp"

namespace mi {
namespace {

template <int NTQ, int TARGET>
size_t dense_m_doubles(uint32_t d, uint64_t C)
{
    using G = LogitGeo<NTQ>;
    const size_t nbm = (d + 15) / 16, n_wg = (C + 31) / 32;
    return 2 * nbm * G::XBUF_PAD + (TARGET == LOGIT_TARGET_DENSE ? 0 : n_wg * 2 * 4 * G::NSQ * 64);
}

template <int NTQ, int TARGET>
int launch_dense_m(LogitParams& prm, const double* X_dev, const double* y_dev, void* workspace, void* mws, hipStream_t st)
{
    using G = LogitGeo<NTQ>;
    const size_t n_wg = (prm.C + 31) / 32;
    const uint32_t nbm = (prm.d + 15) / 16;
    double* xp = static_cast<double*>(workspace);
    prm.state = xp + (size_t)prm.NB * G::XBUF_PAD;
    double* mi = static_cast<double*>(mws);
    double* lp = mi + (size_t)nbm * G::XBUF_PAD;
    prm.xexch = (TARGET == LOGIT_TARGET_DENSE) ? prm.state + n_wg * 8 * LOGIT_STATE_VECS * G::NSQ * 64 : lp + (size_t)nbm * G::XBUF_PAD;
    prm.Xp = xp; prm.Mip = mi; prm.Lp = lp;
    hipLaunchKernelGGL((pack_logit_lds_kernel<NTQ, TARGET == LOGIT_TARGET_DENSE>), dim3(prm.NB), dim3(256), 0, st, X_dev, y_dev, prm.d, prm.n_rows, xp);
    hipLaunchKernelGGL((pack_logit_lds_kernel<NTQ, true>), dim3(nbm), dim3(256), 0, st, prm.Minv_rm, nullptr, prm.d, prm.d, mi);
    hipLaunchKernelGGL((pack_logit_lds_kernel<NTQ, true>), dim3(nbm), dim3(256), 0, st, prm.L_rm, nullptr, prm.d, prm.d, lp);
    auto kern = logit_lds_kernel<NTQ, LOGIT_HMC, TARGET, false, false, true>;
    note_kernel("logit_lds_kernel<%d, %d, %d, false, false, true>", NTQ, (int)LOGIT_HMC, TARGET);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(512), G::LDS_BYTES, st, prm);
    return (int)hipGetLastError();
}

}  // namespace

size_t logit_lds_dense_m_bytes(uint32_t d, uint64_t C, int target)
{
    const size_t n = (target == LOGIT_TARGET_DENSE)
                         ? ((d <= 192) ? dense_m_doubles<3, LOGIT_TARGET_DENSE>(d, C) : (d <= 256) ? dense_m_doubles<4, LOGIT_TARGET_DENSE>(d, C)
                            : (d <= 384) ? dense_m_doubles<6, LOGIT_TARGET_DENSE>(d, C) : dense_m_doubles<8, LOGIT_TARGET_DENSE>(d, C))
                   : (d <= 64) ? dense_m_doubles<1, LOGIT_TARGET_LOGISTIC>(d, C) : (d <= 128) ? dense_m_doubles<2, LOGIT_TARGET_LOGISTIC>(d, C)
                   : (d <= 256) ? dense_m_doubles<4, LOGIT_TARGET_LOGISTIC>(d, C) : dense_m_doubles<8, LOGIT_TARGET_LOGISTIC>(d, C);
    return n * sizeof(double);
}

int logit_lds_launch_hmc_dense_m(LogitParams prm, const double* X_dev, const double* y_dev, void* workspace, void* mws, hipStream_t st, int target)
{
    if (target == LOGIT_TARGET_DENSE) {
        if (prm.d <= 192) return launch_dense_m<3, LOGIT_TARGET_DENSE>(prm, X_dev, y_dev, workspace, mws, st);
        if (prm.d <= 256) return launch_dense_m<4, LOGIT_TARGET_DENSE>(prm, X_dev, y_dev, workspace, mws, st);
        if (prm.d <= 384) return launch_dense_m<6, LOGIT_TARGET_DENSE>(prm, X_dev, y_dev, workspace, mws, st);
        return launch_dense_m<8, LOGIT_TARGET_DENSE>(prm, X_dev, y_dev, workspace, mws, st);
    }
    if (prm.d <= 64) return launch_dense_m<1, LOGIT_TARGET_LOGISTIC>(prm, X_dev, y_dev, workspace, mws, st);
    if (prm.d <= 128) return launch_dense_m<2, LOGIT_TARGET_LOGISTIC>(prm, X_dev, y_dev, workspace, mws, st);
    if (prm.d <= 256) return launch_dense_m<4, LOGIT_TARGET_LOGISTIC>(prm, X_dev, y_dev, workspace, mws, st);
    return launch_dense_m<8, LOGIT_TARGET_LOGISTIC>(prm, X_dev, y_dev, workspace, mws, st);
}

}  // namespace mi
