// logistic_dense_m_impl.hpp -- launch templates of the DENSEM instantiations of the LDS-streamed kernel (logistic_lds.hpp: hmc / mala with a dense
// precond_mat), shared by logistic_hmc_dense_m.hip and logistic_mala_dense_m.hip
#pragma once
#include "logistic_lds_impl.hpp"

namespace mi {
namespace {

constexpr int dense_m_matrices(int algo) { return algo == LOGIT_MALA ? 3 : 2; }

template <int NTQ, int TARGET>
size_t dense_m_doubles(uint32_t d, uint64_t C, int algo)
{
    using G = LogitGeo<NTQ>;
    const size_t nbm = (d + 15) / 16, n_wg = (C + 31) / 32;
    return (size_t)dense_m_matrices(algo) * nbm * G::XBUF_PAD + (TARGET == LOGIT_TARGET_DENSE ? 0 : n_wg * 2 * 4 * G::NSQ * 64);
}

template <int NTQ, int ALGO, int TARGET>
int launch_dense_m(LogitParams& prm, const double* X_dev, const double* y_dev, void* workspace, void* mws, hipStream_t st)
{
    using G = LogitGeo<NTQ>;
    const size_t n_wg = (prm.C + 31) / 32;
    const uint32_t nbm = (prm.d + 15) / 16;
    const size_t img = (size_t)nbm * G::XBUF_PAD;
    double* xp = static_cast<double*>(workspace);
    prm.state = xp + (size_t)prm.NB * G::XBUF_PAD;
    double* m0 = static_cast<double*>(mws);
    prm.xexch = (TARGET == LOGIT_TARGET_DENSE) ? prm.state + n_wg * 8 * LOGIT_STATE_VECS * G::NSQ * 64 : m0 + (size_t)dense_m_matrices(ALGO) * img;
    prm.Xp = xp;
    hipLaunchKernelGGL((pack_logit_lds_kernel<NTQ, TARGET == LOGIT_TARGET_DENSE>), dim3(prm.NB), dim3(256), 0, st, X_dev, y_dev, prm.d, prm.n_rows, xp);
    auto pack = [&](const double* rm, double* dst) {
        hipLaunchKernelGGL((pack_logit_lds_kernel<NTQ, true>), dim3(nbm), dim3(256), 0, st, rm, nullptr, prm.d, prm.d, dst);
    };
    prm.Lp = m0; pack(prm.L_rm, m0);
    if (ALGO == LOGIT_MALA) {
        prm.Mp = m0 + img; pack(prm.M_rm, m0 + img);
        prm.Sip = m0 + 2 * img; pack(prm.Sinv_rm, m0 + 2 * img);
    } else {
        prm.Mip = m0 + img; pack(prm.Minv_rm, m0 + img);
    }
    auto kern = logit_lds_kernel<NTQ, ALGO, TARGET, false, false, true>;
    note_kernel("logit_lds_kernel<%d, %d, %d, false, false, true>", NTQ, ALGO, TARGET);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(512), G::LDS_BYTES, st, prm);
    return (int)hipGetLastError();
}

template <int ALGO>
int launch_dense_m_any(LogitParams& prm, const double* X_dev, const double* y_dev, void* workspace, void* mws, hipStream_t st, int target)
{
    if (target == LOGIT_TARGET_DENSE) {
        if (prm.d <= 192) return launch_dense_m<3, ALGO, LOGIT_TARGET_DENSE>(prm, X_dev, y_dev, workspace, mws, st);
        if (prm.d <= 256) return launch_dense_m<4, ALGO, LOGIT_TARGET_DENSE>(prm, X_dev, y_dev, workspace, mws, st);
        if (prm.d <= 384) return launch_dense_m<6, ALGO, LOGIT_TARGET_DENSE>(prm, X_dev, y_dev, workspace, mws, st);
        return launch_dense_m<8, ALGO, LOGIT_TARGET_DENSE>(prm, X_dev, y_dev, workspace, mws, st);
    }
    if (prm.d <= 64) return launch_dense_m<1, ALGO, LOGIT_TARGET_LOGISTIC>(prm, X_dev, y_dev, workspace, mws, st);
    if (prm.d <= 128) return launch_dense_m<2, ALGO, LOGIT_TARGET_LOGISTIC>(prm, X_dev, y_dev, workspace, mws, st);
    if (prm.d <= 256) return launch_dense_m<4, ALGO, LOGIT_TARGET_LOGISTIC>(prm, X_dev, y_dev, workspace, mws, st);
    return launch_dense_m<8, ALGO, LOGIT_TARGET_LOGISTIC>(prm, X_dev, y_dev, workspace, mws, st);
}

}  // namespace
}  // namespace mi
