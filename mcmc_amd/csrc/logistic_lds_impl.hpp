// logistic_lds_impl.hpp -- launch templates of the LDS-streamed kernels, shared by the translation units that instantiate them
// (logistic_lds.hip: mala / hmc / rwmh; logistic_hmc_box.hip: hmc with settings.vals_bound)
#pragma once
#include "logistic_lds.hpp"
#include "launch_common.hpp"

namespace mi {
namespace {

template <int NTQ>
size_t ws_doubles(uint32_t NB, uint64_t C, int target, int algo)
{
    return logit_lds_ws_doubles<NTQ>(NB, C, target, algo);
}

template <int NTQ, int ALGO, int TARGET, bool DIAGM = false, bool BOUNDS = false>
int launch(LogitParams& prm, const double* X_dev, const double* y_dev, void* workspace, hipStream_t st)
{
    if constexpr ((ALGO == LOGIT_HMC || ALGO == LOGIT_MALA) && !DIAGM) {    // a diagonal precond_mat: the same launch with the DIAGM instantiation
        if (prm.m_sqrt != nullptr) return launch<NTQ, ALGO, TARGET, true>(prm, X_dev, y_dev, workspace, st);
    }
    using G = LogitGeo<NTQ>;
    const size_t n_wg = (prm.C + 31) / 32;
    double* xp = static_cast<double*>(workspace);
    prm.state = xp + (size_t)prm.NB * G::XBUF_PAD;
    prm.xexch = (TARGET == LOGIT_TARGET_DENSE) ? prm.state + n_wg * 8 * LOGIT_STATE_VECS * G::NSQ * 64 : nullptr;
    prm.Xp = xp;
    hipLaunchKernelGGL((pack_logit_lds_kernel<NTQ, TARGET == LOGIT_TARGET_DENSE>), dim3(prm.NB), dim3(256), 0, st, X_dev, y_dev, prm.d, prm.n_rows, xp);
    auto kern = logit_lds_kernel<NTQ, ALGO, TARGET, DIAGM, BOUNDS>;
    if (BOUNDS) note_kernel("logit_lds_kernel<%d, %d, %d, true, true>", NTQ, ALGO, TARGET);
    else note_kernel("logit_lds_kernel<%d, %d, %d, %s>", NTQ, ALGO, TARGET, DIAGM ? "true" : "false");
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(512), G::LDS_BYTES, st, prm);
    return (int)hipGetLastError();
}

template <int ALGO, bool BOUNDS = false>
int launch_any(LogitParams& prm, const double* X_dev, const double* y_dev, void* workspace, hipStream_t st, int target)
{
    if (target == LOGIT_TARGET_DENSE) {                 // 128 < d <= 512 (smaller d: hmc_dense.hpp keeps P resident in LDS)
        if (prm.d <= 192) return launch<3, ALGO, LOGIT_TARGET_DENSE, BOUNDS, BOUNDS>(prm, X_dev, y_dev, workspace, st);
        if (prm.d <= 256) return launch<4, ALGO, LOGIT_TARGET_DENSE, BOUNDS, BOUNDS>(prm, X_dev, y_dev, workspace, st);
        if (prm.d <= 384) return launch<6, ALGO, LOGIT_TARGET_DENSE, BOUNDS, BOUNDS>(prm, X_dev, y_dev, workspace, st);
        return launch<8, ALGO, LOGIT_TARGET_DENSE, BOUNDS, BOUNDS>(prm, X_dev, y_dev, workspace, st);
    }
    if (prm.d <= 64) return launch<1, ALGO, LOGIT_TARGET_LOGISTIC, BOUNDS, BOUNDS>(prm, X_dev, y_dev, workspace, st);
    if (prm.d <= 128) return launch<2, ALGO, LOGIT_TARGET_LOGISTIC, BOUNDS, BOUNDS>(prm, X_dev, y_dev, workspace, st);
    if (prm.d <= 256) return launch<4, ALGO, LOGIT_TARGET_LOGISTIC, BOUNDS, BOUNDS>(prm, X_dev, y_dev, workspace, st);
    return launch<8, ALGO, LOGIT_TARGET_LOGISTIC, BOUNDS, BOUNDS>(prm, X_dev, y_dev, workspace, st);
}

}  // namespace
// hmc with settings.vals_bound (logistic_hmc_box.hip)
int logit_lds_launch_hmc_box(LogitParams prm, const double* X_dev, const double* y_dev, void* workspace, hipStream_t st, int target);
}  // namespace mi
