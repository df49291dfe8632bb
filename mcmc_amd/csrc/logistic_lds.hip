// logistic_lds.hip -- translation unit of the logistic-regression kernels (see logistic_launch.hpp for why it is separate).
#define MI_KC_MODE 2
#define MI_RNG_NOINLINE 1      // the Philox + Box-Muller pair as an out-of-line leaf function: smaller per-draw code, fewer spills (config 3: 61.2 -> 56.2 ms per 20 draws)
#include "logistic_lds.hpp"
#include "launch_common.hpp"

namespace mi {
namespace {

template <int NTQ>
size_t ws_doubles(uint32_t NB, uint64_t C, int target, int algo)
{
    return logit_lds_ws_doubles<NTQ>(NB, C, target, algo);
}

template <int NTQ, int ALGO, int TARGET, bool DIAGM = false>
int launch(LogitParams& prm, const double* X_dev, const double* y_dev, void* workspace, hipStream_t st)
{
    if constexpr ((ALGO == LOGIT_HMC || ALGO == LOGIT_MALA) && !DIAGM) {    // a diagonal precond_mat: the same launch with the DIAGM instantiation
        if (prm.m_sqrt != nullptr) return launch<NTQ, ALGO, TARGET, true>(prm, X_dev, y_dev, workspace, st);
    }
    using G = LogitGeo<NTQ>;
    const size_t n_wg = (prm.C + 31) / 32;
    double* xp = static_cast<double*>(workspace);
    prm.state = xp + (size_t)prm.NB * G::XBUF_PAD;
    prm.xexch = (TARGET == LOGIT_TARGET_DENSE) ? prm.state + n_wg * 8 * 2 * G::NSQ * 64 : nullptr;
    prm.Xp = xp;
    hipLaunchKernelGGL((pack_logit_lds_kernel<NTQ, TARGET == LOGIT_TARGET_DENSE>), dim3(prm.NB), dim3(256), 0, st, X_dev, y_dev, prm.d, prm.n_rows, xp);
    auto kern = logit_lds_kernel<NTQ, ALGO, TARGET, DIAGM>;
    note_kernel("logit_lds_kernel<%d, %d, %d, %s>", NTQ, ALGO, TARGET, DIAGM ? "true" : "false");
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(512), G::LDS_BYTES, st, prm);
    return (int)hipGetLastError();
}

template <int ALGO>
int launch_any(LogitParams& prm, const double* X_dev, const double* y_dev, void* workspace, hipStream_t st, int target)
{
    if (target == LOGIT_TARGET_DENSE) {                 // 128 < d <= 512 (smaller d: hmc_dense.hpp keeps P resident in LDS)
        if (prm.d <= 192) return launch<3, ALGO, LOGIT_TARGET_DENSE>(prm, X_dev, y_dev, workspace, st);
        if (prm.d <= 256) return launch<4, ALGO, LOGIT_TARGET_DENSE>(prm, X_dev, y_dev, workspace, st);
        if (prm.d <= 384) return launch<6, ALGO, LOGIT_TARGET_DENSE>(prm, X_dev, y_dev, workspace, st);
        return launch<8, ALGO, LOGIT_TARGET_DENSE>(prm, X_dev, y_dev, workspace, st);
    }
    if (prm.d <= 64) return launch<1, ALGO, LOGIT_TARGET_LOGISTIC>(prm, X_dev, y_dev, workspace, st);
    if (prm.d <= 128) return launch<2, ALGO, LOGIT_TARGET_LOGISTIC>(prm, X_dev, y_dev, workspace, st);
    if (prm.d <= 256) return launch<4, ALGO, LOGIT_TARGET_LOGISTIC>(prm, X_dev, y_dev, workspace, st);
    return launch<8, ALGO, LOGIT_TARGET_LOGISTIC>(prm, X_dev, y_dev, workspace, st);
}

}  // namespace

size_t logit_lds_workspace_bytes(uint32_t d, uint32_t NB, uint64_t C, int target, int algo)
{
    const size_t n = (target == LOGIT_TARGET_DENSE)
                         ? ((d <= 192) ? ws_doubles<3>(NB, C, target, algo) : (d <= 256) ? ws_doubles<4>(NB, C, target, algo)
                            : (d <= 384) ? ws_doubles<6>(NB, C, target, algo) : ws_doubles<8>(NB, C, target, algo))
                   : (d <= 64) ? ws_doubles<1>(NB, C, target, algo) : (d <= 128) ? ws_doubles<2>(NB, C, target, algo)
                   : (d <= 256) ? ws_doubles<4>(NB, C, target, algo) : ws_doubles<8>(NB, C, target, algo);
    return n * sizeof(double);
}

int logit_lds_launch(int algo, LogitParams prm, const double* X_dev, const double* y_dev, void* workspace, hipStream_t st, int target)
{
    if (algo == LOGIT_NUTS) return logit_lds_launch_nuts(prm, X_dev, y_dev, workspace, st, target);     // logistic_nuts.hip
    return algo == LOGIT_HMC  ? launch_any<LOGIT_HMC>(prm, X_dev, y_dev, workspace, st, target)
         : algo == LOGIT_RWMH ? launch_any<LOGIT_RWMH>(prm, X_dev, y_dev, workspace, st, target)
                              : launch_any<LOGIT_MALA>(prm, X_dev, y_dev, workspace, st, target);
}

}  // namespace mi
