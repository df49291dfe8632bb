// logistic_lds.hip -- translation unit of the logistic-regression kernels (see logistic_launch.hpp for why it is separate).
#define MI_KC_MODE 2
#define MI_RNG_NOINLINE 1      // the Philox + Box-Muller pair as an out-of-line leaf function: smaller per-draw code, fewer spills (config 3: 61.2 -> 56.2 ms per 20 draws)
#include "logistic_lds.hpp"
#include "launch_common.hpp"

namespace mi {
namespace {

template <int NTQ>
size_t ws_doubles(uint32_t NB, uint64_t C)
{
    using G = LogitGeo<NTQ>;
    const size_t n_wg = (C + 31) / 32;
    return (size_t)NB * G::XBUF_PAD + n_wg * 8 * 2 * G::NSQ * 64;
}

template <int NTQ, int ALGO>
int launch(LogitParams& prm, const double* X_dev, const double* y_dev, void* workspace, hipStream_t st)
{
    using G = LogitGeo<NTQ>;
    const size_t n_wg = (prm.C + 31) / 32;
    double* xp = static_cast<double*>(workspace);
    prm.state = xp + (size_t)prm.NB * G::XBUF_PAD;
    prm.Xp = xp;
    hipLaunchKernelGGL(pack_logit_lds_kernel<NTQ>, dim3(prm.NB), dim3(256), 0, st, X_dev, y_dev, prm.d, prm.n_rows, xp);
    auto kern = logit_lds_kernel<NTQ, ALGO>;
    note_kernel("logit_lds_kernel<%d, %d>", NTQ, ALGO);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(512), G::LDS_BYTES, st, prm);
    return (int)hipGetLastError();
}

template <int ALGO>
int launch_any(LogitParams& prm, const double* X_dev, const double* y_dev, void* workspace, hipStream_t st)
{
    if (prm.d <= 64) return launch<1, ALGO>(prm, X_dev, y_dev, workspace, st);
    if (prm.d <= 128) return launch<2, ALGO>(prm, X_dev, y_dev, workspace, st);
    if (prm.d <= 256) return launch<4, ALGO>(prm, X_dev, y_dev, workspace, st);
    return launch<8, ALGO>(prm, X_dev, y_dev, workspace, st);
}

}  // namespace

size_t logit_lds_workspace_bytes(uint32_t d, uint32_t NB, uint64_t C)
{
    const size_t n = (d <= 64) ? ws_doubles<1>(NB, C) : (d <= 128) ? ws_doubles<2>(NB, C) : (d <= 256) ? ws_doubles<4>(NB, C) : ws_doubles<8>(NB, C);
    return n * sizeof(double);
}

int logit_lds_launch(int algo, LogitParams prm, const double* X_dev, const double* y_dev, void* workspace, hipStream_t st)
{
    return algo == LOGIT_HMC  ? launch_any<LOGIT_HMC>(prm, X_dev, y_dev, workspace, st)
         : algo == LOGIT_RWMH ? launch_any<LOGIT_RWMH>(prm, X_dev, y_dev, workspace, st)
                              : launch_any<LOGIT_MALA>(prm, X_dev, y_dev, workspace, st);
}

}  // namespace mi
