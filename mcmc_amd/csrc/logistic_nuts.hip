// logistic_nuts.hip -- translation unit of the nuts instantiations of the LDS-streamed kernel (logistic_lds.hpp + nuts_lds.hpp); same
// compile modes as logistic_lds.hip (see logistic_launch.hpp for why these kernels are apart from mi_mcmc.hip).
#define MI_KC_MODE 2
#define MI_RNG_NOINLINE 1
#include "logistic_lds.hpp"
#include "launch_common.hpp"

namespace mi {
namespace {

template <int NTQ, int TARGET, bool DIAGM = false>
int launch_nuts(LogitParams& prm, const double* X_dev, const double* y_dev, void* workspace, hipStream_t st)
{
    if constexpr (!DIAGM) {                              // a diagonal precond_mat: the same launch with the DIAGM instantiation
        if (prm.m_sqrt != nullptr) return launch_nuts<NTQ, TARGET, true>(prm, X_dev, y_dev, workspace, st);
    }
    using G = LogitGeo<NTQ>;
    const size_t n_wg = (prm.C + 31) / 32;
    double* xp = static_cast<double*>(workspace);
    prm.state = xp + (size_t)prm.NB * G::XBUF_PAD;
    double* nxt = prm.state + n_wg * 8 * 2 * G::NSQ * 64;
    prm.xexch = nullptr;
    if (TARGET == LOGIT_TARGET_DENSE) { prm.xexch = nxt; nxt += n_wg * 2 * 4 * G::NSQ * 64; }
    prm.nuts_ws = nxt;                                   // every vector is stored before it is loaded: no memset
    prm.nuts_sc = nxt + n_wg * 8 * lds_nuts::vec_doubles_per_wave(G::NSQ);
    prm.Xp = xp;
    hipLaunchKernelGGL((pack_logit_lds_kernel<NTQ, TARGET == LOGIT_TARGET_DENSE>), dim3(prm.NB), dim3(256), 0, st, X_dev, y_dev, prm.d, prm.n_rows, xp);
    auto kern = logit_lds_kernel<NTQ, LOGIT_NUTS, TARGET, DIAGM>;
    note_kernel("logit_lds_kernel<%d, nuts, %d, %s>", NTQ, TARGET, DIAGM ? "true" : "false");
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(512), G::LDS_BYTES, st, prm);
    return (int)hipGetLastError();
}

}  // namespace

int logit_lds_launch_nuts(LogitParams prm, const double* X_dev, const double* y_dev, void* workspace, hipStream_t st, int target)
{
    if (target == LOGIT_TARGET_DENSE) {                 // 128 < d <= 512 (smaller d: nuts_reg.hpp keeps P resident in LDS)
        if (prm.d <= 192) return launch_nuts<3, LOGIT_TARGET_DENSE>(prm, X_dev, y_dev, workspace, st);
        if (prm.d <= 256) return launch_nuts<4, LOGIT_TARGET_DENSE>(prm, X_dev, y_dev, workspace, st);
        if (prm.d <= 384) return launch_nuts<6, LOGIT_TARGET_DENSE>(prm, X_dev, y_dev, workspace, st);
        return launch_nuts<8, LOGIT_TARGET_DENSE>(prm, X_dev, y_dev, workspace, st);
    }
    if (prm.d <= 64) return launch_nuts<1, LOGIT_TARGET_LOGISTIC>(prm, X_dev, y_dev, workspace, st);
    if (prm.d <= 128) return launch_nuts<2, LOGIT_TARGET_LOGISTIC>(prm, X_dev, y_dev, workspace, st);
    if (prm.d <= 256) return launch_nuts<4, LOGIT_TARGET_LOGISTIC>(prm, X_dev, y_dev, workspace, st);
    return launch_nuts<8, LOGIT_TARGET_LOGISTIC>(prm, X_dev, y_dev, workspace, st);
}

}  // namespace mi
