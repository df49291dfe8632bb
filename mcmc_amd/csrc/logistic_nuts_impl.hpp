// logistic_nuts_impl.hpp -- launch templates of the nuts instantiations of the LDS-streamed kernel, shared by logistic_nuts.hip (identity /
// diagonal precond_mat) and logistic_nuts_box.hip (settings.vals_bound)
#pragma once
#include "logistic_lds.hpp"
#include "launch_common.hpp"
#include "lds_nuts_pieces.hpp"

namespace mi {
namespace {

// the persistent grid: as many workgroups as the chip holds at once (LDS: one per CU for the wide tiles, more for the narrow ones), or
// fewer if the chains are few
template <int NTQ, int TARGET>
uint64_t grid_of(uint64_t C)
{
    using G = LogitGeo<NTQ>;
    int dev = 0, n_cu = 256, per_cu = 1;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (n_cu <= 0) n_cu = 256;
    auto kern = logit_lds_kernel<NTQ, LOGIT_NUTS, TARGET, false>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 512, G::LDS_BYTES) != hipSuccess || per_cu < 1) per_cu = 1;
    const uint64_t need = (C + 31) / 32, cap = (uint64_t)n_cu * (uint64_t)per_cu;
    return cap_grid(need < cap ? need : cap);
}

template <int NTQ, int TARGET, bool DIAGM = false, bool BOUNDS = false>
int launch_nuts(LogitParams& prm, const double* X_dev, const double* y_dev, void* workspace, hipStream_t st)
{
    using G = LogitGeo<NTQ>;
    const size_t n_wg = (size_t)grid_of<NTQ, TARGET>(prm.C);
    double* xp = static_cast<double*>(workspace);
    prm.state = xp + (size_t)prm.NB * G::XBUF_PAD;       // (the hmc / mala kernels' accepted state: not used by nuts, kept in the layout)
    double* nxt = prm.state + n_wg * 8 * LOGIT_STATE_VECS * G::NSQ * 64;
    prm.xexch = nullptr;
    if (TARGET == LOGIT_TARGET_DENSE) { prm.xexch = nxt; nxt += n_wg * 2 * 4 * G::NSQ * 64; }
    prm.nuts_ws = nxt;                                   // every vector is stored before it is loaded: no memset
    prm.nuts_sc = nxt + n_wg * 8 * lds_nuts::vec_doubles_per_wave(G::NSQ);
    prm.nuts_next = reinterpret_cast<uint32_t*>(prm.nuts_sc + n_wg * 8 * lds_nuts::sc_doubles_per_wave());
    prm.Xp = xp;
    hipError_t e = hipMemsetAsync(prm.nuts_next, 0, 64, st);
    if (e != hipSuccess) return (int)e;
    double* theta_backup = nullptr;                      // more chains than chain slots: the runs are cut into pieces (lds_nuts_pieces.hpp)
    if (int ep = lds_nuts_setup_pieces<NTQ>(prm, n_wg, st, &theta_backup)) return ep;
    hipLaunchKernelGGL((pack_logit_lds_kernel<NTQ, TARGET == LOGIT_TARGET_DENSE>), dim3(prm.NB), dim3(256), 0, st, X_dev, y_dev, prm.d, prm.n_rows, xp);
    auto kern = logit_lds_kernel<NTQ, LOGIT_NUTS, TARGET, DIAGM, BOUNDS>;
    if (BOUNDS) note_kernel("logit_lds_kernel<%d, nuts, %d, true, true>", NTQ, TARGET);
    else note_kernel("logit_lds_kernel<%d, nuts, %d, %s>", NTQ, TARGET, DIAGM ? "true" : "false");
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(512), G::LDS_BYTES, st, prm);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    return lds_nuts_restore_flagged(prm, theta_backup, st);
}

template <bool DIAGM, bool BOUNDS>
int dispatch_nuts(LogitParams& prm, const double* X_dev, const double* y_dev, void* workspace, hipStream_t st, int target)
{
    if (target == LOGIT_TARGET_DENSE) {                 // 128 < d <= 512 (smaller d: nuts_memo.hpp keeps P resident in LDS)
        if (prm.d <= 192) return launch_nuts<3, LOGIT_TARGET_DENSE, DIAGM, BOUNDS>(prm, X_dev, y_dev, workspace, st);
        if (prm.d <= 256) return launch_nuts<4, LOGIT_TARGET_DENSE, DIAGM, BOUNDS>(prm, X_dev, y_dev, workspace, st);
        if (prm.d <= 384) return launch_nuts<6, LOGIT_TARGET_DENSE, DIAGM, BOUNDS>(prm, X_dev, y_dev, workspace, st);
        return launch_nuts<8, LOGIT_TARGET_DENSE, DIAGM, BOUNDS>(prm, X_dev, y_dev, workspace, st);
    }
    if (prm.d <= 64) return launch_nuts<1, LOGIT_TARGET_LOGISTIC, DIAGM, BOUNDS>(prm, X_dev, y_dev, workspace, st);
    if (prm.d <= 128) return launch_nuts<2, LOGIT_TARGET_LOGISTIC, DIAGM, BOUNDS>(prm, X_dev, y_dev, workspace, st);
    if (prm.d <= 256) return launch_nuts<4, LOGIT_TARGET_LOGISTIC, DIAGM, BOUNDS>(prm, X_dev, y_dev, workspace, st);
    return launch_nuts<8, LOGIT_TARGET_LOGISTIC, DIAGM, BOUNDS>(prm, X_dev, y_dev, workspace, st);
}

}  // namespace
// nuts with settings.vals_bound (logistic_nuts_box.hip)
int logit_lds_launch_nuts_box(LogitParams prm, const double* X_dev, const double* y_dev, void* workspace, hipStream_t st, int target);
}  // namespace mi
