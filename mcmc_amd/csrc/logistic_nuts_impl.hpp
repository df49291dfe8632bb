// logistic_nuts_impl.hpp -- launch templates of the nuts instantiations of the LDS-streamed kernel, shared by logistic_nuts.hip (identity /
// diagonal precond_mat) and logistic_nuts_box.hip (settings.vals_bound)
#pragma once
#include "logistic_lds.hpp"
#include "launch_common.hpp"

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

#ifndef MI_LDS_NUTS_PIECES
#define MI_LDS_NUTS_PIECES 4
#endif
constexpr uint32_t LDS_NUTS_PIECES = MI_LDS_NUTS_PIECES;
inline size_t lds_nuts_queue_bytes(uint64_t C) { return ((size_t)(LDS_NUTS_PIECES - 1u) * C * sizeof(uint32_t) + 255) & ~(size_t)255; }
// a chain that is flagged in a later piece is replayed from its INITIAL values, which its earlier pieces have overwritten in prm.theta: the launcher's copy comes back
__global__ void lds_nuts_restore_flagged_theta_kernel(const uint32_t* __restrict__ flag, const double* __restrict__ backup, double* __restrict__ theta, uint64_t C)
{
    if (flag[C] == 0u) return;                           // (the "any chain flagged" word)
    const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (c < C && flag[c] != 0u) theta[(size_t)blockIdx.y * C + c] = backup[(size_t)blockIdx.y * C + c];
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
    // More chains than chain slots: the runs are cut into pieces, as nuts_launch.hip cuts those of nuts_gauss_memo_kernel (the reasoning is there; the protocol
    // in nuts_lds.hpp).  With bounds the hand-over carries theta in the transformed space (nuts_lds.hpp).  Pieces of two draws
    // and more: a hand-over costs one evaluation, a draw here tens of them
    prm.n_pieces = 1; prm.piece_len = 0; prm.piece_q = nullptr; prm.piece_tail = nullptr;
    double* theta_backup = nullptr;
    if constexpr (NTQ > 1) {                  // (NTQ = 1, d <= 64: measured 2 % SLOWER cut -- 65 536 chains of d = 20: 56.1 -> 57.1 ms; the wider tiles gain 4-9 %)
        const uint32_t n_total = prm.n_burnin + prm.n_keep;
        if (prm.split_ws != nullptr && prm.nf_flag != nullptr && prm.C > (uint64_t)n_wg * 32u && n_total >= 2u * LDS_NUTS_PIECES && prm.C < (1ull << 28)) {
            prm.piece_len = (n_total + LDS_NUTS_PIECES - 1u) / LDS_NUTS_PIECES;
            prm.n_pieces = (n_total + prm.piece_len - 1u) / prm.piece_len;
            char* b = static_cast<char*>(prm.split_ws);
            prm.piece_tail = reinterpret_cast<uint32_t*>(b);
            prm.piece_q = reinterpret_cast<uint32_t*>(b + 256);
            const size_t q_bytes = lds_nuts_queue_bytes(prm.C);
            if ((e = hipMemsetAsync(prm.piece_tail, 0, 256, st)) != hipSuccess) return (int)e;
            if ((e = hipMemsetAsync(prm.piece_q, 0xff, q_bytes, st)) != hipSuccess) return (int)e;
            uint64_t* u = reinterpret_cast<uint64_t*>(b + 256 + q_bytes);       // stand-ins for what the hand-over goes through
            if (!prm.n_accept) prm.n_accept = u;
            if (!prm.n_leap_out) prm.n_leap_out = u + prm.C;
            if (!prm.n_exec_out) prm.n_exec_out = u + 2 * prm.C;
            double* dd = reinterpret_cast<double*>(u + 3 * prm.C);
            if (!prm.step_out) prm.step_out = dd;
            if (!prm.adapt_state) prm.adapt_state = dd + prm.C;
            theta_backup = dd + 4 * prm.C + 32;
            if ((e = hipMemcpyAsync(theta_backup, prm.theta, (size_t)prm.d * prm.C * sizeof(double), hipMemcpyDeviceToDevice, st)) != hipSuccess) return (int)e;
        }
    }
    hipLaunchKernelGGL((pack_logit_lds_kernel<NTQ, TARGET == LOGIT_TARGET_DENSE>), dim3(prm.NB), dim3(256), 0, st, X_dev, y_dev, prm.d, prm.n_rows, xp);
    auto kern = logit_lds_kernel<NTQ, LOGIT_NUTS, TARGET, DIAGM, BOUNDS>;
    if (BOUNDS) note_kernel("logit_lds_kernel<%d, nuts, %d, true, true>", NTQ, TARGET);
    else note_kernel("logit_lds_kernel<%d, nuts, %d, %s>", NTQ, TARGET, DIAGM ? "true" : "false");
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(512), G::LDS_BYTES, st, prm);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    if (theta_backup != nullptr)
        hipLaunchKernelGGL(lds_nuts_restore_flagged_theta_kernel, dim3((unsigned)((prm.C + 255) / 256), prm.d), dim3(256), 0, st, prm.nf_flag, theta_backup, prm.theta, prm.C);
    return (int)hipGetLastError();
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
