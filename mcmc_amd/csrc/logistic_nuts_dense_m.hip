// logistic_nuts_dense_m.hip -- translation unit of the nuts instantiations with a DENSE precond_mat of the LDS-streamed kernel (logistic_lds.hpp + nuts_lds.hpp:
// DENSEM); same compile modes as logistic_lds.hip.  The launch is launch_nuts's (logistic_nuts_impl.hpp: persistent grid, workspace sized by its chain
// slots) plus the block images of INV(M) and CHOL_LOWER(M) in a buffer of their own, as for hmc / mala (logistic_dense_m_impl.hpp).
#define MI_KC_MODE 2
#define MI_RNG_NOINLINE 1
#include "logistic_lds.hpp"
#include "launch_common.hpp"
#include "lds_nuts_pieces.hpp"

namespace mi {
namespace {

// the persistent grid of THIS instantiation (grid_of of logistic_nuts_impl.hpp would instantiate the kernel without the dense matrix here as well)
template <int NTQ, int TARGET>
uint64_t grid_dm(uint64_t C)
{
    using G = LogitGeo<NTQ>;
    int dev = 0, n_cu = 256, per_cu = 1;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (n_cu <= 0) n_cu = 256;
    auto kern = logit_lds_kernel<NTQ, LOGIT_NUTS, TARGET, false, false, true>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 512, G::LDS_BYTES) != hipSuccess || per_cu < 1) per_cu = 1;
    const uint64_t need = (C + 31) / 32, cap = (uint64_t)n_cu * (uint64_t)per_cu;
    return cap_grid(need < cap ? need : cap);
}

template <int NTQ, int TARGET>
size_t nuts_dense_m_doubles(uint32_t d, uint64_t C)
{
    using G = LogitGeo<NTQ>;
    const size_t nbm = (d + 15) / 16, n_wg = (size_t)grid_dm<NTQ, TARGET>(C);
    return 2 * nbm * G::XBUF_PAD + (TARGET == LOGIT_TARGET_DENSE ? 0 : n_wg * 2 * 4 * G::NSQ * 64);      // + the exchange vectors of the streamed products (logistic target)
}

template <int NTQ, int TARGET>
int launch_nuts_dense_m(LogitParams& prm, const double* X_dev, const double* y_dev, void* workspace, void* mws, hipStream_t st)
{
    using G = LogitGeo<NTQ>;
    const size_t n_wg = (size_t)grid_dm<NTQ, TARGET>(prm.C);
    const uint32_t nbm = (prm.d + 15) / 16;
    const size_t img = (size_t)nbm * G::XBUF_PAD;
    double* xp = static_cast<double*>(workspace);
    prm.state = xp + (size_t)prm.NB * G::XBUF_PAD;
    double* nxt = prm.state + n_wg * 8 * LOGIT_STATE_VECS * G::NSQ * 64;
    double* m0 = static_cast<double*>(mws);
    if (TARGET == LOGIT_TARGET_DENSE) { prm.xexch = nxt; nxt += n_wg * 2 * 4 * G::NSQ * 64; }
    else prm.xexch = m0 + 2 * img;
    prm.nuts_ws = nxt;
    prm.nuts_sc = nxt + n_wg * 8 * lds_nuts::vec_doubles_per_wave(G::NSQ);
    prm.nuts_next = reinterpret_cast<uint32_t*>(prm.nuts_sc + n_wg * 8 * lds_nuts::sc_doubles_per_wave());
    prm.Xp = xp;
    hipError_t e = hipMemsetAsync(prm.nuts_next, 0, 64, st);
    if (e != hipSuccess) return (int)e;
    double* theta_backup = nullptr;                      // more chains than chain slots: the runs are cut into pieces (lds_nuts_pieces.hpp)
    if (int ep = lds_nuts_setup_pieces<NTQ>(prm, n_wg, st, &theta_backup)) return ep;
    hipLaunchKernelGGL((pack_logit_lds_kernel<NTQ, TARGET == LOGIT_TARGET_DENSE>), dim3(prm.NB), dim3(256), 0, st, X_dev, y_dev, prm.d, prm.n_rows, xp);
    auto pack = [&](const double* rm, double* dst) {
        hipLaunchKernelGGL((pack_logit_lds_kernel<NTQ, true>), dim3(nbm), dim3(256), 0, st, rm, nullptr, prm.d, prm.d, dst);
    };
    prm.Lp = m0; pack(prm.L_rm, m0);
    prm.Mip = m0 + img; pack(prm.Minv_rm, m0 + img);
    auto kern = logit_lds_kernel<NTQ, LOGIT_NUTS, TARGET, false, false, true>;
    note_kernel("logit_lds_kernel<%d, nuts, %d, false, false, true>", NTQ, TARGET);
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(512), G::LDS_BYTES, st, prm);
    return lds_nuts_restore_flagged(prm, theta_backup, st);
}

}  // namespace

uint64_t logit_lds_nuts_dense_m_workgroups(uint32_t d, uint64_t C, int target)
{
    if (target == LOGIT_TARGET_DENSE)
        return d <= 192 ? grid_dm<3, LOGIT_TARGET_DENSE>(C) : d <= 256 ? grid_dm<4, LOGIT_TARGET_DENSE>(C)
             : d <= 384 ? grid_dm<6, LOGIT_TARGET_DENSE>(C) : grid_dm<8, LOGIT_TARGET_DENSE>(C);
    return d <= 64 ? grid_dm<1, LOGIT_TARGET_LOGISTIC>(C) : d <= 128 ? grid_dm<2, LOGIT_TARGET_LOGISTIC>(C)
         : d <= 256 ? grid_dm<4, LOGIT_TARGET_LOGISTIC>(C) : grid_dm<8, LOGIT_TARGET_LOGISTIC>(C);
}

size_t logit_lds_nuts_dense_m_bytes(uint32_t d, uint64_t C, int target)
{
    const size_t n = (target == LOGIT_TARGET_DENSE)
                         ? ((d <= 192) ? nuts_dense_m_doubles<3, LOGIT_TARGET_DENSE>(d, C) : (d <= 256) ? nuts_dense_m_doubles<4, LOGIT_TARGET_DENSE>(d, C)
                            : (d <= 384) ? nuts_dense_m_doubles<6, LOGIT_TARGET_DENSE>(d, C) : nuts_dense_m_doubles<8, LOGIT_TARGET_DENSE>(d, C))
                   : (d <= 64) ? nuts_dense_m_doubles<1, LOGIT_TARGET_LOGISTIC>(d, C) : (d <= 128) ? nuts_dense_m_doubles<2, LOGIT_TARGET_LOGISTIC>(d, C)
                   : (d <= 256) ? nuts_dense_m_doubles<4, LOGIT_TARGET_LOGISTIC>(d, C) : nuts_dense_m_doubles<8, LOGIT_TARGET_LOGISTIC>(d, C);
    return n * sizeof(double);
}

int logit_lds_launch_nuts_dense_m(LogitParams prm, const double* X_dev, const double* y_dev, void* workspace, void* mws, hipStream_t st, int target)
{
    if (target == LOGIT_TARGET_DENSE) {
        if (prm.d <= 192) return launch_nuts_dense_m<3, LOGIT_TARGET_DENSE>(prm, X_dev, y_dev, workspace, mws, st);
        if (prm.d <= 256) return launch_nuts_dense_m<4, LOGIT_TARGET_DENSE>(prm, X_dev, y_dev, workspace, mws, st);
        if (prm.d <= 384) return launch_nuts_dense_m<6, LOGIT_TARGET_DENSE>(prm, X_dev, y_dev, workspace, mws, st);
        return launch_nuts_dense_m<8, LOGIT_TARGET_DENSE>(prm, X_dev, y_dev, workspace, mws, st);
    }
    if (prm.d <= 64) return launch_nuts_dense_m<1, LOGIT_TARGET_LOGISTIC>(prm, X_dev, y_dev, workspace, mws, st);
    if (prm.d <= 128) return launch_nuts_dense_m<2, LOGIT_TARGET_LOGISTIC>(prm, X_dev, y_dev, workspace, mws, st);
    if (prm.d <= 256) return launch_nuts_dense_m<4, LOGIT_TARGET_LOGISTIC>(prm, X_dev, y_dev, workspace, mws, st);
    return launch_nuts_dense_m<8, LOGIT_TARGET_LOGISTIC>(prm, X_dev, y_dev, workspace, mws, st);
}

}  // namespace mi
