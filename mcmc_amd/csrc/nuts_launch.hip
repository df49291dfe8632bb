// nuts_launch.hip -- translation unit of the NUTS MFMA kernels of the plain case (nuts_memo.hpp; the tick-local nuts_async.hpp, lock-step
// predecessor nuts_dense.hpp); the bounded / preconditioned variants compile in nuts_general_launch.hip and nuts_dense_launch.hip
#include "nuts_async_launch.hpp"
#include "nuts_memo.hpp"
#include "launchers.hpp"
#include "launch_common.hpp"

namespace mi {
namespace {

// every doubling on a memoised trajectory (nuts_memo.hpp): a persistent grid, chains handed out dynamically
template <int NT, bool DIAGM>
size_t memo_lds()
{
    // fragments | 52 per-chain rows of 64 eight-byte columns (nuts_memo.hpp: R_END) | the test table | the mass tables
    return ((size_t)NT * 4 * NT * 64 + 52 * 64) * sizeof(double) + 10 * 48 * sizeof(uint16_t) + (DIAGM ? 32 * NT : 0) * sizeof(double);
}
// waves per workgroup and workgroups of the persistent grid: 4 waves (64 chain slots) per workgroup when there are enough chains to fill the chip
// that way; with fewer chains 2, then 1 -- every CU gets a workgroup, and a wave its SIMD, the LDS port and the L1 to itself
struct MemoShape { uint32_t waves; uint64_t grid; };
template <int NT, bool DIAGM>
MemoShape memo_shape(uint64_t C)
{
    const size_t lds = memo_lds<NT, DIAGM>();
    auto kern = nuts_gauss_memo_kernel<NT, DIAGM>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int dev = 0, n_cu = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (n_cu <= 0) n_cu = 256;
    const uint32_t waves = (C <= (uint64_t)16 * n_cu) ? 1u : (C <= (uint64_t)32 * n_cu) ? 2u : 4u;
    int per_cu = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), (int)(64 * waves), lds) != hipSuccess || per_cu < 1) per_cu = 1;
    const uint64_t need = (C + 16 * waves - 1) / (16 * waves), cap = (uint64_t)n_cu * (uint64_t)per_cu;
    return MemoShape{waves, cap_grid(need < cap ? need : cap)};
}
// Runs cut into pieces: a piece that ends stores its chain's theta over prm.theta, but a chain that is FLAGGED in a later piece (non-finite regime) is replayed
// from its INITIAL values -- the launcher keeps a copy of prm.theta, and the flagged chains' columns come back from it before the replay reads them
__global__ void restore_flagged_theta_kernel(const uint32_t* __restrict__ flag, const double* __restrict__ backup, double* __restrict__ theta, uint64_t C)
{
    if (flag[C] == 0u) return;                           // (the "any chain flagged" word)
    const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (c < C && flag[c] != 0u) theta[(size_t)blockIdx.y * C + c] = backup[(size_t)blockIdx.y * C + c];
}
template <int NT, bool DIAGM, bool PRE>
int run_memo_k(const NutsParams& prm_in, hipStream_t st)
{
    NutsParams prm = prm_in;
    const size_t lds = memo_lds<NT, DIAGM>();
    auto kern = nuts_gauss_memo_kernel<NT, DIAGM, PRE>;
    note_kernel("nuts_gauss_memo_kernel<%d, %s, %s>", NT, DIAGM ? "true" : "false", PRE ? "true" : "false");
    const MemoShape sh = memo_shape<NT, DIAGM>(prm_in.C);       // (the occupancy of the in-tick instantiation: the same LDS, the same launch bounds)
    MI_LAUNCH_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    MI_LAUNCH_TRY(hipMemsetAsync(prm.next_chain, 0, sizeof(uint32_t), st));
    // More chains than chain slots: a slot runs several chains one after the other, and the run ends when the slot with the most work does -- up to one
    // whole chain after the mean load (configs[3]: 4 chains per slot, ~13 % of the run).  Cut into MEMO_PIECES pieces, the work items are a quarter as long:
    // a piece that is not the first is its chain's continuation in whatever slot is free (nuts_memo_core.hpp, SPLIT; same draws: a continuation call's hand-over)
    double* theta_backup = nullptr;
    MI_LAUNCH_TRY((hipError_t)memo_setup_pieces(prm, prm.split_ws, sh.grid * (uint64_t)sh.waves * 16u, st, &theta_backup));
    if (theta_backup != nullptr) {
        if (prm.nf_flag != nullptr)                      // (the initial values of chains that may be flagged after their first piece: see restore_flagged_theta_kernel)
            MI_LAUNCH_TRY(hipMemcpyAsync(theta_backup, prm.theta, (size_t)prm.d * prm.C * sizeof(double), hipMemcpyDeviceToDevice, st));
        else theta_backup = nullptr;
    }
    if constexpr (PRE) {                                 // every momentum of the run, at full occupancy, before the latency-bound tick starts
        const uint64_t n_waves = ((prm.C + 15) / 16) * (uint64_t)(prm.n_burnin + prm.n_keep);
        if (n_waves > 0) hipLaunchKernelGGL((nuts_momenta_kernel<NT, DIAGM>), dim3((unsigned)((n_waves + 3) / 4)), dim3(256), 0, st, prm);
        MI_LAUNCH_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)sh.grid), dim3(64 * sh.waves), lds, st, prm);
    MI_LAUNCH_TRY(hipGetLastError());
    if (theta_backup != nullptr)
        hipLaunchKernelGGL(restore_flagged_theta_kernel, dim3((unsigned)((prm.C + 255) / 256), prm.d), dim3(256), 0, st, prm.nf_flag, theta_backup, prm.theta, prm.C);
    return (int)hipGetLastError();
}
template <int NT, bool DIAGM>
int run_memo(const NutsParams& prm, hipStream_t st)
{
    // (the table only without a mass matrix: with the two mass tables the d = 128 instantiation that reads it spills -- 80 B of scratch, 664 ms against
    //  631 ms generating in the tick on a diagonal precond_mat, round 5: 652 -- so the host does not offer it there)
    if constexpr (!DIAGM) { if (prm.mom != nullptr) return run_memo_k<NT, false, true>(prm, st); }
    return run_memo_k<NT, DIAGM, false>(prm, st);
}

#ifdef MI_WITH_LEGACY_KERNELS   // the lock-step first-generation kernel (nuts_dense.hpp: 2.4 KB of scratch per lane at d = 128) ships in the
                                // A/B library only (`make prof`); the shipped library ignores its hint, as mi_mcmc.h says of a hint it cannot take
template <int NT>
int lockstep(const NutsParams& prm, hipStream_t st)
{
    const size_t lds = ((size_t)NT * 4 * NT * 64 + (size_t)NUTS_LVLS * 4 * 64) * sizeof(double);
    auto kern = nuts_gauss_mfma_kernel<NT>;
    note_kernel("nuts_gauss_mfma_kernel<%d>", NT);
    MI_LAUNCH_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)((prm.C + 63) / 64)), dim3(256), lds, st, prm);
    return (int)hipGetLastError();
}
#endif

}  // namespace

int launch_nuts_gauss(const NutsParams& prm, int nt, bool gen, bool dense_m, bool ls, uint32_t batch, hipStream_t st)
{
    if (batch < 1) batch = 1;
    if (dense_m) return launch_nuts_gauss_dense_m(prm, nt, batch, st);       // nuts_dense_launch.hip
    if (gen) return launch_nuts_gauss_general(prm, nt, batch, st);          // nuts_general_launch.hip
#ifdef MI_WITH_LEGACY_KERNELS
    if (ls) return MI_DISPATCH_NT(nt, lockstep<1>(prm, st), lockstep<2>(prm, st), lockstep<4>(prm, st), lockstep<8>(prm, st));
#else
    (void)ls;                                             // not in this build: the tick-local asynchronous kernel serves the request, same bits
#endif
    return MI_DISPATCH_NT(nt, (async<1, false, false>(prm, batch, st)), (async<2, false, false>(prm, batch, st)), (async<4, false, false>(prm, batch, st)), (async<8, false, false>(prm, batch, st)));
}

int launch_nuts_gauss_memo(const NutsParams& prm, int nt, hipStream_t st, bool diag_m)
{
    if (diag_m) return MI_DISPATCH_NT(nt, (run_memo<1, true>(prm, st)), (run_memo<2, true>(prm, st)), (run_memo<4, true>(prm, st)), (run_memo<8, true>(prm, st)));
    return MI_DISPATCH_NT(nt, (run_memo<1, false>(prm, st)), (run_memo<2, false>(prm, st)), (run_memo<4, false>(prm, st)), (run_memo<8, false>(prm, st)));
}

size_t nuts_memo_momenta_bytes(uint64_t C, uint32_t n_total, int nt)
{
    const int ns = 4 * (nt <= 1 ? 1 : nt == 2 ? 2 : nt <= 4 ? 4 : 8);
    return memo_momenta_bytes(C, n_total, ns);
}

size_t nuts_split_workspace_bytes(uint64_t C, uint32_t d) { return memo_split_bytes(C, d); }

size_t nuts_memo_workspace_bytes(uint64_t C, int nt, bool diag_m)
{
    const MemoShape sh = diag_m ? MI_DISPATCH_NT(nt, (memo_shape<1, true>(C)), (memo_shape<2, true>(C)), (memo_shape<4, true>(C)), (memo_shape<8, true>(C)))
                                : MI_DISPATCH_NT(nt, (memo_shape<1, false>(C)), (memo_shape<2, false>(C)), (memo_shape<4, false>(C)), (memo_shape<8, false>(C)));
    const int ns = 4 * (nt <= 1 ? 1 : nt == 2 ? 2 : nt <= 4 ? 4 : 8);
    return (size_t)sh.grid * sh.waves * memo_wave_bytes(ns);
}

}  // namespace mi
