// launchers.hpp -- host-side interface between the C ABI (mi_mcmc.hip) and the kernel translation units.
// Every kernel family is compiled in its own .hip file (hmc_launch.hip, mala_launch.hip, nuts_launch.hip, rwmh_launch.hip,
// small_launch.hip, logistic_lds.hip) so that the gfx950 code generation of the ~50 template instantiations runs in parallel; the functions
// below pick the instantiation for (padded dimension tiles nt, variant) and enqueue it on `st`.
// Return value: a hipError_t as int (0 = enqueued).
#pragma once

#include <hip/hip_runtime.h>

namespace mi {

struct HmcParams;
struct MalaParams;
struct NutsParams;
struct RwmhParams;
struct SmallParams;
namespace lit { struct LitParams; }

// nt = ceil(d / 16) in {1, 2, 3..4, 5..8}; general: bounds and / or diagonal precond; dense_m: dense precond (nt <= 4)
int launch_hmc_gauss(const HmcParams& prm, int nt, bool general, bool dense_m, hipStream_t st);
int launch_hmc_gauss_general(const HmcParams& prm, int nt, hipStream_t st);      // hmc_general_launch.hip
int launch_hmc_gauss_dense_m(const HmcParams& prm, int nt, hipStream_t st);      // hmc_dense_launch.hip
// plain case with nt = 8 (64 < d <= 128) when the chains do not fill the chip at two waves per SIMD (strong scaling):
// shape 1 = one wave per SIMD (hmc_gauss_mfma_kernel<8, 4>); hmc_split.hpp: 2 = two waves per 16-chain tile, 3 = four waves per
// tile at two waves per SIMD, 4 = four waves per tile at one wave per SIMD
int launch_hmc_gauss_few_chains(const HmcParams& prm, int shape, hipStream_t st);
int launch_hmc_gauss_diagm(const HmcParams& prm, int nt, hipStream_t st);        // diagonal precond_mat, no bounds: the plain kernel's shape
// variant: 0 plain, 1 general (bounds / diagonal precond), 2 dense precond (unbounded, nt <= 4)
int launch_mala_gauss(const MalaParams& prm, int nt, int variant, hipStream_t st);
// lockstep: the first-generation kernel (plain variant only); batch: momentum-refresh batch of the asynchronous kernel
int launch_nuts_gauss(const NutsParams& prm, int nt, bool general, bool dense_m, bool lockstep, uint32_t batch, hipStream_t st);
// the plain case (unbounded; identity or diagonal precond_mat): nuts_memo.hpp -- every doubling on a memoised trajectory, chains handed out dynamically
// (prm.next_chain: one uint32_t of device memory); its workspace is sized by the chain SLOTS of the
// persistent grid (prm.ws must hold nuts_memo_workspace_bytes)
int launch_nuts_gauss_memo(const NutsParams& prm, int nt, hipStream_t st, bool diag_m = false);
size_t nuts_memo_workspace_bytes(uint64_t C, int nt, bool diag_m);
// bytes behind NutsParams::split_ws: the piece queues of a run cut into pieces and stand-ins for outputs the caller did not ask for (nuts_launch.hip)
size_t nuts_split_workspace_bytes(uint64_t C, uint32_t d);
struct TileParams;
int nuts_tile_setup_pieces(TileParams& p, void* split_ws, hipStream_t st);          // nuts_bounded_launch.hip: the same cut on the tile route's persistent grid
// bytes of the table of momenta the memoised kernel reads when prm.mom is set (filled by its launcher's pre-pass: nuts_memo.hpp)
size_t nuts_memo_momenta_bytes(uint64_t C, uint32_t n_total, int nt);
uint64_t nuts_tile_grid(uint64_t C);                                // nuts_bounded_launch.hip: workgroups of the persistent grid of the tile-policy tick
int launch_nuts_gauss_general(const NutsParams& prm, int nt, uint32_t batch, hipStream_t st);     // nuts_general_launch.hip
// settings.vals_bound (and / or a diagonal precond_mat next to it) on the memoised tick: the built-in Gaussian as a tile target with TileGen
// (nuts_bounded_launch.hip); prm.ws must hold nuts_bounded_workspace_bytes, prm.btype / lb / ub / m_sqrt / m_inv the tables
int launch_nuts_gauss_bounded(const NutsParams& prm, int nt, hipStream_t st);
size_t nuts_bounded_workspace_bytes(uint64_t C, int nt);
int launch_nuts_gauss_dense_m(const NutsParams& prm, int nt, uint32_t batch, hipStream_t st);     // nuts_dense_launch.hip
int launch_rwmh_gauss(const RwmhParams& prm, int nt, bool general, bool dense_c, hipStream_t st);
// one lane per chain, d = 2 normal model (rmhmc_small.hpp, small_samplers.hpp); algo: 0 hmc, 1 mala, 2 nuts, 3 rwmh, 4 rmhmc
int launch_small_normal_model(int algo, const SmallParams& prm, hipStream_t st);

// the same engine on LogisticSmallModel<d> (small_targets.hpp), d = 1..8; algo: 0 hmc, 1 mala, 2 nuts, 3 rwmh
int launch_small_logistic(int algo, int d, const SmallParams& prm, const double* X_dev, const double* y_dev, uint32_t n_rows, hipStream_t st);


// literal.hpp: replay of flagged chains, or (prm.flag == nullptr) the run itself; algo 0 hmc, 1 mala, 2 nuts, 3 rwmh; n_wg workgroups of 256 threads
int launch_literal(int algo, const lit::LitParams& prm, unsigned n_wg, hipStream_t st);

}  // namespace mi
