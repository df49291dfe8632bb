// logistic_launch.hpp -- host-side interface of the logistic-regression kernels (logistic_lds.hpp).
// The kernels live in their own translation unit (logistic_lds.hip) because they are compiled with MI_KC_MODE 2
// (det_math.hpp): their SGPR file has no room for hoisted polynomial coefficients, the RNG-bound kernels of mi_mcmc.hip
// want exactly that.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace mi {

struct LogitParams {
    const double* Xp;       // [NB][XBUF_PAD]  blocks of 16 rows (and their labels) in the LDS image layout (see LogitGeo)
    uint32_t d, n_rows, NB;
    uint64_t C, chain0;
    double* theta;          // [d][C] in/out
    double* state;          // workspace: accepted (beta, grad) of every chain, wave-local layout (see kernel)
    double* draws;
    uint64_t* n_accept;
    uint64_t seed;
    uint32_t n_burnin, n_keep, n_leap;
    uint32_t draw0;         // index of this call's first draw in the chains' random streams (mi_chains.draw0)
    double eps, s2, rs, cons_term, log_det;
    uint32_t* nf_flag;      // [C + 1] or nullptr: chains that reached the non-finite regime are flagged and left to literal.hpp
    const double* m_sqrt;   // hmc with a DIAGONAL precond_mat (no bounds): diagonal of CHOL_LOWER(M) and of INV(M) on the device, PADDED with ones to 512 entries, or nullptr = identity
    const double* m_inv;
    const double* m;        // mala with a diagonal precond_mat: its diagonal and the diagonal of INV(eps^2 M) (padded likewise); m_sqrt as above
    const double* s_inv;
    const int* btype;       // hmc / nuts with settings.vals_bound (lds_box.hpp): bounds type 1..4, lower / upper bound of every dimension on the device, PADDED to
    const double* lb;       // 512 entries with type 1 -- or nullptr = unbounded.  With them m_sqrt / m_inv are set too (ones for the identity).
    const double* ub;
    double* xexch;          // dense Gaussian target (and DENSEM): [chain tile][4 NSQ][64] the vector of a streamed product, shared by the tile's four waves
    // nuts (nuts_lds.hpp): workspace vectors / per-chain scalars of every wave (set by the launcher), outputs and the dual-averaging settings
    double* nuts_ws;
    double* nuts_sc;
    uint32_t* nuts_next;    // chains handed out beyond the first gridDim.x * 32 (zeroed by the launcher)
    uint64_t* n_leap_out;   // [C] leapfrog steps of every chain as the reference counts them (one per leaf), or nullptr
    uint64_t* n_exec_out;   // [C] leapfrogs really made (one per distinct point of a doubling's trajectory + the step-size search), or nullptr
    double* step_out;       // [C] step sizes: out (and in, for a continuation: draw0 > 0), or nullptr
    uint32_t* depth_trace;  // [n_total][C] tree depth of every draw, or nullptr
    double* adapt_state;    // [3][C] dual-averaging state (h, eps_bar, mu): out (and in, for a continuation inside the window), or nullptr
    uint32_t n_adapt, max_depth;
    double delta, eps_bar0, gamma, t0, kappa;
    // hmc / mala with a DENSE precond_mat (logit_lds_kernel<.., DENSEM>), d*d row-major on the device (in): hmc INV(M) and CHOL_LOWER(M); mala
    // M, CHOL_LOWER(M) and INV(eps^2 M) -- and their transposed block images (set by the launcher)
    const double* Minv_rm;
    const double* L_rm;
    const double* M_rm;
    const double* Sinv_rm;
    const double* Mip;
    const double* Lp;
    const double* Mp;
    const double* Sip;
    // nuts without bounds / a dense precond_mat (nuts_lds.hpp, "runs cut into pieces"): n_pieces > 1 cuts every chain's run into pieces of piece_len draws handed
    // out as separate work items; piece_q [n_pieces - 1][C] (0xffffffff = not published) and piece_tail [n_pieces] are set up by the launcher inside split_ws
    // (logit_lds_nuts_split_bytes(C, d) bytes from the caller, or nullptr: no pieces), which then also makes sure n_accept, n_leap_out, n_exec_out, step_out and
    // adapt_state exist (the hand-over goes through them) and keeps a copy of the initial values for the replay of chains flagged after their first piece
    uint32_t n_pieces, piece_len;
    uint32_t* piece_q;
    uint32_t* piece_tail;
    void* split_ws;
};

enum { LOGIT_MALA = 0, LOGIT_HMC = 1, LOGIT_RWMH = 2, LOGIT_NUTS = 3 };   // RWMH: eps carries par_scale (identity cov_mat); NUTS: nuts_lds.hpp
// The target the streamed matrix belongs to.  DENSE: the Gaussian log K = -1/2 x'Px with 128 < d <= 512 -- "X" is P (n_rows = d,
// X_dev = P row-major, y_dev = nullptr), the block images hold P transposed and the evaluation is the X^T r phase alone with r = x.
enum { LOGIT_TARGET_LOGISTIC = 0, LOGIT_TARGET_DENSE = 1 };

// bytes of device workspace a launch needs (block images of X, accepted state of every chain)
size_t logit_lds_workspace_bytes(uint32_t d, uint32_t NB, uint64_t C, int target = LOGIT_TARGET_LOGISTIC, int algo = LOGIT_HMC);
// nuts: workgroups of the persistent grid (32 chain slots each; the workspace is sized by chains = 32 * this)
uint64_t logit_lds_nuts_workgroups(uint32_t d, uint64_t C, int target);
// nuts: bytes behind LogitParams::split_ws
size_t logit_lds_nuts_split_bytes(uint64_t C, uint32_t d);
// hmc / mala with a dense precond_mat (prm.L_rm and Minv_rm resp. M_rm / Sinv_rm set): bytes of the second workspace `mws` of
// logit_lds_launch_dense_m -- the block images of the two resp. three matrices and, for the logistic target, the exchange vectors of the
// streamed products
size_t logit_lds_dense_m_bytes(uint32_t d, uint64_t C, int target, int algo);
int logit_lds_launch_dense_m(int algo, LogitParams prm, const double* X_dev, const double* y_dev, void* workspace, void* mws, hipStream_t st, int target);
// nuts with a dense precond_mat (prm.L_rm, prm.Minv_rm set; logistic_nuts_dense_m.hip): bytes of its `mws` and the launch (workspace: as for nuts)
size_t logit_lds_nuts_dense_m_bytes(uint32_t d, uint64_t C, int target);
uint64_t logit_lds_nuts_dense_m_workgroups(uint32_t d, uint64_t C, int target);        // workgroups of ITS persistent grid (the workspace is sized by 32 x this)
int logit_lds_launch_nuts_dense_m(LogitParams prm, const double* X_dev, const double* y_dev, void* workspace, void* mws, hipStream_t st, int target);
int logit_lds_launch_hmc_dense_m(LogitParams prm, const double* X_dev, const double* y_dev, void* workspace, void* mws, hipStream_t st, int target);     // logistic_hmc_dense_m.hip
int logit_lds_launch_mala_dense_m(LogitParams prm, const double* X_dev, const double* y_dev, void* workspace, void* mws, hipStream_t st, int target);    // logistic_mala_dense_m.hip
// packs X / y into `workspace` and runs the sampler on `st`; returns a hipError_t value (0 = launched)
int logit_lds_launch(int algo, LogitParams prm, const double* X_dev, const double* y_dev, void* workspace, hipStream_t st,
                     int target = LOGIT_TARGET_LOGISTIC);

}  // namespace mi
