// logistic_lds.hpp -- many-chain MALA and HMC for the Bayesian logistic-regression target, X staged through LDS
//   log K(beta) = sum_r [ y_r eta_r - log(1 + e^eta_r) ] - 1/2 |beta|^2,   eta = X beta,
//   grad        = X^T (y - sigmoid(eta)) - beta
// (BASELINE config 3: d = 512, N = 1024 rows, 262 144 chains) on the fp64 matrix cores.
//
// Replaces the draw loops of mcmc::internal::mala_impl (/root/reference/src/mala.cpp:149-186, with mala_mean_fn :97-125,
// mala_prop_adjustment /root/reference/include/mcmc/mala.ipp:30-70, stats_mcmc::dmvnorm
// /root/reference/include/stats/dmvnorm.hpp:28-54) and mcmc::internal::hmc_impl (/root/reference/src/hmc.cpp:155-205),
// identity preconditioner, no bounds (DIAGM: a diagonal precond_mat; BOUNDS, hmc: settings.vals_bound, lds_box.hpp; nuts: nuts_lds.hpp).  The fused
// value+gradient evaluation is the reference's target_log_kernel callback.
//
// Why LDS: with X fragments streamed from L2 into registers (the first version of this kernel) every 8-byte operand fed one
// MFMA of one 16-chain tile: 4 flop per L2 byte, and the kernel sat on the L2/MALL stream (6 TB/s at 25 TFLOP/s).  Here a workgroup of
// 8 waves = 2 chain tiles x 4 dimension quarters shares ONE copy of each 16-row block of X in LDS: the same bytes serve
// the eta = X beta product (A = X rows, 16 rows x 4 dims per fragment) and the X^T r product (A = X columns, 4 rows x 16
// dims per fragment) of both tiles -- 16 flop per L2 byte -- and arrive by direct-to-LDS loads (global_load_lds_dwordx4,
// no register staging) one block ahead.  Row pair p, parity e, dim j of a block lives at p*RSP + e*(DP+16) + j doubles,
// RSP = 2*DP + 34: the bank slot is (j + 16 e + 2 p) mod 32, conflict-free for both fragment shapes.
//
// Mapping and reduction orders (the oracle states the same, oracle/mcmc_oracle.c ORC_TARGET_LOGISTIC): wave (g, q) owns chain tile g
// and dims [q*DQ, (q+1)*DQ) in the MFMA B/D register layout; eta_r = ((e0+e1)+e2)+e3 with e_q = h_q0 + h_q1 the two fma
// chains over the halves of wave q's dims; X^T r rows ascending as one fma chain; the log-likelihood row sum 4-strided + butterfly; dot products over
// dimensions ((S0+S1)+S2)+S3 with S_q the 4-strided dot of block q.
#pragma once

#include "hmc_dense.hpp"
#include "logistic_launch.hpp"
#include "nuts_lds.hpp"
#include "lds_box.hpp"

#ifndef MI_LOGIT_ABLATE
#define MI_LOGIT_ABLATE 0
#endif
#ifndef MI_LOGIT_BATCH
#define MI_LOGIT_BATCH 8
#endif

namespace mi {


constexpr int LOGIT_STATE_VECS = 3;      // workspace vectors of a wave: accepted (beta, grad); BOUNDS: theta across an evaluation

template <int NTQ>
struct LogitGeo {
    static constexpr int NSQ = 4 * NTQ;          // 4-dim slices per wave
    static constexpr int DQ = 16 * NTQ;          // dims per wave
    static constexpr int DP = 64 * NTQ;          // padded dims
    static constexpr int RSP = 2 * DP + 34;      // doubles per row pair
    static constexpr int XBUF = 8 * RSP;         // doubles per 16-row block
    static constexpr int CHUNKS = (XBUF * 8 + 1023) / 1024;   // 1 KiB wave-loads per block
    static constexpr int XBUF_PAD = CHUNKS * 128;             // >= XBUF + 16: the 16 labels of the block sit at [XBUF, XBUF+16)
    static_assert(CHUNKS * 128 >= XBUF + 16, "no room for the labels");
    static constexpr int EXCH = 2 * (4 * 4 * 64 + 4 * 2 * 64);          // doubles: [g] partial eta tiles + [g] row terms
    static constexpr size_t LDS_BYTES = (size_t)(2 * XBUF_PAD + EXCH) * sizeof(double);
    __host__ __device__ static constexpr int xaddr(int row, int dim) { return (row >> 1) * RSP + (row & 1) * (DP + 16) + dim; }
};

// doubles of device workspace of a launch: block images | accepted (beta, grad) of every chain | dense: the exchanged positions |
// nuts: workspace vectors and per-chain scalars of every wave (nuts_lds.hpp)
template <int NTQ>
inline size_t logit_lds_ws_doubles(uint32_t NB, uint64_t C, int target, int algo)
{
    using G = LogitGeo<NTQ>;
    const size_t n_wg = (C + 31) / 32;
    return (size_t)NB * G::XBUF_PAD + n_wg * 8 * LOGIT_STATE_VECS * G::NSQ * 64 + (target == LOGIT_TARGET_DENSE ? n_wg * 2 * 4 * G::NSQ * 64 : 0)
           + (algo == LOGIT_NUTS ? n_wg * 8 * (lds_nuts::vec_doubles_per_wave(G::NSQ) + lds_nuts::sc_doubles_per_wave()) + 32 : 0);   // (+ the chain counter)
}
// (logistic_nuts.hip: the nuts instantiations are a translation unit of their own)
int logit_lds_launch_nuts(LogitParams prm, const double* X_dev, const double* y_dev, void* workspace, hipStream_t st, int target);

// pack X (row-major n_rows x d) into per-block LDS images; zero padding outside.  TRANSPOSED (dense Gaussian, X = P, n_rows = d):
// image row r holds column r of P, so that the X^T r phase with r = x yields P x; no labels.
template <int NTQ, bool TRANSPOSED>
__global__ void pack_logit_lds_kernel(const double* __restrict__ X, const double* __restrict__ y, uint32_t d, uint32_t n_rows,
                                      double* Xp)
{
    using G = LogitGeo<NTQ>;
    const uint32_t b = blockIdx.x;
    double* img = Xp + (size_t)b * G::XBUF_PAD;
    for (int i = threadIdx.x; i < G::XBUF_PAD; i += blockDim.x) img[i] = 0.0;
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * G::DP; i += blockDim.x) {
        const int r = i / G::DP, j = i % G::DP;
        const uint32_t row = 16 * b + r;
        if (row < n_rows && (uint32_t)j < d) img[G::xaddr(r, j)] = TRANSPOSED ? X[(size_t)j * d + row] : X[(size_t)row * d + j];
    }
    if (!TRANSPOSED && threadIdx.x < 16) {
        const uint32_t row = 16 * b + threadIdx.x;
        img[G::XBUF + threadIdx.x] = row < n_rows ? y[row] : 0.0;        // labels in the image's tail padding
    }
}

// DIAGM (hmc, mala): a DIAGONAL precond_mat without bounds (hmc.cpp:57-59,158-160,171,184: p = sqrt(m) z, theta += eps (p / m),
// K = p.(p / m) / 2; mala.cpp:57-58,123,159 with mala.ipp:58-64: mean = x + eps^2 (m grad) / 2, noise eps sqrt(m) z, INV(eps^2 M) diagonal),
// the tables read from global memory where they are used (LDS is full of X).  The reference's dense `inv_precond_matrix * mntm` is
// handled like the identity's: the non-finite regime is detected through the energies and replayed by literal.hpp with the same tables.
// BOUNDS (hmc, nuts): settings.vals_bound (lds_box.hpp) -- the sampler runs in the transformed space, the evaluation sees x = inv_transform(theta);
// always together with DIAGM (tables of ones for the identity).
// waves per SIMD the register allocation is made for: the workgroup's 8 waves are two per SIMD.  (MI_LOGIT_NUTS_W1 = 4 compiles the narrowest
// nuts instantiation for 128 registers, two workgroups per CU -- at d <= 64 an evaluation is the latency of the row terms, not matrix work.
// Measured: SLOWER, 421 vs 296 ms at d = 64, N = 1 024, 16 384 chains and 91 vs 77 ms at d = 32, N = 256: 56 spilled registers, and twice
// the chain slots leave nothing for the dynamic hand-out to balance.)
#ifndef MI_LOGIT_NUTS_W1
#define MI_LOGIT_NUTS_W1 2
#endif
template <int NTQ, int ALGO> constexpr int logit_waves_per_simd() { return (ALGO == LOGIT_NUTS && NTQ == 1) ? MI_LOGIT_NUTS_W1 : 2; }
// DENSEM (hmc, mala; no bounds): a DENSE precond_mat (hmc.cpp:57-59: inv_precond_matrix = INV(M), sqrt_precond_matrix = CHOL_LOWER(M), both from the
// host).  `sqrt_precond_matrix * rand_vec` (:158) and `inv_precond_matrix * new_mntm` (:160,171,184) are streamed through LDS block by block
// like P of the dense Gaussian -- their transposed block images follow the target's in the same double buffer, every evaluation and product
// prefetching block 0 of the matrix that comes next --, each element one fma chain over the columns in ascending order (the oracle's orc_gemv).
// mala (mala.cpp:57-58,123,159; mala.ipp:58-64 with dmvnorm.hpp:37-41): `precond_matrix * grad`, `sqrt_precond_matrix * rand_vec` and the two
// `INV(eps^2 M) * (X - mu)` of the proposal densities the same way; INV(eps^2 M) and its LOG_DET from the host once (Sigma is constant).
template <int NTQ, int ALGO, int TARGET, bool DIAGM = false, bool BOUNDS = false, bool DENSEM = false>
__global__ MI_NO_DS_MERGE __launch_bounds__(512, (logit_waves_per_simd<NTQ, ALGO>())) void logit_lds_kernel(const LogitParams prm)
{
    static_assert(!BOUNDS || (DIAGM && (ALGO == LOGIT_HMC || ALGO == LOGIT_NUTS)), "bounds: hmc and nuts, with the mass tables");
    static_assert(!DENSEM || ((ALGO == LOGIT_HMC || ALGO == LOGIT_MALA || ALGO == LOGIT_NUTS) && !DIAGM && !BOUNDS), "a dense precond_mat: hmc, mala and nuts without bounds");
    using G = LogitGeo<NTQ>;
    constexpr int NSQ = G::NSQ, DQ = G::DQ, DP = G::DP, RSP = G::RSP;
    extern __shared__ double smem[];
    double* const Xs = smem;                                   // [2][XBUF_PAD]
    double* const part_all = smem + 2 * G::XBUF_PAD;           // [g][wave q][reg][lane]
    double* const rt_all = part_all + 2 * 4 * 4 * 64;          // [g][q][0/1][lane]

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int g = w >> 2, q = w & 3;
    const int j4 = lane >> 4;
    const uint32_t d = prm.d;
    const uint64_t C = prm.C;
    const double eps = prm.eps;
    const uint32_t NB = prm.NB;
    // timing experiments only (tools/logit_bench.hip): 1 no row-term math, 2 no eta MFMAs, 4 no gradient MFMAs,
    // 8 no block DMA, 16 no barriers in the block loop, 32 no LDS fragment reads, 64 no normal draws, 128 no DMA wait,
    // 256 DMA always from block 0, 512 all DMA pieces after the first barrier of the block, 4096 all at its top (default: one per MFMA group),
    // 1024 report shader cycles and 100 MHz wall ticks of workgroup 0 in n_accept[0..1]
    constexpr uint32_t ablate = MI_LOGIT_ABLATE;
    const uint64_t t0_clk = (ablate & 1024u) ? clock64() : 0, t0_wall = (ablate & 1024u) ? wall_clock64() : 0;
    const uint64_t cl = ((uint64_t)blockIdx.x * 2 + g) * 16 + (lane & 15);
    const bool live = cl < C;
    const uint64_t chain = prm.chain0 + cl;
    double* const part = part_all + g * (4 * 4 * 64);
    double* const rt = rt_all + g * (4 * 2 * 64);
    double* const ws_wave = prm.state + ((size_t)blockIdx.x * 8 + w) * ((size_t)LOGIT_STATE_VECS * NSQ * 64) + lane;
    // Address of slice s of state vector v.  The base of each group of 8 slices is made opaque where it is used: otherwise the
    // compiler hoists all 2*NSQ loop-invariant 64-bit addresses out of the draw loop, spills them, and serialises every
    // workspace access behind a scratch reload of its address (reload, wait, load, wait).  Inside a group the 512-byte
    // slice stride folds into the instruction's immediate offset.
    auto st = [&](int v, int s) -> double* {
        double* b = ws_wave + ((size_t)v * NSQ + (s & ~7)) * 64;
        asm volatile("" : "+v"(b));
        return b + (s & 7) * 64;
    };
    // the dimension of slice s in this lane.  Opaque on purpose: the `dim < d` predicates of the fully unrolled per-draw loops
    // are loop invariants, and the compiler would otherwise keep dozens of 64-bit lane masks alive in SGPRs across the whole
    // kernel (hundreds of SGPR spills); recomputing a compare where it is used costs nothing.
    auto dim_of = [&](int s) -> uint32_t {
        uint32_t j = (uint32_t)j4;
        asm volatile("" : "+v"(j));
        return (uint32_t)(q * DQ + 4 * s) + j;
    };

    // DIAGM: entry of slice s of a mass table for this lane -- wave-uniform base + an opaque lane offset (re-loaded where it is used: as
    // loop invariants the 2 NSQ entries would be spilled; global, not flat, loads).  The tables are padded to 64 NTQ entries with ones.
    [[maybe_unused]] auto mass_at = [&](const double* tab, int s) __attribute__((always_inline)) -> double {
        uint32_t off = (uint32_t)j4 * 8u;
        asm volatile("" : "+v"(off));
        return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(tab + (q * DQ + 4 * s)) + off);
    };

    // per-lane LDS offsets of this wave's fragments inside a block image
    const int eta_off = G::xaddr(lane & 15, q * DQ + j4);                       // + 4 s
    const int grad_off = (j4 >> 1) * RSP + (j4 & 1) * (DP + 16) + q * DQ + (lane & 15);   // + 2 sp RSP + 16 t

    // Direct-to-LDS block copy: wave w moves the 1 KiB chunks w, w+8, ... (lane l -> bytes [16 l, 16 l + 16) of the chunk).
    // Issued from inline asm so that the compiler does not serialise them against its own LDS traffic (it waits vmcnt(0)
    // before every ds_write otherwise); wait_loads() + a barrier is the explicit completion point.
    const uint32_t xs_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)Xs;
    auto issue_block = [&](uint32_t b, int buf) __attribute__((always_inline)) {
        const double* src = prm.Xp + (size_t)b * G::XBUF_PAD + lane * 2;       // (the target's images: the first block of a launch, and the timing experiments)
        const uint32_t dst = xs_lds + (uint32_t)buf * (uint32_t)(G::XBUF_PAD * sizeof(double));
#pragma unroll
        for (int c0 = 0; c0 < G::CHUNKS; c0 += 8) {
            const int c = c0 + w;
            if (c < G::CHUNKS) {
                uint32_t m0_saved;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(m0_saved) : "v"(src + (size_t)c * 128), "s"(dst + (uint32_t)c * 1024u) : "memory");
            }
        }
    };
    // one 1 KiB piece of a block: the LDS-DMA path of a CU moves ~25 B/clk and the issuing wave stalls behind its own
    // queued pieces (measured ~320 cycles per piece when a wave issues its 9 pieces back to back), so inside the block
    // loop the pieces are issued one at a time, spread over the three phases
    const uint32_t lane16 = (uint32_t)lane * 16u;
    auto issue_piece_of = [&](const double* img, uint32_t b, int buf, int i) __attribute__((always_inline)) {
        const int c = i * 8 + w;
        if (c < G::CHUNKS) {
            // wave-uniform source (SGPR pair) + the lane's 16 bytes as a 32-bit offset: with a 64-bit address per lane the compiler
            // kept one VGPR pair per piece as a loop invariant, spilled them, and reloaded three inside the block loop (a scratch
            // reload waits behind the pieces in flight: vmcnt is in order)
            const double* src = img + (size_t)b * G::XBUF_PAD + (size_t)c * 128;
            const uint32_t dst = xs_lds + (uint32_t)buf * (uint32_t)(G::XBUF_PAD * sizeof(double)) + (uint32_t)c * 1024u;
            uint32_t m0_saved;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(m0_saved) : "v"(lane16), "s"(dst), "s"(src) : "memory");
        }
    };
    // DENSEM: the images of the matrix whose product FOLLOWS the running one (its block 0 is the wrap-around prefetch of the last block)
    const double* next_img = prm.Xp;
    auto img_of_next = [&](const double* img, bool last) __attribute__((always_inline)) -> const double* {
        if constexpr (DENSEM) return last ? next_img : img; else return prm.Xp;
    };
    // schedule of the NP pieces a wave issues per block: NP_E behind every other eta MFMA group, one at the head of the
    // row-term phase, the rest behind every other gradient tile
    constexpr int NP = (G::CHUNKS + 7) / 8, NP_E = (NTQ + 1) / 2;
    static_assert(NP_E + 1 + (NTQ + 1) / 2 >= NP, "piece schedule does not cover the block");
    auto wait_loads = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

    // ((S0 + S1) + S2) + S3 of per-wave partial dots (each already butterflied inside the wave)
    auto exchange = [&](double (&v)[2]) __attribute__((always_inline)) {
        __syncthreads();
        part[(0 * 4 + q) * 64 + lane] = v[0];
        part[(1 * 4 + q) * 64 + lane] = v[1];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k)
            v[k] = ((part[(k * 4 + 0) * 64 + lane] + part[(k * 4 + 1) * 64 + lane]) + part[(k * 4 + 2) * 64 + lane])
                   + part[(k * 4 + 3) * 64 + lane];
    };

    // value and gradient at x: lp = log K, gout = gradient on this wave's dims
    auto blk_sync = [&]() __attribute__((always_inline)) { if constexpr (!(ablate & 16u)) __syncthreads(); };
    // phase profile (ablate & 2048): shader cycles of wave (blockIdx 0) per phase, summed over all blocks and evaluations
    uint64_t prof[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&]() __attribute__((always_inline)) -> uint64_t { if constexpr ((ablate & 2048u) != 0) return clock64(); else return 0; };
    auto lap = [&](uint64_t& tp, int i) __attribute__((always_inline)) {
        if constexpr ((ablate & 2048u) != 0) { const uint64_t t = clock64(); prof[i] += t - tp; tp = t; }
    };
    uint32_t xbuf_next = 0;             // buffer that holds block 0 when an evaluation starts
    // (nuts: the parity is loop-carried through loops that end on workgroup votes read from LDS, which the compiler takes for divergent;
    //  the DMA destination must be an SGPR)
    auto xbuf_parity = [&]() __attribute__((always_inline)) -> uint32_t {
        if constexpr (ALGO == LOGIT_NUTS) return (uint32_t)__builtin_amdgcn_readfirstlane((int)xbuf_next); else return xbuf_next;
    };
    auto evaluate_logit = [&](const double (&x)[NSQ], double (&gout)[NSQ], double& lp) __attribute__((always_inline)) {
        double4_t gacc[NTQ];
        double llq = 0.0;
#pragma unroll
        for (int t = 0; t < NTQ; ++t) gacc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
        uint64_t te = stamp();
        const uint32_t xbuf0 = xbuf_parity();
        __syncthreads();                                 // nobody still reads the exchange area
        lap(te, 8);
        // Block 0 is already resident in buffer xbuf0: the last iteration of the previous evaluation fetched it (the
        // block stream wraps around; X does not change), so an evaluation starts without a DMA round trip.
#pragma unroll 1
        for (uint32_t b = 0; b < NB; ++b) {
            const double* xb = Xs + ((xbuf0 + b) & 1u) * G::XBUF_PAD;
            uint64_t tp = stamp();
            const bool prefetch = !(ablate & 8u);
            const uint32_t nblk = (b + 1 < NB) ? b + 1 : 0u;
            const int nbuf = (int)((xbuf0 + b + 1) & 1u);
            const double* const nimg = img_of_next(prm.Xp, b + 1 >= NB);
            auto issue_piece = [&](uint32_t bb, int buf, int i) __attribute__((always_inline)) { issue_piece_of(nimg, bb, buf, i); };
            if (prefetch && (ablate & 4096u)) issue_block(nblk, nbuf);
            // eta tile of this wave's dims as TWO fma chains (slices [0, NSQ/2) and [NSQ/2, NSQ)), summed at the end: a
            // single chain of NSQ dependent MFMAs runs the matrix pipe at half rate (measured 155 cycles per MFMA).
            double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0}, acc1 = double4_t{0.0, 0.0, 0.0, 0.0};
            {   // software pipeline: fragments of group k+1 are read from LDS while the 4 MFMAs of group k issue
                const double* xe = xb + eta_off;
                constexpr int H = NSQ / 2;
                double a_cur[4], a_nxt[4];
                auto frag = [&](int k, int i) -> const double* { return xe + 4 * ((i & 1) * H + 2 * k + (i >> 1)); };
#pragma unroll
                for (int i = 0; i < 4; ++i) a_cur[i] = (ablate & 32u) ? 1.0 : *frag(0, i);
#pragma unroll
                for (int k = 0; k < NSQ / 4; ++k) {
                    if (k + 1 < NSQ / 4) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) a_nxt[i] = (ablate & 32u) ? 1.0 : *frag(k + 1, i);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if ((k & 1) == 0 && k / 2 < NP_E && prefetch && !(ablate & (512u | 4096u))) issue_piece(nblk, nbuf, k / 2);
                    if constexpr (!(ablate & 2u)) {
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[0], x[2 * k], acc, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[1], x[H + 2 * k], acc1, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[2], x[2 * k + 1], acc, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[3], x[H + 2 * k + 1], acc1, 0, 0, 0);
                    } else {
                        acc[0] += a_cur[0] + a_cur[1] + a_cur[2] + a_cur[3];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) a_cur[i] = a_nxt[i];
                }
                acc[0] = acc[0] + acc1[0]; acc[1] = acc[1] + acc1[1]; acc[2] = acc[2] + acc1[2]; acc[3] = acc[3] + acc1[3];
            }
            lap(tp, 0);
            part[(q * 4 + 0) * 64 + lane] = acc[0]; part[(q * 4 + 1) * 64 + lane] = acc[1];
            part[(q * 4 + 2) * 64 + lane] = acc[2]; part[(q * 4 + 3) * 64 + lane] = acc[3];
            blk_sync();
            lap(tp, 1);
            if (prefetch && (ablate & 512u)) issue_block(nblk, nbuf);
            if (prefetch && !(ablate & (512u | 4096u))) issue_piece(nblk, nbuf, NP_E);
            const double* xg = xb + grad_off;
            double g_cur[4], g_nxt[4];                   // first X^T r fragments: in flight across the row-term phase
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) g_cur[sp] = (ablate & 32u) ? 1.0 : xg[2 * sp * RSP];
            {   // row group q: rows 16b + 4q + j4
                const uint32_t row = 16 * b + 4 * q + j4;
                const double yv = xb[G::XBUF + 4 * q + j4];          // labels ride in the block image
                const bool valid = row < prm.n_rows;
                const double eta = ((part[(0 * 4 + q) * 64 + lane] + part[(1 * 4 + q) * 64 + lane]) + part[(2 * 4 + q) * 64 + lane])
                                   + part[(3 * 4 + q) * 64 + lane];
                // softplus / sigmoid share e = exp(-|eta|) (the oracle evaluates it once per function; same bits)
#ifdef MI_LOGIT_ESTRIN      // EXPERIMENT (timing only: not the oracle's arithmetic): exp / log1p of the row term with Estrin-form polynomials
                double e, l1p;
                {
                    const double x = eta > 0.0 ? -eta : eta;
                    const double kf = __builtin_rint(x * INV_LN2);
                    const int k = (int)kf;
                    double r = dfma(-kf, LN2_HI, x);
                    r = dfma(-kf, LN2_LO, r);
                    const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
                    const double a0 = dfma(r, 1.0, 1.0), a1 = dfma(r, MI_KC(1.0 / 6.0), 0.5), a2 = dfma(r, MI_KC(1.0 / 120.0), MI_KC(1.0 / 24.0)), a3 = dfma(r, MI_KC(1.0 / 5040.0), MI_KC(1.0 / 720.0));
                    const double a4 = dfma(r, MI_KC(1.0 / 362880.0), MI_KC(1.0 / 40320.0)), a5 = dfma(r, MI_KC(1.0 / 39916800.0), MI_KC(1.0 / 3628800.0)), a6 = dfma(r, MI_KC(1.0 / 6227020800.0), MI_KC(1.0 / 479001600.0));
                    const double b0 = dfma(r2, a1, a0), b1 = dfma(r2, a3, a2), b2 = dfma(r2, a5, a4), b3 = dfma(r2, MI_KC(1.0 / 87178291200.0), a6);
                    const double d0 = dfma(r4, b1, b0), d1 = dfma(r4, b3, b2);
                    const double pe = dfma(r8, d1, d0);
                    const int k1 = k / 2, k2 = k - k1;
                    e = (x < -745.2) ? 0.0 : (pe * pow2i(k1)) * pow2i(k2);
                    double m = 1.0 + e;
                    const bool big = m > 0x1.6a09e667f3bcdp+0;
                    m = big ? m * 0.5 : m;
                    const double ef = big ? 1.0 : 0.0;
                    const double f = m - 1.0;
                    const double sq = f / (2.0 + f);
                    const double z = sq * sq, z2 = z * z, z4 = z2 * z2, z8 = z4 * z4;
                    const double c0 = dfma(z, MI_KC(1.0 / 3.0), 1.0), c1 = dfma(z, MI_KC(1.0 / 7.0), MI_KC(1.0 / 5.0)), c2 = dfma(z, MI_KC(1.0 / 11.0), MI_KC(1.0 / 9.0)), c3 = dfma(z, MI_KC(1.0 / 15.0), MI_KC(1.0 / 13.0));
                    const double c4 = dfma(z, MI_KC(1.0 / 19.0), MI_KC(1.0 / 17.0)), c5 = dfma(z, MI_KC(1.0 / 23.0), MI_KC(1.0 / 21.0));
                    const double g0 = dfma(z2, c1, c0), g1 = dfma(z2, c3, c2), g2 = dfma(z2, c5, c4);
                    const double h0 = dfma(z4, g1, g0);
                    const double pl = dfma(z8, g2, h0);
                    l1p = dfma(ef, LN2_HI, dfma(ef, LN2_LO, (2.0 * sq) * pl));
                }
#else
                const double e = (ablate & 1u) ? eta * 0.25 : det_exp(eta > 0.0 ? -eta : eta);
                const double l1p = (ablate & 1u) ? e * 0.5 : det_log(1.0 + e);
#endif
                const double sp = (eta > 0.0) ? (eta + l1p) : l1p;
                const double sg = (eta >= 0.0) ? (1.0 / (1.0 + e)) : (e / (1.0 + e));
                rt[(q * 2 + 0) * 64 + lane] = valid ? (yv - sg) : 0.0;
                rt[(q * 2 + 1) * 64 + lane] = valid ? (yv * eta - sp) : 0.0;
            }
            lap(tp, 2);
            blk_sync();
            lap(tp, 3);
            double res[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { res[r] = rt[(r * 2 + 0) * 64 + lane]; llq = llq + rt[(r * 2 + 1) * 64 + lane]; }
            {
#pragma unroll
                for (int t = 0; t < NTQ; ++t) {
                    if (t + 1 < NTQ) {
#pragma unroll
                        for (int sp = 0; sp < 4; ++sp) g_nxt[sp] = (ablate & 32u) ? 1.0 : xg[2 * sp * RSP + 16 * (t + 1)];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if ((t & 1) == 0 && NP_E + 1 + t / 2 < NP && prefetch && !(ablate & (512u | 4096u))) issue_piece(nblk, nbuf, NP_E + 1 + t / 2);
#pragma unroll
                    for (int sp = 0; sp < 4; ++sp)
                    { if constexpr (!(ablate & 4u)) gacc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(g_cur[sp], res[sp], gacc[t], 0, 0, 0); else gacc[t][0] += g_cur[sp]; }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int sp = 0; sp < 4; ++sp) g_cur[sp] = g_nxt[sp];
                }
            }
            lap(tp, 4);
            if constexpr (!(ablate & 128u)) wait_loads();    // block b+1 landed (this wave's chunks) ...
            lap(tp, 5);
            blk_sync();                             // ... everybody's, and buffer b&1 is free for block b+2
            lap(tp, 6);
        }
        xbuf_next = (xbuf0 + NB) & 1u;
        te = stamp();
        llq = llq + __shfl_xor(llq, 32);
        llq = llq + __shfl_xor(llq, 16);
        double v[2];
        {
            double nrm = 0.0;
#pragma unroll
            for (int s = 0; s < NSQ; ++s) nrm = dfma(x[s], x[s], nrm);
            nrm = nrm + __shfl_xor(nrm, 32);
            nrm = nrm + __shfl_xor(nrm, 16);
            v[0] = nrm; v[1] = 0.0;
        }
        exchange(v);
#pragma unroll
        for (int t = 0; t < NTQ; ++t) {
            gout[4 * t + 0] = gacc[t][0] - x[4 * t + 0];
            gout[4 * t + 1] = gacc[t][1] - x[4 * t + 1];
            gout[4 * t + 2] = gacc[t][2] - x[4 * t + 2];
            gout[4 * t + 3] = gacc[t][3] - x[4 * t + 3];
        }
        lp = llq - 0.5 * v[0];
        lap(te, 9);
    };


    // Dense Gaussian (LOGIT_TARGET_DENSE): w = P x by the X^T r machinery alone.  The block images hold P transposed, so block b
    // carries columns 16 b .. 16 b + 15 of P, its "row terms" are x on those dimensions -- slice 4 (b mod NTQ) + sp of wave b / NTQ --
    // and every wave accumulates w on its own dimensions in registers across all blocks: element i of w is ONE fma chain over the
    // columns in ascending order (what gauss_dense_grad of hmc_dense.hpp and the oracle's orc_gemv do).  x travels between the four
    // waves of a tile through xexch (64 KiB per tile at d = 512: LDS is full of P), written once per evaluation; one barrier per
    // block (the double buffer), none for an eta phase or an exchange of partial sums.  lp = -1/2 x.w, gout = -w.
    // gacc = A x for the matrix whose TRANSPOSED block images are img[0 .. nb) (P of the dense target; DENSEM: INV / CHOL_LOWER of precond_mat)
    auto stream_product = [&](const double* img, uint32_t nb, const double (&x)[NSQ], double4_t (&gacc)[NTQ]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < NTQ; ++t) gacc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
        const uint32_t xbuf0 = xbuf_parity();
        // wave-uniform base (SGPR pair, known to be global memory) + the lane's byte offset, opaque so that the NSQ addresses are not
        // hoisted: a laundered POINTER would lose its address space and turn these into flat loads, behind which every LDS wait of
        // the block loop becomes lgkmcnt(0) + vmcnt(0) -- i.e. a wait for the block's own DMA pieces
        double* const xq_base = prm.xexch + ((size_t)blockIdx.x * 2 + g) * ((size_t)4 * NSQ * 64);
        uint32_t xq_off = (uint32_t)lane * 8u;
        asm volatile("" : "+v"(xq_off));
        auto xq_at = [&](uint32_t slot) __attribute__((always_inline)) -> double* {
            return reinterpret_cast<double*>(reinterpret_cast<char*>(xq_base) + ((size_t)slot * 512u + xq_off));
        };
#pragma unroll
        for (int s = 0; s < NSQ; ++s) *xq_at((uint32_t)(q * NSQ + s)) = x[s];
        __syncthreads();                                 // x is visible to the tile's four waves; nobody still reads the exchange area
        double r_cur[4], r_nxt[4];
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) r_cur[sp] = *xq_at((uint32_t)sp);
        constexpr bool prefetch = !(ablate & 8u);
#pragma unroll 1
        for (uint32_t b = 0; b < nb; ++b) {
            const double* xb = Xs + ((xbuf0 + b) & 1u) * G::XBUF_PAD;
            const uint32_t nblk = (b + 1 < nb) ? b + 1 : 0u;
            const int nbuf = (int)((xbuf0 + b + 1) & 1u);
            const uint32_t bn = (b + 1 < nb) ? b + 1 : b;
            const double* const nimg = img_of_next(img, b + 1 >= nb);
            auto issue_piece = [&](uint32_t bb, int buf, int i) __attribute__((always_inline)) { issue_piece_of(nimg, bb, buf, i); };
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) r_nxt[sp] = *xq_at(4u * bn + (uint32_t)sp);
            const double* xg = xb + grad_off;
#ifndef MI_DENSE_TT
#define MI_DENSE_TT 1      // measured: 2 and 4 tiles interleaved are 3 % and 9 % slower (registers), two waves per SIMD hide the chain already
#endif
            // TT dimension tiles at a time: the four MFMAs of one tile form a dependent chain (one fma chain per element of w), and
            // back-to-back dependent MFMAs leave the matrix pipe half idle; interleaved with the chain of the next tile they do not.
            constexpr int TT = (NTQ % MI_DENSE_TT == 0) ? MI_DENSE_TT : 1;
            double g_cur[TT][4], g_nxt[TT][4];
#pragma unroll
            for (int u = 0; u < TT; ++u)
#pragma unroll
                for (int sp = 0; sp < 4; ++sp) g_cur[u][sp] = (ablate & 32u) ? 1.0 : xg[2 * sp * RSP + 16 * u];
#pragma unroll
            for (int t = 0; t < NTQ; t += TT) {
                if (t + TT < NTQ) {
#pragma unroll
                    for (int u = 0; u < TT; ++u)
#pragma unroll
                        for (int sp = 0; sp < 4; ++sp) g_nxt[u][sp] = (ablate & 32u) ? 1.0 : xg[2 * sp * RSP + 16 * (t + TT + u)];
                }
                __builtin_amdgcn_sched_barrier(0);
#ifndef MI_DENSE_DMA_TILES
#define MI_DENSE_DMA_TILES 0       // tiles of a block over which the wave spreads its DMA pieces of the next block (0: all NTQ)
#endif
                if constexpr (prefetch) {
                    constexpr int DT = (MI_DENSE_DMA_TILES > 0 && MI_DENSE_DMA_TILES < NTQ) ? MI_DENSE_DMA_TILES : NTQ;
                    if (t < DT) {
#pragma unroll
                        for (int i = (t * NP) / DT; i < ((t + TT) * NP) / DT && i < NP; ++i) issue_piece(nblk, nbuf, i);
                    }
                }
#pragma unroll
                for (int sp = 0; sp < 4; ++sp)
#pragma unroll
                    for (int u = 0; u < TT; ++u)
                    { if constexpr (!(ablate & 4u)) gacc[t + u] = __builtin_amdgcn_mfma_f64_16x16x4f64(g_cur[u][sp], r_cur[sp], gacc[t + u], 0, 0, 0); else gacc[t + u][0] += g_cur[u][sp] * r_cur[sp]; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < TT; ++u)
#pragma unroll
                    for (int sp = 0; sp < 4; ++sp) g_cur[u][sp] = g_nxt[u][sp];
            }
            if constexpr (!(ablate & 128u)) wait_loads();   // block b+1 landed (this wave's pieces; r_nxt too) ...
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) r_cur[sp] = r_nxt[sp];
            blk_sync();                             // ... everybody's, and buffer b&1 is free for block b+2
        }
        xbuf_next = (xbuf0 + nb) & 1u;
    };
    auto evaluate_dense = [&](const double (&x)[NSQ], double (&gout)[NSQ], double& lp) __attribute__((always_inline)) {
        double4_t gacc[NTQ];
        stream_product(prm.Xp, NB, x, gacc);
        double v[2];
        {
            double a = 0.0;
#pragma unroll
            for (int s = 0; s < NSQ; ++s) a = dfma(x[s], gacc[s >> 2][s & 3], a);
            a = a + __shfl_xor(a, 32);
            a = a + __shfl_xor(a, 16);
            v[0] = a; v[1] = 0.0;
        }
        exchange(v);
#pragma unroll
        for (int s = 0; s < NSQ; ++s) gout[s] = -gacc[s >> 2][s & 3];
        lp = -0.5 * v[0];
    };
    auto evaluate = [&](const double (&x)[NSQ], double (&gout)[NSQ], double& lp) __attribute__((always_inline)) {
        if constexpr (TARGET == LOGIT_TARGET_DENSE) evaluate_dense(x, gout, lp); else evaluate_logit(x, gout, lp);
    };

    double bp[NSQ], gp[NSQ];            // position / proposal and its gradient (this wave's dims)
#pragma unroll
    for (int s = 0; s < NSQ; ++s) {
        const uint32_t dim = dim_of(s);
        const double v = prm.theta[(size_t)(dim < d ? dim : 0u) * C + (live ? cl : C - 1)];
        bp[s] = (dim < d) ? v : 0.0;
    }
    [[maybe_unused]] LdsBox<NTQ> box;
    [[maybe_unused]] double* const lj_rel = part + (3 * 4) * 64 + 32;       // BOUNDS: the log-Jacobian relay of the tile (16 doubles)
    if constexpr (BOUNDS && ALGO == LOGIT_HMC) {
        box.init(prm.btype, prm.lb, prm.ub, d, q, lane);
#pragma unroll
        for (int s = 0; s < NSQ; ++s) bp[s] = (dim_of(s) < d) ? box.enter(bp[s], s) : 0.0;       // hmc.cpp:134-136
    }
    // BOUNDS: the evaluation at x = inv_transform(theta) (hmc.cpp:108); theta crosses it in the wave's third workspace vector
    auto evaluate_at = [&](double (&t)[NSQ], double (&gout)[NSQ], double& lp) __attribute__((always_inline)) {
        if constexpr (BOUNDS) {
#pragma unroll
            for (int s = 0; s < NSQ; ++s) *st(2, s) = t[s];
            box.x_inplace(t);
            evaluate(t, gout, lp);
#pragma unroll
            for (int s = 0; s < NSQ; ++s) t[s] = *st(2, s);
        } else evaluate(t, gout, lp);
    };
    issue_block(0, 0);
    wait_loads();
    if constexpr (ALGO == LOGIT_NUTS) {
        // NUTS (nuts.cpp:30-332): the per-chain tree state machine on this kernel's evaluation and exchange (nuts_lds.hpp); chains are
        // handed to the workgroup's 32 slots dynamically, their first evaluation is a state of that machine
        // DENSEM (round 6): the streamed products of INV(M) / CHOL_LOWER(M) as the tick's collectives (nuts_lds.hpp: dm)
        struct DenseMOps {
            decltype(stream_product)& sp; decltype(issue_piece_of)& ip; decltype(wait_loads)& wl; decltype(xbuf_parity)& xp;
            const double*& nimg; uint32_t nbm;
            __device__ __forceinline__ void product(const double* img, const double* next, const double (&x)[NSQ], double4_t (&acc)[NTQ]) const { nimg = next; sp(img, nbm, x, acc); }
            __device__ __forceinline__ void next(const double* img) const { nimg = img; }
            // block 0 of img into the buffer the next product starts from: the last product prefetched another matrix there (its DMA has landed and
            // nobody reads that buffer: the product ended on a wait + barrier)
            __device__ __forceinline__ void reload0(const double* img) const
            {
                const int buf = (int)xp();
#pragma unroll
                for (int i = 0; i < NP; ++i) ip(img, 0u, buf, i);
                wl();
                __syncthreads();
            }
        };
        DenseMOps dm{stream_product, issue_piece_of, wait_loads, xbuf_parity, next_img, (d + 15u) / 16u};
        nuts_lds_body<NTQ, DIAGM, BOUNDS, DENSEM>(prm, evaluate, part_all, dm);
        return;
    }
    double first_lp;
    if constexpr (DENSEM) next_img = (ALGO == LOGIT_MALA) ? prm.Mp : prm.Lp;     // what follows the first evaluation: mala M grad, hmc the first draw's L z
    evaluate_at(bp, gp, first_lp);      // box_log_kernel(first_draw): mala.cpp:138 / hmc.cpp:140
    if constexpr (BOUNDS) first_lp = first_lp + box.log_jacobian(bp, lj_rel, [&]() { __syncthreads(); });      // hmc.cpp:84-95
#pragma unroll
    for (int s = 0; s < NSQ; ++s) { *st(0, s) = bp[s]; *st(1, s) = gp[s]; }
    uint64_t n_acc = 0;
    const uint32_t n_total = prm.n_burnin + prm.n_keep;
    // non-finite regime: detected through the proposal densities (mala) / energies (hmc), the same bits in
    // the four waves of a chain tile; the chain is flagged and replayed by literal.hpp instead of finished here.  The flag is bit 63
    // of the accept counter: this kernel has no register to spare for a second loop-carried value (as a bool of its own it cost the
    // d = 512 mala instantiation nine more spilled VGPRs).
    constexpr uint64_t NF_BIT = 1ull << 63;

    constexpr int SB0 = (MI_LOGIT_BATCH == 0) ? 2 : ((NSQ < MI_LOGIT_BATCH) ? NSQ : MI_LOGIT_BATCH);
    constexpr int SB = (NSQ % SB0 == 0) ? SB0 : 4;      // slices per batch of workspace loads (NTQ = 3: NSQ = 12)
#ifndef MI_LOGIT_NT_DRAWS
#define MI_LOGIT_NT_DRAWS 0      // (timing experiment) 1: kept rows -- never read again by the kernel -- stored non-temporal, so that they do not push the streamed matrix out of the L2:
                                 // measured on configs[2], nothing (profiles/r6_mala_nt_ab.log: 2 146 against 2 146 ms)
#endif
    auto put_row = [&](double* p_, double v_) __attribute__((always_inline)) {
#if MI_LOGIT_NT_DRAWS
        __builtin_nontemporal_store(v_, p_);
#else
        *p_ = v_;
#endif
    };
    auto keep_draw = [&](uint32_t draw, bool accept) __attribute__((always_inline)) {
        if (draw >= prm.n_burnin) {
            n_acc += accept ? 1u : 0u;
            if (prm.draws != nullptr) {
                double* out = prm.draws + (size_t)(draw - prm.n_burnin) * d * C + cl;
                if (__any(!accept)) {                    // some chain of the wave repeats its previous draw: fetch it
#pragma unroll
                    for (int s0 = 0; s0 < NSQ; s0 += SB) {
                        double old[SB];
#pragma unroll
                        for (int i = 0; i < SB; ++i) old[i] = *st(0, s0 + i);
#pragma unroll
                        for (int i = 0; i < SB; ++i) {
                            const uint32_t dim = dim_of(s0 + i);
                            double v = accept ? bp[s0 + i] : old[i];
                            if constexpr (BOUNDS) v = box.leave(v, s0 + i);      // rows are reported in the constrained space (hmc.cpp:211-218)
                            if (live && dim < d) put_row(out + (size_t)dim * C, v);
                        }
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < NSQ; ++s) {
                        const uint32_t dim = dim_of(s);
                        double v = bp[s];
                        if constexpr (BOUNDS) v = box.leave(v, s);
                        if (live && dim < d) put_row(out + (size_t)dim * C, v);
                    }
                }
            }
        }
    };

    if constexpr (ALGO == LOGIT_MALA && DENSEM) {
        // Workspace vectors of a wave: 0 the accepted beta, 2 M grad at it (what the mean needs of the gradient), 1 M grad at the proposal
        // while the density products run.  Never more than three vectors in registers: the proposal, a product's vector and its result.
        const uint32_t NBM = (d + 15u) / 16u;
        const double s2 = prm.s2;
        double prev_LP = first_lp, prop_LP;
        double4_t acc[NTQ];
        next_img = prm.Lp;                               // the first draw's L z follows
        stream_product(prm.Mp, NBM, gp, acc);            // precond_matrix * grad at the initial values (mala.cpp:123)
#pragma unroll
        for (int s = 0; s < NSQ; ++s) *st(2, s) = acc[s >> 2][s & 3];
        auto quad = [&]() __attribute__((always_inline)) -> double {      // (X - mu) . (INV(Sigma) (X - mu)), this wave's part (dmvnorm.hpp:39)
            double a = 0.0;
#pragma unroll
            for (int s = 0; s < NSQ; ++s) a = dfma(gp[s], acc[s >> 2][s & 3], a);
            a = a + __shfl_xor(a, 32);
            a = a + __shfl_xor(a, 16);
            return a;
        };
#pragma unroll 1
        for (uint32_t draw = 0; draw < n_total; ++draw) {
#pragma unroll
            for (int m = 0; m < NSQ / 2; ++m) {          // rand_vec (:150)
                double z0, z1;
                const uint32_t slot = (uint32_t)(q * DQ / 2 + 4 * m + j4);
                rng_normal_pair(prm.seed, chain, draw + prm.draw0, slot, STREAM_NORMAL, z0, z1);
                bp[2 * m] = (dim_of(2 * m) < d) ? z0 : 0.0;
                bp[2 * m + 1] = (dim_of(2 * m + 1) < d) ? z1 : 0.0;
                __builtin_amdgcn_sched_barrier(0);
            }
            next_img = prm.Xp;
            stream_product(prm.Lp, NBM, bp, acc);        // sqrt_precond_matrix * rand_vec (:159)
#pragma unroll
            for (int s = 0; s < NSQ; ++s) bp[s] = (*st(0, s) + (s2 * *st(2, s)) / 2.0) + eps * acc[s >> 2][s & 3];     // :123, :159
            next_img = prm.Mp;
            evaluate(bp, gp, prop_LP);                   // :162
            next_img = prm.Sip;
            stream_product(prm.Mp, NBM, gp, acc);        // precond_matrix * grad at the proposal
            // mala_prop_adjustment (mala.ipp:59-64): dmvnorm(prev | mean(prop), Sigma) - dmvnorm(prop | mean(prev), Sigma); the gradient's registers
            // carry X - mu (dmvnorm.hpp:37)
#pragma unroll
            for (int s = 0; s < NSQ; ++s) {
                const double mgp = acc[s >> 2][s & 3];
                *st(1, s) = mgp;
                gp[s] = *st(0, s) - (bp[s] + (s2 * mgp) / 2.0);
            }
            stream_product(prm.Sip, NBM, gp, acc);       // (the second density's product follows: next_img is Sip already)
            double qv[2];
            qv[0] = quad();
#pragma unroll
            for (int s = 0; s < NSQ; ++s) gp[s] = bp[s] - (*st(0, s) + (s2 * *st(2, s)) / 2.0);
            next_img = prm.Lp;                           // the next draw starts with L z
            stream_product(prm.Sip, NBM, gp, acc);
            qv[1] = quad();
            exchange(qv);
            double pl = prop_LP;
            if (!is_finite(pl)) pl = -INF;               // mala.cpp:164-166
            const double da = prm.cons_term - 0.5 * (prm.log_det + qv[0]);       // dmvnorm.hpp:41
            const double db = prm.cons_term - 0.5 * (prm.log_det + qv[1]);
            if (!is_finite(da) || !is_finite(db)) n_acc |= NF_BIT;
            const double x = pl - prev_LP + (da - db);
            const double comp_val = (x < 0.01) ? x : 0.01;                       // mala.cpp:170
            const double z = rng_uniform(prm.seed, chain, draw + prm.draw0, 0u);             // :171
            const bool accept = z < det_exp(comp_val);                           // :173
            if (accept) {
                prev_LP = pl;
                if (live) {
#pragma unroll
                    for (int s = 0; s < NSQ; ++s) { *st(0, s) = bp[s]; *st(2, s) = *st(1, s); }
                }
            }
            keep_draw(draw, accept);
        }
    } else if constexpr (ALGO == LOGIT_MALA) {
        const double s2 = prm.s2, rs = prm.rs;
        double prev_LP = first_lp, prop_LP;
#pragma unroll 1
        for (uint32_t draw = 0; draw < n_total; ++draw) {
            uint64_t td = stamp();
            // proposal = mala_mean_fn(prev) + eps * z   (mala.cpp:150,159).  The accepted (beta, grad) come from the wave's
            // workspace in batches of SB slices, one batch ahead of the normals that consume them.
            {
                double be_c[SB], gr_c[SB], be_n[SB], gr_n[SB];
#pragma unroll
                for (int i = 0; i < SB; ++i) { be_c[i] = *st(0, i); gr_c[i] = *st(1, i); }
#pragma unroll
                for (int s0 = 0; s0 < NSQ; s0 += SB) {
                    if (s0 + SB < NSQ) {
#pragma unroll
                        for (int i = 0; i < SB; ++i) { be_n[i] = *st(0, s0 + SB + i); gr_n[i] = *st(1, s0 + SB + i); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#ifndef MI_LOGIT_RNG_GROUP
#define MI_LOGIT_RNG_GROUP 2
#endif
                    constexpr int RG = (MI_LOGIT_RNG_GROUP < SB / 2) ? MI_LOGIT_RNG_GROUP : ((SB / 2 >= 4) ? 4 : 2);   // Philox slots per out-of-line call (det_math.hpp): independent chains the scheduler interleaves
                    static_assert((RG == 2 || RG == 4) && (SB / 2) % RG == 0, "pairs are drawn RG at a time");
#pragma unroll
                    for (int m = 0; m < SB / 2; m += RG) {
                        const int mm = s0 / 2 + m;
                        double zz[2 * RG];
                        const uint32_t slot = (uint32_t)(q * DQ / 2 + 4 * mm + j4);
                        if constexpr ((ablate & 64u) != 0) {
#pragma unroll
                            for (int h = 0; h < 2 * RG; ++h) zz[h] = (h & 1) ? -0.5 : 0.5;
                        } else if constexpr (RG == 2) {
                            const rng_double4 t4 = rng_normal_two_pairs(prm.seed, chain, draw + prm.draw0, slot, slot + 4u, STREAM_NORMAL);
                            zz[0] = t4[0]; zz[1] = t4[1]; zz[2] = t4[2]; zz[3] = t4[3];
                        } else {
                            const rng_double8 t8 = rng_normal_four_pairs(prm.seed, chain, draw + prm.draw0, slot, 4u, STREAM_NORMAL);
#pragma unroll
                            for (int h = 0; h < 8; ++h) zz[h] = t8[h];
                        }
#pragma unroll
                        for (int h = 0; h < RG; ++h) {
                            const double za = (dim_of(2 * (mm + h)) < d) ? zz[2 * h] : 0.0;
                            const double zb = (dim_of(2 * (mm + h) + 1) < d) ? zz[2 * h + 1] : 0.0;
                            if constexpr (DIAGM) {       // mean = x + eps^2 (M grad) / 2 (:123), proposal = mean + eps (sqrt(M) z) (:159), M diagonal
                                const int sa = 2 * (mm + h), sb = sa + 1;
                                bp[sa] = (be_c[2 * (m + h)] + (s2 * (mass_at(prm.m, sa) * gr_c[2 * (m + h)])) / 2.0) + eps * (mass_at(prm.m_sqrt, sa) * za);
                                bp[sb] = (be_c[2 * (m + h) + 1] + (s2 * (mass_at(prm.m, sb) * gr_c[2 * (m + h) + 1])) / 2.0) + eps * (mass_at(prm.m_sqrt, sb) * zb);
                            } else {
                            bp[2 * (mm + h)] = (be_c[2 * (m + h)] + (s2 * gr_c[2 * (m + h)]) / 2.0) + eps * za;           // :123, :159
                            bp[2 * (mm + h) + 1] = (be_c[2 * (m + h) + 1] + (s2 * gr_c[2 * (m + h) + 1]) / 2.0) + eps * zb;
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int i = 0; i < SB; ++i) { be_c[i] = be_n[i]; gr_c[i] = gr_n[i]; }
                }
            }
            lap(td, 7);
            evaluate(bp, gp, prop_LP);                   // :162
            td = stamp();
            // mala_prop_adjustment (mala.ipp:59-64)
            double qv[2];
            {
                double qa = 0.0, qb = 0.0;
                double be_c[SB], gr_c[SB], be_n[SB], gr_n[SB];
#pragma unroll
                for (int i = 0; i < SB; ++i) { be_c[i] = *st(0, i); gr_c[i] = *st(1, i); }
#pragma unroll
                for (int s0 = 0; s0 < NSQ; s0 += SB) {
                    if (s0 + SB < NSQ) {
#pragma unroll
                        for (int i = 0; i < SB; ++i) { be_n[i] = *st(0, s0 + SB + i); gr_n[i] = *st(1, s0 + SB + i); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < SB; ++i) {
                        const int s = s0 + i;
                        const double be = be_c[i], gr = gr_c[i];
                        if constexpr (DIAGM) {           // Sigma = eps^2 M: INV(Sigma)_ii from the host (s_inv), the means with M grad
                            const double mi_ = mass_at(prm.m, s), si_ = mass_at(prm.s_inv, s);
                            const double mean_prop = bp[s] + (s2 * (mi_ * gp[s])) / 2.0;
                            const double xa = be - mean_prop;
                            qa = dfma(xa, si_ * xa, qa);
                            const double mean_prev = be + (s2 * (mi_ * gr)) / 2.0;
                            const double xb = bp[s] - mean_prev;
                            qb = dfma(xb, si_ * xb, qb);
                        } else {
                        const double mean_prop = bp[s] + (s2 * gp[s]) / 2.0;
                        const double xa = be - mean_prop;    // dmvnorm.hpp:37
                        qa = dfma(xa, rs * xa, qa);
                        const double mean_prev = be + (s2 * gr) / 2.0;
                        const double xb = bp[s] - mean_prev;
                        qb = dfma(xb, rs * xb, qb);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < SB; ++i) { be_c[i] = be_n[i]; gr_c[i] = gr_n[i]; }
                }
                qa = qa + __shfl_xor(qa, 32); qa = qa + __shfl_xor(qa, 16);
                qb = qb + __shfl_xor(qb, 32); qb = qb + __shfl_xor(qb, 16);
                qv[0] = qa; qv[1] = qb;
            }
            exchange(qv);
            double pl = prop_LP;
            if (!is_finite(pl)) pl = -INF;               // mala.cpp:164-166
            const double da = prm.cons_term - 0.5 * (prm.log_det + qv[0]);       // dmvnorm.hpp:41
            const double db = prm.cons_term - 0.5 * (prm.log_det + qv[1]);
            if (!is_finite(da) || !is_finite(db)) n_acc |= NF_BIT;
            const double x = pl - prev_LP + (da - db);
            const double comp_val = (x < 0.01) ? x : 0.01;                       // mala.cpp:170
            const double z = rng_uniform(prm.seed, chain, draw + prm.draw0, 0u);             // :171
            const bool accept = z < det_exp(comp_val);                           // :173
            if (accept) {
                prev_LP = pl;
                if (live) {
#pragma unroll
                    for (int s = 0; s < NSQ; ++s) { *st(0, s) = bp[s]; *st(1, s) = gp[s]; }
                }
            }
            keep_draw(draw, accept);
            lap(td, 10);
        }
    } else if constexpr (ALGO == LOGIT_RWMH) {
        // RWMH (rwmh.cpp:123-151), identity cov_mat: proposal = prev + par_scale * z, value-only use of the fused evaluation
        double prev_LP = first_lp, prop_LP;
#pragma unroll 1
        for (uint32_t draw = 0; draw < n_total; ++draw) {
#pragma unroll
            for (int m = 0; m < NSQ / 2; ++m) {
                double z0, z1;
                const uint32_t slot = (uint32_t)(q * DQ / 2 + 4 * m + j4);
                rng_normal_pair(prm.seed, chain, draw + prm.draw0, slot, STREAM_NORMAL, z0, z1);
                const double za = (dim_of(2 * m) < d) ? z0 : 0.0;
                const double zb = (dim_of(2 * m + 1) < d) ? z1 : 0.0;
                bp[2 * m] = *st(0, 2 * m) + eps * za;                            // :126
                bp[2 * m + 1] = *st(0, 2 * m + 1) + eps * zb;
                __builtin_amdgcn_sched_barrier(0);
            }
            evaluate(bp, gp, prop_LP);                                           // :128
            double pl = prop_LP;
            if (!is_finite(pl)) pl = -INF;                                       // :130-132
            const double x = pl - prev_LP;
            const double comp_val = (x < 0.0) ? x : 0.0;                         // std::min(0.0, x): NaN -> 0 (:136)
            const double z = rng_uniform(prm.seed, chain, draw + prm.draw0, 0u); // :137
            const bool accept = z < det_exp(comp_val);                           // :139
            if (accept) {
                prev_LP = pl;
                if (live) {
#pragma unroll
                    for (int s = 0; s < NSQ; ++s) { *st(0, s) = bp[s]; *st(1, s) = gp[s]; }
                }
            }
            keep_draw(draw, accept);
        }
    } else if constexpr (ALGO == LOGIT_NUTS) {
        // (handled above)
    } else {
        // HMC (hmc.cpp:155-205): one evaluation per leapfrog step -- the second half-kick of step k and the first of step
        // k+1 are at the same position -- and the value of the last one is prop_U.
        const uint32_t n_leap = prm.n_leap;
        double pm[NSQ];
        double prev_U = -first_lp;                       // hmc.cpp:140
        const uint32_t NBM = (d + 15u) / 16u;            // DENSEM: blocks of the d x d matrices
        auto kinetic = [&]() __attribute__((always_inline)) -> double {   // p.(Minv p) / 2, block order (:160,184)
            double v[2];
            double a = 0.0;
            if constexpr (DENSEM) {
                double4_t mp[NTQ];
                stream_product(prm.Mip, NBM, pm, mp);      // inv_precond_matrix * mntm
#pragma unroll
                for (int s = 0; s < NSQ; ++s) a = dfma(pm[s], mp[s >> 2][s & 3], a);
            } else {
#pragma unroll
            for (int s = 0; s < NSQ; ++s) {
                if constexpr (DIAGM) a = dfma(pm[s], mass_at(prm.m_inv, s) * pm[s], a);
                else a = dfma(pm[s], pm[s], a);
            }
            }
            a = a + __shfl_xor(a, 32);
            a = a + __shfl_xor(a, 16);
            v[0] = a; v[1] = 0.0;
            exchange(v);
            return v[0] / 2.0;
        };
#pragma unroll 1
        for (uint32_t draw = 0; draw < n_total; ++draw) {
#pragma unroll
            for (int m = 0; m < NSQ / 2; ++m) {          // momentum ~ N(0, I), :156-158
                double z0, z1;
                const uint32_t slot = (uint32_t)(q * DQ / 2 + 4 * m + j4);
                if constexpr (!(ablate & 64u)) rng_normal_pair(prm.seed, chain, draw + prm.draw0, slot, STREAM_NORMAL, z0, z1); else { z0 = 0.5; z1 = -0.5; }
                pm[2 * m] = (dim_of(2 * m) < d) ? z0 : 0.0;
                pm[2 * m + 1] = (dim_of(2 * m + 1) < d) ? z1 : 0.0;
                if constexpr (DIAGM) {                       // p = L z with a diagonal L (:158)
                    pm[2 * m] = mass_at(prm.m_sqrt, 2 * m) * pm[2 * m];
                    pm[2 * m + 1] = mass_at(prm.m_sqrt, 2 * m + 1) * pm[2 * m + 1];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (DENSEM) {                       // p = CHOL_LOWER(M) z (:158); then Minv p for prev_K
                double4_t lz[NTQ];
                next_img = prm.Mip;
                stream_product(prm.Lp, NBM, pm, lz);
#pragma unroll
                for (int s = 0; s < NSQ; ++s) pm[s] = lz[s >> 2][s & 3];
            }
            double prev_K;
            if constexpr (DENSEM) {                       // (the product first: three vectors live across it, not five)
                next_img = prm.Mip;                       // the first step's product (n_leap = 0: prop_K's) follows
                prev_K = kinetic();
#pragma unroll
                for (int s = 0; s < NSQ; ++s) { bp[s] = *st(0, s); gp[s] = *st(1, s); }
            } else {
#pragma unroll
            for (int s = 0; s < NSQ; ++s) { bp[s] = *st(0, s); gp[s] = *st(1, s); }   // new_draw = prev_draw (:162)
            prev_K = kinetic();
            }
            double lp = -prev_U;
#pragma unroll 1
            for (uint32_t k = 0; k < n_leap; ++k) {      // :164-176
                if constexpr (DENSEM) {
#pragma unroll
                    for (int s = 0; s < NSQ; ++s) pm[s] = pm[s] + (eps * gp[s]) / 2.0;     // first half-step (:126)
                    double4_t mp[NTQ];
                    next_img = prm.Xp;
                    stream_product(prm.Mip, NBM, pm, mp);
#pragma unroll
                    for (int s = 0; s < NSQ; ++s) bp[s] = bp[s] + eps * mp[s >> 2][s & 3];   // (:171) theta += eps (Minv p)
                    next_img = prm.Mip;                   // behind the evaluation: the next step's product, or prop_K's
                } else {
#pragma unroll
                for (int s = 0; s < NSQ; ++s) {
                    if constexpr (BOUNDS) pm[s] = pm[s] + (eps * box.jgrad(bp[s], gp[s], s)) / 2.0;     // (:122,126) with the inverse Jacobian
                    else pm[s] = pm[s] + (eps * gp[s]) / 2.0; // first half-step (:126)
                    if constexpr (DIAGM) bp[s] = bp[s] + eps * (mass_at(prm.m_inv, s) * pm[s]);   // (:171) theta += eps Minv p
                    else bp[s] = bp[s] + eps * pm[s];    // (:171)
                }
                }
                evaluate_at(bp, gp, lp);
#pragma unroll
                for (int s = 0; s < NSQ; ++s) {
                    if constexpr (BOUNDS) pm[s] = pm[s] + (eps * box.jgrad(bp[s], gp[s], s)) / 2.0;
                    else pm[s] = pm[s] + (eps * gp[s]) / 2.0;     // second half-step (:175)
                }
            }
            if constexpr (DENSEM) {                       // the end point's gradient waits in the wave's third workspace vector
                next_img = prm.Lp;                        // the next draw starts with L z
#pragma unroll
                for (int s = 0; s < NSQ; ++s) *st(2, s) = gp[s];
            }
            const double prop_K = kinetic();
            if constexpr (DENSEM) {
#pragma unroll
                for (int s = 0; s < NSQ; ++s) gp[s] = *st(2, s);
            }
            // BOUNDS: box_log_kernel adds log_jacobian(theta) (:84-95) -- only the end point's value is used (n_leap = 0: prev_U, exactly)
            if constexpr (BOUNDS) { const double lj = box.log_jacobian(bp, lj_rel, [&]() { __syncthreads(); }); if (n_leap != 0u) lp = lp + lj; }
            double prop_U = -lp;                         // :178 (n_leap = 0: the value at the unchanged position)
            const bool u_nf = !is_finite(prop_U);
            if (u_nf) prop_U = INF;                      // :180-182
            if (u_nf || !is_finite(prop_K)) n_acc |= NF_BIT;
            const double x = -(prop_U + prop_K) + (prev_U + prev_K);
            const double comp_val = (x < 0.01) ? x : 0.01;                       // :188
            const double z = rng_uniform(prm.seed, chain, draw + prm.draw0, 0u);             // :189
            const bool accept = z < det_exp(comp_val);                           // :191
            if (accept) {
                prev_U = prop_U;
                if (live) {
#pragma unroll
                    for (int s = 0; s < NSQ; ++s) { *st(0, s) = bp[s]; *st(1, s) = gp[s]; }
                }
            }
            keep_draw(draw, accept);
        }
    }

    if constexpr ((ablate & 2048u) != 0) {
        if (blockIdx.x == 0 && lane == 0 && prm.n_accept) { for (int i = 0; i < 11; ++i) prm.n_accept[2 + w * 11 + i] = prof[i]; }
    }
    if constexpr ((ablate & 1024u) != 0) {           // shader clock over the 100 MHz wall clock, workgroup 0 (n_accept[0..1])
        if (blockIdx.x == 0 && threadIdx.x == 0 && prm.n_accept) { prm.n_accept[0] = clock64() - t0_clk; prm.n_accept[1] = wall_clock64() - t0_wall; }
        return;
    }
    const bool replay = (n_acc & NF_BIT) != 0 && prm.nf_flag != nullptr;
    n_acc &= ~NF_BIT;
    if (live && replay && q == 0 && j4 == 0) { prm.nf_flag[cl] = 1u; prm.nf_flag[C] = 1u; }
    if (live && !replay) {
#pragma unroll
        for (int s = 0; s < NSQ; ++s) {
            const uint32_t dim = dim_of(s);
            double v = *st(0, s);
            if constexpr (BOUNDS) v = box.leave(v, s);
            if (dim < d) prm.theta[(size_t)dim * C + cl] = v;
        }
        if (q == 0 && j4 == 0 && prm.n_accept) prm.n_accept[cl] = n_acc;
    }
}

}  // namespace mi
