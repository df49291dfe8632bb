// draw_stats.hpp -- reducers over the engine's draws_out slabs [n_keep][d][C] on the device: per dimension the
// pooled mean, the autocovariance function pooled over chains (what ESS needs), and the Gelman-Rubin R-hat ingredients.
// The reference has no ESS / R-hat code; these are the caller-side reductions next to the path, so that ESS/sec (the second
// half of BASELINE.json's metric) does not need the draws on the host.  Definitions are those of mcmc_amd/ess.py.
//
// One wave per (dimension j, chain group g): for 64 chains at a time the centred series v[t] = x[t][j][c] - mean_j sit in
// LDS as [t][lane] (conflict-free columns), every lane forms its own lagged products s_k = sum_t v[t] v[t+k] for all lags and
// adds them to per-lag LDS accumulators; at the end the 64 lanes of each lag are summed with a fixed butterfly.  The host
// adds the G partials in order: the result does not depend on scheduling.  Up to 160 kept draws the whole series and all lags
// sit in LDS (two [n][64] fp64 arrays); longer series (the reference's default is 1 000 kept draws) are streamed through LDS in
// time tiles with a halo, and the autocovariance is computed up to lag STATS_TILED_LAGS - 1 (Geyer's sum stops at the first
// non-positive pair, long before that for any usable chain).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace mi {

constexpr int STATS_MAX_N = 160;          // series length up to which every lag is computed
constexpr int STATS_TILED_LAGS = 128;     // lags computed for longer series
constexpr int STATS_TILE_T = 64;          // time steps per tile of the streamed variant: (TILE_T + 2 LAGS) x 64 doubles = 160 KB

// partial sums of x over (t = 0, t_step, 2 t_step, ... < n; chains of group g) per dimension: out[g][j]
__global__ __launch_bounds__(64) void stats_sum_kernel(const double* __restrict__ draws, uint32_t n, uint32_t t_step, uint32_t d, uint64_t C,
                                                       uint32_t G, double* __restrict__ out)
{
    const uint32_t j = blockIdx.x, g = blockIdx.y;
    const uint64_t c_lo = (uint64_t)g * ((C + G - 1) / G), c_hi = (c_lo + (C + G - 1) / G < C) ? c_lo + (C + G - 1) / G : C;
    double s = 0.0;
    for (uint64_t c = c_lo + threadIdx.x; c < c_hi; c += 64)
        for (uint32_t t = 0; t < n; t += t_step) s += draws[((size_t)t * d + j) * C + c];
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if (threadIdx.x == 0) out[(size_t)g * d + j] = s;
}

// Lags [16 b_lo, 16 b_hi) of the series of the dimensions listed in `dims` (nullptr: all d), n <= STATS_MAX_N:
//   out[g][jj][0 .. 16 (b_hi - b_lo)): sum over the group's chains of sum_t v[t] v[t-k];  with want_mom also out2[g][jj][0..3):
//   sum_c m_c, sum_c m_c^2, sum_c var_c (m_c = chain mean of the centred series, var_c = its unbiased variance) for R-hat.
// One wave per (dimension, chain group); per tile of 64 chains the centred series sit in LDS as [t][lane] (conflict-free columns,
// n x 512 bytes) and the lags are formed 16 at a time in registers: the lane walks t once per block with a 16-deep window of
// v[t - k0 - i] (a circular buffer with compile-time indices), two LDS reads and 16 fmas per step; per-lane sums of the pass's lags
// accumulate in LDS over the tiles and are summed across the 64 lanes by a fixed butterfly at the end: the result does not depend on
// scheduling.  The host asks for the lags in passes, and only for the dimensions whose Geyer sum has not ended yet.
// (The first version held an [n][64] accumulator array next to the series -- one wave per CU -- and every lane formed all n lags
// from LDS operands: 124 ms for the 6.7 GB of configs[1]'s kept draws, more than the sampler took to produce them.)
constexpr int STATS_LB = 16;              // lags per register block
constexpr int STATS_PASS_BLOCKS = 3;      // blocks per launch at most: series + accumulators stay near 75 KB of LDS at n = 100
__global__ __launch_bounds__(64) void stats_acov_kernel(const double* __restrict__ draws, const double* __restrict__ mean,
                                                        uint32_t n, uint32_t d, uint64_t C, uint32_t G,
                                                        const uint32_t* __restrict__ dims, uint32_t b_lo, uint32_t b_hi, int want_mom,
                                                        double* __restrict__ out, double* __restrict__ out2)
{
    constexpr int LB = STATS_LB;
    extern __shared__ double lds[];
    double* v = lds;                       // [n][64]
    double* accl = lds + (size_t)n * 64;   // [(b_hi - b_lo) * LB][64]
    const uint32_t jj = blockIdx.x, g = blockIdx.y, lane = threadIdx.x;
    const uint32_t j = dims ? dims[jj] : jj;
    const uint32_t nd = gridDim.x;
    const uint64_t per = (C + G - 1) / G;
    const uint64_t c_lo = (uint64_t)g * per, c_hi = (c_lo + per < C) ? c_lo + per : C;
    const double mj = mean[j];
    const uint32_t npl = (b_hi - b_lo) * LB;            // lags of this pass
    for (uint32_t k = 0; k < npl; ++k) accl[(size_t)k * 64 + lane] = 0.0;
    double sm = 0.0, sm2 = 0.0, sv = 0.0;
    const size_t row = (size_t)d * C;
    for (uint64_t c0 = c_lo; c0 < c_hi; c0 += 64) {
        const uint64_t c = c0 + lane;
        const bool on = c < c_hi;
        const double* p = draws + (size_t)j * C + (on ? c : c_hi - 1);
        double s1 = 0.0;
        for (uint32_t t0 = 0; t0 < n; t0 += LB) {        // LB rows in flight per lane (a row of the wave = one 512-byte segment)
            double xr[LB];
#pragma unroll
            for (int u = 0; u < LB; ++u) xr[u] = p[(size_t)((t0 + u < n) ? t0 + u : n - 1) * row];
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                if (t0 + u < n) {
                    const double x = on ? xr[u] - mj : 0.0;
                    v[(size_t)(t0 + u) * 64 + lane] = x;
                    s1 += x;
                }
            }
        }
        if (want_mom) {
            const double mc = s1 / (double)n;
            double ss = 0.0;
            for (uint32_t t = 0; t < n; ++t) { const double e = v[(size_t)t * 64 + lane] - mc; ss = __builtin_fma(e, e, ss); }
            if (on) { sm += mc; sm2 = __builtin_fma(mc, mc, sm2); sv += (n > 1) ? ss / (double)(n - 1) : 0.0; }
        }
        for (uint32_t b = b_lo; b < b_hi; ++b) {
            const uint32_t k0 = b * LB;
            double acc[LB], win[LB];
#pragma unroll
            for (int i = 0; i < LB; ++i) { acc[i] = 0.0; win[i] = 0.0; }
            // t runs from k0 (the first time step whose lag-k0 partner exists) in chunks of LB: slot u of the window receives
            // v[t - k0], and lag k0 + i pairs v[t] with slot (u - i) mod LB = v[t - k0 - i] (zero while that is before the series)
            for (uint32_t tc = k0; tc < n; tc += LB) {
#pragma unroll
                for (int u = 0; u < LB; ++u) {
                    const uint32_t t = tc + (uint32_t)u;
                    const bool in = t < n;
                    const uint32_t tcl = in ? t : n - 1;
                    const double a = in ? v[(size_t)tcl * 64 + lane] : 0.0;
                    win[u] = in ? v[(size_t)(tcl - k0) * 64 + lane] : 0.0;
#pragma unroll
                    for (int i = 0; i < LB; ++i) acc[i] = __builtin_fma(a, win[(u - i + LB) % LB], acc[i]);
                }
            }
            double* al = accl + (size_t)(b - b_lo) * LB * 64 + lane;
#pragma unroll
            for (int i = 0; i < LB; ++i) al[(size_t)i * 64] += acc[i];
        }
    }
    for (uint32_t k = 0; k < npl; ++k) {
        double s_ = accl[(size_t)k * 64 + lane];
        for (int m = 32; m >= 1; m >>= 1) s_ += __shfl_xor(s_, m);
        if (lane == 0) out[((size_t)g * nd + jj) * npl + k] = s_;
    }
    if (want_mom) {
        for (int m = 32; m >= 1; m >>= 1) { sm += __shfl_xor(sm, m); sm2 += __shfl_xor(sm2, m); sv += __shfl_xor(sv, m); }
        if (lane == 0) { double* o = out2 + ((size_t)g * nd + jj) * 3; o[0] = sm; o[1] = sm2; o[2] = sv; }
    }
}

// The same for n > STATS_MAX_N: lags 0 .. STATS_TILED_LAGS - 1, the series streamed through LDS in tiles of STATS_TILE_T steps plus a
// halo of STATS_TILED_LAGS.  out[g][j][0..STATS_TILED_LAGS), out2 as above.
__global__ __launch_bounds__(64) void stats_acov_tiled_kernel(const double* __restrict__ draws, const double* __restrict__ mean,
                                                              uint32_t n, uint32_t d, uint64_t C, uint32_t G,
                                                              double* __restrict__ out, double* __restrict__ out2)
{
    constexpr uint32_t L = STATS_TILED_LAGS, T = STATS_TILE_T;
    extern __shared__ double lds[];
    double* v = lds;                             // [T + L][64]
    double* acc = lds + (size_t)(T + L) * 64;    // [L][64]
    const uint32_t j = blockIdx.x, g = blockIdx.y, lane = threadIdx.x;
    const uint64_t per = (C + G - 1) / G;
    const uint64_t c_lo = (uint64_t)g * per, c_hi = (c_lo + per < C) ? c_lo + per : C;
    const double mj = mean[j];
    for (uint32_t k = 0; k < L; ++k) acc[(size_t)k * 64 + lane] = 0.0;
    double sm = 0.0, sm2 = 0.0, sv = 0.0;
    for (uint64_t c0 = c_lo; c0 < c_hi; c0 += 64) {
        const uint64_t c = c0 + lane;
        const bool on = c < c_hi;
        double s1 = 0.0;
        for (uint32_t t = 0; t < n; ++t) s1 += on ? draws[((size_t)t * d + j) * C + c] - mj : 0.0;
        const double mc = s1 / (double)n;
        double ss = 0.0;
        for (uint32_t t0 = 0; t0 < n; t0 += T) {
            const uint32_t rows = (n - t0 < T + L) ? n - t0 : T + L;       // tile + halo
            const uint32_t own = (n - t0 < T) ? n - t0 : T;
            for (uint32_t r = 0; r < rows; ++r) v[(size_t)r * 64 + lane] = on ? draws[((size_t)(t0 + r) * d + j) * C + c] - mj : 0.0;
            for (uint32_t r = 0; r < own; ++r) { const double e = v[(size_t)r * 64 + lane] - mc; ss = __builtin_fma(e, e, ss); }
            for (uint32_t k = 0; k < L; ++k) {
                double s_ = 0.0;
                for (uint32_t r = 0; r < own && r + k < rows; ++r) s_ = __builtin_fma(v[(size_t)r * 64 + lane], v[(size_t)(r + k) * 64 + lane], s_);
                acc[(size_t)k * 64 + lane] += s_;
            }
        }
        if (on) { sm += mc; sm2 = __builtin_fma(mc, mc, sm2); sv += (n > 1) ? ss / (double)(n - 1) : 0.0; }
    }
    for (uint32_t k = 0; k < L; ++k) {
        double s_ = acc[(size_t)k * 64 + lane];
        for (int m = 32; m >= 1; m >>= 1) s_ += __shfl_xor(s_, m);
        if (lane == 0) out[((size_t)g * d + j) * L + k] = s_;
    }
    for (int m = 32; m >= 1; m >>= 1) { sm += __shfl_xor(sm, m); sm2 += __shfl_xor(sm2, m); sv += __shfl_xor(sv, m); }
    if (lane == 0) { double* o = out2 + ((size_t)g * d + j) * 3; o[0] = sm; o[1] = sm2; o[2] = sv; }
}

// ---- the fast path of ESS / R-hat (no full autocovariance requested): lags 0 .. L-1 only, straight from HBM.
// Geyer's initial positive sequence stops at the first non-positive pair of autocorrelations -- a handful of lags for a sampler
// that mixes -- so computing all n lags (n^2 / 2 LDS-fed fmas per series, one wave per CU: the kernels above) is wasted work, and
// measured slower than the sampler whose draws it summarises.  Here a lane streams its chain's series through a window of L
// registers (a circular buffer with compile-time indices): one coalesced 512-byte load per time step and wave, L fmas per value,
// no LDS; many waves per SIMD.  out[g][j][0..L): sum over the group's chains of sum_t v[t] v[t-k]; out2 as above, with the chain
// variance from sum v^2 - n m_c^2 (v is centred by the pooled mean, so nothing cancels).  The host checks whether Geyer's sum
// ended inside the L lags and asks for more (2 L, then every lag with the kernels above) if not.
template <int L>
__global__ __launch_bounds__(64) void stats_window_kernel(const double* __restrict__ draws, const double* __restrict__ mean,
                                                          uint32_t n, uint32_t d, uint64_t C, uint32_t G,
                                                          double* __restrict__ out, double* __restrict__ out2)
{
    const uint32_t j = blockIdx.x, g = blockIdx.y, lane = threadIdx.x;
    const uint64_t per = (C + G - 1) / G;
    const uint64_t c_lo = (uint64_t)g * per, c_hi = (c_lo + per < C) ? c_lo + per : C;
    const double mj = mean[j];
    const size_t row = (size_t)d * C;                   // doubles between consecutive draws of one (dimension, chain)
    double acc[L];
#pragma unroll
    for (int k = 0; k < L; ++k) acc[k] = 0.0;
    double sm = 0.0, sm2 = 0.0, sv = 0.0;
    for (uint64_t c0 = c_lo; c0 < c_hi; c0 += 64) {
        const uint64_t c = c0 + lane;
        const bool on = c < c_hi;
        const double* p = draws + (size_t)j * C + (on ? c : c_hi - 1);
        double win[L];
#pragma unroll
        for (int u = 0; u < L; ++u) win[u] = 0.0;
        double s1 = 0.0, q0 = 0.0;
        for (uint32_t t0 = 0; t0 < n; t0 += L) {
            double x[L];
#pragma unroll
            for (int u = 0; u < L; ++u) {               // L loads in flight per lane
                const uint32_t t = t0 + (uint32_t)u;
                x[u] = p[(size_t)(t < n ? t : n - 1) * row];
            }
#pragma unroll
            for (int u = 0; u < L; ++u) {
                const double v = (on && t0 + (uint32_t)u < n) ? x[u] - mj : 0.0;
                win[u] = v;
                s1 += v;
                q0 = __builtin_fma(v, v, q0);
#pragma unroll
                for (int k = 0; k < L; ++k) acc[k] = __builtin_fma(v, win[(u - k + L) % L], acc[k]);
            }
        }
        const double mc = s1 / (double)n;
        if (on) { sm += mc; sm2 = __builtin_fma(mc, mc, sm2); sv += (n > 1) ? (q0 - (double)n * mc * mc) / (double)(n - 1) : 0.0; }
    }
#pragma unroll
    for (int k = 0; k < L; ++k) {
        double s_ = acc[k];
        for (int m = 32; m >= 1; m >>= 1) s_ += __shfl_xor(s_, m);
        if (lane == 0) out[((size_t)g * d + j) * L + k] = s_;
    }
    for (int m = 32; m >= 1; m >>= 1) { sm += __shfl_xor(sm, m); sm2 += __shfl_xor(sm2, m); sv += __shfl_xor(sv, m); }
    if (lane == 0) { double* o = out2 + ((size_t)g * d + j) * 3; o[0] = sm; o[1] = sm2; o[2] = sv; }
}

// ---- every lag in ONE pass, on the fp64 matrix cores (round 4; n <= 128).  For a slowly mixing sampler Geyer's sum needs every lag
// (configs[2]: 107 GB of kept draws, 272 ms through the window + lag-block passes above, 13 times what one read of the slab costs).
// The autocovariances of a dimension are the diagonal sums of a Gram matrix over the chains,
//     acov_k = sum_t G[t][t + k] / (C (n - k)),      G[t][s] = sum_c v[t][c] v[s][c],      v = x - pooled mean,
// and G = V V' is exactly a v_mfma_f64_16x16x4_f64 product contracting over 4 chains per instruction: with the centred series of a
// tile of 32 chains in LDS as [t][chain] (row stride 34: conflict-free fragment reads), fragment tb of chain group cg -- lane l holds
// v[16 tb + (l & 15)][4 cg + (l >> 4)] -- is the A operand of row block tb AND the B operand of column block tb.  A wave keeps the
// NB (NB + 1) / 2 upper blocks of G in its accumulators (NB = n / 16 rounded up: 28 blocks = 224 registers at n = 100) over all its
// tiles; per 4 chains that is 3.5 KB of draws for 28 MFMAs of 64 cycles: 2 bytes per SIMD cycle, the chip's HBM rate -- the pass
// is bound by the read of the slab and by the matrix pipe at the same time.  The R-hat moments ride along (per-chain sums as the tile
// is loaded).  At the end the four waves of a workgroup spill their blocks through LDS one after the other and thread k adds diagonal
// k in a fixed order; the host adds the G chain groups in order: the result does not depend on scheduling.
//   out[g][j][0 .. 16 NB): sum over the group's chains of sum_t v[t] v[t + k];  out[g][j][16 NB .. 32 NB): the row sums sum_c v[t][c];
//   out2[g][j][0..3) as in the kernels above.  v = x - mean[j], where `mean` may be a PROVISIONAL centre (see the host).
constexpr int STATS_GRAM_MAX_N = 128;
constexpr int STATS_GRAM_TC = 64;                       // chains per tile
constexpr int STATS_GRAM_RS = 66;                       // LDS row stride in doubles: (66 i + k) mod 32 = 2 i + k, conflict-free fragment reads
constexpr size_t stats_gram_lds_bytes(int nb)
{
    const size_t tiles = (size_t)2 * 16 * nb * STATS_GRAM_RS + 2 * 8 * 64 * 2, mat = (size_t)16 * nb * (16 * nb + 1);
    return ((tiles > mat ? tiles : mat) + 32) * sizeof(double);
}
// upper-triangle block b of an NB x NB block matrix, row-major: (tb, sb), tb <= sb
template <int NB> constexpr int gram_tb(int b) { int tb = 0; while (b >= NB - tb) { b -= NB - tb; ++tb; } return tb; }
template <int NB> constexpr int gram_sb(int b) { int tb = 0; while (b >= NB - tb) { b -= NB - tb; ++tb; } return tb + b; }

typedef double double4_gram __attribute__((ext_vector_type(4)));
// the blocks b = Q, Q + 4, ... of quarter Q, with (tb, sb) as true compile-time constants (as values of a loop variable the compiler
// indexed the fragment array dynamically: scratch, and a vmcnt(0) per MFMA that also waited for the next tile's loads)
template <int NB, int Q, int I = 0>
__device__ __forceinline__ void gram_run(const double (&f)[NB], double4_gram (&acc)[(NB * (NB + 1) / 2 + 3) / 4])
{
    constexpr int b = Q + 4 * I;
    if constexpr (b < NB * (NB + 1) / 2) {
        constexpr int tb = gram_tb<NB>(b), sb = gram_sb<NB>(b);
        acc[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[tb], f[sb], acc[I], 0, 0, 0);
        gram_run<NB, Q, I + 1>(f, acc);
    }
}
template <int NB, int Q, int I = 0>
__device__ __forceinline__ void gram_spill(double* M, int lane, const double4_gram (&acc)[(NB * (NB + 1) / 2 + 3) / 4])
{
    constexpr int b = Q + 4 * I, NR = 16 * NB;
    if constexpr (b < NB * (NB + 1) / 2) {
        constexpr int tb = gram_tb<NB>(b), sb = gram_sb<NB>(b);
#pragma unroll
        for (int r = 0; r < 4; ++r) M[(size_t)(16 * tb + 4 * r + (lane >> 4)) * (NR + 1) + 16 * sb + (lane & 15)] = acc[I][r];
        gram_spill<NB, Q, I + 1>(M, lane, acc);
    }
}

// Workgroup = 8 waves (two per SIMD) sharing ONE tile of 64 chains at a time, double-buffered in LDS: while the waves run the MFMAs
// of tile i out of one buffer, the rows of tile i + 1 are in flight (14 coalesced 512-byte loads per lane) and land in the other --
// one barrier per tile.  Wave w = (quarter q = w >> 1, parity w & 1) owns the blocks b = q (mod 4) of G for the chain groups of its
// parity: 7 accumulator blocks at n = 100 (the first version gave every wave a tile of its own and all 28 blocks: 446 registers, one
// wave per SIMD, loads and MFMAs taking turns -- 2 TB/s).
template <int NB>
__global__ __launch_bounds__(512, 2) void stats_gram_kernel(const double* __restrict__ draws, const double* __restrict__ mean,
                                                            uint32_t n, uint32_t d, uint64_t C, uint32_t G,
                                                            double* __restrict__ out, double* __restrict__ out2)
{
    typedef double4_gram double4_g;
    constexpr int NR = 16 * NB, RS = STATS_GRAM_RS, TC = STATS_GRAM_TC, NBLK = NB * (NB + 1) / 2, BPW = (NBLK + 3) / 4;
    constexpr int RPW = (NR + 7) / 8;                   // rows a wave loads per tile: w, w + 8, ...
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const uint32_t j = blockIdx.x, g = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = __builtin_amdgcn_readfirstlane(wave >> 1), par = __builtin_amdgcn_readfirstlane(wave & 1);
    double* const buf0 = lds;                           // [2][NR][RS]
    double* const momb = lds + (size_t)2 * NR * RS;     // [2][8 waves][64][2]: per-chain partial sums of a tile's rows
    const uint64_t per = (C + G - 1) / G;
    const uint64_t c_lo = (uint64_t)g * per, c_hi = (c_lo + per < C) ? c_lo + per : C;
    const double mj = mean[j];
    const size_t row = (size_t)d * C;
    for (int i = threadIdx.x; i < 2 * NR * RS; i += 512) buf0[i] = 0.0;     // rows n .. NR - 1 and the padding columns stay zero
    double4_g acc[BPW];
#pragma unroll
    for (int b = 0; b < BPW; ++b) acc[b] = double4_g{0.0, 0.0, 0.0, 0.0};
    double sm = 0.0, sm2 = 0.0, sv = 0.0;
    const uint64_t n_tiles = (c_hi > c_lo) ? (c_hi - c_lo + TC - 1) / TC : 0;
    // Two tiles ahead: the rows of tiles i + 1 and i + 2 are in flight (registers) while tile i runs out of LDS.  One tile ahead kept
    // 57 KB in flight per CU, 15 MB on the chip: at the ~7 us a 512-byte row segment takes under load that is 2 TB/s (measured);
    // HBM wants twice as many bytes in flight.  (NB = 8 has no registers for the second set and stays one tile ahead.)
    constexpr bool DEEP = false;                        // (measured at n = 100: 52.2 ms either way once the barrier stopped draining the loads -- the pass is not latency-bound)
    // The barrier of the tile loop orders LDS traffic only.  __syncthreads() is a workgroup-scope fence as well and drains the GLOBAL
    // loads in flight (s_waitcnt vmcnt(0)): the prefetched rows would be waited for at every tile, which is what held the first
    // versions at 2 TB/s however far ahead they requested.
    auto lds_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    double xrA[RPW], xrB[DEEP ? RPW : 1];
    // this lane's share of the ROW sums R_t = sum_c v[t][c], rows wave + 8 u: the host needs them to move the centre from the provisional
    // `mean` the kernel is given to the pooled mean it yields (one pass over the slab instead of a mean pass and this one)
    double rs[RPW];
#pragma unroll
    for (int u = 0; u < RPW; ++u) rs[u] = 0.0;
    // rows wave, wave + 8, ... of tile `tile`: requested into registers
    auto request = [&](uint64_t tile, auto& xr) __attribute__((always_inline)) {
        const uint64_t c = c_lo + tile * TC + lane;
        const double* p = draws + (size_t)j * C + ((c < c_hi) ? c : c_hi - 1);
#pragma unroll
        for (int u = 0; u < RPW; ++u) {
            const uint32_t t = (uint32_t)(wave + 8 * u);
            xr[u] = p[(size_t)(t < n ? t : n - 1) * row];
        }
    };
    // ... centred, written to buffer `bi`, with this wave's part of the per-chain sums
    auto deposit = [&](uint64_t tile, int bi, const auto& xr) __attribute__((always_inline)) {
        const bool on = c_lo + tile * TC + lane < c_hi;
        double* const V = buf0 + (size_t)bi * NR * RS;
        double s1 = 0.0, q0 = 0.0;
#pragma unroll
        for (int u = 0; u < RPW; ++u) {
            const uint32_t t = (uint32_t)(wave + 8 * u);
            if (t < n) {
                const double v = on ? xr[u] - mj : 0.0;
                V[(size_t)t * RS + lane] = v;
                s1 += v;
                q0 = __builtin_fma(v, v, q0);
                rs[u] += v;
            }
        }
        double* m = momb + (((size_t)bi * 8 + wave) * 64 + lane) * 2;
        m[0] = s1; m[1] = q0;
    };
    // the MFMAs and the R-hat moments of the tile in buffer bi
    auto compute = [&](uint64_t tile, int bi) __attribute__((always_inline)) {
        const double* const V = buf0 + (size_t)bi * NR * RS;
        if ((uint64_t)wave == (tile & 7)) {             // one wave adds the 8 partial per-chain sums, in order
            double s1 = 0.0, q0 = 0.0;
            for (int wv = 0; wv < 8; ++wv) { const double* m = momb + (((size_t)bi * 8 + wv) * 64 + lane) * 2; s1 += m[0]; q0 += m[1]; }
            if (c_lo + tile * TC + lane < c_hi) {
                const double mc = s1 / (double)n;
                sm += mc; sm2 = __builtin_fma(mc, mc, sm2);
                sv += (n > 1) ? (q0 - (double)n * mc * mc) / (double)(n - 1) : 0.0;
            }
        }
        const double* fr = V + (size_t)(lane & 15) * RS + (lane >> 4) + 4 * par;
#pragma unroll 1
        for (int cg2 = 0; cg2 < TC / 8; ++cg2) {        // chain groups cg = 2 cg2 + par
            double f[NB];
#pragma unroll
            for (int tb = 0; tb < NB; ++tb) f[tb] = fr[(size_t)(16 * tb) * RS + 8 * cg2];
            if (q == 0) gram_run<NB, 0>(f, acc);
            else if (q == 1) gram_run<NB, 1>(f, acc);
            else if (q == 2) gram_run<NB, 2>(f, acc);
            else gram_run<NB, 3>(f, acc);
        }
    };
    __syncthreads();                                    // the zeroed buffers
    if (n_tiles > 0) { request(0, xrA); deposit(0, 0, xrA); }
    if (n_tiles > 1) request(1, xrA);
    __syncthreads();
    if constexpr (DEEP) {
        // invariant at the top of a step: tile `tile` in buffer bi, tile + 1 in flight in the first register set
        auto step = [&](uint64_t tile, auto& x_next, auto& x_after) __attribute__((always_inline)) {
            const int bi = (int)(tile & 1);
            if (tile + 2 < n_tiles) request(tile + 2, x_after);
            compute(tile, bi);
            if (tile + 1 < n_tiles) deposit(tile + 1, 1 - bi, x_next);
            lds_barrier();                              // tile + 1 is in its buffer, everybody is done with tile
        };
        for (uint64_t tile = 0; tile < n_tiles; tile += 2) {
            step(tile, xrA, xrB);
            if (tile + 1 < n_tiles) step(tile + 1, xrB, xrA);
        }
    } else {
        for (uint64_t tile = 0; tile < n_tiles; ++tile) {
            const int bi = (int)(tile & 1);
            compute(tile, bi);                          // (tile + 1 was requested before this tile's MFMAs began)
            if (tile + 1 < n_tiles) deposit(tile + 1, 1 - bi, xrA);
            if (tile + 2 < n_tiles) request(tile + 2, xrA);
            lds_barrier();
        }
    }
    // ---- diagonal sums: the 8 waves spill their blocks through LDS one after the other, thread k adds diagonal k (fixed order)
    double* const M = lds;                              // [NR][NR + 1]
    double* const red = lds + ((size_t)2 * NR * RS + 2 * 8 * 64 * 2 > (size_t)NR * (NR + 1) ? (size_t)2 * NR * RS + 2 * 8 * 64 * 2 : (size_t)NR * (NR + 1));
    double lag_sum = 0.0;
    for (int m = 32; m >= 1; m >>= 1) { sm += __shfl_xor(sm, m); sm2 += __shfl_xor(sm2, m); sv += __shfl_xor(sv, m); }
    if (lane == 0) { red[wave * 3 + 0] = sm; red[wave * 3 + 1] = sm2; red[wave * 3 + 2] = sv; }
    for (int pr = 0; pr < 2; ++pr) {                    // the four quarters of one parity together fill the whole upper triangle
        __syncthreads();
        if (par == pr) {
            if (q == 0) gram_spill<NB, 0>(M, lane, acc);
            else if (q == 1) gram_spill<NB, 1>(M, lane, acc);
            else if (q == 2) gram_spill<NB, 2>(M, lane, acc);
            else gram_spill<NB, 3>(M, lane, acc);
        }
        __syncthreads();
        const uint32_t k = threadIdx.x;
        if (k < n) {
            double s_ = 0.0;
            for (uint32_t t = 0; t + k < n; ++t) s_ += M[(size_t)t * (NR + 1) + t + k];
            lag_sum += s_;
        }
    }
    if (threadIdx.x < (uint32_t)NR) out[((size_t)g * d + j) * (2 * NR) + threadIdx.x] = (threadIdx.x < n) ? lag_sum : 0.0;
#pragma unroll
    for (int u = 0; u < RPW; ++u) {                     // row sums: the 64 chains-lanes of the wave that owns the row, fixed butterfly
        double r_ = rs[u];
        for (int m = 32; m >= 1; m >>= 1) r_ += __shfl_xor(r_, m);
        const uint32_t t = (uint32_t)(wave + 8 * u);
        if (lane == 0 && t < (uint32_t)NR) out[((size_t)g * d + j) * (2 * NR) + NR + t] = (t < n) ? r_ : 0.0;
    }
    if (threadIdx.x == 0) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        for (int wv = 0; wv < 8; ++wv) { a0 += red[wv * 3 + 0]; a1 += red[wv * 3 + 1]; a2 += red[wv * 3 + 2]; }
        double* o = out2 + ((size_t)g * d + j) * 3;
        o[0] = a0; o[1] = a1; o[2] = a2;
    }
}

// draws_out slab [n][d][C] -> per chain the column-major n x d matrix Eigen would hold: out[c][j][k] (SURVEY 8 f-3 "layout
// converters").  One workgroup per (dimension j, tile of 64 chains): rows of 64 chains in (512-byte coalesced), through an LDS tile,
// out as runs of n consecutive doubles per chain.
constexpr int TRANSPOSE_KB = 128;                      // draws per LDS tile
__global__ __launch_bounds__(256) void draws_to_chain_major_slice_kernel(const double* __restrict__ in, uint32_t n, uint32_t d, uint64_t C_all,
                                                                        uint64_t c_first, uint64_t c_cnt, double* __restrict__ out)
{
    __shared__ double tile[TRANSPOSE_KB * 65];
    const uint32_t j = blockIdx.x;
    const uint64_t c0 = c_first + (uint64_t)blockIdx.y * 64;
    const uint64_t C = c_first + c_cnt;                 // one past the last chain of this launch; rows keep the stride C_all
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint32_t kb = 0; kb < n; kb += TRANSPOSE_KB) {
        const uint32_t rows = (n - kb < (uint32_t)TRANSPOSE_KB) ? n - kb : (uint32_t)TRANSPOSE_KB;
        for (uint32_t k = wv; k < rows; k += 4) {
            const uint64_t c = c0 + lane;
            tile[(size_t)k * 65 + lane] = (c < C) ? in[((size_t)(kb + k) * d + j) * C_all + c] : 0.0;
        }
        __syncthreads();
        for (uint32_t cc = wv; cc < 64; cc += 4) {
            const uint64_t c = c0 + cc;
            if (c < C) {
                double* o = out + ((size_t)c * d + j) * n + kb;
                for (uint32_t k = lane; k < rows; k += 64) o[k] = tile[(size_t)k * 65 + cc];
            }
        }
        __syncthreads();
    }
}

}  // namespace mi
