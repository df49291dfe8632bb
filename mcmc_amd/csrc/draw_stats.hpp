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

namespace mi {

constexpr int STATS_MAX_N = 160;          // series length up to which every lag is computed
constexpr int STATS_TILED_LAGS = 128;     // lags computed for longer series
constexpr int STATS_TILE_T = 64;          // time steps per tile of the streamed variant: (TILE_T + 2 LAGS) x 64 doubles = 160 KB

// partial sums of x over (t, chains of group g) per dimension: out[g][j]
__global__ __launch_bounds__(64) void stats_sum_kernel(const double* __restrict__ draws, uint32_t n, uint32_t d, uint64_t C,
                                                       uint32_t G, double* __restrict__ out)
{
    const uint32_t j = blockIdx.x, g = blockIdx.y;
    const uint64_t c_lo = (uint64_t)g * ((C + G - 1) / G), c_hi = (c_lo + (C + G - 1) / G < C) ? c_lo + (C + G - 1) / G : C;
    double s = 0.0;
    for (uint64_t c = c_lo + threadIdx.x; c < c_hi; c += 64)
        for (uint32_t t = 0; t < n; ++t) s += draws[((size_t)t * d + j) * C + c];
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
