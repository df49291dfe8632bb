// lds_box.hpp -- settings.vals_bound on the LDS-streamed kernels (logistic_lds.hpp: hmc; nuts_lds.hpp: nuts): the reference samples in
// the transformed space (ref: src/hmc.cpp:84-95,107-122,134-136,211-218 and the same lines of src/nuts.cpp; include/misc/transform_vals.hpp,
// inv_transform_vals, jacobian_adjust.hpp, log_jacobian.hpp:36-57), element by element with the formulas of hmc_dense.hpp (box_*), for a chain
// whose dimensions are split over the four waves of its tile: wave q holds dims [q DQ, (q+1) DQ) in the MFMA B / D register layout.
//
//   * tables (type 1 none / 2 lower / 3 upper / 4 both of determine_bounds_type.hpp:27-57, the bounds) are read from global memory where they
//     are used, padded to 512 entries with type 1; a wave-uniform bit mask names the slices that hold a bounded dimension at all -- the
//     others skip every transform (exp / log);
//   * log_jacobian is a SCALAR loop over the dimensions in ascending order in the reference, i.e. one running sum that passes through the
//     dims of wave 0, then wave 1, ...: the four waves relay it through LDS (four barriers per evaluation that needs the potential);
//   * the reference's `jacob_matrix * grad` is a dense product with a diagonal matrix: its NaN rule (0 * inf in the off-diagonal) is not
//     reproduced here -- like every other product of these kernels the regime is detected through the energies and the chain replayed by
//     literal.hpp.
#pragma once

#include "hmc_dense.hpp"

namespace mi {

template <int NTQ>
struct LdsBox {
    static constexpr int NS = 4 * NTQ, DQ = 16 * NTQ;
    const int* bt; const double* lb; const double* ub;
    uint32_t bmask, d;
    int q, j4, lane;

    __device__ __forceinline__ void init(const int* bt_, const double* lb_, const double* ub_, uint32_t d_, int q_, int lane_)
    {
        bt = bt_; lb = lb_; ub = ub_; d = d_; q = q_; lane = lane_; j4 = lane_ >> 4;
        uint32_t m = 0;
#pragma unroll
        for (int s = 0; s < NS; ++s) m |= (__ballot(bt[q * DQ + 4 * s + j4] != 1) != 0ull ? 1u : 0u) << s;
        bmask = (uint32_t)__builtin_amdgcn_readfirstlane((int)m);        // bit s: slice s of this wave holds a bounded dimension
    }
    __device__ __forceinline__ bool bounded(int s) const { return ((bmask >> s) & 1u) != 0u; }
    // this lane's entry of slice s (opaque offset: the 3 NS loop-invariant table entries would be spilled otherwise; global, not flat, loads)
    template <class T> __device__ __forceinline__ T at(const T* tab, int s) const
    {
        uint32_t off = (uint32_t)j4 * (uint32_t)sizeof(T);
        asm volatile("" : "+v"(off));
        return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(tab + (q * DQ + 4 * s)) + off);
    }
    __device__ __forceinline__ double enter(double v, int s) const     // transform (hmc.cpp:134-136)
    {
        return bounded(s) ? box_transform(v, at(bt, s), at(lb, s), at(ub, s)) : v;
    }
    __device__ __forceinline__ double leave(double v, int s) const     // inv_transform (hmc.cpp:211-218)
    {
        return bounded(s) ? box_inv_transform(v, at(bt, s), at(lb, s), at(ub, s)) : v;
    }
    __device__ __forceinline__ void x_inplace(double (&th)[NS]) const   // theta -> x = inv_transform(theta) (hmc.cpp:108)
    {
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (bounded(s)) {
                const bool in = (uint32_t)(q * DQ + 4 * s + j4) < d;
                const double x = box_inv_transform(th[s], at(bt, s), at(lb, s), at(ub, s));
                th[s] = in ? x : 0.0;
            }
    }
    // [J^-1]_ii grad_i (hmc.cpp:122: jacob_matrix * grad, a gemv with a diagonal matrix; 1.0 * g on an unbounded dimension: the same bits)
    __device__ __forceinline__ double jgrad(double th_s, double g_s, int s) const
    {
        return bounded(s) ? box_inv_jacobian(th_s, at(bt, s), at(lb, s), at(ub, s)) * g_s : g_s;
    }
    // log_jacobian(theta) (log_jacobian.hpp:36-57): the running sum over the dimensions in ascending order, relayed wave 0 -> 1 -> 2 -> 3 through
    // rel[16] (one double per chain of the tile).  COLLECTIVE: every wave of the workgroup calls it (four barriers).
    template <class Sync>
    __device__ __forceinline__ double log_jacobian(const double (&th)[NS], double* rel, Sync&& sync) const
    {
        const int c = lane & 15;
#pragma unroll 1
        for (int stage = 0; stage < 4; ++stage) {
            if (q == stage) {
                double lj = (stage == 0) ? 0.0 : rel[c];
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    if (!bounded(s)) continue;
                    const double term = box_log_jacobian_term(th[s], at(bt, s), at(lb, s), at(ub, s));
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const double tg = __shfl(term, c + 16 * g);
                        const int dim = q * DQ + 4 * s + g;
                        if ((uint32_t)dim < d && bt[dim] != 1) lj = lj + tg;
                    }
                }
                rel[c] = lj;
            }
            sync();
        }
        return rel[c];
    }
};

}  // namespace mi
