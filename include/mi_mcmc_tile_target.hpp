// mi_mcmc_tile_target.hpp -- user-defined DEVICE targets on the TILED engine: d up to 128 at the speed of the built-in MFMA kernels.
//
// include/mi_mcmc_target.hpp gives a user target one lane per chain (d <= 8).  This header is the same idea one level up: the
// reference's callback (ref: include/mcmc/hmc.hpp:42-48) as a functor that evaluates value and gradient for a TILE of 16 chains at
// once, on vectors that live in registers in the fp64 MFMA B / D lane layout
//     lane l, register s  <->  dimension 4 s + (l >> 4) of chain (l & 15),      s = 0 .. 4 NT - 1,      d <= 16 NT
// so that a dense mat-vec inside the target runs on the matrix cores with the helpers below, exactly as the built-in Gaussian
// kernels do (mcmc_amd/csrc/hmc_dense.hpp explains the layout; mcmc_amd/csrc/tile_samplers.hpp states the contract):
//
//     struct MyTile {
//         static constexpr int NT = 8;                    // 1, 2, 4 or 8: the padded dimension is 16 NT
//         static constexpr int WPB = 8;                   // optional: waves per workgroup, 4 (default) or 8 (two per SIMD)
//         const double* P; uint32_t d;                    // anything trivially copyable; pointers are DEVICE pointers
//         size_t lds_doubles() const { return mi::tile::matrix_doubles<NT>(); }
//         __device__ void stage(double* lds) const { mi::tile::stage_matrix<NT>(P, d, lds); }
//         __device__ void grad_tile(const double* lds, const double (&th)[4 * NT], double (&g)[4 * NT], double& value, bool want_value) const
//         {
//             double w[4 * NT];
//             mi::tile::matvec<NT>(lds, th, w);           // w = P theta: each row one fma chain, k ascending (MFMA f64 16x16x4)
//             for (int s = 0; s < 4 * NT; ++s) g[s] = -w[s];
//             if (want_value) value = -0.5 * mi::tile::dot<4 * NT>(th, w);   // four strided fma chains, (q0 + q2) + (q1 + q3): same bits in the chain's 4 lanes
//         }
//     };
//     MI_MCMC_DEFINE_TILE_TARGET(my_tile, MyTile)
//
//     hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I<repo>/include -shared my_tile.hip \
//           -L<repo>/mcmc_amd -lmi_mcmc -o libmy_tile.so
//
// The macro defines  extern "C" int my_tile_run(int algo, const MyTile* target, uint64_t d, const mi_settings*, mi_chains*, void* stream)
// with algo 0 = mcmc::hmc, 1 = mcmc::mala, 2 = mcmc::nuts (max_tree_depth <= 10); hmc and nuts also with settings.vals_bound and / or a
// DIAGONAL precond_mat, mala with a diagonal precond_mat (anything else -- mala with bounds, a dense precond_mat -- returns MI_ERR_UNSUPPORTED with the reason),
// the settings / chains contract of include/mi_mcmc.h (host or device memory, global chain ids, draw0).  The target above IS the
// built-in dense Gaussian: it reproduces hmc_gauss_mfma_kernel's draws bit for bit (tests/test_user_tile_target.py), and a
// non-Gaussian target is checked the way every target is -- the oracle driven by a host function with the same operation order.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "mi_mcmc.h"
#include "mi_mcmc_engine/tile_samplers.hpp"
#include "mi_mcmc_engine/nuts_tile.hpp"

namespace mi {
namespace tile {

// doubles of LDS one d_pad x d_pad matrix takes in MFMA A-fragment order
template <int NT> constexpr size_t matrix_doubles() { return (size_t)NT * 4 * NT * 64; }
// stage a row-major d x d device matrix into LDS in fragment order (zero padded); every thread of the workgroup calls it
template <int NT> __device__ __forceinline__ void stage_matrix(const double* M, uint32_t d, double* lds) { stage_precision<NT>(M, d, lds); }
// y = M x for the wave's 16 chains, M staged at `lds` by stage_matrix: v_mfma_f64_16x16x4_f64, each row a sequential fma chain
template <int NT> __device__ __forceinline__ void matvec(const double* lds, const double (&x)[4 * NT], double (&y)[4 * NT])
{
    matvec_mfma<NT>(lds + (threadIdx.x & 63), x, y);
}
// x . y over the chain's dimensions: four strided fma chains combined (q0 + q2) + (q1 + q3); every lane of the chain gets the result
template <int NS> __device__ __forceinline__ double dot(const double (&x)[NS], const double (&y)[NS]) { return dot4<NS>(x, y); }
// this lane's dimension of register s, and which of the wave's 16 chains the lane belongs to
__device__ __forceinline__ int dim_of(int s) { return 4 * s + (int)((threadIdx.x & 63) >> 4); }
__device__ __forceinline__ int chain_in_tile() { return (int)(threadIdx.x & 15); }

}  // namespace tile

template <class T>
int tile_target_launch(int algo, const void* tile_params, const void* target_pod, uint64_t lds_bytes, void* stream)
{
    static_assert(std::is_trivially_copyable<T>::value, "a device target is passed to the kernels by value");
    static_assert(T::NT == 1 || T::NT == 2 || T::NT == 4 || T::NT == 8, "NT is 1, 2, 4 or 8");
    constexpr int WPB = tile_wpb<T>();
    static_assert(WPB == 4 || WPB == 8, "WPB is 4 or 8");
    const TileParams& prm = *static_cast<const TileParams*>(tile_params);
    const T& tgt = *static_cast<const T*>(target_pod);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((prm.C + 16 * WPB - 1) / (16 * WPB)));
    hipError_t e;
    const bool gen = prm.btype != nullptr;               // settings.vals_bound and / or a diagonal precond_mat (TileGen)
    if (algo == 0 && gen) {
        auto kern = hmc_tile_gen_kernel<T>;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3((unsigned)((prm.C + 63) / 64)), dim3(256), lds_bytes, st, prm, tgt);
    } else if (algo == 2 && gen) {
        auto kern = nuts_tile_kernel<T, true>;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3(prm.nuts_grid ? (unsigned)prm.nuts_grid : (unsigned)((prm.C + 63) / 64)), dim3(256), lds_bytes, st, prm, tgt);
    } else if (algo == 0) {
        auto kern = hmc_tile_kernel<T, WPB>;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, grid, dim3(64 * WPB), lds_bytes, st, prm, tgt);
    } else if (algo == 1 && prm.m != nullptr) {
        auto kern = mala_tile_kernel<T, true>;           // a diagonal precond_mat
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3((unsigned)((prm.C + 63) / 64)), dim3(256), lds_bytes, st, prm, tgt);
    } else if (algo == 1) {
        auto kern = mala_tile_kernel<T>;                 // always one wave per SIMD
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3((unsigned)((prm.C + 63) / 64)), dim3(256), lds_bytes, st, prm, tgt);
    } else if (algo == 2) {
        auto kern = nuts_tile_kernel<T, false>;          // one wave per SIMD: register-carried leaf state (nuts_tile.hpp)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3(prm.nuts_grid ? (unsigned)prm.nuts_grid : (unsigned)((prm.C + 63) / 64)), dim3(256), lds_bytes, st, prm, tgt);
    } else return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}

}  // namespace mi

#define MI_MCMC_DEFINE_TILE_TARGET(NAME, TARGET_T)                                                                                  \
    extern "C" int NAME##_run(int algo, const TARGET_T* target, uint64_t d, const mi_settings* settings, mi_chains* chains, void* stream) \
    {                                                                                                                               \
        return mi_mcmc_run_tile_target(algo, d, TARGET_T::NT, mi::tile_wpb<TARGET_T>(), (uint64_t)(target->lds_doubles() * sizeof(double)),  \
                                       &mi::tile_target_launch<TARGET_T>, target, (uint64_t)sizeof(mi::TileParams), MI_MCMC_VERSION,  \
                                       settings, chains, stream);                                                                   \
    }
