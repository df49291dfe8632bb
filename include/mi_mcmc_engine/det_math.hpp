// det_math.hpp -- deterministic scalar math + per-chain counter-based RNG for the device path.
//
// Replaces the RNG / transcendental side of the BaseMatrixOps shim the reference samplers are
// written against (bmo::stats::runif, bmo::stats::internal::rnorm_vec_inplace over
// std::mt19937_64: /root/reference/src/hmc.cpp:156,189; src/mala.cpp:150,171;
// src/nuts.cpp:166,200,206,233,261; include/mcmc/nuts.ipp:214) by Philox4x32-10 keyed per chain,
// as BASELINE.json's north_star asks.  exp/log/sincos use only IEEE +,-,*,/,fma,sqrt,rint so the
// same inputs give the same bits on the host and on gfx950 (compile with -ffp-contract=off; every
// fused operation is an explicit fma).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define MI_HD __host__ __device__ __forceinline__

// Kernels whose MFMA operands come from LDS: keep every fragment read a ds_read_b64 (2 LDS cycles, 64 banks, 32-lane groups).
// The DS load/store merger would pair them into ds_read2_b64 / ds_read2st64_b64, which the LDS services as two 4 x 16-lane
// accesses over 32 banks (8 cycles, half the bandwidth, and conflicts for layouts built for the 64-bank mapping).
#if defined(__HIP_DEVICE_COMPILE__)
#define MI_NO_DS_MERGE __attribute__((target("no-load-store-opt")))
#else
#define MI_NO_DS_MERGE      // the host pass does not know the feature (and compiles no kernel body)
#endif

namespace mi {

// Polynomial coefficients as scalar operands (device code).  Left alone (MI_KC_MODE 0) the compiler hoists the 64-bit
// literals of the inlined exp / log / sincos polynomials out of the loops into dozens of long-lived VGPR pairs and spills
// some of them inside the hot loops.  The value is the same in every mode.
//   MI_KC_MODE 1 (default): each coefficient passes through an SGPR pair (volatile asm "+s"); the literal moves feeding them
//     are still hoisted, into SGPRs -- best for the RNG-heavy kernels, whose SGPR file has room.
//   MI_KC_MODE 2: each coefficient is materialised by two s_mov_b32 literals AT its point of use, ordered after the Horner
//     value that needs it: nothing lives across loops -- for kernels whose SGPR file is full (logistic_lds.hpp).
#ifndef MI_KC_MODE
#define MI_KC_MODE 1
#endif
#if defined(__HIP_DEVICE_COMPILE__) && MI_KC_MODE == 1
__device__ __forceinline__ double mi_kc(double c) { asm volatile("" : "+s"(c)); return c; }
#define MI_KC(c) ::mi::mi_kc(c)
#define MI_KCD(c, dep) ::mi::mi_kc(c)
#elif defined(__HIP_DEVICE_COMPILE__) && MI_KC_MODE == 2
template <uint64_t BITS>
__device__ __forceinline__ double mi_kc()
{
    uint32_t lo, hi;
    asm volatile("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3" : "=s"(lo), "=s"(hi) : "n"((uint32_t)BITS), "n"((uint32_t)(BITS >> 32)));
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | (uint64_t)lo);
}
template <uint64_t BITS>
__device__ __forceinline__ double mi_kcd(double dep)
{
    uint32_t lo, hi;
    asm volatile("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3" : "=s"(lo), "=s"(hi) : "n"((uint32_t)BITS), "n"((uint32_t)(BITS >> 32)), "v"(dep));
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | (uint64_t)lo);
}
#define MI_KC(c) (::mi::mi_kc<__builtin_bit_cast(uint64_t, (double)(c))>())
#define MI_KCD(c, dep) (::mi::mi_kcd<__builtin_bit_cast(uint64_t, (double)(c))>(dep))
#else
#define MI_KC(c) (c)
#define MI_KCD(c, dep) (c)
#endif


MI_HD uint64_t d2u(double x) { return __builtin_bit_cast(uint64_t, x); }
MI_HD double u2d(uint64_t u) { return __builtin_bit_cast(double, u); }
MI_HD double pow2i(int k) { return u2d((uint64_t)(k + 1023) << 52); }
MI_HD double dfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
MI_HD bool is_finite(double x) { return (d2u(x) & 0x7ff0000000000000ULL) != 0x7ff0000000000000ULL; }

constexpr double LN2_HI = 0x1.62e42fee00000p-1;
constexpr double LN2_LO = 0x1.a39ef35793c76p-33;
constexpr double INV_LN2 = 0x1.71547652b82fep+0;
constexpr double PI_4 = 0x1.921fb54442d18p-1;
constexpr double INF = __builtin_huge_val();

// exp: x = k ln2 + r, degree-14 Taylor polynomial of exp(r), two-step power-of-two scaling.
MI_HD double det_exp(double x)
{
    if (x != x) return x;
    if (x > 709.782712893384) return INF;
    if (x < -745.2) return 0.0;
    const double kf = __builtin_rint(x * INV_LN2);
    const int k = (int)kf;
    double r = dfma(-kf, LN2_HI, x);
    r = dfma(-kf, LN2_LO, r);
    double p = MI_KCD(1.0 / 87178291200.0, r);
    p = dfma(p, r, MI_KCD(1.0 / 6227020800.0, p));
    p = dfma(p, r, MI_KCD(1.0 / 479001600.0, p));
    p = dfma(p, r, MI_KCD(1.0 / 39916800.0, p));
    p = dfma(p, r, MI_KCD(1.0 / 3628800.0, p));
    p = dfma(p, r, MI_KCD(1.0 / 362880.0, p));
    p = dfma(p, r, MI_KCD(1.0 / 40320.0, p));
    p = dfma(p, r, MI_KCD(1.0 / 5040.0, p));
    p = dfma(p, r, MI_KCD(1.0 / 720.0, p));
    p = dfma(p, r, MI_KCD(1.0 / 120.0, p));
    p = dfma(p, r, MI_KCD(1.0 / 24.0, p));
    p = dfma(p, r, MI_KCD(1.0 / 6.0, p));
    p = dfma(p, r, 0.5);
    p = dfma(p, r, 1.0);
    p = dfma(p, r, 1.0);
    const int k1 = k / 2, k2 = k - k1;
    return (p * pow2i(k1)) * pow2i(k2);
}

// log: x = 2^e m, m in [sqrt(1/2), sqrt(2)); s = (m-1)/(m+1); log m = 2 s P(s^2).
MI_HD double det_log(double x)
{
    if (x != x) return x;
    if (x < 0.0) return __builtin_nan("");
    if (x == 0.0) return -INF;
    if (x == INF) return x;
    int e = 0;
    uint64_t u = d2u(x);
    if ((u >> 52) == 0) { x = x * 0x1p54; u = d2u(x); e = -54; }
    e += (int)(u >> 52) - 1023;
    u = (u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = u2d(u);
    if (m > 0x1.6a09e667f3bcdp+0) { m = m * 0.5; e += 1; }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    double p = MI_KCD(1.0 / 23.0, z);
    p = dfma(p, z, MI_KCD(1.0 / 21.0, p));
    p = dfma(p, z, MI_KCD(1.0 / 19.0, p));
    p = dfma(p, z, MI_KCD(1.0 / 17.0, p));
    p = dfma(p, z, MI_KCD(1.0 / 15.0, p));
    p = dfma(p, z, MI_KCD(1.0 / 13.0, p));
    p = dfma(p, z, MI_KCD(1.0 / 11.0, p));
    p = dfma(p, z, MI_KCD(1.0 / 9.0, p));
    p = dfma(p, z, MI_KCD(1.0 / 7.0, p));
    p = dfma(p, z, MI_KCD(1.0 / 5.0, p));
    p = dfma(p, z, MI_KCD(1.0 / 3.0, p));
    p = dfma(p, z, 1.0);
    const double lm = (2.0 * s) * p;
    const double ef = (double)e;
    return dfma(ef, LN2_HI, dfma(ef, LN2_LO, lm));
}

MI_HD double det_pow(double x, double y) { return det_exp(y * det_log(x)); }

MI_HD void sincos_kernel(double a, double& s, double& c)
{
    const double z = a * a;
    double ps = MI_KCD(-1.0 / 121645100408832000.0, z);
    ps = dfma(ps, z, MI_KCD(1.0 / 355687428096000.0, ps));
    ps = dfma(ps, z, MI_KCD(-1.0 / 1307674368000.0, ps));
    ps = dfma(ps, z, MI_KCD(1.0 / 6227020800.0, ps));
    ps = dfma(ps, z, MI_KCD(-1.0 / 39916800.0, ps));
    ps = dfma(ps, z, MI_KCD(1.0 / 362880.0, ps));
    ps = dfma(ps, z, MI_KCD(-1.0 / 5040.0, ps));
    ps = dfma(ps, z, MI_KCD(1.0 / 120.0, ps));
    ps = dfma(ps, z, MI_KCD(-1.0 / 6.0, ps));
    ps = dfma(ps, z, 1.0);
    s = a * ps;
    double pc = MI_KCD(1.0 / 6402373705728000.0, z);
    pc = dfma(pc, z, MI_KCD(-1.0 / 20922789888000.0, pc));
    pc = dfma(pc, z, MI_KCD(1.0 / 87178291200.0, pc));
    pc = dfma(pc, z, MI_KCD(-1.0 / 479001600.0, pc));
    pc = dfma(pc, z, MI_KCD(1.0 / 3628800.0, pc));
    pc = dfma(pc, z, MI_KCD(-1.0 / 40320.0, pc));
    pc = dfma(pc, z, MI_KCD(1.0 / 720.0, pc));
    pc = dfma(pc, z, MI_KCD(-1.0 / 24.0, pc));
    pc = dfma(pc, z, 0.5);
    c = dfma(-pc, z, 1.0);
}

// sin, cos of 2 pi u for u in [0,1): octant reduction, odd octants reflected.
MI_HD void det_sincos2pi(double u, double& sn, double& cs)
{
    const double v = u * 8.0;
    const double qf = __builtin_floor(v);
    const int q = (int)qf & 7;
    double t = v - qf;
    if (q & 1) t = 1.0 - t;
    double s, c;
    sincos_kernel(t * PI_4, s, c);
    // octant table written branch-free: swap for q in {1,2,5,6}, signs by quadrant
    const bool swap = ((q + 1) & 2) != 0;
    const double cc = swap ? s : c;
    const double ss = swap ? c : s;
    const bool neg_c = (q >= 2) && (q <= 5);
    const bool neg_s = (q >= 4);
    cs = neg_c ? -cc : cc;
    sn = neg_s ? -ss : ss;
}

MI_HD double softplus(double eta)
{
    if (eta > 0.0) return eta + det_log(1.0 + det_exp(-eta));
    return det_log(1.0 + det_exp(eta));
}
MI_HD double sigmoid(double eta)
{
    if (eta >= 0.0) return 1.0 / (1.0 + det_exp(-eta));
    const double e = det_exp(eta);
    return e / (1.0 + e);
}

// ------------------------------------------------------------------ Philox4x32-10
struct u32x4 { uint32_t x, y, z, w; };

MI_HD uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

MI_HD u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 product per multiplier (a single v_mad_u64_u32 on the device) instead of a mul_hi and a mul_lo
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
        c = u32x4{(uint32_t)(p1 >> 32) ^ c.y ^ k0, (uint32_t)p1, (uint32_t)(p0 >> 32) ^ c.w ^ k1, (uint32_t)p0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// 52 random bits -> (2k+1) 2^-53 in (0,1), exact
MI_HD double u01(uint32_t lo, uint32_t hi)
{
    const uint64_t k = (((uint64_t)hi << 32) | lo) >> 12;
    return (double)(2 * k + 1) * 0x1p-53;
}

enum : uint32_t { STREAM_NORMAL = 0u, STREAM_UNIFORM = 1u, STREAM_INIT = 2u };

// counter = (chain lo32, draw, slot, stream | chain hi bits << 8), key = seed
MI_HD u32x4 rng_block(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t slot, uint32_t stream)
{
    const u32x4 c{(uint32_t)chain, draw, slot, stream | ((uint32_t)(chain >> 32) << 8)};
    return philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}

MI_HD double rng_uniform(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t slot)
{
    const u32x4 w = rng_block(seed, chain, draw, slot, STREAM_UNIFORM);
    return u01(w.x, w.y);
}

// Box-Muller pair of one slot: z0 = r cos(2 pi u2), z1 = r sin(2 pi u2), r = sqrt(-2 log u1)
MI_HD void rng_normal_pair_core(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t slot, uint32_t stream, double& z0, double& z1)
{
    const u32x4 w = rng_block(seed, chain, draw, slot, stream);
    const double u1 = u01(w.x, w.y);
    const double u2 = u01(w.z, w.w);
    const double r = __builtin_sqrt(-2.0 * det_log(u1));
    double s, c;
    det_sincos2pi(u2, s, c);
    z0 = r * c;
    z1 = r * s;
}
#ifdef MI_RNG_NOINLINE
__attribute__((noinline))
#endif
MI_HD void rng_normal_pair(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t slot, uint32_t stream,
                           double& z0, double& z1)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // keep the slot opaque: in the fully unrolled per-draw loops the compiler otherwise hoists the slot-only part of the first
    // Philox round of every slot out of the draw loop (3 registers per slot), spills it, and reloads it from scratch per draw
    asm volatile("" : "+v"(slot));
#endif
    rng_normal_pair_core(seed, chain, draw, slot, stream, z0, z1);
}

// Two slots in one out-of-line call, results by value (four registers): the two Philox / log / sincos chains are independent, so
// the scheduler interleaves them -- a Box-Muller pair is ~250 dependent operations, and a wave that walks them one chain at a time
// waits out every result latency (the logistic kernels: 16 pairs per lane and draw with the matrix pipe idle).
typedef double rng_double4 __attribute__((ext_vector_type(4)));
__device__ __attribute__((noinline)) inline rng_double4 rng_normal_two_pairs(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t slot_a,
                                                                             uint32_t slot_b, uint32_t stream)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(slot_a), "+v"(slot_b));
#endif
    double a0, a1, b0, b1;
    rng_normal_pair_core(seed, chain, draw, slot_a, stream, a0, a1);
    rng_normal_pair_core(seed, chain, draw, slot_b, stream, b0, b1);
    return rng_double4{a0, a1, b0, b1};
}
typedef double rng_double8 __attribute__((ext_vector_type(8)));
__device__ __attribute__((noinline)) inline rng_double8 rng_normal_four_pairs(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t slot_a,
                                                                              uint32_t slot_step, uint32_t stream)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(slot_a));
#endif
    double z[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) rng_normal_pair_core(seed, chain, draw, slot_a + (uint32_t)k * slot_step, stream, z[2 * k], z[2 * k + 1]);
    return rng_double8{z[0], z[1], z[2], z[3], z[4], z[5], z[6], z[7]};
}

// The same with slot = base + j, j the lane's part of the slot (lane-varying, fixed for the whole kernel) made opaque BEFORE the
// sum: otherwise `base + j` of every unrolled call is a loop invariant of the draw loop -- one VGPR per slot, spilled, and
// reloaded from scratch in front of every pair (a reload waits for every store in flight: vmcnt is in order).
MI_HD void rng_normal_pair_at(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t base, uint32_t j, uint32_t stream,
                              double& z0, double& z1)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(j));
#endif
    rng_normal_pair(seed, chain, draw, base + j, stream, z0, z1);
}

// Canonical dimension <-> slot map: i = 8b + 4h + j (j<4, h<2) -> slot 4b + j, component h.
// A lane of the MFMA layout owns dims {4s + j}: both halves of a pair stay in the lane.

}  // namespace mi
