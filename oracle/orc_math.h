/*
 * oracle/orc_math.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Deterministic scalar math + counter-based RNG used by the CPU oracle.
 *
 * The reference (kthohr/mcmc) takes every random number from BaseMatrixOps
 * (bmo::stats::runif / rnorm_vec_inplace over std::mt19937_64;
 * /root/reference/src/hmc.cpp:156,189) -- an un-vendored submodule whose
 * arithmetic is not recoverable here.  BASELINE.json's north_star replaces it
 * by a per-chain counter-based generator ("Philox/xoshiro per-chain RNG") and
 * asks for bit-exact accept decisions for identical streams.  To make that
 * checkable the oracle and the HIP engine both use
 *   - Philox4x32-10 (Salmon et al., SC'11; Random123 v1.x reference KATs are
 *     checked in tests/test_oracle_math.py),
 *   - exp / log / sincos built ONLY from IEEE-754 +,-,*,/,fma,sqrt,rint so that
 *     glibc and ROCm ocml cannot disagree in the last bit.
 * Build with -ffp-contract=off: every fused operation below is an explicit fma().
 *
 * This file is an independent statement of the algorithms in
 * mcmc_amd/csrc/det_math.hpp; the two are compared bit-for-bit by the GPU tests.
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

/* ---------------------------------------------------------------- bit casts */
static inline uint64_t orc_d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double   orc_u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

/* 2^k for -1022 <= k <= 1023 */
static inline double orc_pow2i(int k) { return orc_u2d((uint64_t)(k + 1023) << 52); }

#define ORC_LN2_HI   0x1.62e42fee00000p-1   /* ln2 rounded to 33 bits */
#define ORC_LN2_LO   0x1.a39ef35793c76p-33  /* ln2 - LN2_HI */
#define ORC_INV_LN2  0x1.71547652b82fep+0
#define ORC_PI_4     0x1.921fb54442d18p-1

/* ---------------------------------------------------------------- exp
 * x = k ln2 + r, |r| <= ln2/2; exp(r) by its degree-14 Taylor polynomial in
 * Horner form (truncation < 2^-57); result scaled by 2^k in two exact-or-single-
 * rounding steps.  ~1 ulp. */
static inline double orc_exp(double x)
{
    if (x != x) return x;
    if (x > 709.782712893384) return INFINITY;
    if (x < -745.2) return 0.0;
    const double kf = rint(x * ORC_INV_LN2);
    const int k = (int)kf;
    double r = fma(-kf, ORC_LN2_HI, x);
    r = fma(-kf, ORC_LN2_LO, r);
    double p = 1.0 / 87178291200.0;            /* 1/14! */
    p = fma(p, r, 1.0 / 6227020800.0);
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const int k1 = k / 2, k2 = k - k1;         /* |k| <= 1075 -> both in range */
    return (p * orc_pow2i(k1)) * orc_pow2i(k2);
}

/* ---------------------------------------------------------------- log
 * x = 2^e m, m in [sqrt(1/2), sqrt(2)); s = (m-1)/(m+1);
 * log m = 2 s (1 + z/3 + z^2/5 + ... + z^11/23), z = s^2 (truncation < 2^-58). */
static inline double orc_log(double x)
{
    if (x != x) return x;
    if (x < 0.0) return NAN;
    if (x == 0.0) return -INFINITY;
    if (x == INFINITY) return x;
    int e = 0;
    uint64_t u = orc_d2u(x);
    if ((u >> 52) == 0) { x = x * 0x1p54; u = orc_d2u(x); e = -54; }   /* subnormal */
    e += (int)(u >> 52) - 1023;
    u = (u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = orc_u2d(u);
    if (m > 0x1.6a09e667f3bcdp+0) { m = m * 0.5; e += 1; }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    double p = 1.0 / 23.0;
    p = fma(p, z, 1.0 / 21.0);
    p = fma(p, z, 1.0 / 19.0);
    p = fma(p, z, 1.0 / 17.0);
    p = fma(p, z, 1.0 / 15.0);
    p = fma(p, z, 1.0 / 13.0);
    p = fma(p, z, 1.0 / 11.0);
    p = fma(p, z, 1.0 / 9.0);
    p = fma(p, z, 1.0 / 7.0);
    p = fma(p, z, 1.0 / 5.0);
    p = fma(p, z, 1.0 / 3.0);
    p = fma(p, z, 1.0);
    const double lm = (2.0 * s) * p;
    const double ef = (double)e;
    return fma(ef, ORC_LN2_HI, fma(ef, ORC_LN2_LO, lm));
}

/* pow restated through exp/log (only used by the NUTS dual-averaging schedule,
 * /root/reference/src/nuts.cpp:299 and nuts.ipp:74). */
static inline double orc_pow(double x, double y) { return orc_exp(y * orc_log(x)); }

/* ---------------------------------------------------------------- sincos(2 pi u), u in [0,1)
 * octant q = floor(8u), t = 8u - q; odd octants are reflected (t -> 1-t); the
 * reduced angle a = t*pi/4 in [0, pi/4] goes through Taylor polynomials. */
static inline void orc_sincos_kernel(double a, double* s, double* c)
{
    const double z = a * a;
    double ps = -1.0 / 121645100408832000.0;   /* -1/19! */
    ps = fma(ps, z, 1.0 / 355687428096000.0);  /* 1/17! */
    ps = fma(ps, z, -1.0 / 1307674368000.0);   /* -1/15! */
    ps = fma(ps, z, 1.0 / 6227020800.0);       /* 1/13! */
    ps = fma(ps, z, -1.0 / 39916800.0);        /* -1/11! */
    ps = fma(ps, z, 1.0 / 362880.0);           /* 1/9! */
    ps = fma(ps, z, -1.0 / 5040.0);            /* -1/7! */
    ps = fma(ps, z, 1.0 / 120.0);              /* 1/5! */
    ps = fma(ps, z, -1.0 / 6.0);               /* -1/3! */
    ps = fma(ps, z, 1.0);
    *s = a * ps;
    double pc = 1.0 / 6402373705728000.0;      /* 1/18! */
    pc = fma(pc, z, -1.0 / 20922789888000.0);  /* -1/16! */
    pc = fma(pc, z, 1.0 / 87178291200.0);      /* 1/14! */
    pc = fma(pc, z, -1.0 / 479001600.0);       /* -1/12! */
    pc = fma(pc, z, 1.0 / 3628800.0);          /* 1/10! */
    pc = fma(pc, z, -1.0 / 40320.0);           /* -1/8! */
    pc = fma(pc, z, 1.0 / 720.0);              /* 1/6! */
    pc = fma(pc, z, -1.0 / 24.0);              /* -1/4! */
    pc = fma(pc, z, 0.5);
    *c = fma(-pc, z, 1.0);
}

static inline void orc_sincos2pi(double u, double* sn, double* cs)
{
    const double v = u * 8.0;
    const double qf = floor(v);
    const int q = (int)qf & 7;
    double t = v - qf;
    if (q & 1) t = 1.0 - t;
    double s, c;
    orc_sincos_kernel(t * ORC_PI_4, &s, &c);
    switch (q) {
    case 0: *cs =  c; *sn =  s; break;
    case 1: *cs =  s; *sn =  c; break;
    case 2: *cs = -s; *sn =  c; break;
    case 3: *cs = -c; *sn =  s; break;
    case 4: *cs = -c; *sn = -s; break;
    case 5: *cs = -s; *sn = -c; break;
    case 6: *cs =  s; *sn = -c; break;
    default: *cs = c; *sn = -s; break;
    }
}

/* ---------------------------------------------------------------- Philox4x32-10 */
static inline void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* 52 random bits -> (2k+1) 2^-53, strictly inside (0,1), exact in binary64 */
static inline double orc_u01(uint32_t lo, uint32_t hi)
{
    const uint64_t k = (((uint64_t)hi << 32) | lo) >> 12;
    return (double)(2 * k + 1) * 0x1p-53;
}

/* stream ("purpose") tags: 4th counter word */
#define ORC_STREAM_NORMAL   0u   /* momentum / proposal normals, slot = pair index          */
#define ORC_STREAM_UNIFORM  1u   /* uniforms, slot = running index inside the draw            */
#define ORC_STREAM_INIT     2u   /* NUTS: the extra normal vector of src/nuts.cpp:166         */

/* counter = (chain lo32, draw, slot, stream | chain hi bits<<8), key = seed */
static inline void orc_rng_block(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t slot,
                                 uint32_t stream, uint32_t out[4])
{
    const uint32_t ctr[4] = { (uint32_t)chain, draw, slot, stream | ((uint32_t)(chain >> 32) << 8) };
    const uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    orc_philox4x32(ctr, key, out);
}

static inline double orc_rng_uniform(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t slot)
{
    uint32_t w[4];
    orc_rng_block(seed, chain, draw, slot, ORC_STREAM_UNIFORM, w);
    return orc_u01(w[0], w[1]);
}

/* Box-Muller pair for one slot: z0 = r cos, z1 = r sin */
static inline void orc_rng_normal_pair(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t slot,
                                       uint32_t stream, double* z0, double* z1)
{
    uint32_t w[4];
    orc_rng_block(seed, chain, draw, slot, stream, w);
    const double u1 = orc_u01(w[0], w[1]);
    const double u2 = orc_u01(w[2], w[3]);
    const double r = sqrt(-2.0 * orc_log(u1));
    double s, c;
    orc_sincos2pi(u2, &s, &c);
    *z0 = r * c;
    *z1 = r * s;
}

/* Canonical dimension <-> slot map (chosen so that the MFMA-layout kernels, where a
 * lane owns dims {4s + j}, get both halves of a Box-Muller pair in one lane):
 *   i = 8b + 4h + j (j<4, h<2)  ->  slot = 4b + j, component h. */
static inline void orc_rng_normal_vec(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t stream,
                                      size_t d, double* out)
{
    const size_t nslots = 4 * ((d + 7) / 8);
    for (size_t slot = 0; slot < nslots; ++slot) {
        const size_t b = slot / 4, j = slot % 4;
        const size_t i0 = 8 * b + j, i1 = i0 + 4;
        if (i0 >= d) continue;
        double z0, z1;
        orc_rng_normal_pair(seed, chain, draw, (uint32_t)slot, stream, &z0, &z1);
        out[i0] = z0;
        if (i1 < d) out[i1] = z1;
    }
}

#endif /* ORC_MATH_H */
