// examples/user_tile_target.hip -- user-defined targets on the TILED engine (include/mi_mcmc_tile_target.hpp): value + gradient for a
// wavefront's 16 chains at once, vectors in the MFMA register layout, dense mat-vecs on the fp64 matrix cores.
//
// GaussTile    the built-in dense Gaussian written as a user target (d <= 128): reproduces hmc_gauss_mfma_kernel bit for bit.
// TwistedTile  a non-Gaussian target, d <= 64: u = v except u_1 = v_1 + b (v_0^2 - s^2) (a "banana" twist of the first two
//              coordinates),  log K(v) = -1/2 u' P u,  grad = -J' (P u) with J = du/dv (identity plus dJ_10 = 2 b v_0).
//              twisted_host_kernel is the same arithmetic as a plain C callback (the reference's contract, ref:
//              include/mcmc/hmc.hpp:42-48): rows of P u as fma chains with k ascending, the quadratic form as four strided fma chains
//              combined (q0 + q2) + (q1 + q3) -- tests/ hand it to the oracle and compare draws bit for bit.
//
//   hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Iinclude -shared examples/user_tile_target.hip \
//         -Lmcmc_amd -lmi_mcmc -Wl,-rpath,$PWD/mcmc_amd -o libuser_tile_target.so
#include "mi_mcmc_tile_target.hpp"

struct GaussTile {
    static constexpr int NT = 8;
    static constexpr int WPB = 8;
    const double* P;        // device, d x d row-major
    uint32_t d;
    size_t lds_doubles() const { return mi::tile::matrix_doubles<NT>(); }
    __device__ void stage(double* lds) const { mi::tile::stage_matrix<NT>(P, d, lds); }
    __device__ void grad_tile(const double* lds, const double (&th)[4 * NT], double (&g)[4 * NT], double& value, bool want_value) const
    {
        double w[4 * NT];
        mi::tile::matvec<NT>(lds, th, w);
#pragma unroll
        for (int s = 0; s < 4 * NT; ++s) g[s] = -w[s];
        if (want_value) value = -0.5 * mi::tile::dot<4 * NT>(th, w);
    }
};
MI_MCMC_DEFINE_TILE_TARGET(gauss_tile, GaussTile)

struct TwistedTile {
    static constexpr int NT = 4;
    const double* P;        // device (kernels) / host (twisted_host_kernel), d x d row-major
    uint32_t d;             // >= 2
    double b, s2;
    size_t lds_doubles() const { return mi::tile::matrix_doubles<NT>(); }
    __device__ void stage(double* lds) const { mi::tile::stage_matrix<NT>(P, d, lds); }
    __device__ void grad_tile(const double* lds, const double (&v)[4 * NT], double (&g)[4 * NT], double& value, bool want_value) const
    {
        const int lane = threadIdx.x & 63;
        // dimension 0 sits in register 0 of the lanes with (lane >> 4) == 0, dimension 1 in register 0 of (lane >> 4) == 1
        const double v0 = __shfl(v[0], lane & 15);
        double u[4 * NT];
#pragma unroll
        for (int s = 0; s < 4 * NT; ++s) u[s] = v[s];
        if ((lane >> 4) == 1) u[0] = v[0] + b * (v0 * v0 - s2);
        double w[4 * NT];
        mi::tile::matvec<NT>(lds, u, w);
        if (want_value) value = -0.5 * mi::tile::dot<4 * NT>(u, w);
        const double w1 = __shfl(w[0], 16 + (lane & 15));
#pragma unroll
        for (int s = 0; s < 4 * NT; ++s) g[s] = -w[s];
        if ((lane >> 4) == 0) g[0] = -(w[0] + ((2.0 * b) * v0) * w1);
    }
};
MI_MCMC_DEFINE_TILE_TARGET(twisted_tile, TwistedTile)

// the same target as the reference's host callback (mi_log_kernel_cb); target_data points to a TwistedTile whose P is a HOST pointer
extern "C" double twisted_host_kernel(const double* vals, double* grad_out, void* target_data)
{
    const TwistedTile& t = *static_cast<const TwistedTile*>(target_data);
    const uint32_t d = t.d;
    double u[64], w[64];
    for (uint32_t i = 0; i < d; ++i) u[i] = vals[i];
    u[1] = vals[1] + t.b * (vals[0] * vals[0] - t.s2);
    for (uint32_t i = 0; i < d; ++i) {
        double acc = 0.0;
        for (uint32_t k = 0; k < d; ++k) acc = __builtin_fma(t.P[(size_t)i * d + k], u[k], acc);
        w[i] = acc;
    }
    double q[4] = {0.0, 0.0, 0.0, 0.0};
    for (uint32_t i = 0; i < d; ++i) q[i & 3] = __builtin_fma(u[i], w[i], q[i & 3]);
    const double value = -0.5 * ((q[0] + q[2]) + (q[1] + q[3]));
    if (grad_out) {
        for (uint32_t i = 0; i < d; ++i) grad_out[i] = -w[i];
        grad_out[0] = -(w[0] + ((2.0 * t.b) * vals[0]) * w[1]);
    }
    return value;
}
