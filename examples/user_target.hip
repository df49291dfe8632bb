// examples/user_target.hip -- a user-defined device target (include/mi_mcmc_target.hpp): the reference's callback contract
// (ref: include/mcmc/hmc.hpp:42-48) as a __device__ functor, compiled into its own small library next to libmi_mcmc.so.
//
// Target: a three-dimensional "twisted Gaussian" (banana), not one of the engine's built-in kinds:
//     u = (v0,  v1 + b (v0^2 - s^2),  v2),   log K(v) = -1/2 [ u0^2 / s^2 + u1^2 + c u2^2 ] - 1/2 rho u0 u2
// with the analytic gradient.  kernel() uses only IEEE + - * and explicit fma, so host and device give the same bits: the same
// member function is exported as a plain C callback (banana_host_kernel) with which tests/ drive the CPU oracle.
//
//   hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Iinclude -shared examples/user_target.hip \
//         -Lmcmc_amd -lmi_mcmc -Wl,-rpath,$PWD/mcmc_amd -o libuser_target.so
#include "mi_mcmc_target.hpp"

struct Banana {
    static constexpr int D = 3;
    double s2, b, c, rho;

    __host__ __device__ double kernel(const double (&v)[3], double (&g)[3], bool want_grad) const
    {
        const double t = v[0] * v[0] - s2;
        const double u1 = v[1] + b * t;
        const double q = (v[0] * v[0]) / s2 + u1 * u1 + c * (v[2] * v[2]);
        const double val = -0.5 * q - 0.5 * (rho * (v[0] * v[2]));
        if (want_grad) {
            g[0] = -(v[0] / s2) - (u1 * ((2.0 * b) * v[0])) - 0.5 * (rho * v[2]);
            g[1] = -u1;
            g[2] = -(c * v[2]) - 0.5 * (rho * v[0]);
        }
        return val;
    }
};

MI_MCMC_DEFINE_TARGET(banana, Banana)

// the same function as the reference's host callback (mi_log_kernel_cb): target_data points to a Banana
extern "C" double banana_host_kernel(const double* vals, double* grad_out, void* target_data)
{
    const Banana& t = *static_cast<const Banana*>(target_data);
    const double v[3] = {vals[0], vals[1], vals[2]};
    double g[3] = {0.0, 0.0, 0.0};
    const double r = t.kernel(v, g, grad_out != nullptr);
    if (grad_out) { grad_out[0] = g[0]; grad_out[1] = g[1]; grad_out[2] = g[2]; }
    return r;
}
