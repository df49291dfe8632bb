// literal_launch.hip -- translation unit of the literal replay kernels (literal.hpp): mcmc::hmc / mcmc::mala, one workgroup per
// chain, the reference's dense operations as written.  Launched behind every throughput kernel of the plain paths (it returns at
// once unless a chain was flagged) and, on all chains, for bounded mala with a dense precond_mat.
#include "literal.hpp"
#include "launchers.hpp"
#include "launch_common.hpp"

namespace mi {

int launch_literal(int algo, const lit::LitParams& prm, unsigned n_wg, hipStream_t st)
{
    if (n_wg == 0) return 0;
    if (prm.flag == nullptr) note_kernel("literal_kernel<%d>", algo);     // the run itself, not a replay behind another kernel
    if (algo == 0) hipLaunchKernelGGL(lit::literal_kernel<0>, dim3(n_wg), dim3(256), 0, st, prm);
    else if (algo == 1) hipLaunchKernelGGL(lit::literal_kernel<1>, dim3(n_wg), dim3(256), 0, st, prm);
    else if (algo == 2) hipLaunchKernelGGL(lit::literal_kernel<2>, dim3(n_wg), dim3(256), 0, st, prm);
    else if (algo == 3) hipLaunchKernelGGL(lit::literal_kernel<3>, dim3(n_wg), dim3(256), 0, st, prm);
    else hipLaunchKernelGGL(lit::literal_kernel<4>, dim3(n_wg), dim3(256), 0, st, prm);
    return (int)hipGetLastError();
}

}  // namespace mi
