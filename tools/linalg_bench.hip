// tools/linalg_bench.hip -- launch shapes of the device INV / CHOL_LOWER (mcmc_amd/csrc/linalg_device.hip): kernel time by grid size and by
// barrier (cooperative_groups grid.sync() against the arrival-counter barrier), results checked against each other bit for bit.
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Imcmc_amd/csrc -o /tmp/linalg_bench tools/linalg_bench.hip && /tmp/linalg_bench [d]
#include "../mcmc_amd/csrc/linalg_device.hip"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

namespace mi { namespace host {
std::string& last_error() { static std::string e; return e; }
std::string& last_kernel() { static std::string e; return e; }
} }

int main(int argc, char** argv)
{
    const size_t d = argc > 1 ? (size_t)atoi(argv[1]) : 512;
    std::mt19937_64 g(7);
    std::normal_distribution<double> nd;
    std::vector<double> G(d * d), M(d * d, 0.0), ref, out(d * d);
    for (auto& v : G) v = nd(g) / std::sqrt((double)d);
    for (size_t i = 0; i < d; ++i)
        for (size_t j = 0; j < d; ++j) {
            double s = (i == j) ? 1.0 : 0.0;
            for (size_t k = 0; k < d; ++k) s += G[i * d + k] * G[j * d + k];
            M[i * d + j] = s;
        }
    for (int what = 0; what < 2; ++what)
        for (int cg = 1; cg >= 0; --cg)
            for (uint32_t cap : {256u, 128u, 64u, 32u, 16u}) {
                mi::LaTiming tm;
                int rc = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    if (what == 0) rc = cg ? mi::inverse_impl<true>(M.data(), d, out.data(), cap, &tm) : mi::inverse_impl<false>(M.data(), d, out.data(), cap, &tm);
                    else rc = cg ? mi::cholesky_impl<true>(M.data(), d, out.data(), cap, &tm) : mi::cholesky_impl<false>(M.data(), d, out.data(), cap, &tm);
                }
                if (rc) { printf("rc %d: %s\n", rc, mi::host::last_error().c_str()); return 1; }
                if (cg == 1 && cap == 256u) ref = out;
                printf("%s d=%zu barrier=%s grid<=%u: kernel %.2f ms (%.1f us per step) same_bits=%d\n", what ? "CHOL_LOWER" : "INV", d, cg ? "cg::grid.sync" : "counter", cap,
                       tm.kernel_ms, tm.kernel_ms * 1e3 / d, (int)(std::memcmp(ref.data(), out.data(), d * d * 8) == 0));
            }
    return 0;
}
