// tests/lit_host.hip -- TEST INFRASTRUCTURE: the HOST instantiation of the literal replay (mcmc_amd/csrc/literal.hpp: the same
// __host__ __device__ functions the GPU runs, one "thread" per chain here) behind a C entry point, so that the CPU test suite can
// hold the product's replay logic against the oracle on non-finite cases without a GPU.  Built by tests/lit_host.py with
//   hipcc -O2 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -shared
#include "../mcmc_amd/csrc/literal_host.hpp"

// what the extended entry point adds: per-chain diagonal masses (hmc), host callbacks as the target (kind = LIT_CALLBACK: the host
// instantiation of the mailbox simply calls them), the nuts dual-averaging state in / out
struct LitHostExtra {
    const double* mass_diag = nullptr;        // [d][C]
    mi::lit::lit_kernel_cb kernel = nullptr; void* kernel_data = nullptr;
    mi::lit::lit_tensor_cb tensor = nullptr; void* tensor_data = nullptr;
    double* adapt_state = nullptr;            // [3][C]
};

static int lit_host_run_impl(int algo, int kind, uint32_t d, uint32_t n_rows, const double* prec, const double* X, const double* y,
                             uint64_t C, uint64_t chain0, double* theta, double* draws, uint64_t* n_accept, uint64_t* n_leap,
                             uint64_t seed, uint32_t n_burnin, uint32_t n_keep, uint32_t n_leap_steps, uint32_t draw0, double eps,
                             int vals_bound, const double* lower, const double* upper, const double* precond_mat,
                             uint32_t n_adapt, uint32_t max_depth, double delta, double gamma, double t0, double kappa,
                             double* step_out, uint32_t* depth_trace, uint32_t n_fp_steps, const LitHostExtra& ex)
{
    using namespace mi::lit;
    LitPrep pr;
    if (algo == 2 && max_depth > (uint32_t)LIT_NUTS_MAX_DEPTH) return 1;
    lit_prepare(algo, d, eps, vals_bound, lower, upper, algo == 4 ? nullptr : precond_mat, pr);
    LitParams p{};
    std::vector<double> prec_t, Xt;
    if (kind == LIT_DENSE) lit_transpose(prec, d, d, prec_t);
    if (kind == LIT_LOGISTIC) lit_transpose(X, n_rows, d, Xt);
    p.t.kind = kind; p.t.d = d; p.t.n_rows = n_rows; p.t.prec = kind == LIT_DENSE ? prec_t.data() : prec; p.t.prec_stride = 1;
    p.t.X = X; p.t.Xt = Xt.empty() ? nullptr : Xt.data(); p.t.y = y;
    lit_orders(p.t);
    p.C = C; p.chain0 = chain0; p.theta = theta; p.draws = draws; p.n_accept = n_accept; p.n_leap = n_leap;
    p.seed = seed; p.n_burnin = n_burnin; p.n_keep = n_keep; p.n_leap_steps = n_leap_steps; p.draw0 = draw0; p.eps = eps;
    p.vals_bound = vals_bound; p.btype = pr.bt.data(); p.lb = pr.lb.data(); p.ub = pr.ub.data();
    p.precond = pr.precond;
    p.m = pr.m.data(); p.m_sqrt = pr.m_sqrt.data(); p.m_inv = pr.m_inv.data();
    p.Mfull = pr.Mfull.data(); p.Lchol = pr.Lchol.data(); p.Minv = pr.Minv.data();
    p.sinv_diag = pr.sinv_diag.empty() ? nullptr : pr.sinv_diag.data();
    p.Sinv = pr.Sinv.empty() ? nullptr : pr.Sinv.data();
    p.rs = pr.rs; p.log_det = pr.log_det; p.cons_term = pr.cons_term;
    p.n_adapt = n_adapt; p.max_depth = max_depth; p.delta = delta; p.gamma = gamma; p.t0 = t0; p.kappa = kappa;
    p.step_out = step_out; p.depth_trace = depth_trace; p.n_fp_steps = n_fp_steps;
    p.adapt_state = ex.adapt_state;
    std::vector<double> ms, mi_, mb_x, mb_out;
    double mb_value = 0.0;
    uint32_t mb_ctl[LIT_MB_WORDS] = {0};
    if (ex.mass_diag) {                                  // per-chain diagonal masses: the tables mi_mcmc.hip forms on the device
        ms.resize((size_t)d * C); mi_.resize((size_t)d * C);
        for (size_t e = 0; e < (size_t)d * C; ++e) { ms[e] = __builtin_sqrt(ex.mass_diag[e]); mi_[e] = 1.0 / ex.mass_diag[e]; }
        p.precond = 1; p.m = ex.mass_diag; p.m_sqrt = ms.data(); p.m_inv = mi_.data(); p.m_chain_stride = C;
    }
    if (kind == LIT_CALLBACK) {
        mb_x.resize(d); mb_out.resize(algo == 4 ? (size_t)d * d + (size_t)d * d * d : d);
        p.t.mb.ctl = mb_ctl; p.t.mb.value = &mb_value; p.t.mb.x = mb_x.data(); p.t.mb.out = mb_out.data();
        p.t.mb.kernel = ex.kernel; p.t.mb.kernel_data = ex.kernel_data; p.t.mb.tensor = ex.tensor; p.t.mb.tensor_data = ex.tensor_data;
    }
    if (algo == 4 && d > (uint32_t)LIT_RMHMC_MAX_D) return 1;
    std::vector<double> work(lit_work_doubles(d, n_rows, algo == 1 && vals_bound != 0, max_depth, algo == 2, algo == 4));
    const Par par{0, 1};
    for (uint64_t c = 0; c < C; ++c) {
        if (algo == 0) hmc_chain(par, p, c, work.data()); else if (algo == 1) mala_chain(par, p, c, work.data());
        else if (algo == 2) nuts_chain(par, p, c, work.data()); else if (algo == 3) rwmh_chain(par, p, c, work.data());
        else rmhmc_chain(par, p, c, work.data());
    }
    return 0;
}

extern "C" int lit_host_run(int algo, int kind, uint32_t d, uint32_t n_rows, const double* prec, const double* X, const double* y,
                            uint64_t C, uint64_t chain0, double* theta, double* draws, uint64_t* n_accept, uint64_t* n_leap,
                            uint64_t seed, uint32_t n_burnin, uint32_t n_keep, uint32_t n_leap_steps, uint32_t draw0, double eps,
                            int vals_bound, const double* lower, const double* upper, const double* precond_mat,
                            uint32_t n_adapt, uint32_t max_depth, double delta, double gamma, double t0, double kappa,
                            double* step_out, uint32_t* depth_trace, uint32_t n_fp_steps)
{
    return lit_host_run_impl(algo, kind, d, n_rows, prec, X, y, C, chain0, theta, draws, n_accept, n_leap, seed, n_burnin, n_keep, n_leap_steps,
                             draw0, eps, vals_bound, lower, upper, precond_mat, n_adapt, max_depth, delta, gamma, t0, kappa, step_out,
                             depth_trace, n_fp_steps, LitHostExtra{});
}

extern "C" int lit_host_run_ext(int algo, int kind, uint32_t d, uint32_t n_rows, const double* prec, const double* X, const double* y,
                                uint64_t C, uint64_t chain0, double* theta, double* draws, uint64_t* n_accept, uint64_t* n_leap,
                                uint64_t seed, uint32_t n_burnin, uint32_t n_keep, uint32_t n_leap_steps, uint32_t draw0, double eps,
                                int vals_bound, const double* lower, const double* upper, const double* precond_mat,
                                uint32_t n_adapt, uint32_t max_depth, double delta, double gamma, double t0, double kappa,
                                double* step_out, uint32_t* depth_trace, uint32_t n_fp_steps,
                                const double* mass_diag, void* kernel_cb, void* kernel_data, void* tensor_cb, void* tensor_data, double* adapt_state)
{
    LitHostExtra ex;
    ex.mass_diag = mass_diag; ex.kernel = reinterpret_cast<mi::lit::lit_kernel_cb>(kernel_cb); ex.kernel_data = kernel_data;
    ex.tensor = reinterpret_cast<mi::lit::lit_tensor_cb>(tensor_cb); ex.tensor_data = tensor_data; ex.adapt_state = adapt_state;
    return lit_host_run_impl(algo, kind, d, n_rows, prec, X, y, C, chain0, theta, draws, n_accept, n_leap, seed, n_burnin, n_keep, n_leap_steps,
                             draw0, eps, vals_bound, lower, upper, precond_mat, n_adapt, max_depth, delta, gamma, t0, kappa, step_out,
                             depth_trace, n_fp_steps, ex);
}
