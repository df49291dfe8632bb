// tests/lit_host.hip -- TEST INFRASTRUCTURE: the HOST instantiation of the literal replay (mcmc_amd/csrc/literal.hpp: the same
// __host__ __device__ functions the GPU runs, one "thread" per chain here) behind a C entry point, so that the CPU test suite can
// hold the product's replay logic against the oracle on non-finite cases without a GPU.  Built by tests/lit_host.py with
//   hipcc -O2 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -shared
#include "../mcmc_amd/csrc/literal_host.hpp"

extern "C" int lit_host_run(int algo, int kind, uint32_t d, uint32_t n_rows, const double* prec, const double* X, const double* y,
                            uint64_t C, uint64_t chain0, double* theta, double* draws, uint64_t* n_accept, uint64_t* n_leap,
                            uint64_t seed, uint32_t n_burnin, uint32_t n_keep, uint32_t n_leap_steps, uint32_t draw0, double eps,
                            int vals_bound, const double* lower, const double* upper, const double* precond_mat,
                            uint32_t n_adapt, uint32_t max_depth, double delta, double gamma, double t0, double kappa,
                            double* step_out, uint32_t* depth_trace, uint32_t n_fp_steps)
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
