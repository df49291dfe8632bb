// examples/hmc_plumbing.cpp -- the call pattern of the reference examples
// (/root/reference/examples/eigen/hmc_normal.cpp:83-118: settings, mcmc::hmc(initial_val, target, draws_out, &data, settings),
// column means, acceptance rate) against this repository's mcmc.hpp.  BASELINE config[0]: 3-D isotropic Gaussian, 1 chain,
// host std::function target; then the device-target route with many chains for hmc / mala / nuts; then mcmc::mala and
// mcmc::nuts with the host std::function target (examples/eigen/mala_normal.cpp:108, nuts_normal.cpp:107).
//
//   g++ -std=c++17 -O2 -Iinclude examples/hmc_plumbing.cpp -Lmcmc_amd -lmi_mcmc -Wl,-rpath,$PWD/mcmc_amd -o hmc_plumbing
#define MCMC_ENABLE_EIGEN_WRAPPERS
#include "mcmc.hpp"

#include <cmath>
#include <cstdio>
#include <limits>
#include <vector>

struct iso_data_t { int n_calls_grad = 0, n_calls_value = 0; };

// log K(theta) = -1/2 |theta|^2 ; grad = -theta   (user code: runs on the host)
static double log_target_dens(const mcmc::ColVec_t& vals_inp, mcmc::ColVec_t* grad_out, void* ll_data)
{
    iso_data_t* dta = reinterpret_cast<iso_data_t*>(ll_data);
    double ss = 0.0;
    for (size_t i = 0; i < size_t(vals_inp.size()); ++i) ss += vals_inp(i) * vals_inp(i);
    if (grad_out) {
        grad_out->resize(vals_inp.size(), 1);
        for (size_t i = 0; i < size_t(vals_inp.size()); ++i) (*grad_out)(i, 0) = -vals_inp(i);
        dta->n_calls_grad++;
    } else {
        dta->n_calls_value++;
    }
    return -0.5 * ss;
}

// the model of the reference's rmhmc example as HOST callbacks: log-likelihood of N(mu, sigma^2) data with its gradient, and the Fisher
// information diag(n / sigma^2, 2 n / sigma^2) with its derivatives (user code: runs on the host)
struct norm_data_t { const double* x; size_t n; int n_grad, n_value, n_tensor; };
static double normal_ll(const mcmc::ColVec_t& v, mcmc::ColVec_t* grad_out, void* data)
{
    norm_data_t* dta = reinterpret_cast<norm_data_t*>(data);
    const double mu = v(0), sigma = v(1), n = double(dta->n);
    double s1 = 0.0, s2 = 0.0;
    for (size_t i = 0; i < dta->n; ++i) { const double e = dta->x[i] - mu; s1 += e; s2 += e * e; }
    if (grad_out) {
        grad_out->resize(2, 1);
        (*grad_out)(0, 0) = s1 / (sigma * sigma);
        (*grad_out)(1, 0) = -n / sigma + s2 / (sigma * sigma * sigma);
        dta->n_grad++;
    } else dta->n_value++;
    return -0.5 * n * std::log(2.0 * 3.14159265358979323846) - n * std::log(sigma) - s2 / (2.0 * sigma * sigma);
}
static mcmc::Mat_t normal_tensor(const mcmc::ColVec_t& v, mcmc::Cube_t* deriv_out, void* data)
{
    norm_data_t* dta = reinterpret_cast<norm_data_t*>(data);
    const double sigma = v(1), n = double(dta->n);
    dta->n_tensor++;
    mcmc::Mat_t G(2, 2);
    G.setZero();
    G(0, 0) = n / (sigma * sigma); G(1, 1) = 2.0 * n / (sigma * sigma);
    if (deriv_out) {
        deriv_out->setZero(2, 2, 2);                                   // mat(0) = dG/dmu = 0
        deriv_out->mat(1)(0, 0) = -2.0 * n / (sigma * sigma * sigma);
        deriv_out->mat(1)(1, 1) = -4.0 * n / (sigma * sigma * sigma);
    }
    return G;
}

static double col_mean(const mcmc::Mat_t& m, size_t j)
{
    double s = 0.0;
    for (size_t i = 0; i < size_t(m.rows()); ++i) s += m(i, j);
    return s / double(m.rows());
}

int main()
{
    // ---- host-callback route (the reference contract), one chain
    iso_data_t dta;
    mcmc::ColVec_t initial_val(3);
    initial_val(0) = 1.0; initial_val(1) = 1.0; initial_val(2) = 1.0;

    mcmc::algo_settings_t settings;
    settings.rng_seed_value = 1234;
    settings.hmc_settings.step_size = 0.2;
    settings.hmc_settings.n_leap_steps = 10;
    settings.hmc_settings.n_burnin_draws = 1000;
    settings.hmc_settings.n_keep_draws = 1000;

    mcmc::Mat_t draws_out;
    const bool ok = mcmc::hmc(initial_val, log_target_dens, draws_out, &dta, settings);
    std::printf("callback ok=%d rows=%zu cols=%zu mean=%.6f %.6f %.6f acc=%.4f grad_calls=%d value_calls=%d\n",
                int(ok), size_t(draws_out.rows()), size_t(draws_out.cols()),
                col_mean(draws_out, 0), col_mean(draws_out, 1), col_mean(draws_out, 2),
                double(settings.hmc_settings.n_accept_draws) / double(settings.hmc_settings.n_keep_draws),
                dta.n_calls_grad, dta.n_calls_value);

    // ---- device-target route: d = 16 dense Gaussian, 64 chains, hmc / mala / nuts
    const size_t d = 16, C = 64;
    std::vector<double> P(d * d, 0.0);
    for (size_t i = 0; i < d; ++i) { P[i * d + i] = 2.0; if (i + 1 < d) { P[i * d + i + 1] = -0.5; P[(i + 1) * d + i] = -0.5; } }
    mcmc::mi355x::target_t tgt = mcmc::mi355x::gaussian_dense(d, P.data());
    tgt.n_chains = C;
    mcmc::ColVec_t init(d);
    for (size_t i = 0; i < d; ++i) init(i) = 0.1 * double(i);

    mcmc::algo_settings_t s2;
    s2.rng_seed_value = 7;
    s2.hmc_settings.step_size = 0.1;  s2.hmc_settings.n_leap_steps = 8;
    s2.hmc_settings.n_burnin_draws = 50; s2.hmc_settings.n_keep_draws = 50;
    s2.mala_settings.step_size = 0.3;
    s2.mala_settings.n_burnin_draws = 50; s2.mala_settings.n_keep_draws = 50;
    s2.nuts_settings.n_burnin_draws = 50; s2.nuts_settings.n_keep_draws = 50; s2.nuts_settings.n_adapt_draws = 50;
    s2.rwmh_settings.par_scale = 0.3;
    s2.rwmh_settings.n_burnin_draws = 50; s2.rwmh_settings.n_keep_draws = 50;

    mcmc::Mat_t dr;
    bool ok2 = mcmc::hmc(init, mcmc::mi355x::device_kernel, dr, &tgt, s2);
    std::printf("device hmc ok=%d rows=%zu cols=%zu acc0=%.3f\n", int(ok2), size_t(dr.rows()), size_t(dr.cols()),
                double(s2.hmc_settings.n_accept_draws) / 50.0);
    ok2 = mcmc::mala(init, mcmc::mi355x::device_kernel, dr, &tgt, s2);
    std::printf("device mala ok=%d rows=%zu cols=%zu acc0=%.3f\n", int(ok2), size_t(dr.rows()), size_t(dr.cols()),
                double(s2.mala_settings.n_accept_draws) / 50.0);
    ok2 = mcmc::nuts(init, mcmc::mi355x::device_kernel, dr, &tgt, s2);
    std::printf("device nuts ok=%d rows=%zu cols=%zu acc0=%.3f eps0=%.4f\n", int(ok2), size_t(dr.rows()), size_t(dr.cols()),
                double(s2.nuts_settings.n_accept_draws) / 50.0, tgt.step_size[0]);

    bool ok3 = mcmc::rwmh(init, mcmc::mi355x::device_value_kernel, dr, &tgt, s2);
    std::printf("device rwmh ok=%d rows=%zu cols=%zu acc0=%.3f\n", int(ok3), size_t(dr.rows()), size_t(dr.cols()),
                double(s2.rwmh_settings.n_accept_draws) / 50.0);
    if (!ok3) return 1;

    // mcmc::rmhmc on the reference's own example model (examples/eigen/rmhmc_normal.cpp): (mu, sigma) of normal data,
    // Fisher-information metric; 64 chains
    std::vector<double> x_obs(1000);
    { unsigned long long st = 88172645463325252ULL;      // xorshift -> Irwin-Hall(12) - 6: near-normal observations
      for (auto& v : x_obs) { double a = 0; for (int k = 0; k < 12; ++k) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; a += double(st >> 11) / 9007199254740992.0; } v = 2.0 + 2.0 * (a - 6.0); } }
    mcmc::mi355x::target_t nm = mcmc::mi355x::normal_model(x_obs.size(), x_obs.data());
    nm.n_chains = 64;
    mcmc::ColVec_t init2(2); init2(0) = 3.0; init2(1) = 3.0;
    mcmc::algo_settings_t s4;
    s4.rng_seed_value = 9;
    s4.rmhmc_settings.step_size = 0.02; s4.rmhmc_settings.n_burnin_draws = 200; s4.rmhmc_settings.n_keep_draws = 200;
    const bool ok4 = mcmc::rmhmc(init2, mcmc::mi355x::device_kernel, mcmc::mi355x::device_tensor, dr, &nm, &nm, s4);
    std::printf("device rmhmc ok=%d rows=%zu cols=%zu acc0=%.3f mu0=%.3f sigma0=%.3f\n", int(ok4), size_t(dr.rows()), size_t(dr.cols()),
                double(s4.rmhmc_settings.n_accept_draws) / 200.0, ok4 ? dr.col_mean(0) : 0.0, ok4 ? dr.col_mean(1) : 0.0);
    if (!ok4) return 1;

    // ---- mcmc::mala and mcmc::nuts with the SAME host std::function target: the call pattern of the reference's
    //      examples/eigen/mala_normal.cpp:108 and nuts_normal.cpp:107 (settings, sampler call, column means, acceptance rate)
    mcmc::algo_settings_t s5;
    s5.rng_seed_value = 1234;
    s5.mala_settings.step_size = 0.8;
    s5.mala_settings.n_burnin_draws = 500; s5.mala_settings.n_keep_draws = 1000;
    s5.nuts_settings.n_burnin_draws = 300; s5.nuts_settings.n_keep_draws = 600; s5.nuts_settings.n_adapt_draws = 300;
    iso_data_t dm;
    mcmc::Mat_t dmala;
    const bool okm = mcmc::mala(initial_val, log_target_dens, dmala, &dm, s5);
    std::printf("callback mala ok=%d rows=%zu cols=%zu mean=%.6f %.6f %.6f acc=%.4f grad_calls=%d value_calls=%d\n",
                int(okm), size_t(dmala.rows()), size_t(dmala.cols()), okm ? col_mean(dmala, 0) : 0.0, okm ? col_mean(dmala, 1) : 0.0,
                okm ? col_mean(dmala, 2) : 0.0, double(s5.mala_settings.n_accept_draws) / 1000.0, dm.n_calls_grad, dm.n_calls_value);
    iso_data_t dn;
    mcmc::Mat_t dnuts;
    const bool okn = mcmc::nuts(initial_val, log_target_dens, dnuts, &dn, s5);
    std::printf("callback nuts ok=%d rows=%zu cols=%zu mean=%.6f %.6f %.6f acc=%.4f grad_calls=%d value_calls=%d\n",
                int(okn), size_t(dnuts.rows()), size_t(dnuts.cols()), okn ? col_mean(dnuts, 0) : 0.0, okn ? col_mean(dnuts, 1) : 0.0,
                okn ? col_mean(dnuts, 2) : 0.0, double(s5.nuts_settings.n_accept_draws) / 600.0, dn.n_calls_grad, dn.n_calls_value);

    // mcmc::rwmh with a host std::function (value only, ref: include/mcmc/rwmh.hpp:42-47; the call pattern of examples/eigen/rwmh_normal.cpp)
    s5.rwmh_settings.par_scale = 0.8;
    s5.rwmh_settings.n_burnin_draws = 500; s5.rwmh_settings.n_keep_draws = 4000;
    iso_data_t dw;
    mcmc::Mat_t drw;
    const bool okw = mcmc::rwmh(initial_val, [](const mcmc::ColVec_t& v, void* p) {
        static_cast<iso_data_t*>(p)->n_calls_value++;
        double ss = 0.0;
        for (size_t i = 0; i < size_t(v.size()); ++i) ss += v(i) * v(i);
        return -0.5 * ss; }, drw, &dw, s5);
    std::printf("callback rwmh ok=%d rows=%zu cols=%zu mean=%.6f %.6f %.6f acc=%.4f value_calls=%d\n", int(okw), size_t(drw.rows()), size_t(drw.cols()),
                okw ? col_mean(drw, 0) : 0.0, okw ? col_mean(drw, 1) : 0.0, okw ? col_mean(drw, 2) : 0.0,
                double(s5.rwmh_settings.n_accept_draws) / 4000.0, dw.n_calls_value);

    // NOT in the reference: the diagonal mass matrix adapted during burn-in (mcmc::mi355x::hmc_mass_adapted) on an ill-scaled diagonal
    // Gaussian (precisions 1 .. 1e4), 256 chains started from different points -- pooled over the chains, then per chain
    {
        const size_t dd = 32, CC = 256;
        std::vector<double> lam(dd);
        for (size_t i = 0; i < dd; ++i) lam[i] = std::pow(10.0, 4.0 * double(i) / double(dd - 1));
        mcmc::mi355x::target_t td = mcmc::mi355x::gaussian_diag(dd, lam.data());
        td.n_chains = CC;
        mcmc::ColVec_t i0(dd * CC);
        unsigned long long stt = 88172645463325252ULL;
        for (size_t c = 0; c < CC; ++c)
            for (size_t i = 0; i < dd; ++i) { stt ^= stt << 13; stt ^= stt >> 7; stt ^= stt << 17; i0(c * dd + i) = (double(stt >> 11) / 9007199254740992.0 - 0.5) * 3.0 / std::sqrt(lam[i]); }
        mcmc::algo_settings_t sm;
        sm.rng_seed_value = 3;
        sm.hmc_settings.step_size = 0.2; sm.hmc_settings.n_leap_steps = 8; sm.hmc_settings.n_burnin_draws = 60; sm.hmc_settings.n_keep_draws = 20;
        mcmc::Mat_t dmass;
        std::vector<double> mass;
        const bool okp = mcmc::mi355x::hmc_mass_adapted(i0, td, dmass, sm, 3, false, 0.0, &mass);
        std::printf("mass adapted (pooled) ok=%d cols=%zu acc0=%.2f mass0/prec0=%.2f massLast/precLast=%.2f\n", int(okp), size_t(dmass.cols()),
                    double(sm.hmc_settings.n_accept_draws) / 20.0, okp ? mass[0] / lam[0] : 0.0, okp ? mass[dd - 1] / lam[dd - 1] : 0.0);
        const bool okc = mcmc::mi355x::hmc_mass_adapted(i0, td, dmass, sm, 2, true, 0.01, &mass);
        std::printf("mass adapted (per chain) ok=%d cols=%zu acc0=%.2f masses=%zu\n", int(okc), size_t(dmass.cols()),
                    double(sm.hmc_settings.n_accept_draws) / 20.0, mass.size());
        if (!okp || !okc) { std::printf("reason=\"%s\"\n", td.last_error.c_str()); return 1; }
    }

    // mcmc::rmhmc with HOST std::function callbacks, the flow of the reference's examples/eigen/rmhmc_normal.cpp: (mu, sigma) of normal
    // data, the Fisher information as metric tensor, sigma bounded below.  The sampler runs on the device and asks for every evaluation.
    norm_data_t nd{x_obs.data(), x_obs.size(), 0, 0, 0};
    mcmc::algo_settings_t s6;
    s6.rng_seed_value = 11;
    s6.vals_bound = true;
    s6.lower_bounds = mcmc::ColVec_t(2); s6.upper_bounds = mcmc::ColVec_t(2);
    s6.lower_bounds(0) = -std::numeric_limits<double>::infinity(); s6.lower_bounds(1) = 0.01;
    s6.upper_bounds(0) = std::numeric_limits<double>::infinity(); s6.upper_bounds(1) = std::numeric_limits<double>::infinity();
    s6.rmhmc_settings.step_size = 0.02;                  // (n_leap_steps = 1, n_fp_steps = 5: the struct's defaults, as in the device run above)
    s6.rmhmc_settings.n_burnin_draws = 100; s6.rmhmc_settings.n_keep_draws = 200;
    mcmc::Mat_t drm;
    const bool okr = mcmc::rmhmc(init2, normal_ll, normal_tensor, drm, &nd, &nd, s6);
    std::printf("callback rmhmc ok=%d rows=%zu cols=%zu mean_mu=%.4f mean_sigma=%.4f acc=%.3f grad_calls=%d value_calls=%d tensor_calls=%d reason=\"%s\"\n",
                int(okr), size_t(drm.rows()), size_t(drm.cols()), okr ? col_mean(drm, 0) : 0.0, okr ? col_mean(drm, 1) : 0.0,
                double(s6.rmhmc_settings.n_accept_draws) / 200.0, nd.n_grad, nd.n_value, nd.n_tensor, okr ? "" : mcmc::mi355x::last_error().c_str());
    // mixing the routes (a device target with a host tensor) is refused with a reason, never run on the CPU
    const bool refused = !mcmc::rmhmc(init2, mcmc::mi355x::device_kernel, normal_tensor, drm, &nm, &nd, s6);
    std::printf("rmhmc with mixed routes refused=%d reason=\"%s\"\n", int(refused), mcmc::mi355x::last_error().c_str());
    return (ok && ok2 && okm && okn && okw && okr && refused) ? 0 : 1;
}
