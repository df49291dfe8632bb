// examples/normal_model.cpp -- the reference's example programs (/root/reference/examples/eigen/{hmc,mala,nuts,rmhmc}_normal.cpp:
// posterior of (mu, sigma) given normal observations; settings, mcmc::<algo>(initial_val, target, draws_out, &data, settings),
// column means, acceptance rate) against this repository's mcmc.hpp, with the device target in place of the host callback and
// 256 chains per call instead of one.  Settings are the examples' own (step 0.08, 2000 + 2000 draws; rmhmc: step 0.2).
//
//   g++ -std=c++17 -O2 -Iinclude examples/normal_model.cpp -Lmcmc_amd -lmi_mcmc -Wl,-rpath,$PWD/mcmc_amd -o normal_model
#include "mcmc.hpp"

#include <cmath>
#include <cstdio>
#include <vector>

static void report(const char* name, bool ok, const mcmc::Mat_t& draws, size_t n_chains, double acc)
{
    double mu = 0.0, sigma = 0.0;                 // draws: n_keep x (2 * C), chain c in columns 2c, 2c+1
    if (ok) {
        for (size_t c = 0; c < n_chains; ++c) { mu += draws.col_mean(2 * c); sigma += draws.col_mean(2 * c + 1); }
        mu /= double(n_chains); sigma /= double(n_chains);
    }
    std::printf("%s ok=%d rows=%zu cols=%zu mean_mu=%.4f mean_sigma=%.4f acc0=%.3f\n", name, int(ok), size_t(draws.rows()),
                size_t(draws.cols()), mu, sigma, acc);
}

int main()
{
    const size_t n_data = 1000, C = 256;
    std::vector<double> x(n_data);                // mu = 2, sigma = 2 (xorshift -> Irwin-Hall(12) - 6: near-normal observations)
    unsigned long long st = 88172645463325252ULL;
    double xbar = 0.0;
    for (auto& v : x) {
        double a = 0;
        for (int k = 0; k < 12; ++k) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; a += double(st >> 11) / 9007199254740992.0; }
        v = 2.0 + 2.0 * (a - 6.0); xbar += v;
    }
    xbar /= double(n_data);
    double s2 = 0.0;
    for (double v : x) s2 += (v - xbar) * (v - xbar);
    std::printf("data n=%zu xbar=%.4f sd=%.4f\n", n_data, xbar, std::sqrt(s2 / double(n_data)));

    mcmc::mi355x::target_t tgt = mcmc::mi355x::normal_model(n_data, x.data());
    tgt.n_chains = C;
    mcmc::ColVec_t initial_val(2);
    initial_val(0) = 3.0;                         // mu + 1
    initial_val(1) = 3.0;                         // sigma + 1
    mcmc::Mat_t draws_out;
    bool all = true;

    {   // hmc_normal.cpp:99-113
        mcmc::algo_settings_t settings;
        settings.rng_seed_value = 1;
        settings.hmc_settings.step_size = 0.08;
        settings.hmc_settings.n_burnin_draws = 2000;
        settings.hmc_settings.n_keep_draws = 2000;
        const bool ok = mcmc::hmc(initial_val, mcmc::mi355x::device_kernel, draws_out, &tgt, settings);
        report("hmc", ok, draws_out, C, double(settings.hmc_settings.n_accept_draws) / 2000.0); all = all && ok;
    }
    {   // mala_normal.cpp:99-113
        mcmc::algo_settings_t settings;
        settings.rng_seed_value = 2;
        settings.mala_settings.step_size = 0.08;
        settings.mala_settings.n_burnin_draws = 2000;
        settings.mala_settings.n_keep_draws = 2000;
        const bool ok = mcmc::mala(initial_val, mcmc::mi355x::device_kernel, draws_out, &tgt, settings);
        report("mala", ok, draws_out, C, double(settings.mala_settings.n_accept_draws) / 2000.0); all = all && ok;
    }
    {   // nuts_normal.cpp:99-112 (default nuts settings)
        mcmc::algo_settings_t settings;
        settings.rng_seed_value = 3;
        settings.nuts_settings.n_burnin_draws = 2000;
        settings.nuts_settings.n_keep_draws = 2000;
        const bool ok = mcmc::nuts(initial_val, mcmc::mi355x::device_kernel, draws_out, &tgt, settings);
        report("nuts", ok, draws_out, C, double(settings.nuts_settings.n_accept_draws) / 2000.0); all = all && ok;
    }
    {   // rwmh (the reference's rwmh example is a one-parameter model; same call pattern)
        mcmc::algo_settings_t settings;
        settings.rng_seed_value = 4;
        settings.rwmh_settings.par_scale = 0.1;
        settings.rwmh_settings.n_burnin_draws = 2000;
        settings.rwmh_settings.n_keep_draws = 2000;
        const bool ok = mcmc::rwmh(initial_val, mcmc::mi355x::device_value_kernel, draws_out, &tgt, settings);
        report("rwmh", ok, draws_out, C, double(settings.rwmh_settings.n_accept_draws) / 2000.0); all = all && ok;
    }
    {   // rmhmc_normal.cpp:132-146 (Fisher-information metric; see DESIGN.md section 3 on the reference's momentum sign)
        mcmc::algo_settings_t settings;
        settings.rng_seed_value = 5;
        settings.rmhmc_settings.step_size = 0.2;
        settings.rmhmc_settings.n_burnin_draws = 2000;
        settings.rmhmc_settings.n_keep_draws = 2000;
        const bool ok = mcmc::rmhmc(initial_val, mcmc::mi355x::device_kernel, mcmc::mi355x::device_tensor, draws_out, &tgt, &tgt, settings);
        report("rmhmc", ok, draws_out, C, double(settings.rmhmc_settings.n_accept_draws) / 2000.0); all = all && ok;
    }
    return all ? 0 : 1;
}
