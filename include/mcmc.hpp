// mcmc.hpp -- drop-in C++17 front end of the MI355X engine for the hmc / mala / nuts path of
// kthohr/mcmc (MCMCLib).  Usage is the reference's:
//
//     #define MCMC_ENABLE_EIGEN_WRAPPERS
//     #include "mcmc.hpp"
//     mcmc::algo_settings_t settings;  settings.hmc_settings.step_size = 0.08;  ...
//     mcmc::hmc(initial_vals, log_target_dens, draws_out, &data, settings);
//
// Same namespace, type aliases, settings structs (names, types and defaults of
// /root/reference/include/misc/mcmc_structs.hpp:24-184) and function signatures
// (/root/reference/include/mcmc/hmc.hpp:42-48,65-72,78-85; mala.hpp:43-49,66-73,79-86;
// nuts.hpp:42-48,65-72,78-85).  Behind them everything runs on the GPU through the C ABI of
// mi_mcmc.h (link -lmi_mcmc); there is no CPU sampler in this header.
//
//  * Host std::function targets (the reference contract): mcmc::hmc drives the draw loop and calls
//    the callback where src/hmc.cpp does; all sampler arithmetic is on the device.  mala / nuts need
//    a device target.
//  * Device targets: pass mcmc::mi355x::device_kernel as target_log_kernel and a
//    mcmc::mi355x::target_t* as target_data; n_chains independent chains run in one fused launch.
//    initial_vals holds d values (shared start) or d*n_chains (chain c in [c*d, (c+1)*d));
//    draws_out is n_keep x (d*n_chains), chain c in columns [c*d, (c+1)*d) -- for one chain exactly the
//    reference's n_keep x d.
//
// With Eigen on the include path (and MCMC_ENABLE_EIGEN_WRAPPERS) ColVec_t / Mat_t are the Eigen types
// of mcmc_options.hpp:159-168; otherwise a minimal column-major stand-in with the members the samplers
// and the examples use.
#ifndef MCMC_MI355X_FRONTEND_HPP
#define MCMC_MI355X_FRONTEND_HPP

#include <cstddef>
#include <cstdint>
#include <functional>
#include <limits>
#include <random>
#include <string>
#include <vector>

#include "mi_mcmc.h"

#if defined(MCMC_ENABLE_EIGEN_WRAPPERS) && defined(__has_include)
  #if __has_include(<Eigen/Dense>)
    #include <Eigen/Dense>
    #define MCMC_MI355X_HAVE_EIGEN 1
  #endif
#endif

namespace mcmc
{

using uint_t = unsigned int;
using fp_t = double;                       // MCMC_FPN_TYPE (mcmc_options.hpp:80-81,99): the engine is fp64 only
using rand_engine_t = std::mt19937_64;     // kept for source compatibility; the device uses Philox4x32-10

static const double eps_dbl = std::numeric_limits<fp_t>::epsilon();
static const double posinf  = std::numeric_limits<fp_t>::infinity();
static const double neginf  = - std::numeric_limits<fp_t>::infinity();

#ifdef MCMC_MI355X_HAVE_EIGEN
using ColVec_t = Eigen::Matrix<fp_t, Eigen::Dynamic, 1>;
using RowVec_t = Eigen::Matrix<fp_t, 1, Eigen::Dynamic>;
using ColVecInt_t = Eigen::Matrix<int, Eigen::Dynamic, 1>;
using Mat_t = Eigen::Matrix<fp_t, Eigen::Dynamic, Eigen::Dynamic>;
#else
// Minimal dense types (column-major like Eigen's default) for builds without Eigen.
class Mat_t
{
public:
    Mat_t() = default;
    Mat_t(size_t r, size_t c) : r_(r), c_(c), v_(r * c, 0.0) {}
    size_t rows() const { return r_; }
    size_t cols() const { return c_; }
    size_t size() const { return v_.size(); }
    void resize(size_t r, size_t c) { r_ = r; c_ = c; v_.assign(r * c, 0.0); }
    fp_t& operator()(size_t i, size_t j) { return v_[i + j * r_]; }
    const fp_t& operator()(size_t i, size_t j) const { return v_[i + j * r_]; }
    fp_t* data() { return v_.data(); }
    const fp_t* data() const { return v_.data(); }
    void setZero() { v_.assign(v_.size(), 0.0); }
    fp_t col_mean(size_t j) const { fp_t s = 0; for (size_t i = 0; i < r_; ++i) s += (*this)(i, j); return r_ ? s / fp_t(r_) : 0; }
private:
    size_t r_ = 0, c_ = 0;
    std::vector<fp_t> v_;
};

class ColVec_t
{
public:
    ColVec_t() = default;
    explicit ColVec_t(size_t n) : v_(n, 0.0) {}
    size_t size() const { return v_.size(); }
    size_t rows() const { return v_.size(); }
    size_t cols() const { return 1; }
    void resize(size_t n) { v_.assign(n, 0.0); }
    void resize(size_t n, size_t) { v_.assign(n, 0.0); }
    fp_t& operator()(size_t i) { return v_[i]; }
    const fp_t& operator()(size_t i) const { return v_[i]; }
    fp_t& operator()(size_t i, size_t) { return v_[i]; }
    const fp_t& operator()(size_t i, size_t) const { return v_[i]; }
    fp_t& operator[](size_t i) { return v_[i]; }
    const fp_t& operator[](size_t i) const { return v_[i]; }
    fp_t* data() { return v_.data(); }
    const fp_t* data() const { return v_.data(); }
private:
    std::vector<fp_t> v_;
};
using RowVec_t = ColVec_t;
#endif

// bmo::Cube_t<fp_t> (mcmc_options.hpp:212): n_slice matrices of n_row x n_col; tensor_fn fills mat(i) = dG/dvals_i
class Cube_t
{
public:
    Cube_t() = default;
    Cube_t(size_t r, size_t c, size_t s) { setZero(r, c, s); }
    void setZero(size_t r, size_t c, size_t s) { mats_.assign(s, Mat_t(r, c)); for (auto& m : mats_) m.setZero(); }
    Mat_t& mat(size_t i) { return mats_[i]; }
    const Mat_t& mat(size_t i) const { return mats_[i]; }
    size_t n_mat() const { return mats_.size(); }
private:
    std::vector<Mat_t> mats_;
};

// ---------------------------------------------------------------------------------------------
// settings (field for field the reference's; only hmc / mala / nuts are read by this engine)

struct aees_settings_t
{
    size_t n_initial_draws = 1E03;
    size_t n_burnin_draws = 1E03;
    size_t n_keep_draws = 1E03;
    int omp_n_threads = -1;
    fp_t par_scale = 1.0;
    Mat_t cov_mat;
    size_t n_rings = 5;
    fp_t ee_prob_par = 0.10;
    ColVec_t temper_vec;
};

struct de_settings_t
{
    bool jumps = false;
    size_t n_pop = 100;
    size_t n_burnin_draws = 1E03;
    size_t n_keep_draws = 1E03;
    int omp_n_threads = -1;
    fp_t par_b = 1E-04;
    fp_t par_gamma = 1.0;
    fp_t par_gamma_jump = 2.0;
    ColVec_t initial_lb;
    ColVec_t initial_ub;
    size_t n_accept_draws;
};

struct hmc_settings_t
{
    size_t n_burnin_draws = 1E03;
    size_t n_keep_draws = 1E03;
    int omp_n_threads = -1;      // unused on the device (the reference uses it for the epilogue only)
    size_t n_leap_steps = 1;
    fp_t step_size = 1.0;
    Mat_t precond_mat;
    size_t n_accept_draws;       // written back by the sampler (chain 0 when several chains run)
};

struct nuts_settings_t
{
    size_t n_burnin_draws = 1E03;
    size_t n_keep_draws = 1E03;
    int omp_n_threads = -1;
    size_t n_adapt_draws = 1E03;
    fp_t target_accept_rate = 0.55;
    size_t max_tree_depth = size_t(10);
    fp_t step_size = 1.0;
    fp_t gamma_val = 0.05;
    fp_t t0_val = 10;
    fp_t kappa_val = 0.75;
    Mat_t precond_mat;
    size_t n_accept_draws;
};

struct rmhmc_settings_t
{
    size_t n_burnin_draws = 1E03;
    size_t n_keep_draws = 1E03;
    int omp_n_threads = -1;
    size_t n_leap_steps = 1;
    fp_t step_size = 1.0;
    Mat_t precond_mat;
    size_t n_fp_steps = 5;
    size_t n_accept_draws;
};

struct mala_settings_t
{
    size_t n_burnin_draws = 1E03;
    size_t n_keep_draws = 1E03;
    int omp_n_threads = -1;
    fp_t step_size = 1.0;
    Mat_t precond_mat;
    size_t n_accept_draws;
};

struct rwmh_settings_t
{
    size_t n_burnin_draws = 1E03;
    size_t n_keep_draws = 1E03;
    int omp_n_threads = -1;
    fp_t par_scale = 1.0;
    Mat_t cov_mat;
    size_t n_accept_draws;
};

struct algo_settings_t
{
    size_t rng_seed_value = std::random_device{}();
    bool vals_bound = false;
    ColVec_t lower_bounds;
    ColVec_t upper_bounds;
    aees_settings_t aees_settings;
    de_settings_t de_settings;
    hmc_settings_t hmc_settings;
    nuts_settings_t nuts_settings;
    rmhmc_settings_t rmhmc_settings;
    mala_settings_t mala_settings;
    rwmh_settings_t rwmh_settings;
};

using log_kernel_fn_t = std::function<fp_t (const ColVec_t& vals_inp, ColVec_t* grad_out, void* target_data)>;
using tensor_fn_t = std::function<Mat_t (const ColVec_t& vals_inp, Cube_t* tensor_deriv_out, void* tensor_data)>;

// ---------------------------------------------------------------------------------------------
// device targets

namespace mi355x
{

struct target_t
{
    mi_target desc{};            // kind / d / prec / X / y (mi_mcmc.h)
    size_t n_chains = 1;         // independent chains to run in one launch
    size_t chain0 = 0;           // global id of the first chain (Philox counter)
    std::vector<uint64_t> n_accept_draws;   // out: per chain
    std::vector<double> step_size;          // out (nuts): adapted step size per chain
    std::string last_error;      // out: mi_mcmc_last_error() when a sampler returns false
};

// why the last sampler call of this thread returned false when it had no target_t to report into (host-callback routes)
inline std::string& last_error() { thread_local std::string e; return e; }

inline target_t gaussian_iso(size_t d)
{
    target_t t; t.desc.struct_size = sizeof(mi_target); t.desc.kind = MI_TARGET_GAUSS_ISO; t.desc.d = d; return t;
}
inline target_t gaussian_diag(size_t d, const double* prec)
{
    target_t t = gaussian_iso(d); t.desc.kind = MI_TARGET_GAUSS_DIAG; t.desc.prec = prec; return t;
}
inline target_t gaussian_dense(size_t d, const double* prec_row_major)
{
    target_t t = gaussian_iso(d); t.desc.kind = MI_TARGET_GAUSS_DENSE; t.desc.prec = prec_row_major; return t;
}
// the d = 2 (mu, sigma) model of the reference's example programs (examples/eigen/rmhmc_normal.cpp); x must outlive the run
inline target_t normal_model(size_t n_obs, const double* x)
{
    target_t t; t.desc.struct_size = sizeof(mi_target); t.desc.kind = MI_TARGET_NORMAL_MODEL; t.desc.d = 2;
    t.desc.y = x; t.desc.n_rows = n_obs; t.desc.mem = MI_MEM_HOST; return t;
}
inline target_t logistic_regression(size_t d, size_t n_rows, const double* X_row_major, const double* y)
{
    target_t t = gaussian_iso(d); t.desc.kind = MI_TARGET_LOGISTIC; t.desc.X = X_row_major; t.desc.y = y;
    t.desc.n_rows = n_rows; return t;
}

// Tag callback: selects the device-target route. Never evaluated on the host.
inline fp_t device_kernel(const ColVec_t&, ColVec_t*, void*) { return std::numeric_limits<fp_t>::quiet_NaN(); }

inline bool is_device_route(const log_kernel_fn_t& f)
{
    using fptr_t = fp_t (*)(const ColVec_t&, ColVec_t*, void*);
    const fptr_t* p = f.target<fptr_t>();
    return p && *p == &device_kernel;
}

// the same tag for mcmc::rwmh, whose callback takes no gradient (ref: include/mcmc/rwmh.hpp:42-47)
// mcmc::rmhmc: the metric tensor built into the target kind (normal_model: Fisher information, mi_mcmc.h)
inline Mat_t device_tensor(const ColVec_t&, Cube_t*, void*) { return Mat_t(); }
inline bool is_device_route(const tensor_fn_t& f)
{
    using fn_t = Mat_t (*)(const ColVec_t&, Cube_t*, void*);
    const fn_t* p = f.target<fn_t>();
    return p && *p == &device_tensor;
}
inline fp_t device_value_kernel(const ColVec_t&, void*) { return std::numeric_limits<fp_t>::quiet_NaN(); }

inline bool is_device_route(const std::function<fp_t (const ColVec_t&, void*)>& f)
{
    using fptr_t = fp_t (*)(const ColVec_t&, void*);
    const fptr_t* p = f.target<fptr_t>();
    return p && *p == &device_value_kernel;
}

}  // namespace mi355x

// ---------------------------------------------------------------------------------------------

namespace internal
{

inline mi_settings flatten_common(const algo_settings_t& s)
{
    mi_settings m; mi_settings_default(&m);
    m.rng_seed_value = s.rng_seed_value;
    m.vals_bound = s.vals_bound ? 1 : 0;
    m.lower_bounds = s.vals_bound ? s.lower_bounds.data() : nullptr;
    m.upper_bounds = s.vals_bound ? s.upper_bounds.data() : nullptr;
    return m;
}

// precond_mat counts only when it has d*d elements (src/hmc.cpp:57); symmetric, so the storage order is immaterial
inline const double* precond_or_null(const Mat_t& M, size_t d) { return (size_t(M.size()) == d * d && d > 0) ? M.data() : nullptr; }

// many chains through mi_mcmc_<algo>_run; fills draws_out as n_keep x (d * C).  algo 10 / 11: hmc with the diagonal mass adapted during
// burn-in, pooled over the chains / per chain (NOT reference modes; mi_mcmc.h: mi_mcmc_hmc_run_mass_adapted[_per_chain])
inline bool run_device(int algo, const ColVec_t& initial_vals, mi355x::target_t& tgt, Mat_t& draws_out, mi_settings& m,
                       unsigned n_windows = 0, double first_step_size = 0.0, std::vector<double>* mass_out = nullptr)
{
    const size_t d = tgt.desc.d, C = tgt.n_chains ? tgt.n_chains : 1, n_keep = m.n_keep_draws;
    const size_t n_init = size_t(initial_vals.size());
    if (d == 0 || (n_init != d && n_init != d * C)) { tgt.last_error = "initial_vals must hold d or d*n_chains values"; return false; }
    std::vector<double> theta(d * C), draws(n_keep * d * C);
    for (size_t c = 0; c < C; ++c)
        for (size_t j = 0; j < d; ++j) theta[j * C + c] = initial_vals(n_init == d ? j : c * d + j);
    tgt.n_accept_draws.assign(C, 0);
    tgt.step_size.assign(C, 0.0);
    mi_chains ch{};
    ch.struct_size = sizeof ch; ch.mem = MI_MEM_HOST; ch.n_chains = C; ch.chain0 = tgt.chain0;
    ch.theta = theta.data(); ch.draws = draws.data(); ch.n_accept = tgt.n_accept_draws.data();
    ch.step_size = tgt.step_size.data();
    tgt.desc.struct_size = sizeof(mi_target);
    if (mass_out) mass_out->assign(algo == 11 ? d * C : d, 0.0);
    const int rc = (algo == 10) ? mi_mcmc_hmc_run_mass_adapted(&tgt.desc, &m, &ch, n_windows, mass_out ? mass_out->data() : nullptr, nullptr)
                 : (algo == 11) ? mi_mcmc_hmc_run_mass_adapted_per_chain(&tgt.desc, &m, &ch, n_windows, first_step_size,
                                                                         mass_out ? mass_out->data() : nullptr, nullptr)
                 : (algo == 0) ? mi_mcmc_hmc_run(&tgt.desc, &m, &ch, nullptr)
                 : (algo == 1) ? mi_mcmc_mala_run(&tgt.desc, &m, &ch, nullptr)
                 : (algo == 3) ? mi_mcmc_rwmh_run(&tgt.desc, &m, &ch, nullptr)
                 : (algo == 4) ? mi_mcmc_rmhmc_run(&tgt.desc, &m, &ch, nullptr)
                               : mi_mcmc_nuts_run(&tgt.desc, &m, &ch, nullptr);
    if (rc != MI_OK) { tgt.last_error = mi_mcmc_last_error(); return false; }
    draws_out.resize(n_keep, d * C);                                  // BMO_MATOPS_SET_SIZE(draws_out, n_keep, n_vals)
    for (size_t k = 0; k < n_keep; ++k)
        for (size_t j = 0; j < d; ++j)
            for (size_t c = 0; c < C; ++c) draws_out(k, c * d + j) = draws[(k * d + j) * C + c];
    return true;
}

struct callback_ctx { const log_kernel_fn_t* fn; void* user; size_t d; };

inline double callback_trampoline(const double* vals, double* grad_out, void* p)
{
    callback_ctx* ctx = static_cast<callback_ctx*>(p);
    ColVec_t v(ctx->d);
    for (size_t i = 0; i < ctx->d; ++i) v(i) = vals[i];
    if (!grad_out) return (*ctx->fn)(v, nullptr, ctx->user);
    ColVec_t g(ctx->d);                                               // pre-sized like src/hmc.cpp:105
    const fp_t r = (*ctx->fn)(v, &g, ctx->user);
    for (size_t i = 0; i < ctx->d; ++i) grad_out[i] = g(i);
    return r;
}

inline bool
hmc_impl(const ColVec_t& initial_vals, log_kernel_fn_t target_log_kernel, Mat_t& draws_out, void* target_data,
         algo_settings_t* settings_inp)
{
    algo_settings_t settings;
    if (settings_inp) settings = *settings_inp;                       // copied by value (src/hmc.cpp:44-48)
    mi_settings m = flatten_common(settings);
    m.n_burnin_draws = settings.hmc_settings.n_burnin_draws;
    m.n_keep_draws = settings.hmc_settings.n_keep_draws;
    m.n_leap_steps = settings.hmc_settings.n_leap_steps;
    m.step_size = settings.hmc_settings.step_size;
    size_t n_accept = 0;
    bool ok;
    if (mi355x::is_device_route(target_log_kernel)) {
        mi355x::target_t& tgt = *static_cast<mi355x::target_t*>(target_data);
        m.precond_mat = precond_or_null(settings.hmc_settings.precond_mat, tgt.desc.d);
        ok = run_device(0, initial_vals, tgt, draws_out, m);
        if (ok) n_accept = size_t(tgt.n_accept_draws[0]);
    } else {
        const size_t d = size_t(initial_vals.size());                 // BMO_MATOPS_SIZE(initial_vals), src/hmc.cpp:40
        m.precond_mat = precond_or_null(settings.hmc_settings.precond_mat, d);
        callback_ctx ctx{&target_log_kernel, target_data, d};
        draws_out.resize(m.n_keep_draws, d);
        uint64_t nacc = 0;
        ok = mi_mcmc_hmc_run_callback(initial_vals.data(), d, &callback_trampoline, &ctx, &m, draws_out.data(), &nacc) == MI_OK;
        if (!ok) mi355x::last_error() = mi_mcmc_last_error();
        n_accept = size_t(nacc);
    }
    if (ok && settings_inp) settings_inp->hmc_settings.n_accept_draws = n_accept;     // src/hmc.cpp:220-222
    return ok;
}

struct value_callback_ctx { const std::function<fp_t (const ColVec_t&, void*)>* fn; void* user; size_t d; };

inline double value_callback_trampoline(const double* vals, double*, void* p)     // mcmc::rwmh: no gradient (rwmh.hpp:42-47)
{
    value_callback_ctx* ctx = static_cast<value_callback_ctx*>(p);
    ColVec_t v(ctx->d);
    for (size_t i = 0; i < ctx->d; ++i) v(i) = vals[i];
    return (*ctx->fn)(v, ctx->user);
}

inline bool
mala_impl(const ColVec_t& initial_vals, log_kernel_fn_t target_log_kernel, Mat_t& draws_out, void* target_data,
          algo_settings_t* settings_inp)
{
    algo_settings_t settings;
    if (settings_inp) settings = *settings_inp;
    if (!mi355x::is_device_route(target_log_kernel)) {                // host std::function: one chain, callback on the host
        const size_t d = size_t(initial_vals.size());
        mi_settings m = flatten_common(settings);
        m.n_burnin_draws = settings.mala_settings.n_burnin_draws;
        m.n_keep_draws = settings.mala_settings.n_keep_draws;
        m.step_size = settings.mala_settings.step_size;
        m.precond_mat = precond_or_null(settings.mala_settings.precond_mat, d);
        callback_ctx ctx{&target_log_kernel, target_data, d};
        draws_out.resize(m.n_keep_draws, d);
        uint64_t nacc = 0;
        const bool okc = mi_mcmc_mala_run_callback(initial_vals.data(), d, &callback_trampoline, &ctx, &m, draws_out.data(), &nacc) == MI_OK;
        if (!okc) mi355x::last_error() = mi_mcmc_last_error();
        if (okc && settings_inp) settings_inp->mala_settings.n_accept_draws = size_t(nacc);
        return okc;
    }
    mi355x::target_t& tgt = *static_cast<mi355x::target_t*>(target_data);
    mi_settings m = flatten_common(settings);
    m.n_burnin_draws = settings.mala_settings.n_burnin_draws;
    m.n_keep_draws = settings.mala_settings.n_keep_draws;
    m.step_size = settings.mala_settings.step_size;
    m.precond_mat = precond_or_null(settings.mala_settings.precond_mat, tgt.desc.d);
    const bool ok = run_device(1, initial_vals, tgt, draws_out, m);
    if (ok && settings_inp) settings_inp->mala_settings.n_accept_draws = size_t(tgt.n_accept_draws[0]);
    return ok;
}

inline bool
nuts_impl(const ColVec_t& initial_vals, log_kernel_fn_t target_log_kernel, Mat_t& draws_out, void* target_data,
          algo_settings_t* settings_inp)
{
    algo_settings_t settings;
    if (settings_inp) settings = *settings_inp;
    if (!mi355x::is_device_route(target_log_kernel)) {                // host std::function: one chain, callback on the host
        const size_t d = size_t(initial_vals.size());
        mi_settings m = flatten_common(settings);
        const nuts_settings_t& n = settings.nuts_settings;
        m.n_burnin_draws = n.n_burnin_draws; m.n_keep_draws = n.n_keep_draws; m.n_adapt_draws = n.n_adapt_draws;
        m.target_accept_rate = n.target_accept_rate; m.max_tree_depth = n.max_tree_depth; m.step_size = n.step_size;
        m.gamma_val = n.gamma_val; m.t0_val = n.t0_val; m.kappa_val = n.kappa_val;
        m.precond_mat = precond_or_null(n.precond_mat, d);
        callback_ctx ctx{&target_log_kernel, target_data, d};
        draws_out.resize(m.n_keep_draws, d);
        uint64_t nacc = 0;
        const bool okc = mi_mcmc_nuts_run_callback(initial_vals.data(), d, &callback_trampoline, &ctx, &m, draws_out.data(), &nacc, nullptr) == MI_OK;
        if (!okc) mi355x::last_error() = mi_mcmc_last_error();
        if (okc && settings_inp) settings_inp->nuts_settings.n_accept_draws = size_t(nacc);
        return okc;
    }
    mi355x::target_t& tgt = *static_cast<mi355x::target_t*>(target_data);
    mi_settings m = flatten_common(settings);
    const nuts_settings_t& n = settings.nuts_settings;
    m.n_burnin_draws = n.n_burnin_draws;
    m.n_keep_draws = n.n_keep_draws;
    m.n_adapt_draws = n.n_adapt_draws;
    m.target_accept_rate = n.target_accept_rate;
    m.max_tree_depth = n.max_tree_depth;
    m.step_size = n.step_size;
    m.gamma_val = n.gamma_val;
    m.t0_val = n.t0_val;
    m.kappa_val = n.kappa_val;
    m.precond_mat = precond_or_null(n.precond_mat, tgt.desc.d);
    const bool ok = run_device(2, initial_vals, tgt, draws_out, m);
    if (ok && settings_inp) settings_inp->nuts_settings.n_accept_draws = size_t(tgt.n_accept_draws[0]);
    return ok;
}

inline bool
rwmh_impl(const ColVec_t& initial_vals, std::function<fp_t (const ColVec_t& vals_inp, void* target_data)> target_log_kernel,
          Mat_t& draws_out, void* target_data, algo_settings_t* settings_inp)
{
    // ref: src/rwmh.cpp:30-175.  par_scale and cov_mat travel in the POD mirror's step_size / precond_mat (mi_mcmc.h)
    algo_settings_t settings;
    if (settings_inp) settings = *settings_inp;
    if (!mi355x::is_device_route(target_log_kernel)) {               // host std::function: one chain, one value callback per draw
        const size_t d = size_t(initial_vals.size());
        mi_settings m = flatten_common(settings);
        m.n_burnin_draws = settings.rwmh_settings.n_burnin_draws;
        m.n_keep_draws = settings.rwmh_settings.n_keep_draws;
        m.step_size = settings.rwmh_settings.par_scale;
        m.precond_mat = precond_or_null(settings.rwmh_settings.cov_mat, d);
        value_callback_ctx ctx{&target_log_kernel, target_data, d};
        draws_out.resize(m.n_keep_draws, d);
        uint64_t nacc = 0;
        const bool okc = mi_mcmc_rwmh_run_callback(initial_vals.data(), d, &value_callback_trampoline, &ctx, &m, draws_out.data(), &nacc) == MI_OK;
        if (!okc) mi355x::last_error() = mi_mcmc_last_error();
        if (okc && settings_inp) settings_inp->rwmh_settings.n_accept_draws = size_t(nacc);
        return okc;
    }
    mi355x::target_t& tgt = *static_cast<mi355x::target_t*>(target_data);
    mi_settings m = flatten_common(settings);
    m.n_burnin_draws = settings.rwmh_settings.n_burnin_draws;
    m.n_keep_draws = settings.rwmh_settings.n_keep_draws;
    m.step_size = settings.rwmh_settings.par_scale;
    m.precond_mat = precond_or_null(settings.rwmh_settings.cov_mat, tgt.desc.d);
    const bool ok = run_device(3, initial_vals, tgt, draws_out, m);
    if (ok && settings_inp) settings_inp->rwmh_settings.n_accept_draws = size_t(tgt.n_accept_draws[0]);
    return ok;
}

struct tensor_callback_ctx { const tensor_fn_t* fn; void* user; size_t d; };

// tensor_fn(vals_inp, tensor_deriv_out, tensor_data) -> d*d row-major tensor and, when asked for, the d matrices dG/dvals_i
inline void tensor_callback_trampoline(const double* vals, double* tensor_out, double* deriv_out, void* p)
{
    tensor_callback_ctx* ctx = static_cast<tensor_callback_ctx*>(p);
    const size_t d = ctx->d;
    ColVec_t v(d);
    for (size_t i = 0; i < d; ++i) v(i) = vals[i];
    Cube_t cube;
    if (deriv_out) cube.setZero(d, d, d);                              // pre-sized like src/rmhmc.cpp:187
    const Mat_t G = (*ctx->fn)(v, deriv_out ? &cube : nullptr, ctx->user);
    for (size_t i = 0; i < d; ++i)
        for (size_t j = 0; j < d; ++j) tensor_out[i * d + j] = G(i, j);
    if (deriv_out)
        for (size_t k = 0; k < d; ++k)
            for (size_t i = 0; i < d; ++i)
                for (size_t j = 0; j < d; ++j) deriv_out[(k * d + i) * d + j] = cube.mat(k)(i, j);
}

inline bool
rmhmc_impl(const ColVec_t& initial_vals, log_kernel_fn_t target_log_kernel, tensor_fn_t tensor_fn, Mat_t& draws_out,
           void* target_data, void* tensor_data, algo_settings_t* settings_inp)
{
    // ref: src/rmhmc.cpp:30-287
    algo_settings_t settings;
    if (settings_inp) settings = *settings_inp;
    if (mi355x::is_device_route(target_log_kernel) != mi355x::is_device_route(tensor_fn)) {
        mi355x::last_error() = "mcmc::rmhmc: target_log_kernel and tensor_fn must both be host callbacks or both the device route "
                               "(mcmc::mi355x::device_kernel and device_tensor with a mi355x::target_t)";
        return false;
    }
    if (!mi355x::is_device_route(target_log_kernel)) {
        // host std::function callbacks (the reference's own contract): the sampler runs on the device and asks the host for every
        // evaluation (mi_mcmc_rmhmc_run_callback)
        const size_t d = size_t(initial_vals.size());
        mi_settings m = flatten_common(settings);
        m.n_burnin_draws = settings.rmhmc_settings.n_burnin_draws;
        m.n_keep_draws = settings.rmhmc_settings.n_keep_draws;
        m.n_leap_steps = settings.rmhmc_settings.n_leap_steps;
        m.step_size = settings.rmhmc_settings.step_size;
        m.n_fp_steps = settings.rmhmc_settings.n_fp_steps;
        callback_ctx kctx{&target_log_kernel, target_data, d};
        tensor_callback_ctx tctx{&tensor_fn, tensor_data, d};
        draws_out.resize(m.n_keep_draws, d);
        uint64_t nacc = 0;
        const bool okc = mi_mcmc_rmhmc_run_callback(initial_vals.data(), d, &callback_trampoline, &kctx, &tensor_callback_trampoline, &tctx, &m,
                                                    draws_out.data(), &nacc) == MI_OK;
        if (!okc) mi355x::last_error() = mi_mcmc_last_error();
        if (okc && settings_inp) settings_inp->rmhmc_settings.n_accept_draws = size_t(nacc);
        return okc;
    }
    mi355x::target_t& tgt = *static_cast<mi355x::target_t*>(target_data);
    mi_settings m = flatten_common(settings);
    m.n_burnin_draws = settings.rmhmc_settings.n_burnin_draws;
    m.n_keep_draws = settings.rmhmc_settings.n_keep_draws;
    m.n_leap_steps = settings.rmhmc_settings.n_leap_steps;
    m.step_size = settings.rmhmc_settings.step_size;
    m.n_fp_steps = settings.rmhmc_settings.n_fp_steps;
    const bool ok = run_device(4, initial_vals, tgt, draws_out, m);
    if (ok && settings_inp) settings_inp->rmhmc_settings.n_accept_draws = size_t(tgt.n_accept_draws[0]);
    return ok;
}

}  // namespace internal

// ---------------------------------------------------------------------------------------------
// public wrappers: the two overloads per algorithm of the reference

inline bool hmc(const ColVec_t& initial_vals, log_kernel_fn_t target_log_kernel, Mat_t& draws_out, void* target_data)
{ return internal::hmc_impl(initial_vals, target_log_kernel, draws_out, target_data, nullptr); }

inline bool hmc(const ColVec_t& initial_vals, log_kernel_fn_t target_log_kernel, Mat_t& draws_out, void* target_data,
                algo_settings_t& settings)
{ return internal::hmc_impl(initial_vals, target_log_kernel, draws_out, target_data, &settings); }

inline bool mala(const ColVec_t& initial_vals, log_kernel_fn_t target_log_kernel, Mat_t& draws_out, void* target_data)
{ return internal::mala_impl(initial_vals, target_log_kernel, draws_out, target_data, nullptr); }

inline bool mala(const ColVec_t& initial_vals, log_kernel_fn_t target_log_kernel, Mat_t& draws_out, void* target_data,
                 algo_settings_t& settings)
{ return internal::mala_impl(initial_vals, target_log_kernel, draws_out, target_data, &settings); }

inline bool nuts(const ColVec_t& initial_vals, log_kernel_fn_t target_log_kernel, Mat_t& draws_out, void* target_data)
{ return internal::nuts_impl(initial_vals, target_log_kernel, draws_out, target_data, nullptr); }

inline bool nuts(const ColVec_t& initial_vals, log_kernel_fn_t target_log_kernel, Mat_t& draws_out, void* target_data,
                 algo_settings_t& settings)
{ return internal::nuts_impl(initial_vals, target_log_kernel, draws_out, target_data, &settings); }

inline bool rwmh(const ColVec_t& initial_vals, std::function<fp_t (const ColVec_t& vals_inp, void* target_data)> target_log_kernel,
                 Mat_t& draws_out, void* target_data)
{ return internal::rwmh_impl(initial_vals, target_log_kernel, draws_out, target_data, nullptr); }

inline bool rwmh(const ColVec_t& initial_vals, std::function<fp_t (const ColVec_t& vals_inp, void* target_data)> target_log_kernel,
                 Mat_t& draws_out, void* target_data, algo_settings_t& settings)
{ return internal::rwmh_impl(initial_vals, target_log_kernel, draws_out, target_data, &settings); }

// NOT in the reference: mcmc::hmc on the device route with the DIAGONAL mass matrix adapted during burn-in -- pooled over the chains
// (one matrix for all, estimated from their spread; mass_out: d values) or per chain (each chain from its own draws, Stan's scheme;
// mass_out: [d][n_chains]; first_step_size drives the first part, which runs with M = I on the raw target).  settings.hmc_settings as
// for mcmc::hmc, precond_mat left empty; step_size is in the preconditioned metric.  See include/mi_mcmc.h.
namespace mi355x {
inline bool hmc_mass_adapted(const ColVec_t& initial_vals, target_t& tgt, Mat_t& draws_out, algo_settings_t& settings, unsigned n_windows,
                             bool per_chain = false, double first_step_size = 0.0, std::vector<double>* mass_out = nullptr)
{
    mi_settings m = internal::flatten_common(settings);
    m.n_burnin_draws = settings.hmc_settings.n_burnin_draws;
    m.n_keep_draws = settings.hmc_settings.n_keep_draws;
    m.n_leap_steps = settings.hmc_settings.n_leap_steps;
    m.step_size = settings.hmc_settings.step_size;
    const bool ok = internal::run_device(per_chain ? 11 : 10, initial_vals, tgt, draws_out, m, n_windows, first_step_size, mass_out);
    if (ok) settings.hmc_settings.n_accept_draws = size_t(tgt.n_accept_draws[0]);
    return ok;
}
}  // namespace mi355x

// ref: include/mcmc/rmhmc.hpp (both overloads)
inline bool rmhmc(const ColVec_t& initial_vals, log_kernel_fn_t target_log_kernel, tensor_fn_t tensor_fn, Mat_t& draws_out,
                  void* target_data, void* tensor_data)
{ return internal::rmhmc_impl(initial_vals, target_log_kernel, tensor_fn, draws_out, target_data, tensor_data, nullptr); }

inline bool rmhmc(const ColVec_t& initial_vals, log_kernel_fn_t target_log_kernel, tensor_fn_t tensor_fn, Mat_t& draws_out,
                  void* target_data, void* tensor_data, algo_settings_t& settings)
{ return internal::rmhmc_impl(initial_vals, target_log_kernel, tensor_fn, draws_out, target_data, tensor_data, &settings); }

}  // namespace mcmc

#endif  // MCMC_MI355X_FRONTEND_HPP
