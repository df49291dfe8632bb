// callback_host.hip -- the host-callback forms of mcmc::hmc / mcmc::mala / mcmc::nuts for ONE chain (C ABI: mi_mcmc_*_run_callback):
// the reference's own target contract, a host std::function (/root/reference/include/mcmc/{hmc,mala,nuts}.hpp).  The host drives
// the control flow and calls the callback exactly where the reference does; the chain's state and every vector operation live
// on the GPU (callback_mode.hpp).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "host_common.hpp"
#include "det_math.hpp"
#include "callback_mode.hpp"

using mi::host::fail;
using mi::host::DevBuf;

extern "C" {

int mi_mcmc_hmc_run_callback(const double* initial_vals, uint64_t d, mi_log_kernel_cb cb, void* target_data,
                             const mi_settings* settings, double* draws_out, uint64_t* n_accept_draws)
{
    if (!initial_vals || !cb || !settings || d == 0) return fail(MI_ERR_BAD_ARG, "null / empty argument");
    if (settings->struct_size != sizeof(mi_settings)) return fail(MI_ERR_BAD_ARG, "struct_size mismatch");
    // bounds / precond_mat: the literal kernel with the callback as its target (the kernel asks the host; literal.hpp LIT_CALLBACK)
    if (settings->vals_bound || settings->precond_mat)
        return mi::host::literal_run_callback("hmc", 0, initial_vals, d, cb, target_data, nullptr, nullptr, settings, draws_out, n_accept_draws, nullptr);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(MI_ERR_NO_DEVICE, "no HIP device visible: the engine has no CPU path");
    const uint32_t dd = (uint32_t)d;
    const uint64_t n_burnin = settings->n_burnin_draws, n_keep = settings->n_keep_draws, n_total = n_burnin + n_keep;
    if (n_keep && !draws_out) return fail(MI_ERR_BAD_ARG, "draws_out is required");
    const double eps = settings->step_size;
    const uint64_t seed = settings->rng_seed_value;

    DevBuf prev, cur, mntm, grad, scal, draws, nacc;
    HIP_TRY(prev.alloc(d * 8)); HIP_TRY(cur.alloc(d * 8)); HIP_TRY(mntm.alloc(d * 8)); HIP_TRY(grad.alloc(d * 8));
    HIP_TRY(scal.alloc(4 * 8)); HIP_TRY(draws.alloc(n_keep * d * 8)); HIP_TRY(nacc.alloc(8));
    HIP_TRY(hipMemset(nacc.p, 0, 8));
    HIP_TRY(hipMemcpy(prev.p, initial_vals, d * 8, hipMemcpyHostToDevice));
    std::vector<double> h_pos(d), h_grad(d);
    double h_scal[4] = {0, 0, 0, 0};

    // prev_U = -box_log_kernel(first_draw)  (hmc.cpp:140)
    h_scal[1] = -cb(initial_vals, nullptr, target_data);
    HIP_TRY(hipMemcpy(scal.p, h_scal, sizeof(h_scal), hipMemcpyHostToDevice));

    auto grad_at_cur = [&]() -> int {   // mntm_update_fn's callback (hmc.cpp:124): gradient at new_draw
        HIP_TRY(hipMemcpy(h_pos.data(), cur.p, d * 8, hipMemcpyDeviceToHost));
        (void)cb(h_pos.data(), h_grad.data(), target_data);
        HIP_TRY(hipMemcpy(grad.p, h_grad.data(), d * 8, hipMemcpyHostToDevice));
        return MI_OK;
    };

    for (uint64_t draw = 0; draw < n_total; ++draw) {
        hipLaunchKernelGGL(mi::cb_begin_draw, dim3(1), dim3(64), 0, 0, seed, 0ull, (uint32_t)draw, dd,
                           prev.as<double>(), cur.as<double>(), mntm.as<double>(), scal.as<double>());
        for (uint64_t k = 0; k < settings->n_leap_steps; ++k) {          // hmc.cpp:164-176
            int rc = grad_at_cur(); if (rc) return rc;
            hipLaunchKernelGGL(mi::cb_half_kick, dim3(1), dim3(64), 0, 0, dd, eps, grad.as<double>(), mntm.as<double>());
            hipLaunchKernelGGL(mi::cb_drift, dim3(1), dim3(64), 0, 0, dd, eps, mntm.as<double>(), cur.as<double>());
            rc = grad_at_cur(); if (rc) return rc;
            hipLaunchKernelGGL(mi::cb_half_kick, dim3(1), dim3(64), 0, 0, dd, eps, grad.as<double>(), mntm.as<double>());
        }
        // prop_U = -box_log_kernel(new_draw)  (hmc.cpp:178): value-only callback
        HIP_TRY(hipMemcpy(h_pos.data(), cur.p, d * 8, hipMemcpyDeviceToHost));
        const double prop_U = -cb(h_pos.data(), nullptr, target_data);
        HIP_TRY(hipMemcpy(scal.as<double>() + 2, &prop_U, 8, hipMemcpyHostToDevice));
        double* row = (draw >= n_burnin) ? draws.as<double>() + (draw - n_burnin) : nullptr;   // column-major n_keep x d
        hipLaunchKernelGGL(mi::cb_accept, dim3(1), dim3(64), 0, 0, seed, 0ull, (uint32_t)draw, dd, (uint32_t)n_burnin,
                           cur.as<double>(), mntm.as<double>(), prev.as<double>(), scal.as<double>(), row, n_keep,
                           nacc.as<unsigned long long>());
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipDeviceSynchronize());
    if (n_keep) HIP_TRY(hipMemcpy(draws_out, draws.p, n_keep * d * 8, hipMemcpyDeviceToHost));
    if (n_accept_draws) HIP_TRY(hipMemcpy(n_accept_draws, nacc.p, 8, hipMemcpyDeviceToHost));
    return MI_OK;
}

}  // extern "C"

// ---- host-callback forms of mcmc::mala and mcmc::nuts for ONE chain (mala.hpp:66-73, nuts.hpp:65-72): the reference's own
//      examples (examples/eigen/{mala,nuts}_normal.cpp) pass a std::function.  As in mi_mcmc_hmc_run_callback the host drives the
//      control flow and calls the callback exactly where the reference does; the state and every vector operation live on the GPU.
namespace {

struct CbMachine {                  // device arena of d-vectors + a scalar mailbox
    uint32_t d = 0;
    DevBuf arena, scal;
    std::vector<double> h_pos, h_grad;
    mi_log_kernel_cb cb = nullptr;
    void* user = nullptr;
    uint64_t n_grad = 0, n_value = 0;
    double* vec(int k) const { return arena.as<double>() + (size_t)k * d; }
    int init(uint64_t dd, int n_vec, mi_log_kernel_cb f, void* u)
    {
        d = (uint32_t)dd; cb = f; user = u;
        h_pos.resize(dd); h_grad.resize(dd);
        HIP_TRY(arena.alloc((size_t)n_vec * dd * 8));
        HIP_TRY(hipMemset(arena.p, 0, (size_t)n_vec * dd * 8));
        HIP_TRY(scal.alloc(8 * 8));
        return MI_OK;
    }
    int copy(int dst, int src) const { HIP_TRY(hipMemcpyAsync(vec(dst), vec(src), (size_t)d * 8, hipMemcpyDeviceToDevice, 0)); return MI_OK; }
    int value_at(int v, double* out)                       // kernel(vals, nullptr, data)
    {
        HIP_TRY(hipMemcpy(h_pos.data(), vec(v), (size_t)d * 8, hipMemcpyDeviceToHost));
        *out = cb(h_pos.data(), nullptr, user); ++n_value;
        return MI_OK;
    }
    int grad_at(int v, int g)                              // kernel(vals, &grad, data), gradient to the device
    {
        HIP_TRY(hipMemcpy(h_pos.data(), vec(v), (size_t)d * 8, hipMemcpyDeviceToHost));
        (void)cb(h_pos.data(), h_grad.data(), user); ++n_grad;
        HIP_TRY(hipMemcpy(vec(g), h_grad.data(), (size_t)d * 8, hipMemcpyHostToDevice));
        return MI_OK;
    }
    int fetch(int n, double* out) const { HIP_TRY(hipMemcpy(out, scal.p, (size_t)n * 8, hipMemcpyDeviceToHost)); return MI_OK; }
    int dot(int x, int y, double* out) const
    {
        hipLaunchKernelGGL(mi::cb_dot, dim3(1), dim3(64), 0, 0, d, vec(x), vec(y), scal.as<double>());
        return fetch(1, out);
    }
};

int callback_common_checks(const char* who, const double* initial_vals, uint64_t d, mi_log_kernel_cb cb, const mi_settings* settings,
                           double* draws_out)
{
    if (!initial_vals || !cb || !settings || d == 0) return fail(MI_ERR_BAD_ARG, "%s(callback): null / empty argument", who);
    if (settings->struct_size != sizeof(mi_settings)) return fail(MI_ERR_BAD_ARG, "struct_size mismatch");
    if (settings->n_keep_draws && !draws_out) return fail(MI_ERR_BAD_ARG, "draws_out is required");
    if (settings->n_burnin_draws + settings->n_keep_draws > 0xffffffffULL) return fail(MI_ERR_BAD_ARG, "too many draws");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(MI_ERR_NO_DEVICE, "no HIP device visible: the engine has no CPU path");
    return MI_OK;
}

#define CB_TRY(expr) do { const int rc_ = (expr); if (rc_) return rc_; } while (0)

// leap_frog_fn with one step of signed size e on (pos, mntm) (src/nuts.cpp:139-154): two gradient callbacks
int cb_leapfrog(CbMachine& m, double e, int pos, int mntm, int grad)
{
    CB_TRY(m.grad_at(pos, grad));
    hipLaunchKernelGGL(mi::cb_add_half, dim3(1), dim3(64), 0, 0, m.d, e, m.vec(mntm), m.vec(grad), m.vec(mntm));
    hipLaunchKernelGGL(mi::cb_add_scaled, dim3(1), dim3(64), 0, 0, m.d, e, m.vec(pos), m.vec(mntm), m.vec(pos));
    CB_TRY(m.grad_at(pos, grad));
    hipLaunchKernelGGL(mi::cb_add_half, dim3(1), dim3(64), 0, 0, m.d, e, m.vec(mntm), m.vec(grad), m.vec(mntm));
    return MI_OK;
}

// vector ids of the nuts machine
enum { NV_PREV = 0, NV_MNTM, NV_NEW, NV_POS_T, NV_NEG_T, NV_POS_P, NV_NEG_P, NV_DUMMY_T, NV_DUMMY_P, NV_GRAD, NV_TMP, NV_LEAF_P, NV_FIXED };

struct CbNuts {
    CbMachine m;
    uint64_t seed = 0;
    uint32_t draw = 0, uslot = 0;
    double step = 0.0, log_u = 0.0, prev_U = 0.0, prev_K = 0.0;
    uint64_t n_leap = 0;
    int next_free = NV_FIXED;       // stack of temporaries of the recursion: 5 vectors per level

    struct Out { uint64_t n = 0, s = 0, n_alpha = 0; double alpha = 0.0; };

    int energy(int pos, int mntm, double* U, double* K)
    {
        double v;
        CB_TRY(m.value_at(pos, &v));
        *U = -v;
        if (!std::isfinite(*U)) *U = INFINITY;
        double q;
        CB_TRY(m.dot(mntm, mntm, &q));
        *K = q / 2.0;
        return MI_OK;
    }
    // nuts_build_tree (nuts.ipp:97-241): subtree of the given depth from (draw_v, mntm_v) in direction v; writes the proposal to
    // `prop` and the far / near edges through the (pos, neg) slots it is handed -- the caller crosses them as the reference does
    int build(int v, int draw_v, int mntm_v, uint32_t depth, int prop, int pos_t, int neg_t, int pos_p, int neg_p, Out& o)
    {
        if (depth == 0) {
            CB_TRY(m.copy(NV_TMP, draw_v));                               // the start may alias an output slot
            CB_TRY(m.copy(NV_LEAF_P, mntm_v));
            CB_TRY(m.copy(prop, NV_TMP));
            CB_TRY(cb_leapfrog(m, (double)v * step, prop, NV_LEAF_P, NV_GRAD));   // :132
            ++n_leap;
            double U, K;
            CB_TRY(energy(prop, NV_LEAF_P, &U, &K));                     // :134-140
            o.n = (log_u <= -U - K) ? 1 : 0;                              // :146
            o.s = (log_u < 1000.0 - U - K) ? 1 : 0;                       // :147
            CB_TRY(m.copy(pos_t, prop)); CB_TRY(m.copy(neg_t, prop));    // :151-155
            CB_TRY(m.copy(pos_p, NV_LEAF_P)); CB_TRY(m.copy(neg_p, NV_LEAF_P));
            const double dd = -(U + K) + (prev_U + prev_K);
            o.alpha = mi::det_exp((dd < 0.0) ? dd : 0.0);                 // :157
            o.n_alpha = 1;
            return MI_OK;
        }
        const int base = next_free;                                       // prop of the first half, then the second half's five
        next_free += 6;
        const int prop_p = base, prop_pp = base + 1, dum_t = base + 2, dum_p = base + 3, edge_t = base + 4, edge_p = base + 5;
        Out a;
        CB_TRY(build(v, draw_v, mntm_v, depth - 1, prop_p, pos_t, neg_t, pos_p, neg_p, a));     // :166-171
        if (a.s == 1) {
            Out b;
            if (v == -1) {                                                // :186-196
                CB_TRY(m.copy(dum_t, pos_t)); CB_TRY(m.copy(dum_p, pos_p));
                CB_TRY(m.copy(edge_t, neg_t)); CB_TRY(m.copy(edge_p, neg_p));
                CB_TRY(build(v, edge_t, edge_p, depth - 1, prop_pp, neg_t, dum_t, neg_p, dum_p, b));
            } else {                                                      // :198-208
                CB_TRY(m.copy(dum_t, neg_t)); CB_TRY(m.copy(dum_p, neg_p));
                CB_TRY(m.copy(edge_t, pos_t)); CB_TRY(m.copy(edge_p, pos_p));
                CB_TRY(build(v, edge_t, edge_p, depth - 1, prop_pp, dum_t, pos_t, dum_p, pos_p, b));
            }
            const double prob = (double)b.n / (double)(a.n + b.n);        // :212
            double z;
            hipLaunchKernelGGL(mi::cb_uniform, dim3(1), dim3(1), 0, 0, seed, 0ull, draw, uslot++, m.scal.as<double>());
            CB_TRY(m.fetch(1, &z));                                       // :213
            if (z < prob) CB_TRY(m.copy(prop_p, prop_pp));                // :215-217
            a.n += b.n; a.alpha += b.alpha; a.n_alpha += b.n_alpha;       // :220-222
            double q[2];
            hipLaunchKernelGGL(mi::cb_diff_dots, dim3(1), dim3(64), 0, 0, m.d, m.vec(pos_t), m.vec(neg_t), m.vec(neg_p), m.vec(pos_p),
                               m.vec(NV_TMP), m.scal.as<double>());
            CB_TRY(m.fetch(2, q));
            a.s = b.s * ((q[0] >= 0.0) ? 1 : 0) * ((q[1] >= 0.0) ? 1 : 0);   // :226-229
        }
        o = a;
        CB_TRY(m.copy(prop, prop_p));                                     // :239
        next_free = base;
        return MI_OK;
    }
};

}  // namespace

extern "C" {

int mi_mcmc_mala_run_callback(const double* initial_vals, uint64_t d, mi_log_kernel_cb cb, void* target_data,
                              const mi_settings* settings, double* draws_out, uint64_t* n_accept_draws)
{
    int rc = callback_common_checks("mala", initial_vals, d, cb, settings, draws_out);
    if (rc) return rc;
    if (settings->vals_bound || settings->precond_mat)     // the literal kernel with the callback as its target (literal.hpp LIT_CALLBACK)
        return mi::host::literal_run_callback("mala", 1, initial_vals, d, cb, target_data, nullptr, nullptr, settings, draws_out, n_accept_draws, nullptr);
    const uint64_t n_burnin = settings->n_burnin_draws, n_keep = settings->n_keep_draws, n_total = n_burnin + n_keep;
    const double eps = settings->step_size, s2 = eps * eps, rs = 1.0 / s2;
    double log_det = 0.0;                                // LOG_DET(eps^2 I) = sum_i 2 log sqrt(s2), i ascending (oracle: orc_log_det_from_chol)
    for (uint64_t i = 0; i < d; ++i) log_det = log_det + 2.0 * mi::det_log(__builtin_sqrt(s2));
    const double cons_term = -0.5 * (double)d * 1.83787706640934548356;
    enum { PREV = 0, PROP, Z, GRAD, MEAN_PREV, MEAN_PROP, TMP0, TMP1, NVEC };
    CbMachine m;
    rc = m.init(d, NVEC, cb, target_data);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(m.vec(PREV), initial_vals, d * 8, hipMemcpyHostToDevice));
    std::vector<double> row(d);
    double prev_LP;
    CB_TRY(m.value_at(PREV, &prev_LP));                  // mala.cpp:138
    uint64_t n_acc = 0;
    const uint32_t dd = (uint32_t)d;
    auto mean_of = [&](int v, int dst) -> int {          // mala_mean_fn (mala.cpp:97-125): one gradient callback
        CB_TRY(m.grad_at(v, GRAD));
        hipLaunchKernelGGL(mi::cb_add_half, dim3(1), dim3(64), 0, 0, dd, s2, m.vec(v), m.vec(GRAD), m.vec(dst));
        return MI_OK;
    };
    for (uint64_t draw = 0; draw < n_total; ++draw) {
        hipLaunchKernelGGL(mi::cb_normals, dim3(1), dim3(64), 0, 0, settings->rng_seed_value, 0ull, (uint32_t)draw, (uint32_t)mi::STREAM_NORMAL, dd, m.vec(Z));   // :150
        CB_TRY(mean_of(PREV, MEAN_PREV));                // :159
        hipLaunchKernelGGL(mi::cb_add_scaled, dim3(1), dim3(64), 0, 0, dd, eps, m.vec(MEAN_PREV), m.vec(Z), m.vec(PROP));
        double prop_LP;
        CB_TRY(m.value_at(PROP, &prop_LP));              // :162
        if (!std::isfinite(prop_LP)) prop_LP = -INFINITY;   // :164-166
        CB_TRY(mean_of(PROP, MEAN_PROP));                // mala.ipp:60
        CB_TRY(mean_of(PREV, MEAN_PREV));                // :61 (the reference evaluates it again)
        double qa, qb;
        hipLaunchKernelGGL(mi::cb_quad_form, dim3(1), dim3(64), 0, 0, dd, rs, m.vec(PREV), m.vec(MEAN_PROP), m.vec(TMP0), m.scal.as<double>());
        CB_TRY(m.fetch(1, &qa));
        hipLaunchKernelGGL(mi::cb_quad_form, dim3(1), dim3(64), 0, 0, dd, rs, m.vec(PROP), m.vec(MEAN_PREV), m.vec(TMP0), m.scal.as<double>());
        CB_TRY(m.fetch(1, &qb));
        const double da = cons_term - 0.5 * (log_det + qa), db = cons_term - 0.5 * (log_det + qb);   // dmvnorm.hpp:41
        const double x = prop_LP - prev_LP + (da - db);
        const double comp_val = (x < 0.01) ? x : 0.01;   // mala.cpp:170
        double z;
        hipLaunchKernelGGL(mi::cb_uniform, dim3(1), dim3(1), 0, 0, settings->rng_seed_value, 0ull, (uint32_t)draw, 0u, m.scal.as<double>());
        CB_TRY(m.fetch(1, &z));                          // :171
        if (z < mi::det_exp(comp_val)) {                 // :173
            CB_TRY(m.copy(PREV, PROP));
            prev_LP = prop_LP;
            if (draw >= n_burnin) ++n_acc;
        }
        if (draw >= n_burnin) {                          // row draw - n_burnin of the column-major n_keep x d matrix
            HIP_TRY(hipMemcpy(row.data(), m.vec(PREV), d * 8, hipMemcpyDeviceToHost));
            for (uint64_t j = 0; j < d; ++j) draws_out[(draw - n_burnin) + j * n_keep] = row[j];
        }
    }
    if (n_accept_draws) *n_accept_draws = n_acc;
    return MI_OK;
}

// mcmc::rwmh with a host callback (rwmh.hpp:42-47: the target returns the log-density only; src/rwmh.cpp:123-151): one value
// callback per draw.  par_scale travels in settings.step_size; identity cov_mat.
int mi_mcmc_rwmh_run_callback(const double* initial_vals, uint64_t d, mi_log_kernel_cb cb, void* target_data,
                              const mi_settings* settings, double* draws_out, uint64_t* n_accept_draws)
{
    int rc = callback_common_checks("rwmh", initial_vals, d, cb, settings, draws_out);
    if (rc) return rc;
    if (settings->vals_bound || settings->precond_mat)     // bounds / cov_mat: the literal kernel with the callback as its target
        return mi::host::literal_run_callback("rwmh", 3, initial_vals, d, cb, target_data, nullptr, nullptr, settings, draws_out, n_accept_draws, nullptr);
    const uint64_t n_burnin = settings->n_burnin_draws, n_keep = settings->n_keep_draws, n_total = n_burnin + n_keep;
    const double par_scale = settings->step_size;
    enum { PREV = 0, PROP, Z, NVEC };
    CbMachine m;
    rc = m.init(d, NVEC, cb, target_data);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(m.vec(PREV), initial_vals, d * 8, hipMemcpyHostToDevice));
    std::vector<double> row(d);
    double prev_LP;
    CB_TRY(m.value_at(PREV, &prev_LP));                  // rwmh.cpp:113
    uint64_t n_acc = 0;
    const uint32_t dd = (uint32_t)d;
    for (uint64_t draw = 0; draw < n_total; ++draw) {
        hipLaunchKernelGGL(mi::cb_normals, dim3(1), dim3(64), 0, 0, settings->rng_seed_value, 0ull, (uint32_t)draw, (uint32_t)mi::STREAM_NORMAL, dd, m.vec(Z));   // :124
        hipLaunchKernelGGL(mi::cb_add_scaled, dim3(1), dim3(64), 0, 0, dd, par_scale, m.vec(PREV), m.vec(Z), m.vec(PROP));   // :126, cov_chol = par_scale * I
        double prop_LP;
        CB_TRY(m.value_at(PROP, &prop_LP));              // :128
        if (!std::isfinite(prop_LP)) prop_LP = -INFINITY;   // :130-132
        const double x = prop_LP - prev_LP;
        const double comp_val = (x < 0.0) ? x : 0.0;     // std::min(0.0, x) :136
        double z;
        hipLaunchKernelGGL(mi::cb_uniform, dim3(1), dim3(1), 0, 0, settings->rng_seed_value, 0ull, (uint32_t)draw, 0u, m.scal.as<double>());
        CB_TRY(m.fetch(1, &z));                          // :137
        if (z < mi::det_exp(comp_val)) {                 // :139
            CB_TRY(m.copy(PREV, PROP));
            prev_LP = prop_LP;
            if (draw >= n_burnin) ++n_acc;
        }
        if (draw >= n_burnin) {                          // :148-150
            HIP_TRY(hipMemcpy(row.data(), m.vec(PREV), d * 8, hipMemcpyDeviceToHost));
            for (uint64_t j = 0; j < d; ++j) draws_out[(draw - n_burnin) + j * n_keep] = row[j];
        }
    }
    if (n_accept_draws) *n_accept_draws = n_acc;
    return MI_OK;
}

int mi_mcmc_nuts_run_callback(const double* initial_vals, uint64_t d, mi_log_kernel_cb cb, void* target_data,
                              const mi_settings* settings, double* draws_out, uint64_t* n_accept_draws, double* step_size_out)
{
    int rc = callback_common_checks("nuts", initial_vals, d, cb, settings, draws_out);
    if (rc) return rc;
    if (settings->vals_bound || settings->precond_mat)     // the literal kernel with the callback as its target (literal.hpp LIT_CALLBACK)
        return mi::host::literal_run_callback("nuts", 2, initial_vals, d, cb, target_data, nullptr, nullptr, settings, draws_out, n_accept_draws, step_size_out);
    const uint64_t n_burnin = settings->n_burnin_draws, n_keep = settings->n_keep_draws, n_total = n_burnin + n_keep;
    const uint64_t n_adapt = settings->n_adapt_draws <= n_total ? settings->n_adapt_draws : n_total;      // nuts.cpp:54
    const uint64_t max_depth = settings->max_tree_depth;
    if (max_depth > 24) return fail(MI_ERR_UNSUPPORTED, "nuts(callback): max_tree_depth > 24 not implemented");
    CbNuts t;
    rc = t.m.init(d, NV_FIXED + 6 * ((int)max_depth + 1), cb, target_data);
    if (rc) return rc;
    CbMachine& m = t.m;
    t.seed = settings->rng_seed_value;
    const uint32_t dd = (uint32_t)d;
    HIP_TRY(hipMemcpy(m.vec(NV_PREV), initial_vals, d * 8, hipMemcpyHostToDevice));
    std::vector<double> row(d);
    // nuts_find_initial_step_size (nuts.ipp:30-93) from (first_draw, z_init)
    hipLaunchKernelGGL(mi::cb_normals, dim3(1), dim3(64), 0, 0, t.seed, 0ull, 0u, (uint32_t)mi::STREAM_INIT, dd, m.vec(NV_MNTM));   // nuts.cpp:166-168
    double step = 1.0;
    {
        double U0, K0, U, K;
        CB_TRY(t.energy(NV_PREV, NV_MNTM, &U0, &K0));
        CB_TRY(m.copy(NV_NEW, NV_PREV)); CB_TRY(m.copy(NV_LEAF_P, NV_MNTM));
        CB_TRY(cb_leapfrog(m, step, NV_NEW, NV_LEAF_P, NV_GRAD)); ++t.n_leap;
        CB_TRY(t.energy(NV_NEW, NV_LEAF_P, &U, &K));
        const double log_half = mi::det_log(0.5), neg_log2 = -mi::det_log(2.0);
        int a_val = 2 * ((-(U + K) + (U0 + K0)) > log_half ? 1 : 0) - 1;
        bool cond = (-(U + K) + (U0 + K0)) > neg_log2;
        while (cond) {
            step *= (a_val == 1) ? 2.0 : 0.5;
            CB_TRY(cb_leapfrog(m, step, NV_NEW, NV_LEAF_P, NV_GRAD)); ++t.n_leap;
            CB_TRY(t.energy(NV_NEW, NV_LEAF_P, &U, &K));
            a_val = 2 * ((-(U + K) + (U0 + K0)) > log_half ? 1 : 0) - 1;
            cond = (-(U + K) + (U0 + K0)) > neg_log2;
        }
    }
    const double mu_val = mi::det_log(10 * step);        // nuts.cpp:174
    double h_val = 0.0, eps_bar = settings->step_size;
    double v0;
    CB_TRY(m.value_at(NV_PREV, &v0));
    t.prev_U = -v0;                                      // :181
    uint64_t n_acc = 0;
    for (uint64_t draw = 0; draw < n_total; ++draw) {
        t.draw = (uint32_t)draw; t.uslot = 0; t.step = step;
        hipLaunchKernelGGL(mi::cb_normals, dim3(1), dim3(64), 0, 0, t.seed, 0ull, t.draw, (uint32_t)mi::STREAM_NORMAL, dd, m.vec(NV_MNTM));   // :200-202
        double q, z;
        CB_TRY(m.dot(NV_MNTM, NV_MNTM, &q));
        t.prev_K = q / 2.0;                              // :204
        hipLaunchKernelGGL(mi::cb_uniform, dim3(1), dim3(1), 0, 0, t.seed, 0ull, t.draw, t.uslot++, m.scal.as<double>());
        CB_TRY(m.fetch(1, &z));
        t.log_u = mi::det_log(z) - t.prev_U - t.prev_K;  // :206
        CB_TRY(m.copy(NV_NEW, NV_PREV)); CB_TRY(m.copy(NV_POS_T, NV_PREV)); CB_TRY(m.copy(NV_NEG_T, NV_PREV));   // :210-215
        CB_TRY(m.copy(NV_POS_P, NV_MNTM)); CB_TRY(m.copy(NV_NEG_P, NV_MNTM));
        uint64_t depth = 0, n_val = 1, s_val = 1;
        CbNuts::Out o;
        int good_round = 0;
        while (s_val == 1 && depth < max_depth) {        // :227
            hipLaunchKernelGGL(mi::cb_uniform, dim3(1), dim3(1), 0, 0, t.seed, 0ull, t.draw, t.uslot++, m.scal.as<double>());
            CB_TRY(m.fetch(1, &z));                      // :233
            const int v = (z <= 0.5) ? -1 : 1;           // :235
            t.next_free = NV_FIXED;
            if (v == -1) {                               // :238-246
                CB_TRY(m.copy(NV_DUMMY_T, NV_POS_T)); CB_TRY(m.copy(NV_DUMMY_P, NV_POS_P));
                CB_TRY(t.build(v, NV_PREV, NV_MNTM, (uint32_t)depth, NV_NEW, NV_DUMMY_T, NV_NEG_T, NV_DUMMY_P, NV_NEG_P, o));
            } else {                                     // :248-256
                CB_TRY(m.copy(NV_DUMMY_T, NV_NEG_T)); CB_TRY(m.copy(NV_DUMMY_P, NV_NEG_P));
                CB_TRY(t.build(v, NV_PREV, NV_MNTM, (uint32_t)depth, NV_NEW, NV_POS_T, NV_DUMMY_T, NV_POS_P, NV_DUMMY_P, o));
            }
            if (o.s == 1) {                              // :260
                hipLaunchKernelGGL(mi::cb_uniform, dim3(1), dim3(1), 0, 0, t.seed, 0ull, t.draw, t.uslot++, m.scal.as<double>());
                CB_TRY(m.fetch(1, &z));                  // :261
                if (z < (double)o.n / (double)n_val) {   // :263
                    double v1;
                    CB_TRY(m.value_at(NV_NEW, &v1));     // :264
                    double pu = -v1;
                    if (!std::isfinite(pu)) pu = INFINITY;
                    CB_TRY(m.copy(NV_PREV, NV_NEW));     // :272-273
                    t.prev_U = pu;
                    good_round = 1;
                }
            }
            n_val += o.n;                                // :283
            depth += 1;
            double qq[2];
            hipLaunchKernelGGL(mi::cb_diff_dots, dim3(1), dim3(64), 0, 0, dd, m.vec(NV_POS_T), m.vec(NV_NEG_T), m.vec(NV_NEG_P), m.vec(NV_POS_P),
                               m.vec(NV_TMP), m.scal.as<double>());
            CB_TRY(m.fetch(2, qq));
            s_val = o.s * ((qq[0] >= 0.0) ? 1 : 0) * ((qq[1] >= 0.0) ? 1 : 0);   // :286-289
        }
        if (draw < n_adapt) {                            // :294-302
            const double it = (double)(draw + 1);
            h_val = h_val + (1.0 / (it + settings->t0_val)) * (settings->target_accept_rate - (o.alpha / (double)o.n_alpha) - h_val);
            step = mi::det_exp(mu_val - h_val * std::sqrt(it) / settings->gamma_val);
            eps_bar = eps_bar * mi::det_exp(mi::det_pow(it, -settings->kappa_val) * (mi::det_log(step) - mi::det_log(eps_bar)));
        } else {
            step = eps_bar;
        }
        if (draw >= n_burnin) {                          // :306-309
            n_acc += (uint64_t)good_round;
            HIP_TRY(hipMemcpy(row.data(), m.vec(NV_PREV), d * 8, hipMemcpyDeviceToHost));
            for (uint64_t j = 0; j < d; ++j) draws_out[(draw - n_burnin) + j * n_keep] = row[j];
        }
    }
    HIP_TRY(hipDeviceSynchronize());
    if (n_accept_draws) *n_accept_draws = n_acc;
    if (step_size_out) *step_size_out = step;
    return MI_OK;
}

}  // extern "C"
