// mi_mcmc.hip -- C ABI (include/mi_mcmc.h) of the MI355X many-chain HMC / MALA / NUTS engine.
// Host side of the boundary: validates the POD mirrors of algo_settings_t, stages targets and
// chain state in HBM, launches the gfx950 kernels.  No CPU fallback: anything the device path
// does not implement returns MI_ERR_UNSUPPORTED.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mi_mcmc.h"
#include "host_common.hpp"
#include "host_linalg.hpp"
#include "det_math.hpp"
#include "hmc_dense.hpp"
#include "nuts_dense.hpp"
#include "nuts_tile.hpp"
#include "nuts_async.hpp"
#include "mala_dense.hpp"
#include "rwmh_dense.hpp"
#include "hmc_diag.hpp"
#include "logistic_launch.hpp"
#include "launchers.hpp"
#include "gemm_samplers.hpp"
#include "small_samplers.hpp"
#include "literal_host.hpp"
#include "tile_samplers.hpp"

// round time of the few-chain launch shapes of the plain HMC kernel relative to the default (two waves per SIMD), d = 128
#ifndef MI_HMC_COST_1WAVE
#define MI_HMC_COST_1WAVE 0.535       // 27.1 ms against 50.8 ms per round (tools/hmc_shapes.py)
#define MI_HMC_COST_SPLIT2 0.287      // 14.55 ms
#define MI_HMC_COST_SPLIT4X2 0.278    // 14.12 ms
#define MI_HMC_COST_SPLIT4 0.153      // 7.78 ms
#endif

namespace mi {
namespace host {
std::string& last_error() { thread_local std::string e; return e; }
std::string& last_kernel() { thread_local std::string k; return k; }
int device_inverse(const double* A, size_t d, double* Ainv);                 // linalg_device.hip
int device_cholesky_lower(const double* A, size_t d, double* L);
}  // namespace host
namespace {
// INV / CHOL_LOWER of a dense precond_mat with d >= 64 run on the device (host_linalg.hpp: same operations per element, same bits)
const bool g_linalg_installed = [] {
    LinalgAccel& a = linalg_accel();
    a.inverse = host::device_inverse;
    a.cholesky_lower = host::device_cholesky_lower;
    return true;
}();
}  // namespace
void note_kernel(const char* fmt, ...)
{
    char buf[160];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    host::last_kernel() = buf;
}
// The test hook's cap on the persistent grids is read ONCE per sampler call (snapshot_grid_cap, at the first thing every route does) into a
// thread-local: the workspace of a call is sized and its kernel launched from the same value even if another thread moves the cap in between
// (sized with one grid, launched with a larger one, the kernel would write point records out of bounds).
namespace { std::atomic<uint64_t> g_test_grid_cap{0}; thread_local uint64_t t_grid_cap = 0; }
uint64_t test_grid_cap() { return t_grid_cap; }
namespace { void snapshot_grid_cap() { t_grid_cap = g_test_grid_cap.load(std::memory_order_relaxed); } }
}  // namespace mi

// test hook, declared in mi_mcmc_probes.h (not in the product header): see launch_common.hpp
extern "C" void mi_mcmc_test_set_grid_cap(uint32_t max_workgroups) { mi::g_test_grid_cap.store(max_workgroups, std::memory_order_relaxed); }

namespace {

using mi::host::fail;
using mi::host::DevBuf;

// Per-(device, stream) workspace cache.  Kernels of one stream are ordered, so one buffer serves consecutive calls; it is
// reallocated (after a stream sync) only when a call needs more, and released by mi_mcmc_release_workspace().  A call holds
// the entry's mutex from ws_get() until its kernels are enqueued (WsLease), so a second host thread on the same stream can
// neither free nor regrow the buffer between "pointer handed out" and "kernel enqueued"; after that the stream orders them.
// (hipMallocAsync / hipFreeAsync pool memory was observed to hand the pack -> sample kernel pair of the logistic path
// buffers whose first blocks read back as zeros; a plain cached hipMalloc does not.)
struct WsEntry { void* p = nullptr; size_t cap = 0; std::mutex mu; };
std::mutex g_ws_mu;
std::map<std::pair<int, hipStream_t>, std::unique_ptr<WsEntry>> g_ws;

struct WsLease {
    std::unique_lock<std::mutex> lk;
    void* p = nullptr;
    template <class T> T* as() const { return static_cast<T*>(p); }
};

int ws_get(hipStream_t st, size_t bytes, WsLease& lease)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    WsEntry* w = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_ws_mu);
        std::unique_ptr<WsEntry>& slot = g_ws[std::make_pair(dev, st)];
        if (!slot) slot.reset(new WsEntry);
        w = slot.get();
    }
    lease.lk = std::unique_lock<std::mutex>(w->mu);
    if (w->cap < bytes) {
        if (w->p) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipFree(w->p)); w->p = nullptr; w->cap = 0; }
        HIP_TRY(hipMalloc(&w->p, bytes));
        w->cap = bytes;
    }
    lease.p = w->p;
    return MI_OK;
}

// bytes the cache holds for (current device, st) right now
size_t ws_cached_bytes(hipStream_t st)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    std::lock_guard<std::mutex> lk(g_ws_mu);
    auto it = g_ws.find(std::make_pair(dev, st));
    return (it == g_ws.end() || !it->second) ? 0 : it->second->cap;
}

#ifndef MI_NUTS_MOMENTA_MAX_BYTES
#define MI_NUTS_MOMENTA_MAX_BYTES ((size_t)24 << 30)     // the table of momenta of a nuts run (nuts_memo.hpp): 13.6 GB on BASELINE configs[3]
#endif

// frees the cached workspace of (current device, st); all_streams: every entry of the current device
int ws_release(hipStream_t st, bool all_streams, uint64_t* freed)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::vector<WsEntry*> victims;
    {
        std::lock_guard<std::mutex> lk(g_ws_mu);
        for (auto& kv : g_ws)
            if (kv.first.first == dev && (all_streams || kv.first.second == st)) victims.push_back(kv.second.get());
    }
    uint64_t total = 0;
    for (WsEntry* w : victims) {
        std::lock_guard<std::mutex> lk(w->mu);
        if (!w->p) continue;
        HIP_TRY(all_streams ? hipDeviceSynchronize() : hipStreamSynchronize(st));
        HIP_TRY(hipFree(w->p));
        total += w->cap;
        w->p = nullptr; w->cap = 0;
    }
    if (freed) *freed = total;
    return MI_OK;
}

// ---- literal replay of the non-finite regime (literal.hpp).  The plain throughput kernels flag the chains whose energies /
// proposal densities went non-finite (nf_flag[c] = 1, nf_flag[C] = 1) and leave their outputs alone; the literal kernel, enqueued
// right behind them on the same stream, replays exactly those chains from their initial values (it returns at once when nothing
// was flagged: one scalar load per workgroup).  Flags and the replay's workspace ride behind the caller's own workspace in the
// per-stream cache.
struct ReplayWs {
    uint32_t* flag = nullptr;     // [C + 1]
    double* tbuf = nullptr;       // t_doubles: the target's matrix transposed (literal.hpp reads matrices column-major: coalesced)
    size_t t_doubles = 0;
    double* work = nullptr;
    size_t stride = 0;            // doubles per workgroup
    unsigned n_wg = 0;
    size_t own_bytes = 0;         // the caller's part, rounded up
    size_t total_bytes = 0;
};
// needs_matrix: the replay reads a d x max(d, n_rows) matrix transposed (DENSE precision, LOGISTIC design matrix, a dense precond_mat);
// the separable ISO / DIAG targets of the "any d" elementwise path read none, and their workspace must not grow as d^2
ReplayWs replay_layout(size_t own_bytes, uint64_t C, uint32_t d, uint32_t n_rows, bool mala_bounded, bool needs_matrix = true)
{
    ReplayWs r;
    r.t_doubles = needs_matrix ? (((size_t)d * std::max<size_t>(d, n_rows) + 31) & ~(size_t)31) : 32;
    r.own_bytes = (own_bytes + 255) & ~(size_t)255;
    r.stride = mi::lit::lit_work_doubles(d, n_rows, mala_bounded);
    r.n_wg = (unsigned)std::min<uint64_t>(C, mala_bounded ? 128u : 512u);
    const size_t flag_bytes = ((C + 1) * sizeof(uint32_t) + 255) & ~(size_t)255;
    r.total_bytes = r.own_bytes + flag_bytes + (r.t_doubles + (size_t)r.n_wg * r.stride) * sizeof(double);
    return r;
}
int replay_bind(ReplayWs& r, void* base, uint64_t C, hipStream_t st)
{
    char* b = static_cast<char*>(base);
    r.flag = reinterpret_cast<uint32_t*>(b + r.own_bytes);
    const size_t flag_bytes = ((C + 1) * sizeof(uint32_t) + 255) & ~(size_t)255;
    r.tbuf = reinterpret_cast<double*>(b + r.own_bytes + flag_bytes);
    r.work = r.tbuf + r.t_doubles;
    HIP_TRY(hipMemsetAsync(r.flag, 0, (C + 1) * sizeof(uint32_t), st));
    return MI_OK;
}
// out[c * rows + r] = in[r * cols + c] on the device (small matrices: the d x d precision, the n x d design matrix)
__global__ void transpose_small_kernel(const double* __restrict__ in, double* __restrict__ out, uint32_t rows, uint32_t cols)
{
    const size_t n = (size_t)rows * cols;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t c = e / rows, r = e % rows;        // consecutive threads write consecutive addresses
        out[e] = in[r * cols + c];
    }
}
int transpose_on_device(const double* in, double* out, uint32_t rows, uint32_t cols, hipStream_t st)
{
    const size_t n = (size_t)rows * cols;
    hipLaunchKernelGGL(transpose_small_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 1024)), dim3(256), 0, st, in, out, rows, cols);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

// the Gaussian kinds as the literal kernels read them: ISO / DIAG are ELEMENT-WISE targets in the oracle (prec_i * theta_i), DENSE a
// mat-vec.  P_dense: the d*d device matrix of the MFMA path (diagonal for ISO / DIAG), or nullptr with prec_vec (d values / nullptr)
int lit_gauss_target(mi::lit::LitTarget& t, int kind, uint32_t d, const double* P_dense, const double* prec_vec, double* tbuf, hipStream_t st)
{
    t = mi::lit::LitTarget{};
    t.d = d;
    if (kind == MI_TARGET_GAUSS_DENSE) {                 // the literal kernels read the precision transposed
        t.kind = mi::lit::LIT_DENSE; t.prec = tbuf;
        const int rc = transpose_on_device(P_dense, tbuf, d, d, st);
        if (rc) return rc;
    }
    else if (kind == MI_TARGET_GAUSS_DIAG) {
        t.kind = mi::lit::LIT_DIAG;
        if (P_dense) { t.prec = P_dense; t.prec_stride = d + 1; } else { t.prec = prec_vec; t.prec_stride = 1; }
    } else t.kind = mi::lit::LIT_ISO;
    mi::lit::lit_orders(t);
    return MI_OK;
}
void lit_common(mi::lit::LitParams& p, const mi_settings* s, const mi_chains* dev_chains, const ReplayWs& r, bool all_chains)
{
    p.C = dev_chains->n_chains; p.chain0 = dev_chains->chain0;
    p.theta = dev_chains->theta; p.draws = dev_chains->draws; p.n_accept = dev_chains->n_accept; p.n_leap = dev_chains->n_leapfrogs;
    p.seed = s->rng_seed_value;
    p.n_burnin = (uint32_t)s->n_burnin_draws; p.n_keep = (uint32_t)s->n_keep_draws; p.n_leap_steps = (uint32_t)s->n_leap_steps;
    p.draw0 = (uint32_t)dev_chains->draw0; p.eps = s->step_size;
    p.flag = all_chains ? nullptr : r.flag;
    p.any = all_chains ? nullptr : r.flag + p.C;
    p.work = r.work; p.work_stride = r.stride;
}

// device copies of a LitPrep (literal_host.hpp) and the LitParams pointers into them.  The buffers die with the LitDev, so a caller
// that uploaded anything synchronises the stream before it returns.
struct LitDev {
    DevBuf bt, lb, ub, m, ms, mi, Mfull, Lchol, Minv, sinv_diag, Sinv;
    bool any = false;
};
int lit_upload(const mi::lit::LitPrep& pr, uint32_t d, bool want_bounds, LitDev& dv, mi::lit::LitParams& p)
{
    auto up = [&](DevBuf& b, const void* src, size_t bytes) -> int {
        HIP_TRY(b.alloc(bytes));
        HIP_TRY(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
        dv.any = true;
        return MI_OK;
    };
    int rc;
    p.precond = pr.precond;
    p.rs = pr.rs; p.log_det = pr.log_det; p.cons_term = pr.cons_term;
    if (want_bounds) {
        if ((rc = up(dv.bt, pr.bt.data(), d * sizeof(int)))) return rc;
        if ((rc = up(dv.lb, pr.lb.data(), d * 8))) return rc;
        if ((rc = up(dv.ub, pr.ub.data(), d * 8))) return rc;
        p.vals_bound = 1; p.btype = dv.bt.as<int>(); p.lb = dv.lb.as<double>(); p.ub = dv.ub.as<double>();
    }
    if (pr.precond == 1) {
        if ((rc = up(dv.m, pr.m.data(), d * 8))) return rc;
        if ((rc = up(dv.ms, pr.m_sqrt.data(), d * 8))) return rc;
        if ((rc = up(dv.mi, pr.m_inv.data(), d * 8))) return rc;
        p.m = dv.m.as<double>(); p.m_sqrt = dv.ms.as<double>(); p.m_inv = dv.mi.as<double>();
    } else if (pr.precond == 2) {
        const size_t mb = (size_t)d * d * 8;
        if ((rc = up(dv.Mfull, pr.Mfull.data(), mb))) return rc;
        if ((rc = up(dv.Lchol, pr.Lchol.data(), mb))) return rc;
        if ((rc = up(dv.Minv, pr.Minv.data(), mb))) return rc;
        p.Mfull = dv.Mfull.as<double>(); p.Lchol = dv.Lchol.as<double>(); p.Minv = dv.Minv.as<double>();
    }
    if (!pr.sinv_diag.empty()) { if ((rc = up(dv.sinv_diag, pr.sinv_diag.data(), d * 8))) return rc; p.sinv_diag = dv.sinv_diag.as<double>(); }
    if (!pr.Sinv.empty()) { if ((rc = up(dv.Sinv, pr.Sinv.data(), (size_t)d * d * 8))) return rc; p.Sinv = dv.Sinv.as<double>(); }
    return MI_OK;
}

// ---- the literal kernels as the device path of everything the tiled kernels do not implement: dense-gradient targets with d > 128,
// the logistic target with bounds / a preconditioner / nuts beyond d = 8 or with d > 512, max_tree_depth > 10.  One workgroup per
// chain, the reference's operations as written (literal.hpp): O(d^2) per leapfrog step and chain, no MFMA, no shared tiles -- a
// completeness path, not a throughput path.  algo: 0 hmc, 1 mala, 2 nuts, 3 rwmh.
int run_literal(const char* who, int algo, const mi_target* target, const mi_settings* settings, mi_chains* chains, hipStream_t st);

// nuts continuation rules (mi_mcmc.h: mi_chains.draw0 / nuts_adapt_state); *n_adapt = the RUN's adaptation window
int nuts_continuation(const mi_settings* s, const mi_chains* c, uint32_t* n_adapt)
{
    *n_adapt = (uint32_t)(s->n_adapt_draws > 0xffffffffULL ? 0xffffffffULL : s->n_adapt_draws);
    if (c->draw0 > 0) {
        if (!c->step_size) return fail(MI_ERR_BAD_ARG, "nuts: a continuation needs chains.step_size (the step sizes of the previous call)");
        if (c->draw0 <= s->n_adapt_draws && !c->nuts_adapt_state)
            return fail(MI_ERR_BAD_ARG, "nuts: a continuation that starts inside or at the end of the adaptation window (draw0 <= n_adapt_draws) needs chains.nuts_adapt_state of the previous call");
    }
    return MI_OK;
}

int check_common(const mi_target* t, const mi_settings* s, const mi_chains* c, bool mass_allowed = false)
{
    if (!t || !s || !c) return fail(MI_ERR_BAD_ARG, "null target / settings / chains");
    mi::snapshot_grid_cap();
    if (t->struct_size != sizeof(mi_target) || s->struct_size != sizeof(mi_settings) ||
        c->struct_size != sizeof(mi_chains))
        return fail(MI_ERR_BAD_ARG, "struct_size mismatch (header / library version skew)");
    if (c->n_leapfrogs_executed && !c->n_leapfrogs) return fail(MI_ERR_BAD_ARG, "n_leapfrogs_executed needs n_leapfrogs next to it");
    if (t->d == 0 || c->n_chains == 0) return fail(MI_ERR_BAD_ARG, "d and n_chains must be positive");
    if (!c->theta) return fail(MI_ERR_BAD_ARG, "chains.theta is required");
    if (c->mass_diag && !mass_allowed) return fail(MI_ERR_UNSUPPORTED, "chains.mass_diag (per-chain diagonal masses) is implemented for mi_mcmc_hmc_run only");
    if (c->mass_diag && s->precond_mat) return fail(MI_ERR_BAD_ARG, "chains.mass_diag and settings.precond_mat are two mass matrices: one of them must be NULL");
    if (c->draw0 + s->n_burnin_draws + s->n_keep_draws > 0xffffffffULL) return fail(MI_ERR_BAD_ARG, "draw0 + draws exceeds the 32-bit draw counter");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(MI_ERR_NO_DEVICE, "no HIP device visible: the engine has no CPU path");
    (void)hipGetLastError();      // a stale non-fatal status of the caller's own HIP use (e.g. hipErrorNotReady of an event query) is not ours
    return MI_OK;
}

// Dense d x d precision on the device for the Gaussian kinds (ISO / DIAG expand to a diagonal
// matrix: an fma chain over exact zeros reproduces prec_i * theta_i bit for bit).
int dense_precision_on_device(const mi_target* t, DevBuf& owned, const double** P_dev, hipStream_t st)
{
    const size_t d = t->d;
    if (t->kind == MI_TARGET_GAUSS_DENSE) {
        if (!t->prec) return fail(MI_ERR_BAD_ARG, "GAUSS_DENSE needs prec (d*d)");
        if (t->mem == MI_MEM_DEVICE) { *P_dev = t->prec; return MI_OK; }
        HIP_TRY(owned.alloc(d * d * sizeof(double)));
        HIP_TRY(hipMemcpyAsync(owned.p, t->prec, d * d * sizeof(double), hipMemcpyHostToDevice, st));
        *P_dev = owned.as<double>();
        return MI_OK;
    }
    std::vector<double> diag(d, 1.0);
    if (t->kind == MI_TARGET_GAUSS_DIAG) {
        if (!t->prec) return fail(MI_ERR_BAD_ARG, "GAUSS_DIAG needs prec (d)");
        if (t->mem == MI_MEM_DEVICE)
            HIP_TRY(hipMemcpy(diag.data(), t->prec, d * sizeof(double), hipMemcpyDeviceToHost));
        else
            std::memcpy(diag.data(), t->prec, d * sizeof(double));
    } else if (t->kind != MI_TARGET_GAUSS_ISO) {
        return fail(MI_ERR_UNSUPPORTED, "target kind %d has no dense-precision form", t->kind);
    }
    std::vector<double> P(d * d, 0.0);
    for (size_t i = 0; i < d; ++i) P[i * d + i] = diag[i];
    HIP_TRY(owned.alloc(d * d * sizeof(double)));
    HIP_TRY(hipMemcpy(owned.p, P.data(), d * d * sizeof(double), hipMemcpyHostToDevice));
    *P_dev = owned.as<double>();
    return MI_OK;
}

// host <-> device staging of one mi_chains shard
struct StagedChains {
    DevBuf theta, draws, n_accept, step, n_leap, depth, mass, adapt, n_exec;
    mi_chains dev;   // device-pointer view
    bool exec_written = false;   // the kernel wrote n_leapfrogs_executed itself (nuts_memo.hpp); otherwise it is a copy of n_leapfrogs
};

int stage_in(const mi_chains* c, uint64_t d, uint64_t n_keep, StagedChains& sc, hipStream_t st, uint64_t n_total = 0)
{
    // (every route passes here -- the tile- and user-target entry points do not go through check_common: stage_out copies n_leapfrogs into
    //  n_leapfrogs_executed for every kernel that executes what it counts, so the latter without the former is a bad argument, not a late HIP error)
    if (c->n_leapfrogs_executed && !c->n_leapfrogs) return fail(MI_ERR_BAD_ARG, "n_leapfrogs_executed needs n_leapfrogs next to it");
    mi::snapshot_grid_cap();
    sc.dev = *c;
    if (c->mem == MI_MEM_DEVICE) return MI_OK;
    const size_t C = c->n_chains;
    HIP_TRY(sc.theta.alloc(d * C * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(sc.theta.p, c->theta, d * C * sizeof(double), hipMemcpyHostToDevice, st));
    sc.dev.theta = sc.theta.as<double>();
    if (c->draws) { HIP_TRY(sc.draws.alloc(n_keep * d * C * sizeof(double))); sc.dev.draws = sc.draws.as<double>(); }
    if (c->n_accept) { HIP_TRY(sc.n_accept.alloc(C * sizeof(uint64_t))); sc.dev.n_accept = sc.n_accept.as<uint64_t>(); }
    if (c->step_size) {
        HIP_TRY(sc.step.alloc(C * sizeof(double))); sc.dev.step_size = sc.step.as<double>();
        // in: the adapted step sizes of a nuts continuation; out: written by nuts only -- every other sampler hands the
        // caller's values back unchanged (never uninitialised device memory)
        HIP_TRY(hipMemcpyAsync(sc.step.p, c->step_size, C * sizeof(double), hipMemcpyHostToDevice, st));
    }
    if (c->n_leapfrogs) {
        HIP_TRY(sc.n_leap.alloc(C * sizeof(uint64_t))); sc.dev.n_leapfrogs = sc.n_leap.as<uint64_t>();
        HIP_TRY(hipMemsetAsync(sc.n_leap.p, 0, C * sizeof(uint64_t), st));    // samplers without leapfrog steps (mala, rwmh) report 0
    }
    if (c->n_leapfrogs_executed) {
        HIP_TRY(sc.n_exec.alloc(C * sizeof(uint64_t))); sc.dev.n_leapfrogs_executed = sc.n_exec.as<uint64_t>();
        HIP_TRY(hipMemsetAsync(sc.n_exec.p, 0, C * sizeof(uint64_t), st));
    }
    if (c->nuts_depth) { HIP_TRY(sc.depth.alloc(n_total * C * sizeof(uint32_t))); sc.dev.nuts_depth = sc.depth.as<uint32_t>(); }
    if (c->nuts_adapt_state) {                           // in (a continuation inside the adaptation window) / out
        HIP_TRY(sc.adapt.alloc(3 * C * sizeof(double))); sc.dev.nuts_adapt_state = sc.adapt.as<double>();
        HIP_TRY(hipMemcpyAsync(sc.adapt.p, c->nuts_adapt_state, 3 * C * sizeof(double), hipMemcpyHostToDevice, st));
    }
    if (c->mass_diag) {
        HIP_TRY(sc.mass.alloc(d * C * sizeof(double))); sc.dev.mass_diag = sc.mass.as<double>();
        HIP_TRY(hipMemcpyAsync(sc.mass.p, c->mass_diag, d * C * sizeof(double), hipMemcpyHostToDevice, st));
    }
    sc.dev.mem = MI_MEM_DEVICE;
    return MI_OK;
}

int stage_out(const mi_chains* c, uint64_t d, uint64_t n_keep, StagedChains& sc, hipStream_t st, uint64_t n_total = 0)
{
    // every kernel but nuts_gauss_memo_kernel executes exactly the leapfrogs it counts
    if (c->n_leapfrogs_executed && !sc.exec_written)
        HIP_TRY(hipMemcpyAsync(sc.dev.n_leapfrogs_executed, sc.dev.n_leapfrogs, c->n_chains * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
    if (c->mem == MI_MEM_DEVICE) return MI_OK;
    const size_t C = c->n_chains;
    HIP_TRY(hipMemcpyAsync(c->theta, sc.dev.theta, d * C * sizeof(double), hipMemcpyDeviceToHost, st));
    if (c->draws) HIP_TRY(hipMemcpyAsync(c->draws, sc.dev.draws, n_keep * d * C * sizeof(double), hipMemcpyDeviceToHost, st));
    if (c->n_accept) HIP_TRY(hipMemcpyAsync(c->n_accept, sc.dev.n_accept, C * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    if (c->step_size) HIP_TRY(hipMemcpyAsync(c->step_size, sc.dev.step_size, C * sizeof(double), hipMemcpyDeviceToHost, st));
    if (c->n_leapfrogs) HIP_TRY(hipMemcpyAsync(c->n_leapfrogs, sc.dev.n_leapfrogs, C * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    if (c->n_leapfrogs_executed) HIP_TRY(hipMemcpyAsync(c->n_leapfrogs_executed, sc.dev.n_leapfrogs_executed, C * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    if (c->nuts_depth) HIP_TRY(hipMemcpyAsync(c->nuts_depth, sc.dev.nuts_depth, n_total * C * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    if (c->nuts_adapt_state) HIP_TRY(hipMemcpyAsync(c->nuts_adapt_state, sc.dev.nuts_adapt_state, 3 * C * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return MI_OK;
}

// n_leapfrogs for the samplers whose kernels do not count leapfrog steps themselves (mala / rwmh: 0; hmc on the logistic
// target: n_total * n_leap_steps), so that the caller's array is defined after every call, host or device memory
__global__ void fill_u64_kernel(uint64_t* out, uint64_t n, uint64_t v)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v;
}

// identity tables of the general kernel variants (no bounds, unit mass) in stream-ordered workspace memory: what a replay of the
// plain case through a general variant reads (no host buffer has to outlive the call)
// nuts_gauss_memo_kernel: a flagged chain was replayed by the general variant, which executes every leapfrog it counts
__global__ void copy_flagged_counts_kernel(const uint32_t* flag, const uint64_t* n_leap, uint64_t* n_exec, uint64_t C)
{
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C && flag[c] != 0u) n_exec[c] = n_leap[c];
}

__global__ void fill_identity_tables_kernel(int* bt, double* lb, double* ub, double* ms, double* mi, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { bt[i] = 1; lb[i] = 0.0; ub[i] = 0.0; ms[i] = 1.0; mi[i] = 1.0; }
}

int fill_n_leap(uint64_t* dev_ptr, uint64_t n, uint64_t v, hipStream_t st)
{
    if (!dev_ptr) return MI_OK;
    hipLaunchKernelGGL(fill_u64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dev_ptr, n, v);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

// LDS-staged logistic kernels (logistic_lds.hip): workspace from the per-stream cache, launch in their own translation unit
int launched(const char* what, int hip_err);
struct LdsTables;
void lds_tables_replay(const LdsTables& t, const mi::LogitParams& q, mi::lit::LitParams& lp);
bool lds_tables_active(const LdsTables* t);
int launch_logit(int algo, mi::LogitParams prm, const double* X_dev, const double* y_dev, hipStream_t st,
                 const mi_settings* settings, const mi_chains* dev_chains, int lds_target = mi::LOGIT_TARGET_LOGISTIC, const LdsTables* lt = nullptr)
{
    WsLease base;
    const bool replay = algo != mi::LOGIT_RWMH;          // rwmh forms no product with a vector that can be non-finite (rwmh.cpp:126)
    ReplayWs rp = replay_layout(mi::logit_lds_workspace_bytes(prm.d, prm.NB, prm.C, lds_target), prm.C, prm.d, prm.n_rows, false);
    int rcw = ws_get(st, replay ? rp.total_bytes : rp.own_bytes, base);
    if (rcw) return rcw;
    if (replay) {
        rcw = replay_bind(rp, base.p, prm.C, st);
        if (rcw) return rcw;
        prm.nf_flag = rp.flag;
    }
    // hmc / mala with a DENSE precond_mat (prm.L_rm and the sampler's other matrices from the caller): their block images live in a buffer of their own
    const bool dense_m = prm.L_rm != nullptr;
    DevBuf mws;
    if (dense_m) HIP_TRY(mws.alloc(mi::logit_lds_dense_m_bytes(prm.d, prm.C, lds_target, algo)));
    const int e = dense_m ? mi::logit_lds_launch_dense_m(algo, prm, X_dev, y_dev, base.p, mws.p, st, lds_target)
                          : mi::logit_lds_launch(algo, prm, X_dev, y_dev, base.p, st, lds_target);
    if (e != 0) return fail(MI_ERR_HIP, "LDS-streamed kernel launch: %s", hipGetErrorString((hipError_t)e));
    LitDev ldev;
    if (replay) {                                       // chains that reached the non-finite regime: replayed literally (literal.hpp)
        mi::lit::LitParams lp{};
        rcw = transpose_on_device(X_dev, rp.tbuf, prm.n_rows, prm.d, st);      // eta = X beta resp. P x read the matrix transposed (literal.hpp)
        if (rcw) return rcw;
        if (lds_target == mi::LOGIT_TARGET_DENSE) {
            lp.t.kind = mi::lit::LIT_DENSE; lp.t.d = prm.d; lp.t.prec = rp.tbuf;
        } else {
            lp.t.kind = mi::lit::LIT_LOGISTIC; lp.t.d = prm.d; lp.t.n_rows = prm.n_rows; lp.t.X = X_dev; lp.t.y = y_dev;
            lp.t.Xt = rp.tbuf;
        }
        mi::lit::lit_orders(lp.t);
        lit_common(lp, settings, dev_chains, rp, false);
        lp.rs = prm.rs; lp.log_det = prm.log_det; lp.cons_term = prm.cons_term;
        if (dense_m) {                                   // the replay's own copies (transposed: literal_host.hpp)
            mi::lit::LitPrep prep;
            rcw = mi::lit::lit_prepare(algo == mi::LOGIT_MALA ? 1 : 0, prm.d, settings->step_size, 0, nullptr, nullptr, settings->precond_mat, prep);
            if (rcw) return rcw;
            rcw = lit_upload(prep, prm.d, false, ldev, lp);      // (mala: INV(Sigma), LOG_DET and the constant term with it)
            if (rcw) return rcw;
        }
        else if (lds_tables_active(lt)) lds_tables_replay(*lt, prm, lp);      // hmc: bounds and / or a diagonal precond_mat
        else if (prm.m_sqrt != nullptr) {                // a diagonal precond_mat (mala: m, m_sqrt, s_inv)
            lp.precond = 1; lp.m = prm.m; lp.m_sqrt = prm.m_sqrt; lp.m_inv = prm.m_inv; lp.sinv_diag = prm.s_inv;
        }
        rcw = launched("LDS-streamed kernel (literal replay)", mi::launch_literal(algo == mi::LOGIT_MALA ? 1 : 0, lp, rp.n_wg, st));
        if (rcw) return rcw;
    }
    if (dense_m) HIP_TRY(hipStreamSynchronize(st));      // the images and the replay's matrices are ours
    return MI_OK;
}

// d x d row-major -> MFMA A-fragment order [t][s][lane] = M[16 t + (lane & 15)][4 s + (lane >> 4)], zero padded to 16 nt
// (what stage_precision writes into LDS): the layout of matrices a kernel reads straight from global memory (d > 64:
// INV / CHOL_LOWER of a dense precond_mat do not fit into LDS next to the precision, hmc_dense.hpp: matvec_mfma_g)
std::vector<double> pack_fragments(const std::vector<double>& M, size_t d, int nt)
{
    const int ns = 4 * nt;
    std::vector<double> out((size_t)nt * ns * 64, 0.0);
    for (int t = 0; t < nt; ++t)
        for (int sl = 0; sl < ns; ++sl)
            for (int l = 0; l < 64; ++l) {
                const size_t row = 16 * (size_t)t + (l & 15), col = 4 * (size_t)sl + (l >> 4);
                if (row < d && col < d) out[((size_t)t * ns + sl) * 64 + l] = M[row * d + col];
            }
    return out;
}
// the device copy of a second / third matrix: row-major for d <= 64 (the kernel stages it into LDS), fragment order beyond
int upload_matrix(const std::vector<double>& M, size_t d, DevBuf& buf)
{
    const int nt = (int)((d + 15) / 16);
    if (nt > 4) {
        const std::vector<double> f = pack_fragments(M, d, 8);
        HIP_TRY(buf.alloc(f.size() * 8));
        HIP_TRY(hipMemcpy(buf.p, f.data(), f.size() * 8, hipMemcpyHostToDevice));
    } else {
        HIP_TRY(buf.alloc(d * d * 8));
        HIP_TRY(hipMemcpy(buf.p, M.data(), d * d * 8, hipMemcpyHostToDevice));
    }
    return MI_OK;
}

// Per-dimension tables of the general kernel variants (settings.vals_bound and / or a diagonal precond_mat):
// determine_bounds_type (determine_bounds_type.hpp:27-57: 1 none, 2 lower, 3 upper, 4 both), the bounds, and the diagonal of
// precond_mat with its CHOL_LOWER / INV (element-wise sqrt / reciprocal for a diagonal matrix, as the oracle's BMO shim gives).
struct GeneralTables {
    bool active = false;
    bool dense = false;                       // precond_mat has off-diagonal entries: INV / CHOL_LOWER as dense matrices
    DevBuf minv_full, l_full;
    std::vector<double> m, m_sqrt, m_inv;     // host copies (diagonal)
    DevBuf bt, lb, ub, m_dev, ms_dev, mi_dev;
};

int general_tables(const char* who, const mi_settings* s, uint64_t d, GeneralTables& g, bool allow_dense = false, bool dense_beyond_64 = false)
{
    g.active = s->vals_bound != 0 || s->precond_mat != nullptr;
    if (!g.active) return MI_OK;
    if (d > 128) return fail(MI_ERR_UNSUPPORTED, "%s: vals_bound / precond_mat with d > 128 is not implemented", who);
    if (s->vals_bound && (!s->lower_bounds || !s->upper_bounds)) return fail(MI_ERR_BAD_ARG, "%s: vals_bound needs lower_bounds and upper_bounds", who);
    g.m.assign(d, 1.0); g.m_sqrt.assign(d, 1.0); g.m_inv.assign(d, 1.0);
    if (s->precond_mat)
        for (uint64_t i = 0; i < d; ++i)
            for (uint64_t k = 0; k < d; ++k) {
                const double v = s->precond_mat[i * d + k];
                if (i != k && v != 0.0) {
                    if (!allow_dense) return fail(MI_ERR_UNSUPPORTED, "%s: only a diagonal precond_mat is implemented on the device path", who);
                    g.dense = true;
                }
                if (i == k) { g.m[i] = v; g.m_sqrt[i] = __builtin_sqrt(v); g.m_inv[i] = 1.0 / v; }
            }
    if (g.dense) {
        if (d > 64 && !dense_beyond_64) return fail(MI_ERR_UNSUPPORTED, "%s: a dense precond_mat / cov_mat is implemented for d <= 64 (diagonal: d <= 128)", who);
        std::vector<double> Minv, L;
        int rcu = host_inverse(s->precond_mat, d, Minv); if (rcu) return rcu;
        rcu = host_cholesky_lower(s->precond_mat, d, L); if (rcu) return rcu;
        rcu = upload_matrix(Minv, d, g.minv_full); if (rcu) return rcu;
        rcu = upload_matrix(L, d, g.l_full); if (rcu) return rcu;
    }
    std::vector<int> bt(d, 1);
    std::vector<double> lbv(d, 0.0), ubv(d, 0.0);
    if (s->vals_bound)
        for (uint64_t i = 0; i < d; ++i) {
            lbv[i] = s->lower_bounds[i]; ubv[i] = s->upper_bounds[i];
            const bool fl = std::isfinite(lbv[i]), fu = std::isfinite(ubv[i]);
            bt[i] = (fl && fu) ? 4 : (fl && !fu) ? 2 : (!fl && fu) ? 3 : 1;
        }
    HIP_TRY(g.bt.alloc(d * sizeof(int))); HIP_TRY(g.lb.alloc(d * 8)); HIP_TRY(g.ub.alloc(d * 8));
    HIP_TRY(g.m_dev.alloc(d * 8)); HIP_TRY(g.ms_dev.alloc(d * 8)); HIP_TRY(g.mi_dev.alloc(d * 8));
    HIP_TRY(hipMemcpy(g.bt.p, bt.data(), d * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(g.lb.p, lbv.data(), d * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(g.ub.p, ubv.data(), d * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(g.m_dev.p, g.m.data(), d * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(g.ms_dev.p, g.m_sqrt.data(), d * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(g.mi_dev.p, g.m_inv.data(), d * 8, hipMemcpyHostToDevice));
    return MI_OK;
}

// Kernel families live in their own translation units (launchers.hpp); these wrappers turn their hipError_t into a status.
int launched(const char* what, int hip_err)
{
    if (hip_err != 0) return fail(MI_ERR_HIP, "%s kernel launch: %s", what, hipGetErrorString((hipError_t)hip_err));
    return MI_OK;
}

// per-chain diagonal masses (mi_chains.mass_diag, [d][C]): CHOL_LOWER and INV of a diagonal matrix are the element-wise sqrt(m) and
// 1 / m (what the oracle's Cholesky / Gauss-Jordan give for diag(m)), formed on the device next to the masses
__global__ void chain_mass_tables_kernel(const double* __restrict__ mass, uint64_t n, double* __restrict__ m_sqrt, double* __restrict__ m_inv)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) { const double m = mass[e]; m_sqrt[e] = __builtin_sqrt(m); m_inv[e] = 1.0 / m; }
}
struct ChainMass { DevBuf ms, mi; };
int chain_mass_tables(const double* mass_dev, uint64_t d, uint64_t C, ChainMass& t, hipStream_t st)
{
    const uint64_t n = d * C;
    HIP_TRY(t.ms.alloc(n * 8)); HIP_TRY(t.mi.alloc(n * 8));
    hipLaunchKernelGGL(chain_mass_tables_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mass_dev, n, t.ms.as<double>(), t.mi.as<double>());
    HIP_TRY(hipGetLastError());
    return MI_OK;
}
void lit_set_chain_mass(mi::lit::LitParams& lp, const double* mass_dev, const ChainMass& t, uint64_t C)
{
    lp.precond = 1; lp.m = mass_dev; lp.m_sqrt = t.ms.as<double>(); lp.m_inv = t.mi.as<double>(); lp.m_chain_stride = C;
}

int run_literal(const char* who, int algo, const mi_target* target, const mi_settings* settings, mi_chains* chains, hipStream_t st)
{
    const uint64_t d = target->d, C = chains->n_chains;
    const uint64_t n_total = settings->n_burnin_draws + settings->n_keep_draws;
    if (n_total > 0xffffffffULL) return fail(MI_ERR_BAD_ARG, "too many draws");
    if (d > 0x7fffffffULL) return fail(MI_ERR_BAD_ARG, "%s: d out of range", who);
    if (settings->vals_bound && (!settings->lower_bounds || !settings->upper_bounds)) return fail(MI_ERR_BAD_ARG, "%s: vals_bound needs lower_bounds and upper_bounds", who);
    if (algo == 2 && settings->max_tree_depth > (uint64_t)mi::lit::LIT_NUTS_MAX_DEPTH)
        return fail(MI_ERR_UNSUPPORTED, "nuts: max_tree_depth > %d not implemented (2^%d leapfrog steps per draw)", (int)mi::lit::LIT_NUTS_MAX_DEPTH, (int)mi::lit::LIT_NUTS_MAX_DEPTH);
    mi::lit::LitParams lp{};
    DevBuf t_a, t_b, t_t;                                // staged target (and its transposed matrix)
    lp.t.d = (uint32_t)d;
    auto up = [&](DevBuf& b, const double* src, size_t n_doubles, const double** out) -> int {
        if (target->mem == MI_MEM_DEVICE) { *out = src; return MI_OK; }
        HIP_TRY(b.alloc(n_doubles * 8));
        HIP_TRY(hipMemcpy(b.p, src, n_doubles * 8, hipMemcpyHostToDevice));
        *out = b.as<double>();
        return MI_OK;
    };
    int rc;
    switch (target->kind) {
    case MI_TARGET_GAUSS_ISO: lp.t.kind = mi::lit::LIT_ISO; break;
    case MI_TARGET_GAUSS_DIAG:
        if (!target->prec) return fail(MI_ERR_BAD_ARG, "GAUSS_DIAG needs prec (d)");
        lp.t.kind = mi::lit::LIT_DIAG; lp.t.prec_stride = 1;
        if ((rc = up(t_a, target->prec, d, &lp.t.prec))) return rc;
        break;
    case MI_TARGET_GAUSS_DENSE:
        if (!target->prec) return fail(MI_ERR_BAD_ARG, "GAUSS_DENSE needs prec (d*d)");
        lp.t.kind = mi::lit::LIT_DENSE;
        if ((rc = up(t_a, target->prec, d * d, &lp.t.prec))) return rc;
        HIP_TRY(t_t.alloc(d * d * 8));                     // ... read transposed by the literal kernels
        if ((rc = transpose_on_device(lp.t.prec, t_t.as<double>(), (uint32_t)d, (uint32_t)d, st))) return rc;
        lp.t.prec = t_t.as<double>();
        break;
    case MI_TARGET_LOGISTIC:
        if (!target->X || !target->y || target->n_rows == 0) return fail(MI_ERR_BAD_ARG, "LOGISTIC needs X, y, n_rows");
        lp.t.kind = mi::lit::LIT_LOGISTIC; lp.t.n_rows = (uint32_t)target->n_rows;
        if ((rc = up(t_a, target->X, target->n_rows * d, &lp.t.X))) return rc;
        if ((rc = up(t_b, target->y, target->n_rows, &lp.t.y))) return rc;
        HIP_TRY(t_t.alloc(target->n_rows * d * 8));
        if ((rc = transpose_on_device(lp.t.X, t_t.as<double>(), (uint32_t)target->n_rows, (uint32_t)d, st))) return rc;
        lp.t.Xt = t_t.as<double>();
        break;
    default: return fail(MI_ERR_UNSUPPORTED, "%s: target kind %d not implemented", who, target->kind);
    }
    mi::lit::lit_orders(lp.t);
    StagedChains sc;
    rc = stage_in(chains, d, settings->n_keep_draws, sc, st, n_total);
    if (rc) return rc;
    if (algo == 4 && d > (uint64_t)mi::lit::LIT_RMHMC_MAX_D)
        return fail(MI_ERR_UNSUPPORTED, "rmhmc: d = %llu > %d is not implemented (two d x d x d derivative cubes per chain; O(d^4) per fixed-point step)",
                    (unsigned long long)d, (int)mi::lit::LIT_RMHMC_MAX_D);
    const bool mala_bounded = algo == 1 && settings->vals_bound != 0;
    if (mala_bounded && d > 512) return fail(MI_ERR_UNSUPPORTED, "mala: vals_bound with d > 512 is not implemented (ten d x d matrices per workgroup)");
    ReplayWs rp;
    rp.stride = mi::lit::lit_work_doubles((uint32_t)d, lp.t.n_rows, mala_bounded, (uint32_t)settings->max_tree_depth, algo == 2, algo == 4);
    rp.n_wg = (unsigned)std::min<uint64_t>(C, (mala_bounded || algo == 4) ? 128u : 1024u);
    WsLease ws;
    rc = ws_get(st, (size_t)rp.n_wg * rp.stride * sizeof(double), ws);
    if (rc) return rc;
    rp.work = ws.as<double>();
    lit_common(lp, settings, &sc.dev, rp, true);
    mi::lit::LitPrep prep;
    rc = mi::lit::lit_prepare(algo == 1 ? 1 : 0, (uint32_t)d, settings->step_size, settings->vals_bound ? 1 : 0, settings->lower_bounds,
                              settings->upper_bounds, algo == 4 ? nullptr : settings->precond_mat, prep);     // (rmhmc has no precond_mat)
    if (rc) return rc;
    lp.n_fp_steps = (uint32_t)settings->n_fp_steps;
    LitDev ldev;
    rc = lit_upload(prep, (uint32_t)d, settings->vals_bound != 0, ldev, lp);
    if (rc) return rc;
    ChainMass cm;
    if (sc.dev.mass_diag) {                              // hmc with per-chain diagonal masses (checked by the caller)
        if ((rc = chain_mass_tables(sc.dev.mass_diag, d, C, cm, st))) return rc;
        lit_set_chain_mass(lp, sc.dev.mass_diag, cm, C);
    }
    if (algo == 2) {
        if ((rc = nuts_continuation(settings, chains, &lp.n_adapt))) return rc;
        lp.max_depth = (uint32_t)settings->max_tree_depth;
        lp.delta = settings->target_accept_rate; lp.gamma = settings->gamma_val; lp.t0 = settings->t0_val; lp.kappa = settings->kappa_val;
        lp.step_out = sc.dev.step_size; lp.depth_trace = sc.dev.nuts_depth; lp.adapt_state = sc.dev.nuts_adapt_state;
    }
    rc = launched(who, mi::launch_literal(algo, lp, rp.n_wg, st));
    if (rc) return rc;
    rc = stage_out(chains, d, settings->n_keep_draws, sc, st, n_total);
    if (rc) return rc;
    if (t_a.p || t_b.p || t_t.p || ldev.any || cm.ms.p || chains->mem == MI_MEM_HOST) HIP_TRY(hipStreamSynchronize(st));
    return MI_OK;
}


// ---- mcmc::rmhmc with HOST callbacks (ref: include/mcmc/rmhmc.hpp: target_log_kernel and tensor_fn as std::function): one chain on
// literal_kernel<4> with the LIT_CALLBACK target -- the kernel asks for every evaluation through a mailbox in pinned memory
// (literal.hpp: LitMailbox), this thread serves the requests while the kernel runs.  Everything that is the sampler's own arithmetic
// (fixed-point iterations, INV / CHOL_LOWER / LOG_DET of the metric, the d matrix products of mntm_update_fn, energies, the Metropolis
// test) is on the device; what runs on the host is the user's code, as in the reference.
namespace {
struct PinnedBuf {
    void* p = nullptr;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    hipError_t alloc(size_t bytes) { return hipHostMalloc(&p, bytes ? bytes : 8, hipHostMallocCoherent); }
    template <class T> T* as() const { return static_cast<T*>(p); }
};
// serve the kernel's requests until the stream is idle; returns the number of requests served
uint64_t serve_callbacks(const mi::lit::LitMailbox& mb, uint32_t d, mi_log_kernel_cb kcb, void* kdata, mi_tensor_cb tcb, void* tdata, hipStream_t st)
{
    uint32_t served = 0;
    uint64_t n = 0;
    unsigned idle = 0;
    for (;;) {
        const uint32_t req = __atomic_load_n(&mb.ctl[mi::lit::LIT_MB_REQ], __ATOMIC_ACQUIRE);
        if (req != served) {
            const bool want = mb.ctl[mi::lit::LIT_MB_WANT] != 0u;
            if (mb.ctl[mi::lit::LIT_MB_KIND] == (uint32_t)mi::lit::LIT_REQ_KERNEL) *mb.value = kcb(mb.x, want ? mb.out : nullptr, kdata);
            else tcb(mb.x, mb.out, want ? mb.out + (size_t)d * d : nullptr, tdata);
            __atomic_store_n(&mb.ctl[mi::lit::LIT_MB_ACK], req, __ATOMIC_RELEASE);
            served = req; ++n; idle = 0;
            continue;
        }
        if ((++idle & 0xffu) == 0u) {
            if (hipStreamQuery(st) != hipErrorNotReady) {                            // the kernel has ended (or failed): one last look, then out
                if (__atomic_load_n(&mb.ctl[mi::lit::LIT_MB_REQ], __ATOMIC_ACQUIRE) == served) break;
            }
            std::this_thread::yield();                                              // (between requests the kernel computes: let the core go now and then)
        }
    }
    return n;
}
}  // namespace

// a diagonal precond_mat for the LDS-streamed hmc kernels: sqrt(m) and 1 / m on the device, padded with ones to 512 entries
struct DiagMass { DevBuf ms, mi; };
int diag_mass_upload(const mi_settings* settings, uint64_t d, DiagMass& t)
{
    std::vector<double> ms(512, 1.0), mi_(512, 1.0);
    for (uint64_t i = 0; i < d; ++i) { const double v = settings->precond_mat[i * d + i]; ms[i] = __builtin_sqrt(v); mi_[i] = 1.0 / v; }
    HIP_TRY(t.ms.alloc(512 * 8)); HIP_TRY(t.mi.alloc(512 * 8));
    HIP_TRY(hipMemcpy(t.ms.p, ms.data(), 512 * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(t.mi.p, mi_.data(), 512 * 8, hipMemcpyHostToDevice));
    return MI_OK;
}
// mala with a diagonal precond_mat on the LDS-streamed kernels: m, sqrt(m), INV(eps^2 M) (padded to 512) and LOG_DET(eps^2 M), all in the
// oracle's operation order (literal_host.hpp: lit_prepare)
struct MalaDiagMass { DevBuf m, ms, sinv; double log_det = 0.0; };
int mala_diag_mass_upload(const mi_settings* settings, uint64_t d, MalaDiagMass& t, mi::LogitParams& q)
{
    mi::lit::LitPrep prep;
    const int rcp = mi::lit::lit_prepare(1, (uint32_t)d, settings->step_size, 0, nullptr, nullptr, settings->precond_mat, prep);
    if (rcp) return rcp;
    std::vector<double> m(512, 1.0), ms(512, 1.0), si(512, 1.0);
    for (uint64_t i = 0; i < d; ++i) { m[i] = prep.m[i]; ms[i] = prep.m_sqrt[i]; si[i] = prep.sinv_diag[i]; }
    HIP_TRY(t.m.alloc(512 * 8)); HIP_TRY(t.ms.alloc(512 * 8)); HIP_TRY(t.sinv.alloc(512 * 8));
    HIP_TRY(hipMemcpy(t.m.p, m.data(), 512 * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(t.ms.p, ms.data(), 512 * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(t.sinv.p, si.data(), 512 * 8, hipMemcpyHostToDevice));
    q.m = t.m.as<double>(); q.m_sqrt = t.ms.as<double>(); q.s_inv = t.sinv.as<double>();
    q.log_det = prep.log_det;
    return MI_OK;
}
bool precond_is_diagonal(const mi_settings* settings, uint64_t d)
{
    if (!settings->precond_mat) return false;
    for (uint64_t i = 0; i < d; ++i)
        for (uint64_t k = 0; k < d; ++k)
            if (i != k && settings->precond_mat[i * d + k] != 0.0) return false;
    return true;
}

// settings.vals_bound and / or a DIAGONAL precond_mat for hmc / nuts on the LDS-streamed kernels (lds_box.hpp; logistic_lds.hpp DIAGM / BOUNDS):
// determine_bounds_type (determine_bounds_type.hpp:27-57), the bounds, sqrt(m) and 1 / m -- on the device, padded to 512 entries with type 1 /
// ones; with bounds the mass tables are always there (ones for the identity)
struct LdsTables { DevBuf bt, lb, ub, m, ms, mi; bool bounds = false, mass = false; };
int lds_tables(const char* who, const mi_settings* settings, uint64_t d, LdsTables& t, mi::LogitParams& q)
{
    t.bounds = settings->vals_bound != 0; t.mass = settings->precond_mat != nullptr;
    if (!t.bounds && !t.mass) return MI_OK;
    if (t.bounds && (!settings->lower_bounds || !settings->upper_bounds)) return fail(MI_ERR_BAD_ARG, "%s: vals_bound needs lower_bounds and upper_bounds", who);
    std::vector<double> m(512, 1.0), ms(512, 1.0), mi_(512, 1.0), lbv(512, 0.0), ubv(512, 0.0);
    std::vector<int> bt(512, 1);
    if (t.mass) for (uint64_t i = 0; i < d; ++i) { const double v = settings->precond_mat[i * d + i]; m[i] = v; ms[i] = __builtin_sqrt(v); mi_[i] = 1.0 / v; }
    if (t.bounds)
        for (uint64_t i = 0; i < d; ++i) {
            lbv[i] = settings->lower_bounds[i]; ubv[i] = settings->upper_bounds[i];
            const bool fl = std::isfinite(lbv[i]), fu = std::isfinite(ubv[i]);
            bt[i] = (fl && fu) ? 4 : (fl && !fu) ? 2 : (!fl && fu) ? 3 : 1;
        }
    HIP_TRY(t.m.alloc(512 * 8)); HIP_TRY(t.ms.alloc(512 * 8)); HIP_TRY(t.mi.alloc(512 * 8));
    HIP_TRY(hipMemcpy(t.m.p, m.data(), 512 * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(t.ms.p, ms.data(), 512 * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(t.mi.p, mi_.data(), 512 * 8, hipMemcpyHostToDevice));
    q.m_sqrt = t.ms.as<double>(); q.m_inv = t.mi.as<double>();
    if (t.bounds) {
        HIP_TRY(t.bt.alloc(512 * sizeof(int))); HIP_TRY(t.lb.alloc(512 * 8)); HIP_TRY(t.ub.alloc(512 * 8));
        HIP_TRY(hipMemcpy(t.bt.p, bt.data(), 512 * sizeof(int), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(t.lb.p, lbv.data(), 512 * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(t.ub.p, ubv.data(), 512 * 8, hipMemcpyHostToDevice));
        q.btype = t.bt.as<int>(); q.lb = t.lb.as<double>(); q.ub = t.ub.as<double>();
    }
    return MI_OK;
}
// hmc with a DENSE precond_mat on the LDS-streamed kernel (logistic_lds.hpp: DENSEM): INV(M) and CHOL_LOWER(M) from the host (hmc.cpp:57-59
// through the oracle's Gauss-Jordan / column Cholesky, as everywhere), row-major on the device
// mala (mala.cpp:57-58; mala.ipp:58-64): M, CHOL_LOWER(M), and -- Sigma = eps^2 M is constant without bounds -- INV(Sigma) and LOG_DET(Sigma) =
// sum_i 2 log CHOL_LOWER(Sigma)_ii, i ascending, as the d <= 128 kernel takes them (mala_gauss_dense_m_kernel)
struct LdsDenseM { DevBuf minv, l, m, sinv; };
int lds_dense_m(const mi_settings* settings, uint64_t d, LdsDenseM& t, mi::LogitParams& q, int algo)
{
    auto up = [&](DevBuf& b, const double* src) -> int {
        HIP_TRY(b.alloc(d * d * 8));
        HIP_TRY(hipMemcpy(b.p, src, d * d * 8, hipMemcpyHostToDevice));
        return MI_OK;
    };
    int rc;
    std::vector<double> L;
    if ((rc = host_cholesky_lower(settings->precond_mat, d, L))) return rc;
    if ((rc = up(t.l, L.data()))) return rc;
    q.L_rm = t.l.as<double>();
    if (algo == mi::LOGIT_MALA) {
        const double s2 = settings->step_size * settings->step_size;
        std::vector<double> Sigma(d * d), Sinv, Ls;
        for (uint64_t i = 0; i < d * d; ++i) Sigma[i] = s2 * settings->precond_mat[i];
        if ((rc = host_inverse(Sigma.data(), d, Sinv))) return rc;
        if ((rc = host_cholesky_lower(Sigma.data(), d, Ls))) return rc;
        double ld = 0.0;
        for (uint64_t i = 0; i < d; ++i) ld = ld + 2.0 * mi::det_log(Ls[i * d + i]);
        q.log_det = ld;
        if ((rc = up(t.m, settings->precond_mat))) return rc;
        if ((rc = up(t.sinv, Sinv.data()))) return rc;
        q.M_rm = t.m.as<double>(); q.Sinv_rm = t.sinv.as<double>();
    } else {
        std::vector<double> Minv;
        if ((rc = host_inverse(settings->precond_mat, d, Minv))) return rc;
        if ((rc = up(t.minv, Minv.data()))) return rc;
        q.Minv_rm = t.minv.as<double>();
    }
    return MI_OK;
}
// ... which the hmc and mala front ends route there when this holds (bounds with a dense matrix stay on literal.hpp)
bool lds_dense_m_ok(const mi_target* target, const mi_settings* settings)
{
    return target->kernel_hint != MI_KERNEL_LITERAL && settings->precond_mat && !settings->vals_bound && !precond_is_diagonal(settings, target->d);
}
// the same tables for the literal replay behind such a launch (precond 1: M, sqrt(M), 1 / M element by element; literal_host.hpp)
void lds_tables_replay(const LdsTables& t, const mi::LogitParams& q, mi::lit::LitParams& lp)
{
    if (t.mass) { lp.precond = 1; lp.m = t.m.as<double>(); lp.m_sqrt = q.m_sqrt; lp.m_inv = q.m_inv; }
    if (t.bounds) { lp.vals_bound = 1; lp.btype = q.btype; lp.lb = q.lb; lp.ub = q.ub; }
}
bool lds_tables_active(const LdsTables* t) { return t != nullptr && (t->bounds || t->mass); }
// hmc / nuts cases of settings the LDS-streamed kernels cover: no bounds / bounds, identity / diagonal precond_mat
bool lds_general_ok(const mi_target* target, const mi_settings* settings)
{
    return target->kernel_hint != MI_KERNEL_LITERAL && (!settings->precond_mat || precond_is_diagonal(settings, target->d));
}

// hmc / rwmh on the logistic-regression target (identity preconditioner / cov_mat, no bounds): logit_lds_kernel<., HMC | RWMH>;
// settings->step_size is the leapfrog step resp. par_scale
int run_logit_plain(const char* who, int algo, const mi_target* target, const mi_settings* settings, mi_chains* chains, hipStream_t st)
{
    int rc;
    const uint64_t d = target->d;
    if (!target->X || !target->y || target->n_rows == 0) return fail(MI_ERR_BAD_ARG, "LOGISTIC needs X, y, n_rows");
    if (d > 512) return fail(MI_ERR_UNSUPPORTED, "%s: logistic target with d = %llu > 512 not implemented", who, (unsigned long long)d);
    if (settings->n_burnin_draws + settings->n_keep_draws > 0xffffffffULL) return fail(MI_ERR_BAD_ARG, "too many draws");
    const uint64_t n = target->n_rows;
    DevBuf Xo, yo;
    const double *X_dev = target->X, *y_dev = target->y;
    if (target->mem == MI_MEM_HOST) {
        HIP_TRY(Xo.alloc(n * d * sizeof(double))); HIP_TRY(yo.alloc(n * sizeof(double)));
        HIP_TRY(hipMemcpy(Xo.p, target->X, n * d * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(yo.p, target->y, n * sizeof(double), hipMemcpyHostToDevice));
        X_dev = Xo.as<double>(); y_dev = yo.as<double>();
    }
    StagedChains sc;
    rc = stage_in(chains, d, settings->n_keep_draws, sc, st);
    if (rc) return rc;
    mi::LogitParams q{};
    q.d = (uint32_t)d; q.n_rows = (uint32_t)n; q.NB = (uint32_t)((n + 15) / 16);
    q.C = chains->n_chains; q.chain0 = chains->chain0;
    q.theta = sc.dev.theta; q.draws = sc.dev.draws; q.n_accept = sc.dev.n_accept;
    q.seed = settings->rng_seed_value;
    q.n_burnin = (uint32_t)settings->n_burnin_draws; q.n_keep = (uint32_t)settings->n_keep_draws;
    q.n_leap = (uint32_t)settings->n_leap_steps;
    q.eps = settings->step_size;
    q.draw0 = (uint32_t)chains->draw0;
    LdsTables lt;
    LdsDenseM ldm;
    if (algo == mi::LOGIT_HMC && lds_dense_m_ok(target, settings)) { if ((rc = lds_dense_m(settings, d, ldm, q, algo))) return rc; }      // a dense precond_mat, unbounded
    else if (algo == mi::LOGIT_HMC) { if ((rc = lds_tables(who, settings, d, lt, q))) return rc; }      // (the caller routed bounds / a DIAGONAL matrix here)
    rc = launch_logit(algo, q, X_dev, y_dev, st, settings, &sc.dev, mi::LOGIT_TARGET_LOGISTIC, &lt);
    if (rc) return rc;
    rc = fill_n_leap(sc.dev.n_leapfrogs, chains->n_chains,
                     algo == mi::LOGIT_HMC ? (settings->n_burnin_draws + settings->n_keep_draws) * settings->n_leap_steps : 0, st);
    if (rc) return rc;
    rc = stage_out(chains, d, settings->n_keep_draws, sc, st);
    if (rc) return rc;
    if (Xo.p || lt.ms.p || chains->mem == MI_MEM_HOST) HIP_TRY(hipStreamSynchronize(st));
    return MI_OK;
}

// hmc / mala / rwmh on a dense Gaussian with 128 < d <= 512 (identity preconditioner / cov_mat, no bounds): P no longer fits into LDS
// (hmc_dense.hpp), so it is streamed through LDS block by block like the design matrix of the logistic target --
// logit_lds_kernel<., ., LOGIT_TARGET_DENSE> (logistic_lds.hpp).  settings->step_size is the leapfrog step resp. par_scale.
int run_dense_lds(const char* who, int algo, const mi_target* target, const mi_settings* settings, mi_chains* chains, hipStream_t st)
{
    int rc;
    const uint64_t d = target->d;
    if (!target->prec) return fail(MI_ERR_BAD_ARG, "GAUSS_DENSE needs prec (d*d)");
    if (settings->n_burnin_draws + settings->n_keep_draws > 0xffffffffULL) return fail(MI_ERR_BAD_ARG, "too many draws");
    DevBuf P_owned;
    const double* P_dev = nullptr;
    rc = dense_precision_on_device(target, P_owned, &P_dev, st);
    if (rc) return rc;
    StagedChains sc;
    rc = stage_in(chains, d, settings->n_keep_draws, sc, st);
    if (rc) return rc;
    mi::LogitParams q{};
    q.d = (uint32_t)d; q.n_rows = (uint32_t)d; q.NB = (uint32_t)((d + 15) / 16);
    q.C = chains->n_chains; q.chain0 = chains->chain0;
    q.theta = sc.dev.theta; q.draws = sc.dev.draws; q.n_accept = sc.dev.n_accept;
    q.seed = settings->rng_seed_value;
    q.n_burnin = (uint32_t)settings->n_burnin_draws; q.n_keep = (uint32_t)settings->n_keep_draws;
    q.n_leap = (uint32_t)settings->n_leap_steps;
    q.eps = settings->step_size;
    q.draw0 = (uint32_t)chains->draw0;
    if (algo == mi::LOGIT_MALA) {                        // dmvnorm's constants for Sigma = eps^2 I, as the oracle states them
        const double s2 = settings->step_size * settings->step_size;
        double log_det = 0.0;
        const double lii = __builtin_sqrt(s2);
        for (uint64_t i = 0; i < d; ++i) log_det = log_det + 2.0 * mi::det_log(lii);
        q.s2 = s2; q.rs = 1.0 / s2; q.log_det = log_det;
        q.cons_term = -0.5 * (double)d * 1.83787706640934548356;
    }
    LdsTables lt;
    LdsDenseM ldm;
    MalaDiagMass mdm;
    if ((algo == mi::LOGIT_HMC || algo == mi::LOGIT_MALA) && lds_dense_m_ok(target, settings)) { if ((rc = lds_dense_m(settings, d, ldm, q, algo))) return rc; }      // a dense precond_mat, unbounded
    else if (algo == mi::LOGIT_HMC) { if ((rc = lds_tables(who, settings, d, lt, q))) return rc; }      // (the caller routed bounds / a DIAGONAL matrix here)
    else if (algo == mi::LOGIT_MALA && settings->precond_mat) { if ((rc = mala_diag_mass_upload(settings, d, mdm, q))) return rc; }
    rc = launch_logit(algo, q, P_dev, nullptr, st, settings, &sc.dev, mi::LOGIT_TARGET_DENSE, &lt);
    if (rc) return rc;
    rc = fill_n_leap(sc.dev.n_leapfrogs, chains->n_chains,
                     algo == mi::LOGIT_HMC ? (settings->n_burnin_draws + settings->n_keep_draws) * settings->n_leap_steps : 0, st);
    if (rc) return rc;
    rc = stage_out(chains, d, settings->n_keep_draws, sc, st);
    if (rc) return rc;
    if (P_owned.p || lt.ms.p || mdm.m.p || chains->mem == MI_MEM_HOST) HIP_TRY(hipStreamSynchronize(st));
    return MI_OK;
}

// hmc / mala / rwmh on a dense Gaussian or the logistic-regression target BEYOND d = 512 (identity preconditioner / cov_mat, no bounds): the state of a
// 16-chain tile no longer fits a workgroup, so it lives in HBM and the gradients of ALL chains at one leapfrog step are fp64 matrix products on the matrix
// cores -- W = P Theta, resp. eta = X Theta and X^T (y - sigmoid(eta)) -- with the half-kicks and the drift in the epilogue (gemm_samplers.hip); chains that
// reach the non-finite regime are flagged and replayed by literal.hpp right behind it.  algo: the C ABI's numbers (0 hmc, 1 mala, 3 rwmh).
bool gemm_case(const mi_target* target, const mi_settings* settings, const mi_chains* chains, bool hmc, bool algo_has_mass = true)
{
    return (target->kind == MI_TARGET_GAUSS_DENSE || target->kind == MI_TARGET_LOGISTIC) && target->d > 512 && !settings->vals_bound
           && (!settings->precond_mat || (algo_has_mass && precond_is_diagonal(settings, target->d)))       // identity, or (hmc, mala) a DIAGONAL precond_mat
           && !chains->mass_diag && target->kernel_hint != MI_KERNEL_LITERAL && (!hmc || settings->n_leap_steps >= 1);
}
int run_gemm(const char* who, int algo, const mi_target* target, const mi_settings* settings, mi_chains* chains, hipStream_t st)
{
    int rc;
    const uint64_t d = target->d, C = chains->n_chains;
    const bool logit = target->kind == MI_TARGET_LOGISTIC;
    const uint64_t n = logit ? target->n_rows : 0;
    if (d > 0x7fffffffULL || n > 0x7fffffffULL) return fail(MI_ERR_BAD_ARG, "%s: d / n_rows out of range", who);
    if (settings->n_burnin_draws + settings->n_keep_draws > 0xffffffffULL) return fail(MI_ERR_BAD_ARG, "too many draws");
    if (algo == 0 && settings->n_leap_steps > 0xffffffffULL) return fail(MI_ERR_BAD_ARG, "hmc: too many leapfrog steps");
    DevBuf P_owned, Xo, yo;
    const double *P_dev = nullptr, *X_dev = nullptr, *y_dev = nullptr;
    if (logit) {
        if (!target->X || !target->y || n == 0) return fail(MI_ERR_BAD_ARG, "LOGISTIC needs X, y, n_rows");
        X_dev = target->X; y_dev = target->y;
        if (target->mem == MI_MEM_HOST) {
            HIP_TRY(Xo.alloc(n * d * sizeof(double))); HIP_TRY(yo.alloc(n * sizeof(double)));
            HIP_TRY(hipMemcpy(Xo.p, target->X, n * d * sizeof(double), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(yo.p, target->y, n * sizeof(double), hipMemcpyHostToDevice));
            X_dev = Xo.as<double>(); y_dev = yo.as<double>();
        }
    } else {
        rc = dense_precision_on_device(target, P_owned, &P_dev, st);
        if (rc) return rc;
    }
    StagedChains sc;
    rc = stage_in(chains, d, settings->n_keep_draws, sc, st);
    if (rc) return rc;
    mi::gemm::GemmRun g;
    g.algo = algo; g.d = (uint32_t)d; g.C = C; g.chain0 = chains->chain0; g.P = P_dev; g.X = X_dev; g.y = y_dev; g.n_rows = (uint32_t)n;
    g.theta = sc.dev.theta; g.draws = sc.dev.draws; g.n_accept = sc.dev.n_accept;
    g.seed = settings->rng_seed_value;
    g.n_burnin = (uint32_t)settings->n_burnin_draws; g.n_keep = (uint32_t)settings->n_keep_draws; g.n_leap = (uint32_t)settings->n_leap_steps;
    g.draw0 = (uint32_t)chains->draw0; g.eps = settings->step_size;
    // identity or a DIAGONAL precond_mat (hmc.cpp:57-59, mala.cpp:57-58 with mala.ipp:58-64): diag(M), CHOL_LOWER (sqrt), INV (reciprocal) and, mala, INV(eps^2 M) with
    // LOG_DET(eps^2 M) -- all from lit_prepare, in the oracle's operation order; the identity as tables of ones (1.0 * x is x bit for bit)
    mi::lit::LitPrep prep;
    rc = mi::lit::lit_prepare(algo == 1 ? 1 : 0, (uint32_t)d, settings->step_size, 0, nullptr, nullptr, settings->precond_mat, prep);
    if (rc) return rc;
    const uint32_t dK = mi::gemm::gemm_padded_d((uint32_t)d);
    std::vector<double> tabs(4 * (size_t)dK, 1.0);
    if (algo == 1) {                                     // dmvnorm's constants for Sigma = eps^2 M, as the oracle states them
        g.s2 = settings->step_size * settings->step_size; g.rs = prep.rs; g.log_det = prep.log_det; g.cons_term = prep.cons_term;
        for (uint64_t i = 0; i < d; ++i) tabs[3 * (size_t)dK + i] = prep.precond == 1 ? prep.sinv_diag[i] : prep.rs;
    }
    if (prep.precond == 1)
        for (uint64_t i = 0; i < d; ++i) { tabs[i] = prep.m[i]; tabs[dK + i] = prep.m_sqrt[i]; tabs[2 * (size_t)dK + i] = prep.m_inv[i]; }
    DevBuf tabs_dev;
    HIP_TRY(tabs_dev.alloc(tabs.size() * 8));
    HIP_TRY(hipMemcpy(tabs_dev.p, tabs.data(), tabs.size() * 8, hipMemcpyHostToDevice));
    g.mass_tables = tabs_dev.as<double>(); g.diag_mass = prep.precond == 1;
    // a draw is a handful of launches + one or two per gradient: replayed from a captured graph while a launch is short (few chains); at full size the queue runs ahead anyway
    g.use_graph = (double)d * (double)(logit ? 2 * n : d) * (double)C < 3.0e10;
    const bool replay = algo != 3;                       // rwmh forms no product with a vector that can be non-finite (rwmh.cpp:126)
    WsLease base;
    ReplayWs rp = replay_layout(mi::gemm::gemm_ws_bytes((uint32_t)d, (uint32_t)n, C), C, (uint32_t)d, (uint32_t)(logit ? n : d), false);
    rc = ws_get(st, replay ? rp.total_bytes : rp.own_bytes, base);
    if (rc) return rc;
    if (replay) {
        rc = replay_bind(rp, base.p, C, st);
        if (rc) return rc;
        g.nf_flag = rp.flag;
    }
    g.ws = base.p;
    const char* kname = nullptr;
    const int e = mi::gemm::gemm_run(g, st, &kname);
    if (e != 0) return fail(MI_ERR_HIP, "%s: matrix-product sampler: %s", who, hipGetErrorString((hipError_t)e));
    if (replay) {                                        // chains that reached the non-finite regime: replayed literally (literal.hpp)
        mi::lit::LitParams lp{};
        rc = transpose_on_device(logit ? X_dev : P_dev, rp.tbuf, (uint32_t)(logit ? n : d), (uint32_t)d, st);     // (literal.hpp reads the matrix transposed)
        if (rc) return rc;
        if (logit) { lp.t.kind = mi::lit::LIT_LOGISTIC; lp.t.d = (uint32_t)d; lp.t.n_rows = (uint32_t)n; lp.t.X = X_dev; lp.t.y = y_dev; lp.t.Xt = rp.tbuf; }
        else { lp.t.kind = mi::lit::LIT_DENSE; lp.t.d = (uint32_t)d; lp.t.prec = rp.tbuf; }
        mi::lit::lit_orders(lp.t);
        lit_common(lp, settings, &sc.dev, rp, false);
        lp.rs = g.rs; lp.log_det = g.log_det; lp.cons_term = g.cons_term;
        if (prep.precond == 1) { lp.precond = 1; lp.m = g.mass_tables; lp.m_sqrt = g.mass_tables + dK; lp.m_inv = g.mass_tables + 2 * (size_t)dK; lp.sinv_diag = g.mass_tables + 3 * (size_t)dK; }
        rc = launched("matrix-product sampler (literal replay)", mi::launch_literal(algo, lp, rp.n_wg, st));
        if (rc) return rc;
    }
    mi::note_kernel("%s", kname);
    rc = fill_n_leap(sc.dev.n_leapfrogs, C, algo == 0 ? (settings->n_burnin_draws + settings->n_keep_draws) * settings->n_leap_steps : 0, st);
    if (rc) return rc;
    rc = stage_out(chains, d, settings->n_keep_draws, sc, st);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(st));                  // (the mass tables are ours)
    return MI_OK;
}

// The one-lane-per-chain engine (rmhmc_small.hpp, small_samplers.hpp): hmc, mala, rwmh with any precond_mat / cov_mat and any
// bounds, nuts, rmhmc, for a target policy of dimension d <= SMALL_MAX_D.  Validates, stages the chains, packs the launch
// parameters; `launch` instantiates the kernels for its target type (the built-in policies below, or a user's library through
// mi_mcmc_run_user_target).  algo: 0 hmc, 1 mala, 2 nuts, 3 rwmh, 4 rmhmc.
int run_small(const char* who, int algo, uint64_t d, const mi_settings* settings, mi_chains* chains, hipStream_t st,
              const std::function<int(const mi::SmallParams&, hipStream_t)>& launch)
{
    if (d == 0 || d > (uint64_t)mi::SMALL_MAX_D) return fail(MI_ERR_UNSUPPORTED, "%s: the one-lane-per-chain engine takes 1 <= d <= %d", who, (int)mi::SMALL_MAX_D);
    if (settings->vals_bound && (!settings->lower_bounds || !settings->upper_bounds))
        return fail(MI_ERR_BAD_ARG, "%s: vals_bound needs lower_bounds and upper_bounds", who);
    const uint64_t n_total = settings->n_burnin_draws + settings->n_keep_draws;
    if (n_total > 0xffffffffULL) return fail(MI_ERR_BAD_ARG, "too many draws");
    if (algo == 2) {
        if (settings->max_tree_depth > (uint64_t)mi::NUTS_SMALL_MAX_DEPTH)
            return fail(MI_ERR_UNSUPPORTED, "nuts: max_tree_depth > %d not implemented for this target", (int)mi::NUTS_SMALL_MAX_DEPTH);
    }
    StagedChains sc;
    int rc = stage_in(chains, d, settings->n_keep_draws, sc, st, n_total);
    if (rc) return rc;

    mi::SmallParams prm{};
    prm.d = (uint32_t)d;
    prm.C = chains->n_chains; prm.chain0 = chains->chain0;
    prm.theta = sc.dev.theta; prm.draws = sc.dev.draws; prm.n_accept = sc.dev.n_accept; prm.n_leap = sc.dev.n_leapfrogs;
    prm.seed = settings->rng_seed_value;
    prm.n_burnin = (uint32_t)settings->n_burnin_draws; prm.n_keep = (uint32_t)settings->n_keep_draws;
    prm.n_leap_steps = (uint32_t)settings->n_leap_steps; prm.n_fp_steps = (uint32_t)settings->n_fp_steps;
    prm.draw0 = (uint32_t)chains->draw0;
    prm.eps = settings->step_size;
    prm.vals_bound = settings->vals_bound ? 1 : 0;
    for (int i = 0; i < mi::SMALL_MAX_D; ++i) {
        prm.btype[i] = 1; prm.lb[i] = 0.0; prm.ub[i] = 0.0;
        for (int k = 0; k < mi::SMALL_MAX_D; ++k) prm.M[i][k] = (i == k) ? 1.0 : 0.0;
    }
    if (settings->precond_mat && algo != 4)          // hmc.cpp:57, mala.cpp:57, rwmh.cpp:58 (rmhmc has none)
        for (uint64_t i = 0; i < d; ++i)
            for (uint64_t k = 0; k < d; ++k) prm.M[i][k] = settings->precond_mat[i * d + k];
    if (settings->vals_bound)
        for (uint64_t i = 0; i < d; ++i) {       // determine_bounds_type.hpp:27-57
            prm.lb[i] = settings->lower_bounds[i]; prm.ub[i] = settings->upper_bounds[i];
            const bool fl = std::isfinite(prm.lb[i]), fu = std::isfinite(prm.ub[i]);
            prm.btype[i] = (fl && fu) ? 4 : (fl && !fu) ? 2 : (!fl && fu) ? 3 : 1;
        }
    if (algo == 2) {
        if ((rc = nuts_continuation(settings, chains, &prm.n_adapt))) return rc;
        prm.max_depth = (uint32_t)settings->max_tree_depth;
        prm.delta = settings->target_accept_rate; prm.gamma = settings->gamma_val; prm.t0 = settings->t0_val; prm.kappa = settings->kappa_val;
        prm.step_out = sc.dev.step_size; prm.depth_trace = sc.dev.nuts_depth; prm.adapt_state = sc.dev.nuts_adapt_state;
    }
    rc = launched(who, launch(prm, st));
    if (rc) return rc;
    rc = stage_out(chains, d, settings->n_keep_draws, sc, st, n_total);
    if (rc) return rc;
    if (chains->mem == MI_MEM_HOST) HIP_TRY(hipStreamSynchronize(st));
    return MI_OK;
}

// the d = 2 normal model of the reference's example programs (MI_TARGET_NORMAL_MODEL)
int run_small_normal_model(const char* who, int algo, const mi_target* target, const mi_settings* settings, mi_chains* chains, hipStream_t st)
{
    if (target->d != 2) return fail(MI_ERR_BAD_ARG, "%s: NORMAL_MODEL has d = 2 (mu, sigma)", who);
    if (!target->y || target->n_rows == 0 || target->n_rows > 0xffffffffULL) return fail(MI_ERR_BAD_ARG, "NORMAL_MODEL needs its observations in y[0..n_rows)");
    DevBuf x_owned;
    const double* x_dev = target->y;
    if (target->mem == MI_MEM_HOST) {
        HIP_TRY(x_owned.alloc(target->n_rows * sizeof(double)));
        HIP_TRY(hipMemcpy(x_owned.p, target->y, target->n_rows * sizeof(double), hipMemcpyHostToDevice));
        x_dev = x_owned.as<double>();
    }
    const uint32_t n_rows = (uint32_t)target->n_rows;
    const int rc = run_small(who, algo, 2, settings, chains, st, [&](const mi::SmallParams& p, hipStream_t s_) {
        mi::SmallParams q = p; q.data = x_dev; q.n_rows = n_rows;
        return mi::launch_small_normal_model(algo, q, s_);
    });
    if (rc) return rc;
    if (x_owned.p) HIP_TRY(hipStreamSynchronize(st));
    return MI_OK;
}

// Logistic regression with d <= SMALL_MAX_D coefficients on the one-lane-per-chain engine (small_targets.hpp: the bits of
// logit_lds_kernel): what the LDS kernel does not implement -- nuts, vals_bound, precond_mat / cov_mat
int run_small_logistic(const char* who, int algo, const mi_target* target, const mi_settings* settings, mi_chains* chains, hipStream_t st)
{
    const uint64_t d = target->d, n = target->n_rows;
    if (!target->X || !target->y || n == 0 || n > 0xffffffffULL) return fail(MI_ERR_BAD_ARG, "LOGISTIC needs X, y, n_rows");
    if (d > (uint64_t)mi::SMALL_MAX_D)
        return fail(MI_ERR_UNSUPPORTED, "%s: nuts / vals_bound / precond_mat / cov_mat on the logistic target are implemented for d <= %d "
                                        "(one chain per lane); d = %llu", who, (int)mi::SMALL_MAX_D, (unsigned long long)d);
    DevBuf Xo, yo;
    const double *X_dev = target->X, *y_dev = target->y;
    if (target->mem == MI_MEM_HOST) {
        HIP_TRY(Xo.alloc(n * d * sizeof(double))); HIP_TRY(yo.alloc(n * sizeof(double)));
        HIP_TRY(hipMemcpy(Xo.p, target->X, n * d * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(yo.p, target->y, n * sizeof(double), hipMemcpyHostToDevice));
        X_dev = Xo.as<double>(); y_dev = yo.as<double>();
    }
    const int rc = run_small(who, algo, d, settings, chains, st, [&](const mi::SmallParams& p, hipStream_t s_) {
        return mi::launch_small_logistic(algo, (int)d, p, X_dev, y_dev, (uint32_t)n, s_);
    });
    if (rc) return rc;
    if (Xo.p) HIP_TRY(hipStreamSynchronize(st));
    return MI_OK;
}

}  // namespace

namespace mi {
namespace host {

// One chain of `algo` (0 hmc, 1 mala, 2 nuts, 3 rwmh, 4 rmhmc) on the literal kernel with the HOST callbacks as target (LIT_CALLBACK,
// literal.hpp): what mi_mcmc_rmhmc_run_callback is, and what the callback routes of the other samplers use for everything their
// host-driven form does not implement (bounds, precond_mat / cov_mat, max_tree_depth > 10).  This thread serves the kernel's requests.
int literal_run_callback(const char* who, int algo, const double* initial_vals, uint64_t d, mi_log_kernel_cb target_log_kernel, void* target_data,
                         mi_tensor_cb tensor_fn, void* tensor_data, const mi_settings* settings, double* draws_out,
                         uint64_t* n_accept_draws, double* step_size_out)
{
    if (!initial_vals || !target_log_kernel || !settings) return fail(MI_ERR_BAD_ARG, "%s (callback): null initial_vals / callback / settings", who);
    if (settings->struct_size != sizeof(mi_settings)) return fail(MI_ERR_BAD_ARG, "struct_size mismatch");
    if (d == 0 || d > 0x7fffffffULL) return fail(MI_ERR_BAD_ARG, "%s (callback): d out of range", who);
    if (algo == 4 && d > (uint64_t)mi::lit::LIT_RMHMC_MAX_D)
        return fail(MI_ERR_UNSUPPORTED, "rmhmc (callback): d = %llu outside 1 .. %d (two d x d x d derivative cubes per chain)", (unsigned long long)d, (int)mi::lit::LIT_RMHMC_MAX_D);
    if (settings->vals_bound && (!settings->lower_bounds || !settings->upper_bounds)) return fail(MI_ERR_BAD_ARG, "%s (callback): vals_bound needs lower_bounds and upper_bounds", who);
    const bool mala_bounded = algo == 1 && settings->vals_bound != 0;
    if (mala_bounded && d > 512) return fail(MI_ERR_UNSUPPORTED, "mala (callback): vals_bound with d > 512 is not implemented (ten d x d matrices per workgroup)");
    if (algo == 2 && settings->max_tree_depth > (uint64_t)mi::lit::LIT_NUTS_MAX_DEPTH)
        return fail(MI_ERR_UNSUPPORTED, "nuts (callback): max_tree_depth > %d not implemented", (int)mi::lit::LIT_NUTS_MAX_DEPTH);
    const uint64_t n_keep = settings->n_keep_draws, n_total = settings->n_burnin_draws + n_keep;
    if (n_total > 0xffffffffULL) return fail(MI_ERR_BAD_ARG, "too many draws");
    if (n_keep > 0 && !draws_out) return fail(MI_ERR_BAD_ARG, "%s (callback): draws_out is required", who);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(MI_ERR_NO_DEVICE, "no HIP device visible: the engine has no CPU path");
    (void)hipGetLastError();
    const size_t dd = (size_t)d * d;
    PinnedBuf ctl, val, xb, outb;
    HIP_TRY(ctl.alloc(mi::lit::LIT_MB_WORDS * sizeof(uint32_t))); HIP_TRY(val.alloc(8)); HIP_TRY(xb.alloc(d * 8));
    HIP_TRY(outb.alloc((algo == 4 ? dd + dd * d : d) * 8));
    std::memset(ctl.p, 0, mi::lit::LIT_MB_WORDS * sizeof(uint32_t));
    mi::lit::LitParams lp{};
    lp.t.kind = mi::lit::LIT_CALLBACK; lp.t.d = (uint32_t)d;
    mi::lit::lit_orders(lp.t);
    lp.t.mb.ctl = ctl.as<uint32_t>(); lp.t.mb.value = val.as<double>(); lp.t.mb.x = xb.as<double>(); lp.t.mb.out = outb.as<double>();
    lp.t.mb.timeout_ticks = 60ull * 100000000ull;          // 60 s of the 100 MHz wall clock per request
    DevBuf theta, draws, nacc, work, step;
    HIP_TRY(theta.alloc(d * 8)); HIP_TRY(draws.alloc(n_keep * d * 8)); HIP_TRY(nacc.alloc(8)); HIP_TRY(step.alloc(8));
    const size_t stride = mi::lit::lit_work_doubles((uint32_t)d, 0, mala_bounded, (uint32_t)settings->max_tree_depth, algo == 2, algo == 4);
    HIP_TRY(work.alloc(stride * 8));
    HIP_TRY(hipMemcpy(theta.p, initial_vals, d * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(nacc.p, 0, 8)); HIP_TRY(hipMemset(step.p, 0, 8));
    HIP_TRY(hipDeviceSynchronize());     // the kernel runs on its own NON-BLOCKING stream, which does not order itself behind the null stream's memsets / copies
    lp.C = 1; lp.chain0 = 0; lp.theta = theta.as<double>(); lp.draws = n_keep ? draws.as<double>() : nullptr; lp.n_accept = nacc.as<uint64_t>();
    lp.seed = settings->rng_seed_value;
    lp.n_burnin = (uint32_t)settings->n_burnin_draws; lp.n_keep = (uint32_t)n_keep; lp.n_leap_steps = (uint32_t)settings->n_leap_steps;
    lp.eps = settings->step_size; lp.n_fp_steps = (uint32_t)settings->n_fp_steps;
    lp.work = work.as<double>(); lp.work_stride = stride;
    if (algo == 2) {
        lp.n_adapt = (uint32_t)(settings->n_adapt_draws > n_total ? n_total : settings->n_adapt_draws);
        lp.max_depth = (uint32_t)settings->max_tree_depth;
        lp.delta = settings->target_accept_rate; lp.gamma = settings->gamma_val; lp.t0 = settings->t0_val; lp.kappa = settings->kappa_val;
        lp.step_out = step.as<double>();
    }
    mi::lit::LitPrep prep;
    int rc = mi::lit::lit_prepare(algo == 1 ? 1 : 0, (uint32_t)d, settings->step_size, settings->vals_bound ? 1 : 0, settings->lower_bounds, settings->upper_bounds,
                                  algo == 4 ? nullptr : settings->precond_mat, prep);
    if (rc) return rc;
    LitDev ldev;
    rc = lit_upload(prep, (uint32_t)d, settings->vals_bound != 0, ldev, lp);
    if (rc) return rc;
    hipStream_t st = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    rc = launched(who, mi::launch_literal(algo, lp, 1, st));
    if (!rc) {
        (void)serve_callbacks(lp.t.mb, (uint32_t)d, target_log_kernel, target_data, tensor_fn, tensor_data, st);
        const hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = fail(MI_ERR_HIP, "%s (callback): %s", who, hipGetErrorString(e));
        else if (ctl.as<uint32_t>()[mi::lit::LIT_MB_ABORT] != 0u) rc = fail(MI_ERR_HIP, "%s (callback): the kernel gave up waiting for a callback (60 s)", who);
    }
    (void)hipStreamDestroy(st);
    if (rc) return rc;
    std::vector<double> rows(n_keep * d);
    if (n_keep) HIP_TRY(hipMemcpy(rows.data(), draws.p, n_keep * d * 8, hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < n_keep; ++i)
        for (uint64_t j = 0; j < d; ++j) draws_out[i + j * n_keep] = rows[i * d + j];       // column-major n_keep x d, as Eigen's Mat_t stores draws_out
    if (n_accept_draws) HIP_TRY(hipMemcpy(n_accept_draws, nacc.p, 8, hipMemcpyDeviceToHost));
    if (step_size_out && algo == 2) HIP_TRY(hipMemcpy(step_size_out, step.p, 8, hipMemcpyDeviceToHost));
    return MI_OK;
}

}  // namespace host
}  // namespace mi

extern "C" {

void mi_settings_default(mi_settings* s)
{
    if (!s) return;
    std::memset(s, 0, sizeof(*s));
    s->struct_size = sizeof(mi_settings);
    s->rng_seed_value = 0;
    s->n_burnin_draws = 1000;
    s->n_keep_draws = 1000;
    s->n_leap_steps = 1;
    s->step_size = 1.0;
    s->n_adapt_draws = 1000;
    s->target_accept_rate = 0.55;
    s->max_tree_depth = 10;
    s->gamma_val = 0.05;
    s->t0_val = 10.0;
    s->kappa_val = 0.75;
    s->n_fp_steps = 5;
}

const char* mi_mcmc_last_error(void) { return mi::host::last_error().c_str(); }
const char* mi_mcmc_last_kernel(void) { return mi::host::last_kernel().c_str(); }
int mi_mcmc_version(void) { return MI_MCMC_VERSION; }
int mi_mcmc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mi_mcmc_run_user_target(int algo, uint64_t d, mi_small_launch_fn launch, const void* target_pod, uint64_t small_params_bytes,
                            const mi_settings* settings, mi_chains* chains, void* stream)
{
    if (!launch || !settings || !chains) return fail(MI_ERR_BAD_ARG, "null launch / settings / chains");
    if (algo < 0 || algo > 4) return fail(MI_ERR_BAD_ARG, "algo: 0 hmc, 1 mala, 2 nuts, 3 rwmh, 4 rmhmc");
    if (small_params_bytes != sizeof(mi::SmallParams))
        return fail(MI_ERR_BAD_ARG, "the target library was built against another version of the engine headers (SmallParams %llu vs %llu bytes)",
                    (unsigned long long)small_params_bytes, (unsigned long long)sizeof(mi::SmallParams));
    if (settings->struct_size != sizeof(mi_settings) || chains->struct_size != sizeof(mi_chains))
        return fail(MI_ERR_BAD_ARG, "struct_size mismatch (header / library version skew)");
    if (chains->n_chains == 0 || !chains->theta) return fail(MI_ERR_BAD_ARG, "chains.theta and n_chains are required");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(MI_ERR_NO_DEVICE, "no HIP device visible: the engine has no CPU path");
    (void)hipGetLastError();
    static const char* const names[5] = {"hmc", "mala", "nuts", "rwmh", "rmhmc"};
    return run_small(names[algo], algo, d, settings, chains, static_cast<hipStream_t>(stream), [&](const mi::SmallParams& p, hipStream_t s_) {
        return launch(algo, &p, target_pod, s_);
    });
}

int mi_mcmc_run_tile_target(int algo, uint64_t d, int nt, int wpb, uint64_t lds_bytes, mi_tile_launch_fn launch, const void* target_pod,
                            uint64_t tile_params_bytes, int header_version, const mi_settings* settings, mi_chains* chains, void* stream)
{
    if (!launch || !target_pod || !settings || !chains) return fail(MI_ERR_BAD_ARG, "null launch function / target / settings / chains");
    if (header_version != MI_MCMC_VERSION)
        return fail(MI_ERR_BAD_ARG, "the target library was built against engine headers %#x, this library is %#x: rebuild it", header_version, MI_MCMC_VERSION);
    if (tile_params_bytes != sizeof(mi::TileParams))
        return fail(MI_ERR_BAD_ARG, "the target library was built against another version of the engine headers (TileParams %llu vs %llu bytes)",
                    (unsigned long long)tile_params_bytes, (unsigned long long)sizeof(mi::TileParams));
    if (settings->struct_size != sizeof(mi_settings) || chains->struct_size != sizeof(mi_chains))
        return fail(MI_ERR_BAD_ARG, "struct_size mismatch (header / library version skew)");
    if (algo != 0 && algo != 1 && algo != 2) return fail(MI_ERR_UNSUPPORTED, "tile targets: hmc (0), mala (1) and nuts (2) are implemented");
    if (algo == 2 && settings->max_tree_depth > (uint64_t)mi::tile_nuts::NUTS_MAX_DEPTH)
        return fail(MI_ERR_UNSUPPORTED, "tile targets: nuts with max_tree_depth > %d is not implemented on this route", (int)mi::tile_nuts::NUTS_MAX_DEPTH);
    if (!(nt == 1 || nt == 2 || nt == 4 || nt == 8) || d == 0 || d > (uint64_t)16 * nt) return fail(MI_ERR_BAD_ARG, "tile targets: 1 <= d <= 16 NT, NT in {1, 2, 4, 8}");
    if (wpb != 4 && wpb != 8) return fail(MI_ERR_BAD_ARG, "tile targets: WPB is 4 or 8");
    // vals_bound and / or a DIAGONAL precond_mat: hmc and nuts (TileGen, tile_samplers.hpp); mala with a diagonal precond_mat alone (round 6:
    // mala_tile_kernel<T, true>).  mala with bounds and a dense precond_mat stay with the one-lane targets (include/mi_mcmc_target.hpp), which take
    // every combination: refused here with the reason
    GeneralTables gt;
    const bool mala_diag = algo == 1 && !settings->vals_bound && settings->precond_mat != nullptr;
    if (settings->vals_bound || settings->precond_mat) {
        if (algo == 1 && settings->vals_bound) return fail(MI_ERR_UNSUPPORTED, "tile targets: mala with vals_bound is not implemented on this route (one-lane targets, include/mi_mcmc_target.hpp, take it)");
        const int rcg = general_tables("tile targets", settings, d, gt, false);
        if (rcg) return rcg;
    }
    if (chains->n_chains == 0 || !chains->theta) return fail(MI_ERR_BAD_ARG, "chains.theta and n_chains are required");
    const uint64_t n_total = settings->n_burnin_draws + settings->n_keep_draws;
    if (chains->draw0 + n_total > 0xffffffffULL) return fail(MI_ERR_BAD_ARG, "draw0 + draws exceeds the 32-bit draw counter");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(MI_ERR_NO_DEVICE, "no HIP device visible: the engine has no CPU path");
    (void)hipGetLastError();
    int dev = 0, lds_max = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
    if (lds_bytes > (uint64_t)lds_max) return fail(MI_ERR_BAD_ARG, "tile targets: the target asks for %llu bytes of LDS, a workgroup has %d", (unsigned long long)lds_bytes, lds_max);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint64_t lds_user = lds_bytes;
    if (algo == 2) lds_bytes += mi::tile_nuts::lds_doubles() * sizeof(double);     // the sampler's per-level tables behind the target's own LDS
    if (gt.active && !mala_diag) lds_bytes += (uint64_t)(16 * nt * 4 + 8 * nt) * sizeof(double);  // TileGen<NT>::lds_doubles(): bounds / mass tables
    if (mala_diag) lds_bytes += (uint64_t)(3 * 16 * nt) * sizeof(double);           // mala_tile_kernel<T, true>: m | sqrt(m) | 1 / (eps^2 m)
    if (lds_bytes > (uint64_t)lds_max) return fail(MI_ERR_BAD_ARG, "tile targets: target + nuts tables ask for %llu bytes of LDS, a workgroup has %d", (unsigned long long)lds_bytes, lds_max);
    StagedChains sc;
    int rc = stage_in(chains, d, settings->n_keep_draws, sc, st, n_total);
    if (rc) return rc;
    mi::TileParams p{};
    p.d = (uint32_t)d; p.C = chains->n_chains; p.chain0 = chains->chain0;
    p.theta = sc.dev.theta; p.draws = sc.dev.draws; p.n_accept = sc.dev.n_accept; p.n_leap = sc.dev.n_leapfrogs;
    p.seed = settings->rng_seed_value;
    p.n_burnin = (uint32_t)settings->n_burnin_draws; p.n_keep = (uint32_t)settings->n_keep_draws;
    p.n_leap_steps = (uint32_t)settings->n_leap_steps; p.draw0 = (uint32_t)chains->draw0;
    p.eps = settings->step_size;
    p.s2 = settings->step_size * settings->step_size; p.rs = 1.0 / p.s2;
    p.cons_term = -0.5 * (double)d * 1.83787706640934548356;
    {
        const double lii = __builtin_sqrt(p.s2);
        double ld = 0.0;
        for (uint64_t i = 0; i < d; ++i) ld = ld + 2.0 * mi::det_log(lii);
        p.log_det = ld;
    }
    p.lds_user_doubles = (uint32_t)(lds_user / sizeof(double));
    DevBuf sinv_dev;
    if (mala_diag) {
        // Sigma = eps^2 M is constant: INV(Sigma) = diag(1 / (eps^2 m_i)) and LOG_DET(Sigma) = sum_i 2 log sqrt(eps^2 m_i), i ascending -- what the oracle's
        // Gauss-Jordan / Cholesky give for a diagonal matrix (the built-in general mala kernel, mala_dense.hpp, forms the same values)
        std::vector<double> sinv(d);
        double ld = 0.0;
        for (uint64_t i = 0; i < d; ++i) { const double sig = p.s2 * gt.m[i]; sinv[i] = 1.0 / sig; ld = ld + 2.0 * mi::det_log(__builtin_sqrt(sig)); }
        p.log_det = ld;
        HIP_TRY(sinv_dev.alloc(d * 8));
        HIP_TRY(hipMemcpy(sinv_dev.p, sinv.data(), d * 8, hipMemcpyHostToDevice));
        p.m = gt.m_dev.as<double>(); p.m_sqrt = gt.ms_dev.as<double>(); p.s_inv = sinv_dev.as<double>();
    }
    else if (gt.active) {
        p.vals_bound = settings->vals_bound ? 1 : 0;
        p.btype = gt.bt.as<int>(); p.lb = gt.lb.as<double>(); p.ub = gt.ub.as<double>();
        p.m_sqrt = gt.ms_dev.as<double>(); p.m_inv = gt.mi_dev.as<double>();
    }
    WsLease ws;
    if (algo == 2) {                                       // nuts: point records + scalar table of every chain slot (nuts_memo_core.hpp), one workgroup per 64 chains
        if ((rc = nuts_continuation(settings, chains, &p.n_adapt))) return rc;
        p.max_depth = (uint32_t)settings->max_tree_depth;
        p.delta = settings->target_accept_rate; p.eps_bar0 = settings->step_size;
        p.gamma = settings->gamma_val; p.t0 = settings->t0_val; p.kappa = settings->kappa_val;
        p.step_out = sc.dev.step_size; p.depth_trace = sc.dev.nuts_depth; p.adapt_state = sc.dev.nuts_adapt_state;
        const int nt_pad = nt <= 1 ? 1 : nt == 2 ? 2 : nt <= 4 ? 4 : 8;
        // a persistent grid (one workgroup per CU at most) with the chains handed out dynamically: the workspace is sized by the grid's chain slots
        const uint64_t n_wg = mi::nuts_tile_grid(chains->n_chains);
        const size_t vec_bytes = (mi::tile_nuts::ws_bytes_grid(n_wg, nt_pad) + 255) & ~(size_t)255;
        const size_t split_bytes = chains->n_chains > n_wg * 64 ? mi::nuts_split_workspace_bytes(chains->n_chains, 0) : 0;     // (more chains than chain slots: the runs cut into pieces)
        rc = ws_get(st, vec_bytes + 256 + split_bytes, ws);
        if (rc) return rc;
        p.ws = ws.as<double>();
        p.nuts_grid = (uint32_t)n_wg;
        p.next_chain = reinterpret_cast<uint32_t*>(static_cast<char*>(ws.p) + vec_bytes);
        HIP_TRY(hipMemsetAsync(p.next_chain, 0, sizeof(uint32_t), st));
        p.n_exec = sc.dev.n_leapfrogs_executed; sc.exec_written = p.n_exec != nullptr;     // (each distinct state of a doubling is evaluated once)
        if (split_bytes) { const int e = mi::nuts_tile_setup_pieces(p, static_cast<char*>(ws.p) + vec_bytes + 256, st); if (e != 0) return fail(MI_ERR_HIP, "tile target nuts: %s", hipGetErrorString((hipError_t)e)); }
    } else {
        rc = ws_get(st, (size_t)3 * 16 * nt * ((chains->n_chains + 15) / 16 + 8) * 16 * sizeof(double), ws);
        if (rc) return rc;
        p.wsave = ws.as<double>();
    }
    mi::note_kernel("%s_tile%s_kernel<user target, %d>", algo == 0 ? "hmc" : algo == 1 ? "mala" : "nuts", gt.active ? "_gen" : "", (algo == 0 && !gt.active) ? wpb : 4);     // (mala_tile_gen: mala_tile_kernel<T, true>)
    const int e = launch(algo, &p, target_pod, lds_bytes, stream);
    if (e != 0) return fail(MI_ERR_HIP, "tile target kernel launch: %s", hipGetErrorString((hipError_t)e));
    if (algo == 1) { rc = fill_n_leap(sc.dev.n_leapfrogs, chains->n_chains, 0, st); if (rc) return rc; }
    rc = stage_out(chains, d, settings->n_keep_draws, sc, st, n_total);
    if (rc) return rc;
    if (chains->mem == MI_MEM_HOST || gt.active) HIP_TRY(hipStreamSynchronize(st));      // (the tables are ours)
    return MI_OK;
}

int mi_mcmc_run_user_target_v(int algo, uint64_t d, mi_small_launch_fn launch, const void* target_pod, uint64_t small_params_bytes,
                              int header_version, const mi_settings* settings, mi_chains* chains, void* stream)
{
    if (header_version != MI_MCMC_VERSION)
        return fail(MI_ERR_BAD_ARG, "the target library was built against engine headers %#x, this library is %#x: rebuild it", header_version, MI_MCMC_VERSION);
    return mi_mcmc_run_user_target(algo, d, launch, target_pod, small_params_bytes, settings, chains, stream);
}

int mi_mcmc_mat_inverse(const double* A, uint64_t d, double* Ainv)
{
    if (!A || !Ainv) return fail(MI_ERR_BAD_ARG, "mat_inverse: NULL matrix");
    if (d >= linalg_accel().min_d && mi_mcmc_device_count() < 1) return fail(MI_ERR_NO_DEVICE, "mat_inverse: no GPU visible (d >= 64 runs on the device)");
    std::vector<double> out;
    const int rc = host_inverse(A, (size_t)d, out);
    if (rc) return rc;
    std::memcpy(Ainv, out.data(), (size_t)d * d * sizeof(double));
    return MI_OK;
}

int mi_mcmc_mat_cholesky_lower(const double* A, uint64_t d, double* L)
{
    if (!A || !L) return fail(MI_ERR_BAD_ARG, "mat_cholesky_lower: NULL matrix");
    if (d >= linalg_accel().min_d && mi_mcmc_device_count() < 1) return fail(MI_ERR_NO_DEVICE, "mat_cholesky_lower: no GPU visible (d >= 64 runs on the device)");
    std::vector<double> out;
    const int rc = host_cholesky_lower(A, (size_t)d, out);
    if (rc) return rc;
    std::memcpy(L, out.data(), (size_t)d * d * sizeof(double));
    return MI_OK;
}

int mi_mcmc_release_workspace(void* stream, int all_streams, uint64_t* bytes_freed)
{
    linalg_memo_clear();         // the calling thread's two memoised inverses of a precond_mat (host_linalg.hpp: 2 MiB each at d = 512)
    return ws_release(static_cast<hipStream_t>(stream), all_streams != 0, bytes_freed);
}

int mi_mcmc_hmc_run(const mi_target* target, const mi_settings* settings, mi_chains* chains, void* stream)
{
    int rc = check_common(target, settings, chains, true);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint64_t d = target->d;
    if (chains->mass_diag) {
        // per-chain diagonal masses: separable Gaussians without bounds on the elementwise kernels (below), everything else literally
        const bool sep = target->kind == MI_TARGET_GAUSS_ISO || target->kind == MI_TARGET_GAUSS_DIAG;
        // ... a dense precision with d <= 128, no bounds: the MFMA kernel with per-chain tables (round 4; the literal kernel served it before)
        const bool dense_mfma = target->kind == MI_TARGET_GAUSS_DENSE && d <= 128 && !settings->vals_bound;
        if ((!sep && !dense_mfma) || settings->vals_bound) {
            if (target->kind == MI_TARGET_NORMAL_MODEL) return fail(MI_ERR_UNSUPPORTED, "hmc: chains.mass_diag with the normal-model target is not implemented");
            return run_literal("hmc", 0, target, settings, chains, st);
        }
    }
    if (target->kind == MI_TARGET_NORMAL_MODEL) return run_small_normal_model("hmc", 0, target, settings, chains, st);
    if (target->kind == MI_TARGET_LOGISTIC && gemm_case(target, settings, chains, true)) return run_gemm("hmc", 0, target, settings, chains, st);   // d > 512, plain: two matrix products per leapfrog step
    if (target->kind == MI_TARGET_LOGISTIC) {      // plain: the LDS-staged MFMA kernel (d <= 512); bounds / precond_mat: one chain per lane (d <= 8); else literal.hpp
        // a DIAGONAL precond_mat alone rides the LDS-staged kernel too (its DIAGM instantiation: two tables read from global memory)
        // ... and so do bounds (its BOUNDS instantiation, lds_box.hpp), with the identity or a diagonal matrix
        // ... and, round 5, a DENSE precond_mat without bounds (DENSEM: INV(M) and CHOL_LOWER(M) streamed through LDS like X)
        const bool lds_general = (settings->vals_bound || settings->precond_mat) && (lds_general_ok(target, settings) || lds_dense_m_ok(target, settings))
                                 && d > (uint64_t)mi::SMALL_MAX_D && d <= 512;
        if ((settings->vals_bound || settings->precond_mat) && !lds_general)
            return d <= (uint64_t)mi::SMALL_MAX_D ? run_small_logistic("hmc", 0, target, settings, chains, st) : run_literal("hmc", 0, target, settings, chains, st);
        return d <= 512 ? run_logit_plain("hmc", mi::LOGIT_HMC, target, settings, chains, st) : run_literal("hmc", 0, target, settings, chains, st);
    }
    if (target->kind != MI_TARGET_GAUSS_ISO && target->kind != MI_TARGET_GAUSS_DIAG && target->kind != MI_TARGET_GAUSS_DENSE)
        return fail(MI_ERR_UNSUPPORTED, "hmc: target kind %d not implemented", target->kind);
    // precond_mat (hmc.cpp:57-59): a DIAGONAL matrix is supported (INV and CHOL_LOWER of a diagonal matrix are the
    // element-wise 1/m and sqrt(m), exactly what the oracle's Gauss-Jordan / Cholesky produce); dense is not yet
    std::vector<double> m_sqrt, m_inv;
    bool dense_m = false;
    if (settings->precond_mat) {
        m_sqrt.resize(d); m_inv.resize(d);
        for (uint64_t i = 0; i < d; ++i)
            for (uint64_t k = 0; k < d; ++k) {
                const double v = settings->precond_mat[i * d + k];
                if (i != k && v != 0.0) dense_m = true;
                if (i == k) { m_sqrt[i] = __builtin_sqrt(v); m_inv[i] = 1.0 / v; }
            }
        // a dense matrix: INV / CHOL_LOWER on the host; three fragment sets in LDS (d <= 64), or the two extra ones read from L2
        // in fragment order (64 < d <= 128); target must be an MFMA one
        if (dense_m && target->kind == MI_TARGET_LOGISTIC) return fail(MI_ERR_UNSUPPORTED, "hmc: precond_mat with the logistic target is not implemented");
    }
    // beyond d = 128 the tiled kernels serve separable targets without bounds (identity or diagonal precond_mat: hmc_diag.hpp);
    // everything else there -- dense gradients, bounds, a dense precond_mat -- runs on the literal kernel (literal.hpp)
    if (d > 128 && d <= 512 && target->kind == MI_TARGET_GAUSS_DENSE && !(dense_m && settings->vals_bound) && target->kernel_hint != MI_KERNEL_LITERAL)
        return run_dense_lds("hmc", mi::LOGIT_HMC, target, settings, chains, st);      // P streamed through LDS (logistic_lds.hpp); identity or diagonal precond_mat, with or without bounds; a dense one without
    if (gemm_case(target, settings, chains, true)) return run_gemm("hmc", 0, target, settings, chains, st);      // one matrix product per leapfrog step (gemm_samplers.hip)
    if (d > 128 && (target->kind == MI_TARGET_GAUSS_DENSE || settings->vals_bound || dense_m))
        return run_literal("hmc", 0, target, settings, chains, st);
    const bool bounded = settings->vals_bound != 0 || settings->precond_mat != nullptr;   // the general kernel variant
    if (settings->vals_bound && (!settings->lower_bounds || !settings->upper_bounds))
        return fail(MI_ERR_BAD_ARG, "hmc: vals_bound needs lower_bounds and upper_bounds");
    const bool separable_kind = target->kind == MI_TARGET_GAUSS_ISO || target->kind == MI_TARGET_GAUSS_DIAG;
    // separable target + diagonal precond_mat (no bounds): the elementwise kernel, any d
    const bool diag_precond_elementwise = separable_kind && !settings->vals_bound && settings->precond_mat && !dense_m && d > 128;
    if (bounded && d > 128 && !diag_precond_elementwise) return fail(MI_ERR_UNSUPPORTED, "hmc: vals_bound with d > 128 is not implemented");
    if (settings->n_burnin_draws + settings->n_keep_draws > 0xffffffffULL) return fail(MI_ERR_BAD_ARG, "too many draws");
    const bool separable = target->kind != MI_TARGET_GAUSS_DENSE;
    // kernel_hint (mi_mcmc.h): an explicit request for the elementwise kernels where the MFMA kernel would be picked; every
    // kernel produces the same bits, so the hint changes speed only
    const bool force_diag = !bounded && (target->kernel_hint == MI_KERNEL_ELEMENTWISE_1LANE || target->kernel_hint == MI_KERNEL_ELEMENTWISE_4LANE);
    if (d > 128 && !separable)
        return fail(MI_ERR_UNSUPPORTED, "hmc: d = %llu > 128 not implemented for dense-gradient targets", (unsigned long long)d);
    if (separable && (d > 128 || force_diag || chains->mass_diag)) {
        if (dense_m) return fail(MI_ERR_UNSUPPORTED, "hmc: a dense precond_mat is implemented for d <= 128");
        // no contraction: the elementwise (lane-per-chain) kernel
        DevBuf prec_owned;
        const double* prec_dev = nullptr;
        if (target->kind == MI_TARGET_GAUSS_DIAG) {
            if (!target->prec) return fail(MI_ERR_BAD_ARG, "GAUSS_DIAG needs prec (d)");
            if (target->mem == MI_MEM_DEVICE) prec_dev = target->prec;
            else {
                HIP_TRY(prec_owned.alloc(d * sizeof(double)));
                HIP_TRY(hipMemcpy(prec_owned.p, target->prec, d * sizeof(double), hipMemcpyHostToDevice));
                prec_dev = prec_owned.as<double>();
            }
        }
        StagedChains sc;
        rc = stage_in(chains, d, settings->n_keep_draws, sc, st);
        if (rc) return rc;
        mi::HmcDiagParams q{};
        q.prec = prec_dev; q.d = (uint32_t)d; q.C = chains->n_chains; q.chain0 = chains->chain0;
        q.theta = sc.dev.theta; q.draws = sc.dev.draws; q.n_accept = sc.dev.n_accept; q.n_leap = sc.dev.n_leapfrogs;
        q.seed = settings->rng_seed_value;
        q.n_burnin = (uint32_t)settings->n_burnin_draws; q.n_keep = (uint32_t)settings->n_keep_draws;
        q.n_leap_steps = (uint32_t)settings->n_leap_steps; q.eps = settings->step_size;
        q.draw0 = (uint32_t)chains->draw0;
        WsLease scratch;
        ReplayWs rp = replay_layout(2 * d * chains->n_chains * sizeof(double), chains->n_chains, (uint32_t)d, 0, false, /*needs_matrix=*/false);
        rc = ws_get(st, rp.total_bytes, scratch);
        if (rc) return rc;
        q.scratch = scratch.as<double>();
        rc = replay_bind(rp, scratch.p, chains->n_chains, st);
        if (rc) return rc;
        q.nf_flag = rp.flag;
        int diag_lanes = mi::hmc_diag_pick_lanes(q.C);          // lanes per chain (hmc_diag.hpp)
        if (target->kernel_hint == MI_KERNEL_ELEMENTWISE_1LANE) diag_lanes = 1;
        if (target->kernel_hint == MI_KERNEL_ELEMENTWISE_4LANE) diag_lanes = 4;
        DevBuf ms_d, mi_d;
        ChainMass cm;
        if (sc.dev.mass_diag) {                          // per-chain tables [d][C]
            if ((rc = chain_mass_tables(sc.dev.mass_diag, d, q.C, cm, st))) return rc;
            q.m_sqrt = cm.ms.as<double>(); q.m_inv = cm.mi.as<double>(); q.m_per_chain = 1;
            mi::note_kernel("hmc_diag%d_kernel<true>", diag_lanes == 4 ? 4 : 1);
            if (diag_lanes == 4) hipLaunchKernelGGL(mi::hmc_diag4_kernel<true>, dim3((unsigned)((q.C + 63) / 64)), dim3(256), 0, st, q);
            else hipLaunchKernelGGL(mi::hmc_diag1_kernel<true>, dim3((unsigned)((q.C + 255) / 256)), dim3(256), 0, st, q);
        } else if (diag_precond_elementwise) {
            HIP_TRY(ms_d.alloc(d * 8)); HIP_TRY(mi_d.alloc(d * 8));
            HIP_TRY(hipMemcpy(ms_d.p, m_sqrt.data(), d * 8, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(mi_d.p, m_inv.data(), d * 8, hipMemcpyHostToDevice));
            q.m_sqrt = ms_d.as<double>(); q.m_inv = mi_d.as<double>();
            mi::note_kernel("hmc_diag%d_kernel<true>", diag_lanes == 4 ? 4 : 1);
            if (diag_lanes == 4) hipLaunchKernelGGL(mi::hmc_diag4_kernel<true>, dim3((unsigned)((q.C + 63) / 64)), dim3(256), 0, st, q);
            else hipLaunchKernelGGL(mi::hmc_diag1_kernel<true>, dim3((unsigned)((q.C + 255) / 256)), dim3(256), 0, st, q);
        } else {
        mi::note_kernel("hmc_diag%d_kernel<false>", diag_lanes == 4 ? 4 : 1);
        if (diag_lanes == 4) hipLaunchKernelGGL(mi::hmc_diag4_kernel<false>, dim3((unsigned)((q.C + 63) / 64)), dim3(256), 0, st, q);
        else hipLaunchKernelGGL(mi::hmc_diag1_kernel<false>, dim3((unsigned)((q.C + 255) / 256)), dim3(256), 0, st, q);
        }
        HIP_TRY(hipGetLastError());
        {   // chains that reached the non-finite regime: replayed literally (literal.hpp)
            mi::lit::LitParams lp{};
            rc = lit_gauss_target(lp.t, target->kind, (uint32_t)d, nullptr, prec_dev, rp.tbuf, st);
            if (rc) return rc;
            lit_common(lp, settings, &sc.dev, rp, false);
            if (sc.dev.mass_diag) lit_set_chain_mass(lp, sc.dev.mass_diag, cm, q.C);
            else if (diag_precond_elementwise) { lp.precond = 1; lp.m_sqrt = ms_d.as<double>(); lp.m_inv = mi_d.as<double>(); }
            rc = launched("hmc (literal replay)", mi::launch_literal(0, lp, rp.n_wg, st));
            if (rc) return rc;
        }
        if (diag_precond_elementwise || sc.dev.mass_diag) HIP_TRY(hipStreamSynchronize(st));          // the tables are ours
        rc = stage_out(chains, d, settings->n_keep_draws, sc, st);
        if (rc) return rc;
        if (prec_owned.p || chains->mem == MI_MEM_HOST) HIP_TRY(hipStreamSynchronize(st));
        return MI_OK;
    }

    DevBuf P_owned;
    const double* P_dev = nullptr;
    rc = dense_precision_on_device(target, P_owned, &P_dev, st);
    if (rc) return rc;
    StagedChains sc;
    rc = stage_in(chains, d, settings->n_keep_draws, sc, st);
    if (rc) return rc;

    mi::HmcParams prm{};
    prm.P = P_dev;
    prm.d = (uint32_t)d;
    prm.sep_target = target->kind != MI_TARGET_GAUSS_DENSE;     // ISO / DIAG: the general variants take the gradient element-wise, as the reference's target function does (hmc_dense.hpp: target_times)
    prm.C = chains->n_chains;
    prm.chain0 = chains->chain0;
    prm.theta = sc.dev.theta;
    // stream-ordered workspace: P * theta of the last accepted state, [d][C]
    WsLease wsave;
    const size_t d_pad_h = (d <= 16) ? 16 : (d <= 32) ? 32 : (d <= 64) ? 64 : 128;
    ReplayWs rp = replay_layout(3 * d_pad_h * ((chains->n_chains + 15) / 16 + 8) * 16 * sizeof(double), chains->n_chains, (uint32_t)d, 0, false);
    rc = ws_get(st, rp.total_bytes, wsave);
    if (rc) return rc;
    prm.wsave = wsave.as<double>();
    // a diagonal precond_mat alone (no bounds): the plain kernel's shape with two mass tables (hmc_dense.hpp, DIAGM), replay as the plain kernel;
    // per-chain diagonal masses (mi_chains.mass_diag) ride the same variant with [d][C] tables in global memory (PCM)
    const bool chain_mass = sc.dev.mass_diag != nullptr;
    const bool diag_only = (settings->precond_mat != nullptr && !dense_m && !settings->vals_bound) || chain_mass;
    const bool general = bounded && !diag_only;
    if (!general) {                                     // the general variants reproduce the dense products themselves
        rc = replay_bind(rp, wsave.p, chains->n_chains, st);
        if (rc) return rc;
        prm.nf_flag = rp.flag;
    }
    prm.draws = sc.dev.draws;
    prm.n_accept = sc.dev.n_accept;
    prm.n_leap = sc.dev.n_leapfrogs;
    prm.seed = settings->rng_seed_value;
    prm.n_burnin = (uint32_t)settings->n_burnin_draws;
    prm.n_keep = (uint32_t)settings->n_keep_draws;
    prm.n_leap_steps = (uint32_t)settings->n_leap_steps;
    prm.eps = settings->step_size;
    prm.draw0 = (uint32_t)chains->draw0;
    prm.stagger = 40;
#ifdef MI_PROFILING     // A/B library only (make prof): the shipped library reads no environment variable
    if (const char* e = getenv("MI_HMC_STAGGER")) prm.stagger = (uint32_t)atoi(e);
    if (const char* e = getenv("MI_HMC_ABLATE")) prm.ablate = (uint32_t)atoi(e);
#endif

    const int nt = (int)((d + 15) / 16);
    DevBuf bt_dev, lb_dev, ub_dev;
    if (general) {
        // determine_bounds_type (determine_bounds_type.hpp:27-57): 1 none, 2 lower, 3 upper, 4 both
        std::vector<int> bt(d, 1);
        std::vector<double> lbv(d, 0.0), ubv(d, 0.0);
        if (settings->vals_bound)
            for (uint64_t i = 0; i < d; ++i) {
                lbv[i] = settings->lower_bounds[i]; ubv[i] = settings->upper_bounds[i];
                const bool fl = std::isfinite(lbv[i]), fu = std::isfinite(ubv[i]);
                bt[i] = (fl && fu) ? 4 : (fl && !fu) ? 2 : (!fl && fu) ? 3 : 1;
            }
        if (m_sqrt.empty()) { m_sqrt.assign(d, 1.0); m_inv.assign(d, 1.0); }
        DevBuf ms_dev, mi_dev;
        HIP_TRY(bt_dev.alloc(d * sizeof(int))); HIP_TRY(lb_dev.alloc(d * 8)); HIP_TRY(ub_dev.alloc(d * 8));
        HIP_TRY(ms_dev.alloc(d * 8)); HIP_TRY(mi_dev.alloc(d * 8));
        HIP_TRY(hipMemcpy(bt_dev.p, bt.data(), d * sizeof(int), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(lb_dev.p, lbv.data(), d * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ub_dev.p, ubv.data(), d * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ms_dev.p, m_sqrt.data(), d * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(mi_dev.p, m_inv.data(), d * 8, hipMemcpyHostToDevice));
        prm.btype = bt_dev.as<int>(); prm.lb = lb_dev.as<double>(); prm.ub = ub_dev.as<double>();
        prm.m_sqrt = ms_dev.as<double>(); prm.m_inv = mi_dev.as<double>();
        prm.vals_bound = settings->vals_bound ? 1 : 0;
        DevBuf minv_dev, l_dev;
        if (dense_m) {
            std::vector<double> Minv, L;
            rc = host_inverse(settings->precond_mat, d, Minv); if (rc) return rc;
            rc = host_cholesky_lower(settings->precond_mat, d, L); if (rc) return rc;
            rc = upload_matrix(Minv, d, minv_dev); if (rc) return rc;
            rc = upload_matrix(L, d, l_dev); if (rc) return rc;
            prm.Minv = minv_dev.as<double>(); prm.Lchol = l_dev.as<double>();
        }
        rc = launched("hmc", mi::launch_hmc_gauss(prm, nt, true, dense_m, st));
        if (!rc) HIP_TRY(hipStreamSynchronize(st));     // bounds buffers are ours
    }
    else {
        // launch shape of the plain kernel (64 < d <= 128): with fewer 16-chain tiles than the chip has wave slots, give a tile
        // a whole SIMD, or two / four of them (hmc_split.hpp).  Estimated cost = rounds of workgroups x time of one round,
        // the round times relative to the two-waves-per-SIMD shape as measured on MI355X.
        int shape = 0;
        if (nt > 4) {
            int dev = 0, n_cu = 256;
            HIP_TRY(hipGetDevice(&dev));
            (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
            if (n_cu <= 0) n_cu = 256;
            const uint64_t C = chains->n_chains;
            auto rounds = [&](uint64_t chains_per_wg) { return (double)(((C + chains_per_wg - 1) / chains_per_wg + n_cu - 1) / n_cu); };
            const double cost[5] = {rounds(128) * 1.00, rounds(64) * MI_HMC_COST_1WAVE, rounds(32) * MI_HMC_COST_SPLIT2,
                                    rounds(32) * MI_HMC_COST_SPLIT4X2, rounds(16) * MI_HMC_COST_SPLIT4};
            for (int k = 1; k < 5; ++k) if (cost[k] < cost[shape]) shape = k;
            switch (target->kernel_hint) {
            case MI_KERNEL_HMC_TWO_WAVES_PER_SIMD: shape = 0; break;
            case MI_KERNEL_HMC_ONE_WAVE_PER_SIMD: shape = 1; break;
            case MI_KERNEL_HMC_SPLIT2: shape = 2; break;
            case MI_KERNEL_HMC_SPLIT4_TWO_WAVES: shape = 3; break;
            case MI_KERNEL_HMC_SPLIT4: shape = 4; break;
            default: break;
            }
        }
        DevBuf ms_dev, mi_dev;
        ChainMass cm;
        if (chain_mass) {
            if ((rc = chain_mass_tables(sc.dev.mass_diag, d, chains->n_chains, cm, st))) return rc;
            prm.m_sqrt = cm.ms.as<double>(); prm.m_inv = cm.mi.as<double>(); prm.m_per_chain = 1;
            rc = launched("hmc", mi::launch_hmc_gauss_diagm(prm, nt, st));
        } else if (diag_only) {
            HIP_TRY(ms_dev.alloc(d * 8)); HIP_TRY(mi_dev.alloc(d * 8));
            HIP_TRY(hipMemcpy(ms_dev.p, m_sqrt.data(), d * 8, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(mi_dev.p, m_inv.data(), d * 8, hipMemcpyHostToDevice));
            prm.m_sqrt = ms_dev.as<double>(); prm.m_inv = mi_dev.as<double>();
            rc = launched("hmc", mi::launch_hmc_gauss_diagm(prm, nt, st));
        } else
        rc = shape == 0 ? launched("hmc", mi::launch_hmc_gauss(prm, nt, false, false, st))
                        : launched("hmc", mi::launch_hmc_gauss_few_chains(prm, shape, st));
        if (rc) return rc;
        mi::lit::LitParams lp{};                        // chains that reached the non-finite regime: replayed literally (literal.hpp)
        rc = lit_gauss_target(lp.t, target->kind, (uint32_t)d, P_dev, nullptr, rp.tbuf, st);
        if (rc) return rc;
        lit_common(lp, settings, &sc.dev, rp, false);
        if (chain_mass) lit_set_chain_mass(lp, sc.dev.mass_diag, cm, chains->n_chains);
        else if (diag_only) { lp.precond = 1; lp.m_sqrt = ms_dev.as<double>(); lp.m_inv = mi_dev.as<double>(); }
        rc = launched("hmc (literal replay)", mi::launch_literal(0, lp, rp.n_wg, st));
        if (!rc && diag_only) HIP_TRY(hipStreamSynchronize(st));     // the tables are ours
    }
    if (rc) return rc;

    rc = stage_out(chains, d, settings->n_keep_draws, sc, st);
    if (rc) return rc;
    // buffers we own (staged target / host-mode chains) must outlive the kernel
    if (P_owned.p || chains->mem == MI_MEM_HOST) HIP_TRY(hipStreamSynchronize(st));
    return MI_OK;
}

// per-dimension variance of the chains' current states, pooled over the chains: out[d] (host); theta [d][C] in `mem`
namespace {
__global__ __launch_bounds__(256) void pooled_variance_kernel(const double* __restrict__ theta, uint64_t C, double* __restrict__ out)
{
    // one workgroup per dimension: mean, then the centred second moment (two passes, fixed reduction tree)
    __shared__ double red[256];
    const double* row = theta + (size_t)blockIdx.x * C;
    auto block_sum = [&](double v) -> double {
        red[threadIdx.x] = v;
        __syncthreads();
        for (int m = 128; m >= 1; m >>= 1) { if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m]; __syncthreads(); }
        const double r = red[0];
        __syncthreads();
        return r;
    };
    double s = 0.0;
    for (uint64_t c = threadIdx.x; c < C; c += 256) s += row[c];
    const double mean = block_sum(s) / (double)C;
    double q = 0.0;
    for (uint64_t c = threadIdx.x; c < C; c += 256) { const double e = row[c] - mean; q = __builtin_fma(e, e, q); }
    const double var = block_sum(q) / (double)(C > 1 ? C - 1 : 1);
    if (threadIdx.x == 0) out[blockIdx.x] = var;
}
}  // namespace

int mi_mcmc_hmc_run_mass_adapted(const mi_target* target, const mi_settings* settings, mi_chains* chains, uint32_t n_windows,
                                 double* mass_diag_out, void* stream)
{
    int rc = check_common(target, settings, chains);
    if (rc) return rc;
    if (settings->precond_mat) return fail(MI_ERR_BAD_ARG, "hmc (mass adapted): settings.precond_mat must be NULL, the mass matrix is estimated");
    if (target->kind != MI_TARGET_GAUSS_ISO && target->kind != MI_TARGET_GAUSS_DIAG && target->kind != MI_TARGET_GAUSS_DENSE &&
        target->kind != MI_TARGET_LOGISTIC)
        return fail(MI_ERR_UNSUPPORTED, "hmc (mass adapted): implemented for the Gaussian and the logistic-regression targets (a diagonal precond_mat on the device path)");
    if (chains->n_chains < 2) return fail(MI_ERR_BAD_ARG, "hmc (mass adapted): the mass is pooled over the chains, at least 2 are needed");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint64_t d = target->d, C = chains->n_chains;
    std::vector<double> var(d), M(d * d, 0.0), mass(d, 1.0);
    DevBuf var_dev, theta_stage;
    HIP_TRY(var_dev.alloc(d * 8));
    auto estimate = [&]() -> int {                       // mass_i = 1 / pooled variance of dimension i over the chains' current states
        const double* th = chains->theta;
        if (chains->mem == MI_MEM_HOST) {
            if (!theta_stage.p) HIP_TRY(theta_stage.alloc(d * C * 8));
            HIP_TRY(hipMemcpyAsync(theta_stage.p, chains->theta, d * C * 8, hipMemcpyHostToDevice, st));
            th = theta_stage.as<double>();
        }
        hipLaunchKernelGGL(pooled_variance_kernel, dim3((unsigned)d), dim3(256), 0, st, th, C, var_dev.as<double>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(var.data(), var_dev.p, d * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (uint64_t i = 0; i < d; ++i) {
            const double m = 1.0 / var[i];
            mass[i] = (std::isfinite(m) && m > 0.0) ? m : 1.0;
            M[i * d + i] = mass[i];
        }
        return MI_OK;
    };
    const uint64_t n_burnin = settings->n_burnin_draws;
    if ((uint64_t)n_windows > settings->n_burnin_draws)
        return fail(MI_ERR_BAD_ARG, "hmc (mass adapted): n_windows = %u exceeds n_burnin_draws = %llu (each re-estimation window needs a draw)", n_windows, (unsigned long long)settings->n_burnin_draws);
    const uint64_t n_parts = (uint64_t)n_windows + 1;
    uint64_t done = 0;
    rc = estimate();                                     // from the spread of initial_vals
    if (rc) return rc;
    for (uint64_t part = 0; part < n_parts; ++part) {
        const bool last = part + 1 == n_parts;
        const uint64_t upto = last ? n_burnin : (n_burnin * (part + 1)) / n_parts;
        mi_settings s_ = *settings;
        s_.precond_mat = M.data();
        s_.n_burnin_draws = upto - done;
        s_.n_keep_draws = last ? settings->n_keep_draws : 0;
        mi_chains c_ = *chains;
        c_.draw0 = chains->draw0 + done;
        if (!last) { c_.draws = nullptr; c_.n_accept = nullptr; }
        if (s_.n_burnin_draws + s_.n_keep_draws > 0) {
            rc = mi_mcmc_hmc_run(target, &s_, &c_, stream);
            if (rc) return rc;
            HIP_TRY(hipStreamSynchronize(st));           // M (host) is read during the call's staging; the next estimate reads theta
        }
        done = upto;
        if (!last) { rc = estimate(); if (rc) return rc; }
    }
    if (chains->n_leapfrogs) {                           // every part's call reported its own count: the run's total is what the caller gets
        const uint64_t total = (settings->n_burnin_draws + settings->n_keep_draws) * settings->n_leap_steps;
        if (chains->mem == MI_MEM_DEVICE) { rc = fill_n_leap(chains->n_leapfrogs, C, total, st); if (rc) return rc; }
        else for (uint64_t c = 0; c < C; ++c) chains->n_leapfrogs[c] = total;
        if (chains->n_leapfrogs_executed) {              // hmc executes what it counts: the same total (the header's "equal to n_leapfrogs")
            if (chains->mem == MI_MEM_DEVICE) { rc = fill_n_leap(chains->n_leapfrogs_executed, C, total, st); if (rc) return rc; }
            else for (uint64_t c = 0; c < C; ++c) chains->n_leapfrogs_executed[c] = total;
        }
    }
    if (mass_diag_out) std::memcpy(mass_diag_out, mass.data(), d * 8);
    return MI_OK;
}

// per chain and dimension: variance of the chain's own draws of one burn-in part (slab [n][d][C]; two passes, draws ascending),
// regularised as Stan regularises its windows, and inverted: mass [d][C]
namespace {
__global__ __launch_bounds__(256) void chain_mass_estimate_kernel(const double* __restrict__ slab, uint32_t n, uint64_t dC, double* __restrict__ mass)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;      // e = i * C + c: coalesced over the chains
    if (e >= dC) return;
    double s = 0.0;
    for (uint32_t t = 0; t < n; ++t) s = s + slab[(size_t)t * dC + e];
    const double mean = s / (double)n;
    double q = 0.0;
    for (uint32_t t = 0; t < n; ++t) { const double x = slab[(size_t)t * dC + e] - mean; q = __builtin_fma(x, x, q); }
    const double var = q / (double)(n - 1);
    const double reg = ((double)n * var + 5.0e-3) / ((double)n + 5.0);
    const double m = 1.0 / reg;
    mass[e] = (mi::is_finite(m) && m > 0.0) ? m : 1.0;
}
}  // namespace

int mi_mcmc_hmc_run_mass_adapted_per_chain(const mi_target* target, const mi_settings* settings, mi_chains* chains, uint32_t n_windows,
                                           double first_step_size, double* mass_diag_out, void* stream)
{
    int rc = check_common(target, settings, chains);
    if (rc) return rc;
    if (settings->precond_mat) return fail(MI_ERR_BAD_ARG, "hmc (per-chain mass): settings.precond_mat must be NULL, the mass matrices are estimated");
    if (n_windows == 0) return fail(MI_ERR_BAD_ARG, "hmc (per-chain mass): n_windows must be at least 1");
    const uint64_t d = target->d, C = chains->n_chains, n_burnin = settings->n_burnin_draws;
    const uint64_t n_parts = (uint64_t)n_windows + 1;
    if (n_burnin / n_parts < 3)
        return fail(MI_ERR_BAD_ARG, "hmc (per-chain mass): n_burnin_draws = %llu leaves fewer than 3 draws for each of the %llu parts (a chain estimates its variances from the draws of one part)",
                    (unsigned long long)n_burnin, (unsigned long long)n_parts);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // everything runs in device memory; a caller with host buffers gets them staged once, not once per part
    StagedChains sc;
    rc = stage_in(chains, d, settings->n_keep_draws, sc, st);
    if (rc) return rc;
    DevBuf mass, slab;
    HIP_TRY(mass.alloc(d * C * 8));
    uint64_t max_part = 0;
    for (uint64_t part = 0; part + 1 < n_parts; ++part) {
        const uint64_t len = (n_burnin * (part + 1)) / n_parts - (n_burnin * part) / n_parts;
        max_part = len > max_part ? len : max_part;
    }
    if (hipMalloc(&slab.p, max_part * d * C * 8) != hipSuccess) {
        (void)hipGetLastError();
        return fail(MI_ERR_HIP, "hmc (per-chain mass): no memory for the draws of one burn-in part (%llu x d x C doubles): use more windows",
                    (unsigned long long)max_part);
    }
    uint64_t done = 0;
    for (uint64_t part = 0; part < n_parts; ++part) {
        const bool last = part + 1 == n_parts;
        const uint64_t upto = last ? n_burnin : (n_burnin * (part + 1)) / n_parts;
        mi_settings s_ = *settings;
        mi_chains c_ = sc.dev;
        c_.draw0 = chains->draw0 + done;
        c_.mass_diag = part == 0 ? nullptr : mass.as<double>();       // part 0: M = I, on the raw target's scale
        if (part == 0 && first_step_size != 0.0) s_.step_size = first_step_size;
        if (last) {
            s_.n_burnin_draws = upto - done; s_.n_keep_draws = settings->n_keep_draws;
        } else {                                                     // the part's draws are what the chain learns from
            s_.n_burnin_draws = 0; s_.n_keep_draws = upto - done;
            c_.draws = slab.as<double>(); c_.n_accept = nullptr;
        }
        if (s_.n_burnin_draws + s_.n_keep_draws > 0) {
            rc = mi_mcmc_hmc_run(target, &s_, &c_, stream);
            if (rc) return rc;
        }
        if (!last) {
            const uint64_t n = upto - done, dC = d * C;
            hipLaunchKernelGGL(chain_mass_estimate_kernel, dim3((unsigned)((dC + 255) / 256)), dim3(256), 0, st, slab.as<double>(), (uint32_t)n, dC, mass.as<double>());
            HIP_TRY(hipGetLastError());
        }
        done = upto;
    }
    rc = fill_n_leap(sc.dev.n_leapfrogs, C, (settings->n_burnin_draws + settings->n_keep_draws) * settings->n_leap_steps, st);
    if (rc) return rc;
    if (mass_diag_out)
        HIP_TRY(hipMemcpyAsync(mass_diag_out, mass.p, d * C * 8, chains->mem == MI_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, st));
    rc = stage_out(chains, d, settings->n_keep_draws, sc, st);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(st));                   // mass / slab are ours
    return MI_OK;
}

int mi_mcmc_rmhmc_run_callback(const double* initial_vals, uint64_t d, mi_log_kernel_cb target_log_kernel, void* target_data,
                               mi_tensor_cb tensor_fn, void* tensor_data, const mi_settings* settings, double* draws_out,
                               uint64_t* n_accept_draws)
{
    if (!tensor_fn) return fail(MI_ERR_BAD_ARG, "rmhmc (callback): null tensor_fn");
    return mi::host::literal_run_callback("rmhmc", 4, initial_vals, d, target_log_kernel, target_data, tensor_fn, tensor_data, settings, draws_out,
                                          n_accept_draws, nullptr);
}

int mi_mcmc_mala_run(const mi_target* target, const mi_settings* settings, mi_chains* chains, void* stream)
{
    int rc = check_common(target, settings, chains);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint64_t d = target->d;
    if (settings->n_burnin_draws + settings->n_keep_draws > 0xffffffffULL) return fail(MI_ERR_BAD_ARG, "too many draws");
    if (target->kind == MI_TARGET_NORMAL_MODEL) return run_small_normal_model("mala", 1, target, settings, chains, st);
    // a DIAGONAL precond_mat alone rides the LDS-staged kernel too (its DIAGM instantiation)
    // ... and, round 5, a DENSE one without bounds (DENSEM: M, CHOL_LOWER(M) and INV(eps^2 M) streamed through LDS like X)
    const bool mala_diag_alone = !settings->vals_bound && (precond_is_diagonal(settings, d) || lds_dense_m_ok(target, settings)) && d > (uint64_t)mi::SMALL_MAX_D && d <= 512;
    if (target->kind == MI_TARGET_LOGISTIC && gemm_case(target, settings, chains, false)) return run_gemm("mala", 1, target, settings, chains, st);    // d > 512, identity or a diagonal precond_mat, no bounds
    if (target->kind == MI_TARGET_LOGISTIC && (settings->vals_bound || settings->precond_mat) && !mala_diag_alone)
        return d <= (uint64_t)mi::SMALL_MAX_D ? run_small_logistic("mala", 1, target, settings, chains, st) : run_literal("mala", 1, target, settings, chains, st);
    if (target->kind == MI_TARGET_LOGISTIC && d > 512) return run_literal("mala", 1, target, settings, chains, st);
    // Sigma = eps^2 * I (mala.ipp:41,63): INV by Gauss-Jordan gives diag(1/s2); CHOL gives diag(sqrt(s2));
    // LOG_DET = sum_i 2 log L_ii accumulated sequentially, exactly as the oracle states it.
    const double s2_ = settings->step_size * settings->step_size;
    double log_det_ = 0.0;
    {
        const double lii = __builtin_sqrt(s2_);
        for (uint64_t i = 0; i < d; ++i) log_det_ = log_det_ + 2.0 * mi::det_log(lii);
    }
    if (target->kind == MI_TARGET_LOGISTIC) {
        if (!target->X || !target->y || target->n_rows == 0) return fail(MI_ERR_BAD_ARG, "LOGISTIC needs X, y, n_rows");
        if (d > 512) return fail(MI_ERR_UNSUPPORTED, "mala: logistic target with d = %llu > 512 not implemented", (unsigned long long)d);
        const uint64_t n = target->n_rows;
        DevBuf Xo, yo;
        const double *X_dev = target->X, *y_dev = target->y;
        if (target->mem == MI_MEM_HOST) {
            HIP_TRY(Xo.alloc(n * d * sizeof(double))); HIP_TRY(yo.alloc(n * sizeof(double)));
            HIP_TRY(hipMemcpy(Xo.p, target->X, n * d * sizeof(double), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(yo.p, target->y, n * sizeof(double), hipMemcpyHostToDevice));
            X_dev = Xo.as<double>(); y_dev = yo.as<double>();
        }
        StagedChains sc;
        rc = stage_in(chains, d, settings->n_keep_draws, sc, st);
        if (rc) return rc;
        mi::LogitParams q{};
        q.d = (uint32_t)d; q.n_rows = (uint32_t)n; q.NB = (uint32_t)((n + 15) / 16);
        q.C = chains->n_chains; q.chain0 = chains->chain0;
        q.theta = sc.dev.theta; q.draws = sc.dev.draws; q.n_accept = sc.dev.n_accept;
        q.seed = settings->rng_seed_value;
        q.n_burnin = (uint32_t)settings->n_burnin_draws; q.n_keep = (uint32_t)settings->n_keep_draws;
        q.eps = settings->step_size; q.s2 = s2_; q.rs = 1.0 / s2_;
        q.cons_term = -0.5 * (double)d * 1.83787706640934548356;
        q.log_det = log_det_;
        q.draw0 = (uint32_t)chains->draw0;
        MalaDiagMass mdm;
        LdsDenseM ldm;
        if (lds_dense_m_ok(target, settings)) { if ((rc = lds_dense_m(settings, d, ldm, q, mi::LOGIT_MALA))) return rc; }     // dense, no bounds
        else if (settings->precond_mat) { if ((rc = mala_diag_mass_upload(settings, d, mdm, q))) return rc; }     // (diagonal, no bounds: routed above)
        rc = launch_logit(mi::LOGIT_MALA, q, X_dev, y_dev, st, settings, &sc.dev);
        if (rc) return rc;
        rc = fill_n_leap(sc.dev.n_leapfrogs, chains->n_chains, 0, st);
        if (rc) return rc;
        rc = stage_out(chains, d, settings->n_keep_draws, sc, st);
        if (rc) return rc;
        if (Xo.p || mdm.m.p || chains->mem == MI_MEM_HOST) HIP_TRY(hipStreamSynchronize(st));
        return MI_OK;
    }
    if (target->kind != MI_TARGET_GAUSS_ISO && target->kind != MI_TARGET_GAUSS_DIAG && target->kind != MI_TARGET_GAUSS_DENSE)
        return fail(MI_ERR_UNSUPPORTED, "mala: target kind %d not implemented", target->kind);
    if (d > 128 && d <= 512 && target->kind == MI_TARGET_GAUSS_DENSE && !settings->vals_bound
        && (!settings->precond_mat || precond_is_diagonal(settings, d) || lds_dense_m_ok(target, settings)))
        return run_dense_lds("mala", mi::LOGIT_MALA, target, settings, chains, st);     // P streamed through LDS (logistic_lds.hpp); identity, diagonal or (round 5) dense precond_mat
    if (gemm_case(target, settings, chains, false)) return run_gemm("mala", 1, target, settings, chains, st);    // one matrix product per draw (gemm_samplers.hip)
    if (d > 128) return run_literal("mala", 1, target, settings, chains, st);      // no other tiled kernel beyond d = 128: literal.hpp

    DevBuf P_owned;
    const double* P_dev = nullptr;
    rc = dense_precision_on_device(target, P_owned, &P_dev, st);
    if (rc) return rc;
    StagedChains sc;
    rc = stage_in(chains, d, settings->n_keep_draws, sc, st);
    if (rc) return rc;

    mi::MalaParams prm{};
    prm.P = P_dev;
    prm.d = (uint32_t)d;
    prm.C = chains->n_chains;
    prm.chain0 = chains->chain0;
    prm.theta = sc.dev.theta;
    prm.draws = sc.dev.draws;
    prm.n_accept = sc.dev.n_accept;
    prm.seed = settings->rng_seed_value;
    prm.n_burnin = (uint32_t)settings->n_burnin_draws;
    prm.n_keep = (uint32_t)settings->n_keep_draws;
    prm.eps = settings->step_size;
    prm.draw0 = (uint32_t)chains->draw0;
    // Sigma = eps^2 * I (mala.ipp:41,63): INV by Gauss-Jordan gives diag(1/s2); CHOL gives diag(sqrt(s2));
    // LOG_DET = sum_i 2 log L_ii accumulated sequentially, exactly as the oracle states it.
    prm.s2 = settings->step_size * settings->step_size;
    prm.rs = 1.0 / prm.s2;
    prm.cons_term = -0.5 * (double)d * 1.83787706640934548356;   // MCMC_LOG_2PI, stats/mcmc_stats.hpp:28-30
    {
        const double lii = __builtin_sqrt(prm.s2);
        double ld = 0.0;
        for (uint64_t i = 0; i < d; ++i) ld = ld + 2.0 * mi::det_log(lii);
        prm.log_det = ld;
    }

    const int nt = (int)((d + 15) / 16);
    GeneralTables gt;
    rc = general_tables("mala", settings, d, gt, true, true);
    if (rc) return rc;
    // literal replay (literal.hpp): the chains the element-wise kernels flag as non-finite; bounded mala with a dense precond_mat
    // (INV(eps^2 J(theta') M) per draw, mala.ipp:52-53) runs there entirely
    const bool mala_bounded = settings->vals_bound != 0;
    const bool literal_only = gt.active && gt.dense && mala_bounded;
    const bool dense_unbounded = gt.active && gt.dense && !mala_bounded;   // real dense products (sep_target: element-wise where the reference is): nothing to replay
    WsLease lws;
    ReplayWs rp = replay_layout(0, chains->n_chains, (uint32_t)d, 0, mala_bounded);
    mi::lit::LitParams lp{};
    LitDev ldev;
    if (!dense_unbounded) {
        rc = ws_get(st, rp.total_bytes, lws);
        if (rc) return rc;
        rc = replay_bind(rp, lws.p, chains->n_chains, st);
        if (rc) return rc;
        rc = lit_gauss_target(lp.t, target->kind, (uint32_t)d, P_dev, nullptr, rp.tbuf, st);
        if (rc) return rc;
        lit_common(lp, settings, &sc.dev, rp, literal_only);
        mi::lit::LitPrep prep;
        rc = mi::lit::lit_prepare(1, (uint32_t)d, settings->step_size, settings->vals_bound ? 1 : 0, settings->lower_bounds,
                                  settings->upper_bounds, settings->precond_mat, prep);
        if (rc) return rc;
        rc = lit_upload(prep, (uint32_t)d, mala_bounded, ldev, lp);
        if (rc) return rc;
        prm.nf_flag = rp.flag;
    }
    if (literal_only) {
        rc = launched("mala (literal)", mi::launch_literal(1, lp, rp.n_wg, st));
    }
    else if (dense_unbounded) {
        // dense precond_mat, unbounded: Sigma = eps^2 M is constant, so INV / CHOL_LOWER / LOG_DET come from the host once
        std::vector<double> Sigma(d * d), Sinv, Ls;
        for (uint64_t i = 0; i < d * d; ++i) Sigma[i] = prm.s2 * settings->precond_mat[i];
        rc = host_inverse(Sigma.data(), d, Sinv); if (rc) return rc;
        rc = host_cholesky_lower(Sigma.data(), d, Ls); if (rc) return rc;
        double ld = 0.0;
        for (uint64_t i = 0; i < d; ++i) ld = ld + 2.0 * mi::det_log(Ls[i * d + i]);
        prm.log_det = ld;
        DevBuf m_full, sinv_full;
        rc = upload_matrix(std::vector<double>(settings->precond_mat, settings->precond_mat + d * d), d, m_full); if (rc) return rc;
        rc = upload_matrix(Sinv, d, sinv_full); if (rc) return rc;
        prm.Mfull = m_full.as<double>(); prm.Lchol = gt.l_full.as<double>(); prm.Sinv = sinv_full.as<double>();
        prm.sep_target = target->kind != MI_TARGET_GAUSS_DENSE;     // ISO / DIAG: the gradient is element-wise in the reference's target function
        rc = launched("mala", mi::launch_mala_gauss(prm, nt, 2, st));
        if (!rc) HIP_TRY(hipStreamSynchronize(st));     // the matrices are ours
    }
    else if (gt.active) {
        // unbounded runs hoist LOG_DET(eps^2 M) = sum_i 2 log sqrt(eps^2 M_ii), i ascending (bounded runs sum it per draw)
        double ld = 0.0;
        for (uint64_t i = 0; i < d; ++i) ld = ld + 2.0 * mi::det_log(__builtin_sqrt(prm.s2 * gt.m[i]));
        prm.log_det = ld;
        prm.vals_bound = settings->vals_bound ? 1 : 0;
        prm.btype = gt.bt.as<int>(); prm.lb = gt.lb.as<double>(); prm.ub = gt.ub.as<double>();
        prm.m = gt.m_dev.as<double>(); prm.m_sqrt = gt.ms_dev.as<double>();
        rc = launched("mala", mi::launch_mala_gauss(prm, nt, 1, st));
        if (!rc) rc = launched("mala (literal replay)", mi::launch_literal(1, lp, rp.n_wg, st));
    }
    else {
        rc = launched("mala", mi::launch_mala_gauss(prm, nt, 0, st));
        if (!rc) rc = launched("mala (literal replay)", mi::launch_literal(1, lp, rp.n_wg, st));
    }
    if (rc) return rc;
    if (gt.active || ldev.any) HIP_TRY(hipStreamSynchronize(st));     // the tables are ours
    rc = fill_n_leap(sc.dev.n_leapfrogs, chains->n_chains, 0, st);
    if (rc) return rc;

    rc = stage_out(chains, d, settings->n_keep_draws, sc, st);
    if (rc) return rc;
    if (P_owned.p || chains->mem == MI_MEM_HOST) HIP_TRY(hipStreamSynchronize(st));
    return MI_OK;
}

// mcmc::rwmh (src/rwmh.cpp:30-175) for many chains.  settings->step_size carries rwmh_settings.par_scale and
// settings->precond_mat carries rwmh_settings.cov_mat (identity when NULL, rwmh.cpp:58).
int mi_mcmc_rwmh_run(const mi_target* target, const mi_settings* settings, mi_chains* chains, void* stream)
{
    int rc = check_common(target, settings, chains);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint64_t d = target->d;
    if (target->kind == MI_TARGET_NORMAL_MODEL) return run_small_normal_model("rwmh", 3, target, settings, chains, st);
    if (target->kind == MI_TARGET_LOGISTIC && gemm_case(target, settings, chains, false, false)) return run_gemm("rwmh", 3, target, settings, chains, st);    // d > 512, plain
    if (target->kind == MI_TARGET_LOGISTIC) {
        if (settings->vals_bound || settings->precond_mat)
            return d <= (uint64_t)mi::SMALL_MAX_D ? run_small_logistic("rwmh", 3, target, settings, chains, st) : run_literal("rwmh", 3, target, settings, chains, st);
        return d <= 512 ? run_logit_plain("rwmh", mi::LOGIT_RWMH, target, settings, chains, st) : run_literal("rwmh", 3, target, settings, chains, st);
    }
    if (target->kind != MI_TARGET_GAUSS_ISO && target->kind != MI_TARGET_GAUSS_DIAG && target->kind != MI_TARGET_GAUSS_DENSE)
        return fail(MI_ERR_UNSUPPORTED, "rwmh: target kind %d not implemented", target->kind);
    if (d > 128 && d <= 512 && target->kind == MI_TARGET_GAUSS_DENSE && !settings->vals_bound && !settings->precond_mat)
        return run_dense_lds("rwmh", mi::LOGIT_RWMH, target, settings, chains, st);     // P streamed through LDS (logistic_lds.hpp)
    if (gemm_case(target, settings, chains, false, false)) return run_gemm("rwmh", 3, target, settings, chains, st);
    if (d > 128) return run_literal("rwmh", 3, target, settings, chains, st);      // no other tiled kernel beyond d = 128: literal.hpp

    DevBuf P_owned;
    const double* P_dev = nullptr;
    rc = dense_precision_on_device(target, P_owned, &P_dev, st);
    if (rc) return rc;
    StagedChains sc;
    rc = stage_in(chains, d, settings->n_keep_draws, sc, st);
    if (rc) return rc;

    mi::RwmhParams prm{};
    prm.P = P_dev; prm.d = (uint32_t)d; prm.C = chains->n_chains; prm.chain0 = chains->chain0;
    prm.theta = sc.dev.theta; prm.draws = sc.dev.draws; prm.n_accept = sc.dev.n_accept;
    prm.seed = settings->rng_seed_value;
    prm.n_burnin = (uint32_t)settings->n_burnin_draws; prm.n_keep = (uint32_t)settings->n_keep_draws;
    prm.draw0 = (uint32_t)chains->draw0;
    prm.par_scale = settings->step_size;

    const int nt = (int)((d + 15) / 16);
    GeneralTables gt;
    rc = general_tables("rwmh", settings, d, gt, true, true);
    if (rc) return rc;
    DevBuf c_dev, lc_dev;
    if (gt.active) {
        // cov_mcmc_chol = par_scale * CHOL_LOWER(cov_mat), element by element (rwmh.cpp:119)
        std::vector<double> c(d);
        for (uint64_t i = 0; i < d; ++i) c[i] = prm.par_scale * gt.m_sqrt[i];
        HIP_TRY(c_dev.alloc(d * 8));
        HIP_TRY(hipMemcpy(c_dev.p, c.data(), d * 8, hipMemcpyHostToDevice));
        prm.vals_bound = settings->vals_bound ? 1 : 0;
        prm.btype = gt.bt.as<int>(); prm.lb = gt.lb.as<double>(); prm.ub = gt.ub.as<double>();
        prm.c_diag = c_dev.as<double>();
        if (gt.dense) {
            std::vector<double> L;
            rc = host_cholesky_lower(settings->precond_mat, d, L); if (rc) return rc;
            for (auto& v : L) v = prm.par_scale * v;
            rc = upload_matrix(L, d, lc_dev); if (rc) return rc;
            prm.Lc = lc_dev.as<double>();
        }
        rc = launched("rwmh", mi::launch_rwmh_gauss(prm, nt, true, gt.dense, st));
        if (!rc) HIP_TRY(hipStreamSynchronize(st));     // the tables are ours
    }
    else rc = launched("rwmh", mi::launch_rwmh_gauss(prm, nt, false, false, st));
    if (rc) return rc;
    rc = fill_n_leap(sc.dev.n_leapfrogs, chains->n_chains, 0, st);
    if (rc) return rc;

    rc = stage_out(chains, d, settings->n_keep_draws, sc, st);
    if (rc) return rc;
    if (P_owned.p || chains->mem == MI_MEM_HOST) HIP_TRY(hipStreamSynchronize(st));
    return MI_OK;
}

// nuts on the LDS-streamed evaluation (nuts_lds.hpp): the logistic target with 8 < d <= 512 and dense Gaussians with 128 < d <= 512,
// identity precond_mat, no bounds, 1 <= max_tree_depth <= 10.  Chains that reach the non-finite regime are flagged by the kernel and
// replayed by literal_kernel<2> right behind it.  lds_target: mi::LOGIT_TARGET_LOGISTIC / mi::LOGIT_TARGET_DENSE.
int run_lds_nuts(const mi_target* target, const mi_settings* settings, mi_chains* chains, hipStream_t st, int lds_target)
{
    int rc;
    const uint64_t d = target->d, C = chains->n_chains;
    const uint64_t n_total = settings->n_burnin_draws + settings->n_keep_draws;
    if (n_total > 0xffffffffULL) return fail(MI_ERR_BAD_ARG, "too many draws");
    const bool dense = lds_target == mi::LOGIT_TARGET_DENSE;
    DevBuf Xo, yo, P_owned;
    const double *X_dev = nullptr, *y_dev = nullptr;
    uint64_t n_rows = d;
    if (dense) {
        if (!target->prec) return fail(MI_ERR_BAD_ARG, "GAUSS_DENSE needs prec (d*d)");
        rc = dense_precision_on_device(target, P_owned, &X_dev, st);
        if (rc) return rc;
    } else {
        if (!target->X || !target->y || target->n_rows == 0) return fail(MI_ERR_BAD_ARG, "LOGISTIC needs X, y, n_rows");
        n_rows = target->n_rows;
        X_dev = target->X; y_dev = target->y;
        if (target->mem == MI_MEM_HOST) {
            HIP_TRY(Xo.alloc(n_rows * d * sizeof(double))); HIP_TRY(yo.alloc(n_rows * sizeof(double)));
            HIP_TRY(hipMemcpy(Xo.p, target->X, n_rows * d * sizeof(double), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(yo.p, target->y, n_rows * sizeof(double), hipMemcpyHostToDevice));
            X_dev = Xo.as<double>(); y_dev = yo.as<double>();
        }
    }
    StagedChains sc;
    rc = stage_in(chains, d, settings->n_keep_draws, sc, st, n_total);
    if (rc) return rc;
    mi::LogitParams q{};
    q.d = (uint32_t)d; q.n_rows = (uint32_t)n_rows; q.NB = (uint32_t)((n_rows + 15) / 16);
    q.C = C; q.chain0 = chains->chain0;
    q.theta = sc.dev.theta; q.draws = sc.dev.draws; q.n_accept = sc.dev.n_accept;
    q.seed = settings->rng_seed_value;
    q.n_burnin = (uint32_t)settings->n_burnin_draws; q.n_keep = (uint32_t)settings->n_keep_draws;
    q.draw0 = (uint32_t)chains->draw0;
    q.n_leap_out = sc.dev.n_leapfrogs; q.step_out = sc.dev.step_size; q.depth_trace = sc.dev.nuts_depth;
    q.n_exec_out = sc.dev.n_leapfrogs_executed; sc.exec_written = q.n_exec_out != nullptr;      // (every doubling on a memoised trajectory: nuts_lds.hpp)
    if ((rc = nuts_continuation(settings, chains, &q.n_adapt))) return rc;
    q.adapt_state = sc.dev.nuts_adapt_state;
    q.max_depth = (uint32_t)settings->max_tree_depth;
    q.delta = settings->target_accept_rate; q.eps_bar0 = settings->step_size;
    q.gamma = settings->gamma_val; q.t0 = settings->t0_val; q.kappa = settings->kappa_val;
    // a DENSE precond_mat without bounds (round 6): INV(M) and CHOL_LOWER(M) streamed through LDS like the target's matrix (nuts_lds.hpp: DENSEM)
    const bool dense_m = lds_dense_m_ok(target, settings);
    LdsTables lt;
    LdsDenseM ldm;
    if (dense_m) { if ((rc = lds_dense_m(settings, d, ldm, q, mi::LOGIT_HMC))) return rc; }
    else if ((rc = lds_tables("nuts", settings, d, lt, q))) return rc;      // (the caller routed bounds / a DIAGONAL matrix here)

    // workspace: the kernel's own | non-finite flags | the matrix transposed and the work areas of the literal replay
    ReplayWs rp;
    rp.t_doubles = ((size_t)d * std::max<size_t>(d, n_rows) + 31) & ~(size_t)31;
    // (the kernel's workspace is sized by the chain SLOTS of its persistent grid, not by the chains)
    const uint64_t n_slots = 32 * (dense_m ? mi::logit_lds_nuts_dense_m_workgroups(q.d, C, lds_target) : mi::logit_lds_nuts_workgroups(q.d, C, lds_target));
    rp.own_bytes = (mi::logit_lds_workspace_bytes(q.d, q.NB, n_slots, lds_target, mi::LOGIT_NUTS) + 255) & ~(size_t)255;
    rp.stride = mi::lit::lit_work_doubles((uint32_t)d, dense ? 0u : (uint32_t)n_rows, false, q.max_depth, true, false);
    rp.n_wg = (unsigned)std::min<uint64_t>(C, 512u);
    const size_t flag_bytes = ((C + 1) * sizeof(uint32_t) + 255) & ~(size_t)255;
    rp.total_bytes = rp.own_bytes + flag_bytes + (rp.t_doubles + (size_t)rp.n_wg * rp.stride) * sizeof(double);
    // more chains than chain slots: the launcher cuts the runs into pieces (lds_nuts_pieces.hpp) and needs room for the queues
    const size_t rp_bytes = (rp.total_bytes + 255) & ~(size_t)255;
    const size_t split_bytes = (C > n_slots) ? mi::logit_lds_nuts_split_bytes(C, q.d) : 0;
    WsLease base;
    rc = ws_get(st, rp_bytes + split_bytes, base);
    if (rc) return rc;
    rc = replay_bind(rp, base.p, C, st);
    if (rc) return rc;
    q.nf_flag = rp.flag;
    if (split_bytes) q.split_ws = static_cast<char*>(base.p) + rp_bytes;
    DevBuf mws;                                          // dense_m: the block images of the two matrices (+ the exchange vectors of their products)
    if (dense_m) HIP_TRY(mws.alloc(mi::logit_lds_nuts_dense_m_bytes(q.d, C, lds_target)));
    const int e = dense_m ? mi::logit_lds_launch_nuts_dense_m(q, X_dev, y_dev, base.p, mws.p, st, lds_target)
                          : mi::logit_lds_launch(mi::LOGIT_NUTS, q, X_dev, y_dev, base.p, st, lds_target);
    if (e != 0) return fail(MI_ERR_HIP, "LDS-streamed nuts kernel launch: %s", hipGetErrorString((hipError_t)e));
    const std::string lds_name = mi::host::last_kernel();
    {
        mi::lit::LitParams lp{};
        rc = transpose_on_device(X_dev, rp.tbuf, (uint32_t)n_rows, (uint32_t)d, st);
        if (rc) return rc;
        if (dense) { lp.t.kind = mi::lit::LIT_DENSE; lp.t.d = (uint32_t)d; lp.t.prec = rp.tbuf; }
        else {
            lp.t.kind = mi::lit::LIT_LOGISTIC; lp.t.d = (uint32_t)d; lp.t.n_rows = (uint32_t)n_rows; lp.t.X = X_dev; lp.t.y = y_dev;
            lp.t.Xt = rp.tbuf;
        }
        mi::lit::lit_orders(lp.t);
        lit_common(lp, settings, &sc.dev, rp, false);
        lp.n_adapt = q.n_adapt; lp.max_depth = q.max_depth;
        lp.delta = q.delta; lp.gamma = q.gamma; lp.t0 = q.t0; lp.kappa = q.kappa;
        lp.step_out = sc.dev.step_size; lp.depth_trace = sc.dev.nuts_depth; lp.adapt_state = sc.dev.nuts_adapt_state;
        LitDev ldev;
        if (dense_m) {                                   // the replay's own copies of M, INV(M), CHOL_LOWER(M) (transposed: literal_host.hpp)
            mi::lit::LitPrep prep;
            rc = mi::lit::lit_prepare(0, (uint32_t)d, settings->step_size, 0, nullptr, nullptr, settings->precond_mat, prep);
            if (rc) return rc;
            rc = lit_upload(prep, (uint32_t)d, false, ldev, lp);
            if (rc) return rc;
        }
        else lds_tables_replay(lt, q, lp);
        rc = launched("LDS-streamed nuts kernel (literal replay)", mi::launch_literal(2, lp, rp.n_wg, st));
        if (!rc && q.n_exec_out) {                       // a replayed chain executed every leapfrog it counts
            hipLaunchKernelGGL(copy_flagged_counts_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, st, rp.flag, q.n_leap_out, q.n_exec_out, C);
            HIP_TRY(hipGetLastError());
        }
        if (!rc && dense_m) HIP_TRY(hipStreamSynchronize(st));     // (the replay's matrices are ours)

        if (rc) return rc;
        mi::host::last_kernel() = lds_name;
    }
    rc = stage_out(chains, d, settings->n_keep_draws, sc, st, n_total);
    if (rc) return rc;
    if (Xo.p || P_owned.p || lt.ms.p || dense_m || chains->mem == MI_MEM_HOST) HIP_TRY(hipStreamSynchronize(st));
    return MI_OK;
}
// the cases nuts_lds.hpp covers (everything else on these targets: literal.hpp): no bounds / bounds with the identity or a diagonal precond_mat, and --
// round 6 -- a DENSE precond_mat without bounds
bool lds_nuts_case(const mi_target* target, const mi_settings* settings)
{
    return (lds_general_ok(target, settings) || lds_dense_m_ok(target, settings)) && settings->max_tree_depth >= 1 && settings->max_tree_depth <= 10;
}

// The plain case (unbounded; identity or a diagonal precond_mat) runs on nuts_memo.hpp.  The register-carried kernels of rounds 2-4 (nuts_reg.hpp,
// nuts_dyn.hpp, nuts_split.hpp) are retired: the memoised kernel is faster wherever a leapfrog costs anything (d = 128: 563 ms against 699 at
// 65 536 chains, 135 / 156 at 8 192; d = 64: 209 / 208; d = 32: 71 / 72) and within 13 % at d = 16 (37 / 33 ms) -- one tick to maintain instead of
// four.  Their hints stay valid and are ignored.  MI_KERNEL_NUTS_TICK_LOCAL keeps the independent tick-local kernel for A/B runs.

int mi_mcmc_nuts_run(const mi_target* target, const mi_settings* settings, mi_chains* chains, void* stream)
{
    int rc = check_common(target, settings, chains);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint64_t d = target->d;
    if (target->kind == MI_TARGET_NORMAL_MODEL) return run_small_normal_model("nuts", 2, target, settings, chains, st);
    if (target->kind == MI_TARGET_LOGISTIC) {
        if (d <= (uint64_t)mi::SMALL_MAX_D && settings->max_tree_depth <= 10) return run_small_logistic("nuts", 2, target, settings, chains, st);
        if (d <= 512 && lds_nuts_case(target, settings)) return run_lds_nuts(target, settings, chains, st, mi::LOGIT_TARGET_LOGISTIC);
        return run_literal("nuts", 2, target, settings, chains, st);
    }
    if (target->kind != MI_TARGET_GAUSS_ISO && target->kind != MI_TARGET_GAUSS_DIAG && target->kind != MI_TARGET_GAUSS_DENSE)
        return fail(MI_ERR_UNSUPPORTED, "nuts: target kind %d not implemented", target->kind);
    // the tiled kernels: d <= 128, max_tree_depth <= 10 (per-level records and scalars are sized for that); beyond, literal.hpp
    if (target->kind == MI_TARGET_GAUSS_DENSE && d > 128 && d <= 512 && lds_nuts_case(target, settings))
        return run_lds_nuts(target, settings, chains, st, mi::LOGIT_TARGET_DENSE);     // P streamed through LDS (nuts_lds.hpp)
    if (d > 128 || settings->max_tree_depth > (uint64_t)mi::NUTS_MAX_DEPTH) return run_literal("nuts", 2, target, settings, chains, st);
    const uint64_t n_total = settings->n_burnin_draws + settings->n_keep_draws;
    if (n_total > 0xffffffffULL) return fail(MI_ERR_BAD_ARG, "too many draws");

    DevBuf P_owned;
    const double* P_dev = nullptr;
    rc = dense_precision_on_device(target, P_owned, &P_dev, st);
    if (rc) return rc;
    StagedChains sc;
    rc = stage_in(chains, d, settings->n_keep_draws, sc, st, n_total);
    if (rc) return rc;

    mi::NutsParams prm{};
    prm.P = P_dev;
    prm.d = (uint32_t)d;
    prm.sep_target = target->kind != MI_TARGET_GAUSS_DENSE;     // ISO / DIAG: the general variants (and the replay of flagged chains, which is one) take the gradient element-wise
    prm.C = chains->n_chains;
    prm.chain0 = chains->chain0;
    prm.theta = sc.dev.theta;
    const int nt = (int)((d + 15) / 16);
    GeneralTables gt;
    rc = general_tables("nuts", settings, d, gt, true, true);
    if (rc) return rc;
    const bool lockstep = target->kernel_hint == MI_KERNEL_NUTS_LOCKSTEP && chains->draw0 == 0 && !chains->nuts_adapt_state;   // the first-generation kernel, same bits (it exports no adaptation state)
    const bool tick_local = target->kernel_hint == MI_KERNEL_NUTS_TICK_LOCAL;  // the asynchronous kernel without register-carried state
    // the plain case and a diagonal precond_mat alone can run on the memoised trajectory (nuts_memo.hpp)
    const bool memo = (!gt.active || (!gt.dense && !settings->vals_bound)) && !lockstep && !tick_local;
    WsLease ws;
    const size_t d_pad = (d <= 16) ? 16 : (d <= 32) ? 32 : (d <= 64) ? 64 : 128;
    size_t ws_own = (size_t)mi::NUTS_NVEC_ASYNC * d_pad * ((chains->n_chains + 15) / 16 + 4) * 16 * sizeof(double);
    // (the memoised kernel's workspace is sized by the chain slots of its persistent grid; the replay of flagged chains re-uses the same
    //  bytes in the asynchronous kernel's layout afterwards)
    if (memo) ws_own = std::max(ws_own, mi::nuts_memo_workspace_bytes(chains->n_chains, nt, gt.active));
    // bounds (with the identity or a diagonal precond_mat): the memoised tick with the tile route's policy (nuts_bounded_launch.hip)
    const bool bounded_memo = gt.active && !gt.dense && settings->vals_bound && !lockstep && !tick_local;
    if (bounded_memo) ws_own = std::max(ws_own, mi::nuts_bounded_workspace_bytes(chains->n_chains, nt));
    const size_t ws_own_r = (ws_own + 255) & ~(size_t)255;
    const size_t flag_bytes = ((chains->n_chains + 1) * sizeof(uint32_t) + 255) & ~(size_t)255;   // [C] flags, [C] "any"
    const size_t fixed_bytes = ws_own_r + flag_bytes + 5 * 128 * sizeof(double) + 256 + 256;     // + non-finite flags + identity tables of the replay + the chain counter
    // The memoised kernel reads the momenta of the whole run from a table its launcher fills first (nuts_memo.hpp: nuts_momenta_kernel) when that table
    // is affordable: at most MI_NUTS_MOMENTA_MAX_BYTES and a third of the device memory that is free right now; else -- and under
    // MI_KERNEL_NUTS_MEMO_INTICK -- the momenta are generated inside the tick, as in rounds 2-5.  Same bits either way.
    size_t mom_bytes = 0;
    if (memo && !gt.active && target->kernel_hint != MI_KERNEL_NUTS_MEMO_INTICK) {          // (the plain case; with a diagonal precond_mat: in the tick, nuts_launch.hip)
        const size_t want = mi::nuts_memo_momenta_bytes(chains->n_chains, (uint32_t)n_total, nt);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
        const size_t cached = ws_cached_bytes(st);       // (what this stream's workspace already holds counts as free: it is re-used)
        if (want <= MI_NUTS_MOMENTA_MAX_BYTES && want <= (free_b + cached) / 3) mom_bytes = (want + 255) & ~(size_t)255;
    }
    const size_t split_bytes = (memo || bounded_memo) ? ((mi::nuts_split_workspace_bytes(chains->n_chains, (uint32_t)d) + 255) & ~(size_t)255) : 0;      // (nuts_launch.hip: runs cut into pieces)
    rc = ws_get(st, fixed_bytes + mom_bytes + split_bytes, ws);
    if (rc) return rc;     // every workspace vector is stored by the kernel before it is loaded: no memset needed
    prm.ws = ws.as<double>();
    if (split_bytes) prm.split_ws = static_cast<char*>(ws.p) + fixed_bytes + mom_bytes;
    if (mom_bytes) {
        prm.mom = reinterpret_cast<double*>(static_cast<char*>(ws.p) + fixed_bytes);
        prm.msc = prm.mom + (size_t)n_total * chains->n_chains * (size_t)(16 * (nt <= 1 ? 1 : nt == 2 ? 2 : nt <= 4 ? 4 : 8));
    }
    uint32_t* const nf_flag = reinterpret_cast<uint32_t*>(static_cast<char*>(ws.p) + ws_own_r);
    double* const id_tab = reinterpret_cast<double*>(static_cast<char*>(ws.p) + ws_own_r + flag_bytes);
    prm.next_chain = reinterpret_cast<uint32_t*>(static_cast<char*>(ws.p) + ((ws_own_r + flag_bytes + 5 * 128 * sizeof(double) + 255) & ~(size_t)255));
    prm.draws = sc.dev.draws;
    prm.n_accept = sc.dev.n_accept;
    prm.n_leap = sc.dev.n_leapfrogs;
    prm.step_out = sc.dev.step_size;
    prm.depth_trace = sc.dev.nuts_depth;
#ifdef MI_PROFILING
    DevBuf prof_buf;
    if (getenv("MI_NUTS_PROF")) { HIP_TRY(prof_buf.alloc(96 * 8)); HIP_TRY(hipMemset(prof_buf.p, 0, 96 * 8)); prm.prof = prof_buf.as<unsigned long long>();
}
#endif
    prm.seed = settings->rng_seed_value;
    prm.n_burnin = (uint32_t)settings->n_burnin_draws;
    prm.n_keep = (uint32_t)settings->n_keep_draws;
    prm.draw0 = (uint32_t)chains->draw0;
    // continuation (checkpoint / resume): the step sizes come back in; inside the adaptation window also the dual-averaging state.
    // (The lock-step hint cannot be honoured for a continuation -- that kernel keeps no per-chain step size: as mi_mcmc.h promises for a
    //  hint the request cannot take, it is ignored and the default kernel runs; it also clamps the window to the call, nuts.cpp:54, which is
    //  the same thing for the single call it serves.)
    if ((rc = nuts_continuation(settings, chains, &prm.n_adapt))) return rc;
    prm.adapt_state = sc.dev.nuts_adapt_state;
    prm.max_depth = (uint32_t)settings->max_tree_depth;
    prm.delta = settings->target_accept_rate;
    prm.eps_bar0 = settings->step_size;
    prm.gamma = settings->gamma_val;
    prm.t0 = settings->t0_val;
    prm.kappa = settings->kappa_val;

    uint32_t nuts_batch = 8;                 // momentum-refresh batch of the asynchronous kernel
#ifdef MI_PROFILING
    if (const char* e = getenv("MI_NUTS_BATCH")) nuts_batch = (uint32_t)atoi(e);
#endif
    if (gt.active && gt.dense) {
        prm.btype = gt.bt.as<int>(); prm.lb = gt.lb.as<double>(); prm.ub = gt.ub.as<double>();
        prm.m_sqrt = gt.ms_dev.as<double>(); prm.m_inv = gt.mi_dev.as<double>();
        prm.Minv = gt.minv_full.as<double>(); prm.Lchol = gt.l_full.as<double>();
        prm.vals_bound = settings->vals_bound ? 1 : 0;
        rc = launched("nuts", mi::launch_nuts_gauss(prm, nt, true, true, false, nuts_batch, st));
        if (!rc) HIP_TRY(hipStreamSynchronize(st));     // the tables are ours
    }
    else if (gt.active && memo) {
        // a diagonal precond_mat alone: the plain-case kernel with two mass tables (nuts_memo.hpp, DIAGM); flagged chains are replayed
        // by the general variant with the same tables
        HIP_TRY(hipMemsetAsync(nf_flag, 0, (chains->n_chains + 1) * sizeof(uint32_t), st));
        prm.nf_flag = nf_flag;
        prm.m_sqrt = gt.ms_dev.as<double>(); prm.m_inv = gt.mi_dev.as<double>();
        prm.n_exec = sc.dev.n_leapfrogs_executed; sc.exec_written = prm.n_exec != nullptr;
        rc = launched("nuts", mi::launch_nuts_gauss_memo(prm, nt, st, true));
        if (rc) return rc;
        const std::string reg_name = mi::host::last_kernel();
        mi::NutsParams rp = prm;
        rp.n_exec = nullptr;
        rp.nf_flag = nullptr; rp.replay_flag = nf_flag;
        rp.btype = gt.bt.as<int>(); rp.lb = gt.lb.as<double>(); rp.ub = gt.ub.as<double>();
        rp.vals_bound = 0;
        rc = launched("nuts (replay)", mi::launch_nuts_gauss(rp, nt, true, false, false, nuts_batch, st));
        mi::host::last_kernel() = reg_name;
        if (!rc && prm.n_exec) {
            hipLaunchKernelGGL(copy_flagged_counts_kernel, dim3((unsigned)((chains->n_chains + 255) / 256)), dim3(256), 0, st, nf_flag, prm.n_leap, prm.n_exec, chains->n_chains);
            HIP_TRY(hipGetLastError());
        }
        if (!rc) HIP_TRY(hipStreamSynchronize(st));     // the tables are ours
    }
    else if (gt.active) {
        prm.btype = gt.bt.as<int>(); prm.lb = gt.lb.as<double>(); prm.ub = gt.ub.as<double>();
        prm.m_sqrt = gt.ms_dev.as<double>(); prm.m_inv = gt.mi_dev.as<double>();
        prm.vals_bound = settings->vals_bound ? 1 : 0;
        if (bounded_memo) {
            prm.n_exec = sc.dev.n_leapfrogs_executed; sc.exec_written = prm.n_exec != nullptr;
            rc = launched("nuts", mi::launch_nuts_gauss_bounded(prm, nt, st));
        }
        else rc = launched("nuts", mi::launch_nuts_gauss(prm, nt, true, false, false, nuts_batch, st));     // (MI_KERNEL_NUTS_TICK_LOCAL: the tick-local general kernel)
        if (!rc) HIP_TRY(hipStreamSynchronize(st));     // the tables are ours
    }
    else if (lockstep || tick_local) rc = launched("nuts", mi::launch_nuts_gauss(prm, nt, false, false, lockstep, nuts_batch, st));
    else {
        // the plain case: nuts_gauss_memo_kernel; chains that reach the non-finite regime are flagged there and
        // replayed by the general variant, which reproduces the reference's dense products, with identity tables
        HIP_TRY(hipMemsetAsync(nf_flag, 0, (chains->n_chains + 1) * sizeof(uint32_t), st));
        prm.nf_flag = nf_flag;
        const uint64_t C_ = chains->n_chains;
        prm.n_exec = sc.dev.n_leapfrogs_executed; sc.exec_written = prm.n_exec != nullptr;
        rc = launched("nuts", mi::launch_nuts_gauss_memo(prm, nt, st));
        if (rc) return rc;
        const std::string reg_name = mi::host::last_kernel();
        int* bt_i = reinterpret_cast<int*>(id_tab);
        hipLaunchKernelGGL(fill_identity_tables_kernel, dim3(1), dim3(128), 0, st, bt_i, id_tab + 128, id_tab + 256, id_tab + 384, id_tab + 512, 128u);
        HIP_TRY(hipGetLastError());
        mi::NutsParams rp = prm;
        rp.n_exec = nullptr;
        rp.nf_flag = nullptr; rp.replay_flag = nf_flag;
        rp.btype = bt_i; rp.lb = id_tab + 128; rp.ub = id_tab + 256; rp.m_sqrt = id_tab + 384; rp.m_inv = id_tab + 512;
        rp.vals_bound = 0;
        rc = launched("nuts (replay)", mi::launch_nuts_gauss(rp, nt, true, false, false, nuts_batch, st));
        mi::host::last_kernel() = reg_name;
        if (!rc && prm.n_exec) {
            hipLaunchKernelGGL(copy_flagged_counts_kernel, dim3((unsigned)((C_ + 255) / 256)), dim3(256), 0, st, nf_flag, prm.n_leap, prm.n_exec, C_);
            HIP_TRY(hipGetLastError());
        }
    }
    if (rc) return rc;

    rc = stage_out(chains, d, settings->n_keep_draws, sc, st, n_total);
    if (rc) return rc;
#ifdef MI_PROFILING
    if (prm.prof) {
        unsigned long long h[16];
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(h, prm.prof, sizeof(h), hipMemcpyDeviceToHost));
        // the marks of nuts_memo_core.hpp (MI_MPROF(k) closes phase k); the tick-local kernel (nuts_async.hpp) fills the first eight with its own phases
        const char* names[12] = {"A phase: rows, momenta ahead", "B origin + kick + drift", "mat-vec + kick + energies", "tests (U-turn, memoised)", "C walk",
                                 "init / search", "E record store", "loop head", "point scalars (n', s', alpha)", "D end of doubling", "-", "-"};
        unsigned long long tot = 0;
        for (int k = 0; k < 10; ++k) tot += h[k];            // (slot 10 of the memoised tick is a count: chain-ticks that compute a point)
        for (int k = 0; k < 10; ++k) if (h[k]) fprintf(stderr, "[nuts prof] %-32s %12llu cycles %5.1f%%\n", names[k], h[k], 100.0 * h[k] / (tot ? tot : 1));
        fprintf(stderr, "[nuts prof] ticks %llu (%.0f cycles each), active chain-ticks %llu (%.2f of 16 per tick), of them computing a point %llu (%.2f per tick), walk iterations %llu, test iterations %llu\n", h[12],
                (double)tot / (h[12] ? h[12] : 1), h[13], (double)h[13] / (h[12] ? h[12] : 1), h[10], (double)h[10] / (h[12] ? h[12] : 1), h[14], h[15]);
    }
#endif
    if (P_owned.p || chains->mem == MI_MEM_HOST) HIP_TRY(hipStreamSynchronize(st));
    return MI_OK;
}

// mcmc::rmhmc (src/rmhmc.cpp:30-287) for many chains of a small-dimensional target with a built-in metric tensor.
int mi_mcmc_rmhmc_run(const mi_target* target, const mi_settings* settings, mi_chains* chains, void* stream)
{
    int rc = check_common(target, settings, chains);
    if (rc) return rc;
    if (target->kind == MI_TARGET_LOGISTIC) {           // Fisher information + prior precision (small_targets.hpp), d <= 4
        if (target->d > 4)                                // beyond the one-lane engine's d x d x d cubes per lane: one workgroup per chain (literal.hpp)
            return run_literal("rmhmc", 4, target, settings, chains, static_cast<hipStream_t>(stream));
        return run_small_logistic("rmhmc", 4, target, settings, chains, static_cast<hipStream_t>(stream));
    }
    // the Gaussian kinds: their (constant) precision is the metric, its derivative zero -- the oracle's orc_target_tensor; literal.hpp
    if (target->kind == MI_TARGET_GAUSS_ISO || target->kind == MI_TARGET_GAUSS_DIAG || target->kind == MI_TARGET_GAUSS_DENSE)
        return run_literal("rmhmc", 4, target, settings, chains, static_cast<hipStream_t>(stream));
    if (target->kind != MI_TARGET_NORMAL_MODEL)
        return fail(MI_ERR_UNSUPPORTED, "rmhmc: target kind %d has no built-in metric tensor on the device path (user targets: include/mi_mcmc_target.hpp)", target->kind);
    return run_small_normal_model("rmhmc", 4, target, settings, chains, static_cast<hipStream_t>(stream));
}

}  // extern "C"
