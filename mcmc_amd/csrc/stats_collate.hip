// stats_collate.hip -- what sits on either side of the sampling path in the C ABI: device reducers over draws_out slabs
// (mi_mcmc_draw_stats, draw_stats.hpp), the layout converter, and the multi-GPU helpers (shard arithmetic, RCCL all-gather of
// the kept draws, merge kernel).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

#include "host_common.hpp"
#include "draw_stats.hpp"

using mi::host::fail;
using mi::host::DevBuf;

extern "C" {

int mi_mcmc_draw_stats(const double* draws_kdc, int32_t mem, uint64_t n_keep, uint64_t d, uint64_t n_chains,
                       double* mean, double* acov, double* rhat, double* ess, void* stream)
{
    if (!draws_kdc || n_keep == 0 || d == 0 || n_chains == 0) return fail(MI_ERR_BAD_ARG, "draw_stats: empty input");
    (void)hipGetLastError();
    if (n_keep > 0xffffffffULL) return fail(MI_ERR_BAD_ARG, "draw_stats: n_keep does not fit 32 bits");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t n = n_keep, C = n_chains;
    DevBuf staged;
    const double* x = draws_kdc;
    if (mem == MI_MEM_HOST) {
        HIP_TRY(staged.alloc(n * d * C * 8));
        HIP_TRY(hipMemcpyAsync(staged.p, draws_kdc, n * d * C * 8, hipMemcpyHostToDevice, st));
        x = staged.as<double>();
    }
    // chain groups (partials are added in order on the host).  The one-pass Gram kernel (n <= 128) wants few, long workgroups -- its
    // prologue (first tile) and epilogue (diagonal sums) are per workgroup, and every group is 16 NB x d doubles to fetch --: about 2 048
    // workgroups in all; the streamed kernels want many short ones.
    const bool gram = n <= (size_t)mi::STATS_GRAM_MAX_N;
    const uint32_t G = gram ? (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(64, (2048 + d - 1) / d), (C + 63) / 64))
                            : (uint32_t)std::min<uint64_t>(64, (C + 63) / 64);
    const bool tiled = n > (size_t)mi::STATS_MAX_N;                          // long series: lags below STATS_TILED_LAGS, streamed kernel
    const size_t nlag_max = tiled ? (size_t)mi::STATS_TILED_LAGS : n;
    const size_t pass_lags = (size_t)mi::STATS_PASS_BLOCKS * mi::STATS_LB;
    // one device buffer for everything: [G d] sums, [d] means, [d] dimension list, [G d 3] moments, [G d max(pass, tiled) lags]
    const size_t part_lags = tiled ? nlag_max : std::max<size_t>(std::max<size_t>(pass_lags, 16), (n <= (size_t)mi::STATS_GRAM_MAX_N) ? ((n + 15) / 16) * 32 : 0);
    DevBuf buf;
    const size_t o_sum = 0, o_mean = o_sum + (size_t)G * d, o_dims = o_mean + d, o_mom = o_dims + d, o_part = o_mom + (size_t)G * d * 3;
    HIP_TRY(buf.alloc((o_part + (size_t)G * d * part_lags) * 8));
    double* const B = buf.as<double>();
    // ONE pass over the slab when the Gram kernel serves the request (n <= 128 and more than the mean is asked for): the series are
    // centred by a PROVISIONAL centre a_j -- the mean over the chains of the FIRST, MIDDLE and LAST kept draw, 3 / n of the slab (the first
    // draw alone can sit far from the pooled mean when n_burnin is 0 and the chains share a start far from the mode: with e / sigma large
    // the recentring below would cancel (e / sigma)^2 ulp of the autocovariances) -- and the kernel's row
    // sums R_t = sum_c (x_t - a) move the centre to the pooled mean afterwards (exactly, in the algebra):  sum_c sum_t (v_t - e)(v_{t+k} - e) = S_k - e (sum_{t < n-k} R_t + sum_{t >= k} R_t) + C (n - k) e^2,
    // e = pooled mean - a.  (Otherwise: the pooled mean first, one read of the slab, then the lag kernels.)
    const bool one_pass = gram && (acov || rhat || ess);
    const uint32_t t_step = one_pass ? (uint32_t)std::max<size_t>(1, (n - 1) / 2) : 1u;      // rows 0, (n-1)/2, 2 ((n-1)/2) [= n-1 or n-2] / every row
    const uint32_t n_sum = (uint32_t)((n + t_step - 1) / t_step);
    hipLaunchKernelGGL(mi::stats_sum_kernel, dim3((unsigned)d, G), dim3(64), 0, st, x, (uint32_t)n, t_step, (uint32_t)d, (uint64_t)C, G, B + o_sum);
    std::vector<double> part((size_t)G * d), mean_h(d);
    HIP_TRY(hipMemcpyAsync(part.data(), B + o_sum, part.size() * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (size_t j = 0; j < d; ++j) {
        double s = 0.0;
        for (uint32_t g = 0; g < G; ++g) s += part[(size_t)g * d + j];
        mean_h[j] = s / ((double)n_sum * (double)C);
    }
    if (!one_pass && mean) std::memcpy(mean, mean_h.data(), d * 8);
    if (!acov && !rhat && !ess) return MI_OK;
    HIP_TRY(hipMemcpyAsync(B + o_mean, mean_h.data(), d * 8, hipMemcpyHostToDevice, st));

    std::vector<double> ac((size_t)n * d, std::nan(""));     // ac[k * d + j]: pooled over chains, unbiased per lag
    std::vector<double> mp;                                   // [G][d][3] moments for R-hat
    // adds the G partials of lags [k_lo, k_lo + cnt) of the listed dimensions (stride `stride` lags per (group, dimension))
    auto collect = [&](const std::vector<double>& ap, const std::vector<uint32_t>& dl, size_t stride, size_t k_lo, size_t cnt) {
        const size_t nd = dl.size();
        for (size_t jj = 0; jj < nd; ++jj)
            for (size_t k = 0; k < cnt && k_lo + k < nlag_max; ++k) {
                double s_ = 0.0;
                for (uint32_t g = 0; g < G; ++g) s_ += ap[((size_t)g * nd + jj) * stride + k];
                ac[(k_lo + k) * d + dl[jj]] = s_ / (double)C / (double)(n - (k_lo + k));
            }
    };
    // Geyer's initial positive sequence over the lags [0, nlag) (mcmc_amd/ess.py); *ended: the sum met its first non-positive pair
    // or ran through every lag there is, i.e. more lags would not change it
    auto geyer = [&](size_t nlag, size_t j, bool* ended) -> double {
        if (n < 4) { *ended = true; return (double)n; }
        const double a0 = ac[j];
        const double den = (a0 > 0.0) ? a0 : 1.0;
        double tau = -1.0;
        size_t t = 0;
        *ended = false;
        while (t + 1 < nlag) {
            const double pair = ac[t * d + j] / den + ac[(t + 1) * d + j] / den;
            if (pair <= 0.0) { *ended = true; break; }
            tau += 2.0 * pair;
            t += 2;
        }
        if (nlag >= nlag_max) *ended = true;
        const double e = (tau > 0.0) ? (double)n / std::max(tau, 1.0 / (double)n) : (double)n;
        return std::min(e, (double)n * 10.0);
    };
    std::vector<uint32_t> active(d);
    for (size_t j = 0; j < d; ++j) active[j] = (uint32_t)j;
    std::vector<double> ess_h(d, (double)n);
    size_t computed = 0;                                      // lags [0, computed) are known for the active dimensions
    auto fetch = [&](std::vector<double>& dst, const double* src, size_t cnt) -> int {
        dst.resize(cnt);
        HIP_TRY(hipMemcpyAsync(dst.data(), src, cnt * 8, hipMemcpyDeviceToHost, st));
        return MI_OK;
    };
    auto prune = [&]() {                                      // keep the dimensions whose sum has not ended inside [0, computed)
        std::vector<uint32_t> next;
        for (uint32_t j : active) { bool en; ess_h[j] = geyer(computed, j, &en); if (!en) next.push_back(j); }
        active.swap(next);
    };
    std::vector<double> ap;
    int rc;
    if (tiled) {
        const size_t lds = (size_t)(mi::STATS_TILE_T + 2 * mi::STATS_TILED_LAGS) * 64 * sizeof(double);
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(mi::stats_acov_tiled_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(mi::stats_acov_tiled_kernel, dim3((unsigned)d, G), dim3(64), lds, st, x, B + o_mean, (uint32_t)n, (uint32_t)d,
                           (uint64_t)C, G, B + o_part, B + o_mom);
        HIP_TRY(hipGetLastError());
        if ((rc = fetch(ap, B + o_part, (size_t)G * d * nlag_max))) return rc;
        if ((rc = fetch(mp, B + o_mom, (size_t)G * d * 3))) return rc;
        HIP_TRY(hipStreamSynchronize(st));
        collect(ap, active, nlag_max, 0, nlag_max);
        computed = nlag_max;
        prune();
    } else if (gram) {
        // every lag of every dimension in ONE pass over the slab, on the matrix cores (draw_stats.hpp: stats_gram_kernel)
        const int nb = (int)((n + 15) / 16);
        const size_t lds = mi::stats_gram_lds_bytes(nb);
        auto launch = [&](auto kern) -> int {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3((unsigned)d, G), dim3(512), lds, st, x, B + o_mean, (uint32_t)n, (uint32_t)d, (uint64_t)C, G, B + o_part, B + o_mom);
            HIP_TRY(hipGetLastError());
            return MI_OK;
        };
        switch (nb) {
        case 1: rc = launch(mi::stats_gram_kernel<1>); break;
        case 2: rc = launch(mi::stats_gram_kernel<2>); break;
        case 3: rc = launch(mi::stats_gram_kernel<3>); break;
        case 4: rc = launch(mi::stats_gram_kernel<4>); break;
        case 5: rc = launch(mi::stats_gram_kernel<5>); break;
        case 6: rc = launch(mi::stats_gram_kernel<6>); break;
        case 7: rc = launch(mi::stats_gram_kernel<7>); break;
        default: rc = launch(mi::stats_gram_kernel<8>); break;
        }
        if (rc) return rc;
        const size_t nr = (size_t)16 * nb;
        if ((rc = fetch(ap, B + o_part, (size_t)G * d * 2 * nr))) return rc;
        if ((rc = fetch(mp, B + o_mom, (size_t)G * d * 3))) return rc;
        HIP_TRY(hipStreamSynchronize(st));
        // move the centre from the provisional a_j (mean_h) to the pooled mean (see above); groups added in order
        std::vector<double> R(n);
        for (size_t j = 0; j < d; ++j) {
            double tot = 0.0;
            for (size_t t = 0; t < n; ++t) {
                double r_ = 0.0;
                for (uint32_t g = 0; g < G; ++g) r_ += ap[((size_t)g * d + j) * 2 * nr + nr + t];
                R[t] = r_; tot += r_;
            }
            const double e = tot / ((double)n * (double)C);
            double head = tot, tail = tot;                  // sum_{t < n-k} R_t and sum_{t >= k} R_t, k = 0
            for (size_t k = 0; k < n; ++k) {
                if (k > 0) { head -= R[n - k]; tail -= R[k - 1]; }
                double s_ = 0.0;
                for (uint32_t g = 0; g < G; ++g) s_ += ap[((size_t)g * d + j) * 2 * nr + k];
                const double cen = s_ - e * (head + tail) + (double)C * (double)(n - k) * e * e;
                ac[k * d + j] = cen / (double)C / (double)(n - k);
            }
            mean_h[j] += e;
            // (the R-hat moments are sums over chains of the chain mean, its square and the chain variance about the provisional centre:
            //  the variance of the chain means and the within-chain variances do not depend on the centre)
        }
        if (mean) std::memcpy(mean, mean_h.data(), d * 8);
        computed = n;
        prune();
    } else {
        if (!acov) {
            // ESS / R-hat only: the first 16 lags (and the moments) straight from HBM at stream speed; a dimension whose Geyer sum ends
            // inside them is done
            hipLaunchKernelGGL(mi::stats_window_kernel<16>, dim3((unsigned)d, G), dim3(64), 0, st, x, B + o_mean, (uint32_t)n, (uint32_t)d,
                               (uint64_t)C, G, B + o_part, B + o_mom);
            HIP_TRY(hipGetLastError());
            if ((rc = fetch(ap, B + o_part, (size_t)G * d * 16))) return rc;
            if ((rc = fetch(mp, B + o_mom, (size_t)G * d * 3))) return rc;
            HIP_TRY(hipStreamSynchronize(st));
            collect(ap, active, 16, 0, 16);
            computed = std::min<size_t>(16, n);
            prune();
        }
        // further lags in passes of up to STATS_PASS_BLOCKS blocks of 16, for the dimensions still open (all of them, every lag, when
        // the autocovariance itself is asked for)
        while (!active.empty() && computed < n) {
            const uint32_t b_lo = (uint32_t)(computed / mi::STATS_LB);
            const uint32_t b_all = (uint32_t)((n + mi::STATS_LB - 1) / mi::STATS_LB);
            // (ESS only: the first pass after the 16 streamed lags takes two blocks -- lags 16..47 end most sums of a sampler that mixes)
            const uint32_t b_hi = std::min<uint32_t>(b_all, b_lo + ((!acov && b_lo == 1) ? 2u : (uint32_t)mi::STATS_PASS_BLOCKS));
            const size_t npl = (size_t)(b_hi - b_lo) * mi::STATS_LB;
            const bool all_dims = active.size() == d;
            const int want_mom = (b_lo == 0) ? 1 : 0;
            if (!all_dims) HIP_TRY(hipMemcpyAsync(B + o_dims, active.data(), active.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            const size_t lds = (n * 64 + npl * 64) * sizeof(double);
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(mi::stats_acov_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(mi::stats_acov_kernel, dim3((unsigned)active.size(), G), dim3(64), lds, st, x, B + o_mean, (uint32_t)n, (uint32_t)d,
                               (uint64_t)C, G, all_dims ? nullptr : reinterpret_cast<const uint32_t*>(B + o_dims), b_lo, b_hi, want_mom,
                               B + o_part, B + o_mom);
            HIP_TRY(hipGetLastError());
            if ((rc = fetch(ap, B + o_part, (size_t)G * active.size() * npl))) return rc;
            if (want_mom) { if ((rc = fetch(mp, B + o_mom, (size_t)G * d * 3))) return rc; }
            HIP_TRY(hipStreamSynchronize(st));
            collect(ap, active, npl, (size_t)b_lo * mi::STATS_LB, npl);
            computed = std::min<size_t>(n, (size_t)b_hi * mi::STATS_LB);
            if (!acov) prune();
        }
        if (acov) { computed = n; prune(); }
    }
    if (acov) std::memcpy(acov, ac.data(), ac.size() * 8);
    if (rhat)
        for (size_t j = 0; j < d; ++j) {
            double sm = 0.0, sm2 = 0.0, sv = 0.0;
            for (uint32_t g = 0; g < G; ++g) { const double* o = &mp[((size_t)g * d + j) * 3]; sm += o[0]; sm2 += o[1]; sv += o[2]; }
            const double W = sv / (double)C;                                                  // mean within-chain variance
            const double mbar = sm / (double)C;
            const double B_over_n = (C > 1) ? (sm2 - (double)C * mbar * mbar) / (double)(C - 1) : 0.0;   // variance of the chain means
            const double var_plus = ((double)(n - 1) / (double)n) * W + B_over_n;
            rhat[j] = (W > 0.0) ? std::sqrt(var_plus / W) : 1.0;
        }
    if (ess) std::memcpy(ess, ess_h.data(), d * 8);
    return MI_OK;
}

// ---- multi-GPU helpers of the C ABI (one process per GPU; mcmc_amd/dist.py is the torch.distributed form of the same thing)
void mi_mcmc_shard_bounds(uint64_t n_total, uint32_t world, uint32_t rank, uint64_t* chain0, uint64_t* n_local)
{
    // contiguous, balanced: the first (n % world) ranks get one extra chain
    const uint64_t base = world ? n_total / world : 0, extra = world ? n_total % world : 0;
    if (n_local) *n_local = base + (rank < extra ? 1 : 0);
    if (chain0) *chain0 = (uint64_t)rank * base + (rank < extra ? rank : extra);
}

}  // extern "C"

namespace {

// all[k][j][chain0(r) + c] = rank_major[off(r) + (k d + j) n_local(r) + c]
__global__ __launch_bounds__(256) void merge_shards_kernel(const double* __restrict__ src, uint32_t world, uint64_t rows, uint64_t row0,
                                                          uint64_t C, double* __restrict__ dst)
{
    const uint64_t base = C / world, extra = C % world;
    const uint64_t row = row0 + blockIdx.y;                        // k d + j
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < C; c += (uint64_t)gridDim.x * blockDim.x) {
        // rank of global chain c and its position inside the shard
        const uint64_t cut = extra * (base + 1);
        const uint64_t r = (c < cut) ? c / (base + 1) : extra + (base ? (c - cut) / base : 0);
        const uint64_t c0 = r * base + (r < extra ? r : extra), nl = base + (r < extra ? 1 : 0);
        const uint64_t off = rows * c0;                            // shards before r hold rows * chain0(r) doubles in total
        dst[row * C + c] = src[off + row * nl + (c - c0)];
    }
}

struct Rccl {
    void* h = nullptr;
    int (*group_start)() = nullptr;
    int (*group_end)() = nullptr;
    int (*broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*allgather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*err)(int) = nullptr;
};
Rccl* rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) { r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (r.h) break; }
        if (!r.h) return;
        r.group_start = reinterpret_cast<int (*)()>(dlsym(r.h, "ncclGroupStart"));
        r.group_end = reinterpret_cast<int (*)()>(dlsym(r.h, "ncclGroupEnd"));
        r.broadcast = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, hipStream_t)>(dlsym(r.h, "ncclBroadcast"));
        r.allgather = reinterpret_cast<int (*)(const void*, void*, size_t, int, void*, hipStream_t)>(dlsym(r.h, "ncclAllGather"));
        r.err = reinterpret_cast<const char* (*)(int)>(dlsym(r.h, "ncclGetErrorString"));
    });
    return (r.h && r.group_start && r.group_end && r.broadcast && r.allgather) ? &r : nullptr;
}

}  // namespace

extern "C" {

int mi_mcmc_merge_shards(const double* rank_major, uint32_t world, uint64_t n_keep, uint64_t d, uint64_t C, double* all, void* stream)
{
    if (!rank_major || !all || world == 0) return fail(MI_ERR_BAD_ARG, "merge_shards: null buffer / empty world");
    const uint64_t rows = n_keep * d;
    if (rows == 0 || C == 0) return MI_OK;
    const unsigned gx = (unsigned)std::min<uint64_t>((C + 255) / 256, 4096);
    (void)hipGetLastError();
    for (uint64_t r0 = 0; r0 < rows; r0 += 65535) {        // grid.y limit
        const unsigned gy = (unsigned)std::min<uint64_t>(65535, rows - r0);
        hipLaunchKernelGGL(merge_shards_kernel, dim3(gx, gy), dim3(256), 0, static_cast<hipStream_t>(stream), rank_major, world, rows, r0, C, all);
    }
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

// ragged shards: one broadcast per rank into the rank-major staging buffer, grouped into one RCCL launch
int mi_mcmc_allgather_draws_ragged(void* comm, uint32_t world, uint32_t rank, const double* local, uint64_t n_keep, uint64_t d, uint64_t C,
                                   double* scratch, double* all, void* stream)
{
    if (!comm || !scratch || !all || world == 0 || rank >= world) return fail(MI_ERR_BAD_ARG, "allgather_draws: bad communicator / buffers / rank");
    Rccl* r = rccl();
    if (!r) return fail(MI_ERR_UNSUPPORTED, "allgather_draws: librccl.so could not be loaded");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint64_t rows = n_keep * d;
    uint64_t c0 = 0, nl = 0;
    mi_mcmc_shard_bounds(C, world, rank, &c0, &nl);
    if (nl > 0 && !local) return fail(MI_ERR_BAD_ARG, "allgather_draws: local_draws is required for a non-empty shard");
    int e = r->group_start();
    for (uint32_t q = 0; q < world && e == 0; ++q) {
        uint64_t qc0 = 0, qn = 0;
        mi_mcmc_shard_bounds(C, world, q, &qc0, &qn);
        if (qn == 0) continue;
        // (sendbuff is read on the root only; the other ranks pass their receive slot, a valid device pointer, not NULL)
        double* slot = scratch + rows * qc0;
        e = r->broadcast(q == rank ? static_cast<const void*>(local) : static_cast<const void*>(slot), slot, (size_t)(rows * qn), 8 /* ncclDouble */, (int)q, comm, st);
    }
    const int e2 = r->group_end();
    if (e != 0 || e2 != 0) return fail(MI_ERR_HIP, "allgather_draws: RCCL: %s", r->err ? r->err(e ? e : e2) : "error");
    return mi_mcmc_merge_shards(scratch, world, n_keep, d, C, all, stream);
}

// north_star: "a single RCCL all-gather over xGMI to collate draws_out".  Equal shards (every BASELINE split: 65 536 / 8, 2^20 / 8) are
// exactly one ncclAllGather into the rank-major staging buffer, then the merge kernel puts every chain at its global index; ragged
// shards fall back to the grouped broadcasts above.
int mi_mcmc_allgather_draws(void* comm, uint32_t world, uint32_t rank, const double* local, uint64_t n_keep, uint64_t d, uint64_t C,
                            double* scratch, double* all, void* stream)
{
    if (!comm || !scratch || !all || world == 0 || rank >= world) return fail(MI_ERR_BAD_ARG, "allgather_draws: bad communicator / buffers / rank");
    if (C % world != 0 || C == 0) return mi_mcmc_allgather_draws_ragged(comm, world, rank, local, n_keep, d, C, scratch, all, stream);
    Rccl* r = rccl();
    if (!r) return fail(MI_ERR_UNSUPPORTED, "allgather_draws: librccl.so could not be loaded");
    if (!local) return fail(MI_ERR_BAD_ARG, "allgather_draws: local_draws is required for a non-empty shard");
    const uint64_t per = n_keep * d * (C / world);
    const int e = r->allgather(local, scratch, (size_t)per, 8 /* ncclDouble */, comm, static_cast<hipStream_t>(stream));
    if (e != 0) return fail(MI_ERR_HIP, "allgather_draws: RCCL: %s", r->err ? r->err(e) : "error");
    return mi_mcmc_merge_shards(scratch, world, n_keep, d, C, all, stream);
}

// ---- SURVEY 8(e)'s own receive layout: rank-major [G][n_keep][d][C / G], which equal shards fill with ONE ncclAllGather straight into
// the caller's buffer -- no staging buffer, no merge kernel, half the memory of mi_mcmc_allgather_draws (configs[4], n_keep = 8: 64 GiB
// + the 8 GiB local slab per GPU instead of 64 + 64 + 8).  Ragged shards: the packed form, shard r at offset rows * chain0(r) with its
// own row length n_local(r) (what mi_mcmc_merge_shards reads); one grouped broadcast per rank.
// row0 / n_keep_total: the slab is rows [row0, row0 + n_keep) of a run that keeps n_keep_total draws, gathered INTO the run's one
// rank-major buffer (per kept-draw slab, overlapped with the next trajectory: mi_mcmc_allgather_draws_begin below).  A partial slab
// is not contiguous on the receiving side across ranks, so it travels as one grouped broadcast per rank (one RCCL launch).
namespace {
int gather_rank_major(Rccl* r, void* comm, uint32_t world, uint32_t rank, const double* local, uint64_t n_keep, uint64_t d, uint64_t C,
                      uint64_t row0, uint64_t n_keep_total, double* all, hipStream_t st)
{
    uint64_t c0 = 0, nl = 0;
    mi_mcmc_shard_bounds(C, world, rank, &c0, &nl);
    if (nl > 0 && !local) return fail(MI_ERR_BAD_ARG, "allgather_draws: local_draws is required for a non-empty shard");
    if (C % world == 0 && C > 0 && row0 == 0 && n_keep == n_keep_total) {       // north_star's single all-gather
        const int e = r->allgather(local, all, (size_t)(n_keep * d * (C / world)), 8 /* ncclDouble */, comm, st);
        if (e != 0) return fail(MI_ERR_HIP, "allgather_draws: RCCL: %s", r->err ? r->err(e) : "error");
        return MI_OK;
    }
    int e = r->group_start();
    for (uint32_t q = 0; q < world && e == 0; ++q) {
        uint64_t qc0 = 0, qn = 0;
        mi_mcmc_shard_bounds(C, world, q, &qc0, &qn);
        if (qn == 0 || n_keep == 0) continue;
        double* slot = all + n_keep_total * d * qc0 + row0 * d * qn;             // shard q's block, its rows [row0, row0 + n_keep)
        e = r->broadcast(q == rank ? static_cast<const void*>(local) : static_cast<const void*>(slot), slot, (size_t)(n_keep * d * qn), 8, (int)q, comm, st);
    }
    const int e2 = r->group_end();
    if (e != 0 || e2 != 0) return fail(MI_ERR_HIP, "allgather_draws: RCCL: %s", r->err ? r->err(e ? e : e2) : "error");
    return MI_OK;
}

// one communication stream per device for the overlapped form: collectives of one communicator must be enqueued in the same order
// on every rank, which a single stream per device gives by construction
hipStream_t comm_stream(int dev)
{
    static std::mutex mu;
    static hipStream_t streams[64] = {};
    std::lock_guard<std::mutex> g(mu);
    if (dev < 0 || dev >= 64) return nullptr;
    if (!streams[dev] && hipStreamCreateWithFlags(&streams[dev], hipStreamNonBlocking) != hipSuccess) streams[dev] = nullptr;
    return streams[dev];
}
}  // namespace

struct mi_collation { hipEvent_t done; };

int mi_mcmc_allgather_draws_rank_major(void* comm, uint32_t world, uint32_t rank, const double* local, uint64_t n_keep, uint64_t d, uint64_t C,
                                       double* all_rank_major, void* stream)
{
    if (!comm || !all_rank_major || world == 0 || rank >= world) return fail(MI_ERR_BAD_ARG, "allgather_draws: bad communicator / buffers / rank");
    Rccl* r = rccl();
    if (!r) return fail(MI_ERR_UNSUPPORTED, "allgather_draws: librccl.so could not be loaded");
    return gather_rank_major(r, comm, world, rank, local, n_keep, d, C, 0, n_keep, all_rank_major, static_cast<hipStream_t>(stream));
}

uint64_t mi_mcmc_rank_major_index(uint64_t C, uint32_t world, uint64_t n_keep, uint64_t d, uint64_t i, uint64_t j, uint64_t c)
{
    if (world == 0 || c >= C) return ~(uint64_t)0;
    const uint64_t base = C / world, extra = C % world, cut = extra * (base + 1);
    const uint64_t r = (c < cut) ? c / (base + 1) : extra + (base ? (c - cut) / base : 0);
    const uint64_t c0 = r * base + (r < extra ? r : extra), nl = base + (r < extra ? 1 : 0);
    return n_keep * d * c0 + (i * d + j) * nl + (c - c0);
}

int mi_mcmc_allgather_draws_begin(void* comm, uint32_t world, uint32_t rank, const double* local, uint64_t n_keep, uint64_t d, uint64_t C,
                                  uint64_t row0, uint64_t n_keep_total, double* all_rank_major, void* producer_stream, mi_collation** handle)
{
    if (!handle) return fail(MI_ERR_BAD_ARG, "allgather_draws_begin: null handle");
    *handle = nullptr;
    if (!comm || !all_rank_major || world == 0 || rank >= world || row0 + n_keep > n_keep_total)
        return fail(MI_ERR_BAD_ARG, "allgather_draws_begin: bad communicator / buffers / rank / row range");
    Rccl* r = rccl();
    if (!r) return fail(MI_ERR_UNSUPPORTED, "allgather_draws: librccl.so could not be loaded");
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipStream_t cs = comm_stream(dev);
    if (!cs) return fail(MI_ERR_HIP, "allgather_draws_begin: no communication stream");
    hipEvent_t ready = nullptr, done = nullptr;
    // (every early return below releases what it created: no event outlives a failed call)
    auto try_hip = [&](hipError_t e, const char* what) -> int {
        if (e == hipSuccess) return MI_OK;
        if (ready) (void)hipEventDestroy(ready);
        if (done) (void)hipEventDestroy(done);
        return fail(MI_ERR_HIP, "allgather_draws_begin: %s: %s", what, hipGetErrorString(e));
    };
    int rc;
    if ((rc = try_hip(hipEventCreateWithFlags(&ready, hipEventDisableTiming), "event"))) return rc;
    if ((rc = try_hip(hipEventRecord(ready, static_cast<hipStream_t>(producer_stream)), "record"))) return rc;   // the slab is complete when the producer stream gets here
    if ((rc = try_hip(hipStreamWaitEvent(cs, ready, 0), "wait"))) return rc;
    (void)hipEventDestroy(ready);                                                  // (released once the wait has consumed it)
    ready = nullptr;
    rc = gather_rank_major(r, comm, world, rank, local, n_keep, d, C, row0, n_keep_total, all_rank_major, cs);
    if (rc) return rc;
    if ((rc = try_hip(hipEventCreateWithFlags(&done, hipEventDisableTiming), "event"))) return rc;
    if ((rc = try_hip(hipEventRecord(done, cs), "record"))) return rc;
    *handle = new mi_collation{done};
    return MI_OK;
}

int mi_mcmc_allgather_draws_wait(mi_collation* handle, void* consumer_stream, int block_host)
{
    if (!handle) return fail(MI_ERR_BAD_ARG, "allgather_draws_wait: null handle");
    hipError_t e = block_host ? hipEventSynchronize(handle->done) : hipStreamWaitEvent(static_cast<hipStream_t>(consumer_stream), handle->done, 0);
    (void)hipEventDestroy(handle->done);
    delete handle;
    if (e != hipSuccess) return fail(MI_ERR_HIP, "allgather_draws_wait: %s", hipGetErrorString(e));
    return MI_OK;
}

int mi_mcmc_draws_to_chain_major(const double* kdc, uint64_t n_keep, uint64_t d, uint64_t C, double* out)
{
    if (!kdc || !out) return fail(MI_ERR_BAD_ARG, "null buffer");
    // host arrays: blocked over chains so that both sides stay in cache (the device form below is the one for slabs in HBM)
    constexpr uint64_t CB = 64;
    for (uint64_t c0 = 0; c0 < C; c0 += CB)
        for (uint64_t j = 0; j < d; ++j)
            for (uint64_t i = 0; i < n_keep; ++i) {
                const double* src = kdc + (i * d + j) * C;
                const uint64_t c1 = std::min(C, c0 + CB);
                for (uint64_t c = c0; c < c1; ++c) out[(c * d + j) * n_keep + i] = src[c];
            }
    return MI_OK;
}

int mi_mcmc_draws_to_chain_major_device(const double* kdc_dev, uint64_t n_keep, uint64_t d, uint64_t C, double* out_dev, void* stream)
{
    if (!kdc_dev || !out_dev) return fail(MI_ERR_BAD_ARG, "null buffer");
    if (n_keep == 0 || d == 0 || C == 0) return MI_OK;
    if (n_keep > 0xffffffffULL || d > 0x7fffffffULL || (C + 63) / 64 > 65535ULL * 64ULL) return fail(MI_ERR_BAD_ARG, "draws_to_chain_major: shape out of range");
    hipStream_t st = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    // grid.y is limited to 65 535: chains in slices of 65 535 tiles
    const uint64_t tiles = (C + 63) / 64;
    for (uint64_t t0 = 0; t0 < tiles; t0 += 65535) {
        const uint64_t nt = std::min<uint64_t>(65535, tiles - t0);
        const uint64_t c_first = t0 * 64, c_cnt = std::min<uint64_t>(C - c_first, nt * 64);
        // a slice [c_first, c_first + c_cnt) of the slab has the same row stride C: pass the full C and offset pointers
        hipLaunchKernelGGL(mi::draws_to_chain_major_slice_kernel, dim3((unsigned)d, (unsigned)nt), dim3(256), 0, st, kdc_dev, (uint32_t)n_keep, (uint32_t)d, C, c_first, c_cnt, out_dev);
    }
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

}  // extern "C"
