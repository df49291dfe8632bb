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
    const uint32_t G = (uint32_t)std::min<uint64_t>(64, (C + 63) / 64);      // chain groups (partials are added in order on the host)
    DevBuf sum_dev, mean_dev, acov_dev, mom_dev;
    HIP_TRY(sum_dev.alloc((size_t)G * d * 8)); HIP_TRY(mean_dev.alloc(d * 8));
    HIP_TRY(acov_dev.alloc((size_t)G * d * std::min<size_t>(n, (size_t)mi::STATS_MAX_N) * 8)); HIP_TRY(mom_dev.alloc((size_t)G * d * 3 * 8));
    hipLaunchKernelGGL(mi::stats_sum_kernel, dim3((unsigned)d, G), dim3(64), 0, st, x, (uint32_t)n, (uint32_t)d, (uint64_t)C, G, sum_dev.as<double>());
    std::vector<double> part((size_t)G * d), mean_h(d);
    HIP_TRY(hipMemcpyAsync(part.data(), sum_dev.p, part.size() * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (size_t j = 0; j < d; ++j) {
        double s = 0.0;
        for (uint32_t g = 0; g < G; ++g) s += part[(size_t)g * d + j];
        mean_h[j] = s / ((double)n * (double)C);
    }
    if (mean) std::memcpy(mean, mean_h.data(), d * 8);
    if (!acov && !rhat && !ess) return MI_OK;
    HIP_TRY(hipMemcpyAsync(mean_dev.p, mean_h.data(), d * 8, hipMemcpyHostToDevice, st));
    auto rhat_from = [&](const std::vector<double>& mp) {
        for (size_t j = 0; j < d; ++j) {
            double sm = 0.0, sm2 = 0.0, sv = 0.0;
            for (uint32_t g = 0; g < G; ++g) { const double* o = &mp[((size_t)g * d + j) * 3]; sm += o[0]; sm2 += o[1]; sv += o[2]; }
            const double W = sv / (double)C;                                                  // mean within-chain variance
            const double mbar = sm / (double)C;
            const double B_over_n = (C > 1) ? (sm2 - (double)C * mbar * mbar) / (double)(C - 1) : 0.0;   // variance of the chain means
            const double var_plus = ((double)(n - 1) / (double)n) * W + B_over_n;
            rhat[j] = (W > 0.0) ? std::sqrt(var_plus / W) : 1.0;
        }
    };
    // Geyer's initial positive sequence over the lags [0, nlag) of ac[k * d + j]; *ended: the sum met its first non-positive pair
    // (or ran through all n lags), i.e. more lags would not change it
    auto geyer = [&](const std::vector<double>& ac, size_t nlag, size_t j, bool* ended) -> double {
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
        if (nlag >= n) *ended = true;
        const double e = (tau > 0.0) ? (double)n / std::max(tau, 1.0 / (double)n) : (double)n;
        return std::min(e, (double)n * 10.0);
    };
    if (!acov) {
        // ESS / R-hat only: the first 16 lags straight from HBM (stats_window_kernel); done if Geyer's sum has ended in every dimension
        for (int L : {16}) {
            DevBuf a_dev;
            HIP_TRY(a_dev.alloc((size_t)G * d * L * 8));
            hipLaunchKernelGGL(mi::stats_window_kernel<16>, dim3((unsigned)d, G), dim3(64), 0, st, x, mean_dev.as<double>(), (uint32_t)n, (uint32_t)d, (uint64_t)C, G, a_dev.as<double>(), mom_dev.as<double>());
            HIP_TRY(hipGetLastError());
            std::vector<double> ap((size_t)G * d * L), mp((size_t)G * d * 3);
            HIP_TRY(hipMemcpyAsync(ap.data(), a_dev.p, ap.size() * 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(mp.data(), mom_dev.p, mp.size() * 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            const size_t nlag = std::min<size_t>((size_t)L, n);
            std::vector<double> ac(nlag * d);
            for (size_t j = 0; j < d; ++j)
                for (size_t k = 0; k < nlag; ++k) {
                    double s_ = 0.0;
                    for (uint32_t g = 0; g < G; ++g) s_ += ap[((size_t)g * d + j) * L + k];
                    ac[k * d + j] = s_ / (double)C / (double)(n - k);
                }
            bool all_ended = true;
            std::vector<double> e(d);
            for (size_t j = 0; j < d; ++j) { bool en; e[j] = geyer(ac, nlag, j, &en); all_ended = all_ended && en; }
            if (all_ended || !ess) {
                if (rhat) rhat_from(mp);
                if (ess) std::memcpy(ess, e.data(), d * 8);
                return MI_OK;
            }
        }
        // a slowly mixing series: every lag (n <= 160) or 128 of them, with the LDS-resident kernels below
    }
    // every lag for series of up to STATS_MAX_N draws; beyond, lags below STATS_TILED_LAGS from the streamed kernel
    const bool tiled = n > (size_t)mi::STATS_MAX_N;
    const size_t nlag = tiled ? (size_t)mi::STATS_TILED_LAGS : n;
    DevBuf acov_t;
    if (tiled) { HIP_TRY(acov_t.alloc((size_t)G * d * nlag * 8)); }
    double* const acov_out = tiled ? acov_t.as<double>() : acov_dev.as<double>();
    if (tiled) {
        const size_t lds = (size_t)(mi::STATS_TILE_T + 2 * mi::STATS_TILED_LAGS) * 64 * sizeof(double);
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(mi::stats_acov_tiled_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(mi::stats_acov_tiled_kernel, dim3((unsigned)d, G), dim3(64), lds, st, x, mean_dev.as<double>(), (uint32_t)n, (uint32_t)d,
                           (uint64_t)C, G, acov_out, mom_dev.as<double>());
    } else {
        const size_t lds = (n * 64 + ((n + 15) / 16) * 16) * sizeof(double);
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(mi::stats_acov_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(mi::stats_acov_kernel, dim3((unsigned)d, G), dim3(64), lds, st, x, mean_dev.as<double>(), (uint32_t)n, (uint32_t)d,
                           (uint64_t)C, G, (uint32_t)n, acov_out, mom_dev.as<double>());
    }
    HIP_TRY(hipGetLastError());
    std::vector<double> ap((size_t)G * d * nlag), mp((size_t)G * d * 3), ac((size_t)n * d, std::nan(""));
    HIP_TRY(hipMemcpyAsync(ap.data(), acov_out, ap.size() * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(mp.data(), mom_dev.p, mp.size() * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (size_t j = 0; j < d; ++j)
        for (size_t k = 0; k < nlag; ++k) {
            double s = 0.0;
            for (uint32_t g = 0; g < G; ++g) s += ap[((size_t)g * d + j) * nlag + k];
            ac[k * d + j] = s / (double)C / (double)(n - k);          // pooled over chains, unbiased per lag
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
    if (ess)
        for (size_t j = 0; j < d; ++j) {                               // Geyer's initial positive sequence (mcmc_amd/ess.py)
            if (n < 4) { ess[j] = (double)n; continue; }
            const double a0 = ac[j];
            const double den = (a0 > 0.0) ? a0 : 1.0;
            double tau = -1.0;
            size_t t = 0;
            while (t + 1 < nlag) {
                const double pair = ac[t * d + j] / den + ac[(t + 1) * d + j] / den;
                if (pair <= 0.0) break;
                tau += 2.0 * pair;
                t += 2;
            }
            double e = (tau > 0.0) ? (double)n / std::max(tau, 1.0 / (double)n) : (double)n;
            ess[j] = std::min(e, (double)n * 10.0);
        }
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
        r.err = reinterpret_cast<const char* (*)(int)>(dlsym(r.h, "ncclGetErrorString"));
    });
    return (r.h && r.group_start && r.group_end && r.broadcast) ? &r : nullptr;
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

int mi_mcmc_allgather_draws(void* comm, uint32_t world, uint32_t rank, const double* local, uint64_t n_keep, uint64_t d, uint64_t C,
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
    for (uint32_t q = 0; q < world && e == 0; ++q) {       // ragged shards: one broadcast per rank, grouped into one launch
        uint64_t qc0 = 0, qn = 0;
        mi_mcmc_shard_bounds(C, world, q, &qc0, &qn);
        if (qn == 0) continue;
        e = r->broadcast(q == rank ? local : nullptr, scratch + rows * qc0, (size_t)(rows * qn), 8 /* ncclDouble */, (int)q, comm, st);
    }
    const int e2 = r->group_end();
    if (e != 0 || e2 != 0) return fail(MI_ERR_HIP, "allgather_draws: RCCL: %s", r->err ? r->err(e ? e : e2) : "error");
    return mi_mcmc_merge_shards(scratch, world, n_keep, d, C, all, stream);
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
