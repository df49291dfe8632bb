// probes.hip -- diagnostics of the C ABI used by the GPU tests and the measurement tools (mi_probe_*): one MFMA tile, the
// deterministic math functions, the per-chain RNG, fp64 throughput ceilings.
#include <hip/hip_runtime.h>

#include <cstring>

#include "host_common.hpp"
#include "mi_mcmc_probes.h"
#include "det_math.hpp"
#include "hmc_dense.hpp"

using mi::host::fail;
using mi::host::DevBuf;

// ------------------------------------------------------------------ diagnostics
namespace {

__global__ void probe_mfma_kernel(const double* A, const double* B, const double* Cin, double* D)
{
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];       // A[i][k], 16x4 row-major
    const double b = B[(l >> 4) * 16 + (l & 15)];      // B[k][j], 4x16 row-major
    mi::double4_t c;
    for (int r = 0; r < 4; ++r) c[r] = Cin[((l >> 4) + 4 * r) * 16 + (l & 15)];
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

__global__ void probe_math_kernel(int fn, const double* x, uint64_t n, double* out, double* out2)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0, c = 0.0;
    switch (fn) {
    case 0: s = mi::det_exp(x[i]); break;
    case 1: s = mi::det_log(x[i]); break;
    case 2: mi::det_sincos2pi(x[i], s, c); break;
    case 3: s = mi::softplus(x[i]); break;
    case 4: s = mi::sigmoid(x[i]); break;
    default: s = __builtin_nan("");
    }
    out[i] = s;
    out2[i] = c;
}

__global__ void probe_normals_kernel(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t stream, uint64_t d, double* out)
{
    const uint64_t slot = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nslots = 4 * ((d + 7) / 8);
    if (slot >= nslots) return;
    const uint64_t b = slot / 4, j = slot % 4;
    const uint64_t i0 = 8 * b + j, i1 = i0 + 4;
    double z0, z1;
    mi::rng_normal_pair(seed, chain, draw, (uint32_t)slot, stream, z0, z1);
    if (i0 < d) out[i0] = z0;
    if (i1 < d) out[i1] = z1;
}

__global__ void probe_uniform_kernel(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t slot, double* out)
{
    out[0] = mi::rng_uniform(seed, chain, draw, slot);
}

// fp64 throughput ceilings: 8 independent accumulators per wave, no memory traffic.
__global__ __launch_bounds__(256) void peak_mfma_kernel(int iters, double* sink)
{
    mi::double4_t acc[8];
    for (int t = 0; t < 8; ++t) acc[t] = mi::double4_t{0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
    double s = 0.0;
    for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    if (s == 12345.678) sink[0] = s;
}

// cycles-per-MFMA probe: NACC independent accumulators per wave, optional LDS operand fetch
template <int NACC, bool USE_LDS>
__global__ __launch_bounds__(256) void mfma_cycles_kernel(int iters, unsigned long long* cyc, double* sink)
{
    __shared__ double lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) lds[i] = 1.0 + i * 1e-9;
    __syncthreads();
    mi::double4_t acc[NACC];
    for (int t = 0; t < NACC; ++t) acc[t] = mi::double4_t{0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    const int lane = threadIdx.x & 63;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < NACC; ++t) {
            if (USE_LDS) a = lds[((it + t) & 63) * 64 + lane];
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
    }
    const unsigned long long t1 = clock64();
    double s = 0.0;
    for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    if (s == 12345.678) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void peak_fma_kernel(int iters, double* sink)
{
    double acc[16];
    for (int t = 0; t < 16; ++t) acc[t] = threadIdx.x * 1e-3 + t;
    const double a = 1.0000001, b = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = __builtin_fma(acc[t], a, b);
    }
    double s = 0.0;
    for (int t = 0; t < 16; ++t) s += acc[t];
    if (s == 12345.678) sink[0] = s;
}

}  // namespace

extern "C" {

int mi_probe_mfma_f64(const double* A, const double* B, const double* Cin, double* D)
{
    if (!A || !B || !Cin || !D) return fail(MI_ERR_BAD_ARG, "null buffer");
    DevBuf a, b, c, dd;
    HIP_TRY(a.alloc(64 * 8)); HIP_TRY(b.alloc(64 * 8)); HIP_TRY(c.alloc(256 * 8)); HIP_TRY(dd.alloc(256 * 8));
    HIP_TRY(hipMemcpy(a.p, A, 64 * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b.p, B, 64 * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c.p, Cin, 256 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe_mfma_kernel, dim3(1), dim3(64), 0, 0, a.as<double>(), b.as<double>(), c.as<double>(), dd.as<double>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(D, dd.p, 256 * 8, hipMemcpyDeviceToHost));
    return MI_OK;
}

int mi_probe_math(int fn, const double* x, uint64_t n, double* out, double* out2)
{
    if (!x || !out || !out2) return fail(MI_ERR_BAD_ARG, "null buffer");
    DevBuf dx, d1, d2;
    HIP_TRY(dx.alloc(n * 8)); HIP_TRY(d1.alloc(n * 8)); HIP_TRY(d2.alloc(n * 8));
    HIP_TRY(hipMemcpy(dx.p, x, n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, fn, dx.as<double>(), n, d1.as<double>(), d2.as<double>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d1.p, n * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out2, d2.p, n * 8, hipMemcpyDeviceToHost));
    return MI_OK;
}

int mi_probe_normals(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t stream, uint64_t d, double* out)
{
    if (!out || d == 0) return fail(MI_ERR_BAD_ARG, "bad args");
    DevBuf o;
    HIP_TRY(o.alloc(d * 8));
    const uint64_t nslots = 4 * ((d + 7) / 8);
    hipLaunchKernelGGL(probe_normals_kernel, dim3((unsigned)((nslots + 63) / 64)), dim3(64), 0, 0, seed, chain, draw, stream, d, o.as<double>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, o.p, d * 8, hipMemcpyDeviceToHost));
    return MI_OK;
}

int mi_probe_uniform(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t slot, double* out)
{
    if (!out) return fail(MI_ERR_BAD_ARG, "null buffer");
    DevBuf o;
    HIP_TRY(o.alloc(8));
    hipLaunchKernelGGL(probe_uniform_kernel, dim3(1), dim3(1), 0, 0, seed, chain, draw, slot, o.as<double>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, o.p, 8, hipMemcpyDeviceToHost));
    return MI_OK;
}

// mode: waves per SIMD (1,2,4,8) ; returns shader cycles per MFMA per wave and wall TFLOP/s
int mi_probe_mfma_cycles(int waves_per_simd, int use_lds, int iters, double* cycles_per_mfma, double* tflops_out)
{
    if (!cycles_per_mfma || !tflops_out || iters <= 0) return fail(MI_ERR_BAD_ARG, "bad args");
    DevBuf sink, cyc;
    HIP_TRY(sink.alloc(8)); HIP_TRY(cyc.alloc(8));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    const int grid = 256 * waves_per_simd;      // 256-thread blocks: 1 wave per SIMD each
    const int nacc_req = (use_lds >> 8) & 0xff;  // independent accumulator chains per wave (1, 2, 4; default 8)
    const int nacc = (nacc_req == 1 || nacc_req == 2 || nacc_req == 4) ? nacc_req : 8;
    for (int rep = 0; rep < 2; ++rep) {
        HIP_TRY(hipEventRecord(e0, 0));
        if (nacc == 1) hipLaunchKernelGGL((mfma_cycles_kernel<1, false>), dim3(grid), dim3(256), 0, 0, iters, cyc.as<unsigned long long>(), sink.as<double>());
        else if (nacc == 2) hipLaunchKernelGGL((mfma_cycles_kernel<2, false>), dim3(grid), dim3(256), 0, 0, iters, cyc.as<unsigned long long>(), sink.as<double>());
        else if (nacc == 4) hipLaunchKernelGGL((mfma_cycles_kernel<4, false>), dim3(grid), dim3(256), 0, 0, iters, cyc.as<unsigned long long>(), sink.as<double>());
        else if (use_lds & 1) hipLaunchKernelGGL((mfma_cycles_kernel<8, true>), dim3(grid), dim3(256), 0, 0, iters, cyc.as<unsigned long long>(), sink.as<double>());
        else hipLaunchKernelGGL((mfma_cycles_kernel<8, false>), dim3(grid), dim3(256), 0, 0, iters, cyc.as<unsigned long long>(), sink.as<double>());
        HIP_TRY(hipEventRecord(e1, 0));
        HIP_TRY(hipEventSynchronize(e1));
    }
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c = 0;
    HIP_TRY(hipMemcpy(&c, cyc.p, 8, hipMemcpyDeviceToHost));
    *cycles_per_mfma = (double)c / ((double)iters * nacc);
    *tflops_out = (double)grid * 4 * iters * (double)nacc * 2048.0 / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return MI_OK;
}

int mi_probe_fp64_peak(int use_mfma, int iters, double* tflops_out)
{
    if (!tflops_out || iters <= 0) return fail(MI_ERR_BAD_ARG, "bad args");
    DevBuf sink;
    HIP_TRY(sink.alloc(8));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    const int grid = 256 * 8;   // 8 workgroups (32 waves) per CU
    for (int rep = 0; rep < 2; ++rep) {
        HIP_TRY(hipEventRecord(e0, 0));
        if (use_mfma) hipLaunchKernelGGL(peak_mfma_kernel, dim3(grid), dim3(256), 0, 0, iters, sink.as<double>());
        else hipLaunchKernelGGL(peak_fma_kernel, dim3(grid), dim3(256), 0, 0, iters, sink.as<double>());
        HIP_TRY(hipEventRecord(e1, 0));
        HIP_TRY(hipEventSynchronize(e1));
    }
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    const double waves = (double)grid * 4;
    const double flop = use_mfma ? waves * iters * 8.0 * (16.0 * 16 * 4 * 2) : waves * iters * 16.0 * 64 * 2;
    *tflops_out = flop / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return MI_OK;
}

}  // extern "C"
