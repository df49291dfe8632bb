// linalg_device.hip -- INV and CHOL_LOWER of a dense precond_mat ON THE DEVICE, in the operation order the oracle states for the reference's
// BMO_MATOPS_INV / BMO_MATOPS_CHOL_LOWER (ref: src/hmc.cpp:57-59, src/mala.cpp:57-58, include/stats/dmvnorm.hpp:36-41 through
// include/mcmc/mala.ipp:63-64; oracle/mcmc_oracle.c: orc_inv, orc_chol_lower; the host loops of host_linalg.hpp are the same statements).
//
// Both factorisations are d sequential steps of an embarrassingly parallel update, and every ELEMENT sees exactly the host's sequence of IEEE
// operations (one rounding per multiply, subtract, divide, sqrt; nothing contracted: -ffp-contract=off), so the results are the host's bit for bit:
//
//   INV (Gauss-Jordan with partial pivoting).  Step c: pivot = the first row r >= c with the largest |a[r][c]| (strict >, so NaN never wins
//   unless it sits on the diagonal); rows c and pivot swap; row c is divided by the pivot; every other row r with f = a[r][c] != 0 becomes
//   row_r - f * row_c, element by element, in both the working copy and the inverse.  Here: ONE cooperative launch of a persistent grid, rows
//   dealt cyclically to the workgroups, the matrices ping-ponged between two buffers so that a step reads only what the previous step wrote
//   (one grid barrier per step).  Every workgroup finds the pivot itself (from a compact copy of column c that the previous step's row owners
//   left behind: 8 d contiguous bytes, not d strided cache lines) and scales the pivot row into LDS.
//
//   CHOL_LOWER (column Cholesky: sum = A[j][j] - sum_k L[j][k]^2 in k order; L[i][j] = (A[i][j] - sum_k L[i][k] L[j][k]) / L[j][j]).  The
//   right-looking schedule applies the SAME subtractions to every element in the same order k = 0, 1, ...: step k finalises column k
//   (sqrt, d - k - 1 divisions -- redundantly per workgroup, into LDS) and subtracts L[i][k] L[j][k] from the trailing lower triangle in place.
//
// host: device_inverse / device_cholesky_lower take and return HOST matrices (row-major d x d); kernels of 5.4 / 3.7 ms at d = 512 against 89 ms /
// 40 ms of the one-core host loops that sat inside every blocking call with a dense precond_mat (155 ms per call for a
// 27-35 ms kernel).
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>

#include <cstdint>
#include <vector>

#include "host_common.hpp"

namespace cg = cooperative_groups;

namespace mi {
namespace {

constexpr int LA_THREADS = 256;

// Grid barrier of the resident (cooperatively launched) grid: a monotonic arrival counter in global memory -- barrier k is passed when it
// reaches k * gridDim.x.  One release-increment and an acquire-spin by ONE thread per workgroup between two workgroup barriers: the workgroup
// barrier in front has every wave's stores issued and completed (s_waitcnt vmcnt(0) is part of __syncthreads), the agent-scope release writes the
// XCD's L2 back, the acquire invalidates what this CU may hold of the other XCDs' lines.  (Measured against cooperative_groups' grid.sync() on
// MI355X: tools/linalg_bench.hip.)
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned& epoch)
{
    __syncthreads();
    epoch += gridDim.x;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < epoch) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

// (value, index) of the pivot rule: the sequential scan `best = |a[c][c]|; for r > c: if (|a[r][c]| > best) take r` picks the FIRST index of the
// largest non-NaN magnitude, and keeps c when |a[c][c]| itself is NaN.  key = magnitude, NaN -> -1 (below every magnitude)
struct Piv { double key; uint32_t idx; };
__device__ __forceinline__ Piv piv_better(Piv a, Piv b) { return (b.key > a.key || (b.key == a.key && b.idx < a.idx)) ? b : a; }

template <bool CG>
__global__ __launch_bounds__(LA_THREADS) void gj_inverse_kernel(double* a0, double* a1, double* b0, double* b1, double* colbuf, unsigned* bar, uint32_t d)
{
    unsigned epoch = 0;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* prow_a = lds;                 // [d] the pivot row of the working copy, divided by the pivot
    double* prow_b = lds + d;             // [d] ... of the inverse
    __shared__ Piv red[LA_THREADS / 64];
    __shared__ uint32_t piv_s;
    cg::grid_group grid = cg::this_grid();
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;

    // the inverse starts as the identity, the compact column as column 0 of A (this workgroup's rows), then the first barrier
    for (uint32_t r = blockIdx.x; r < d; r += gridDim.x) {
        for (uint32_t j = tid; j < d; j += LA_THREADS) b0[(size_t)r * d + j] = (j == r) ? 1.0 : 0.0;
        if (tid == 0) colbuf[r] = a0[(size_t)r * d];
    }
    if constexpr (CG) grid.sync(); else grid_barrier(bar, epoch);

    for (uint32_t c = 0; c < d; ++c) {
        const double* sa = (c & 1u) ? a1 : a0; double* da = (c & 1u) ? a0 : a1;
        const double* sb = (c & 1u) ? b1 : b0; double* db = (c & 1u) ? b0 : b1;
        const double* col = colbuf + (size_t)(c & 1u) * d;            // a[r][c] as the previous step left it
        double* coln = colbuf + (size_t)((c + 1u) & 1u) * d;          // a[r][c + 1] for the next step
        // ---- the pivot (every workgroup, the same answer)
        Piv best{-2.0, 0xffffffffu};
        for (uint32_t r = c + tid; r < d; r += LA_THREADS) {
            const double v = __builtin_fabs(col[r]);
            best = piv_better(best, Piv{(v != v) ? -1.0 : v, r});
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            Piv other{__shfl_xor(best.key, o), (uint32_t)__shfl_xor((int)best.idx, o)};
            best = piv_better(best, other);
        }
        if (lane == 0) red[wv] = best;
        __syncthreads();
        if (tid == 0) {
            Piv b = red[0];
            for (int k = 1; k < LA_THREADS / 64; ++k) b = piv_better(b, red[k]);
            const double vc = col[c];
            piv_s = (vc != vc) ? c : b.idx;                       // a NaN on the diagonal is `best` of the scan: nothing compares greater
        }
        __syncthreads();
        const uint32_t piv = piv_s;
        const double pv = col[piv];
        // ---- the pivot row, divided by the pivot
        for (uint32_t j = tid; j < d; j += LA_THREADS) {
            prow_a[j] = sa[(size_t)piv * d + j] / pv;
            prow_b[j] = sb[(size_t)piv * d + j] / pv;
        }
        __syncthreads();
        // ---- this workgroup's rows
        for (uint32_t r = blockIdx.x; r < d; r += gridDim.x) {
            if (r == c) {
                for (uint32_t j = tid; j < d; j += LA_THREADS) {
                    da[(size_t)r * d + j] = prow_a[j];
                    db[(size_t)r * d + j] = prow_b[j];
                    if (j == c + 1u) coln[r] = prow_a[j];
                }
            } else {
                const uint32_t sr = (r == piv) ? c : r;           // the swap: row `piv` now holds what row c held
                const double f = sa[(size_t)sr * d + c];
                const bool skip = f == 0.0;
                for (uint32_t j = tid; j < d; j += LA_THREADS) {
                    const double va = sa[(size_t)sr * d + j], vb = sb[(size_t)sr * d + j];
                    const double na = skip ? va : va - f * prow_a[j];
                    const double nb = skip ? vb : vb - f * prow_b[j];
                    da[(size_t)r * d + j] = na;
                    db[(size_t)r * d + j] = nb;
                    if (j == c + 1u) coln[r] = na;
                }
            }
        }
        if constexpr (CG) grid.sync(); else grid_barrier(bar, epoch);
    }
}

template <bool CG>
__global__ __launch_bounds__(LA_THREADS) void chol_lower_kernel(double* T, double* L, double* colbuf, unsigned* bar, uint32_t d)
{
    unsigned epoch = 0;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* lcol = lds;                   // [d] column k of L
    cg::grid_group grid = cg::this_grid();
    const uint32_t tid = threadIdx.x;
    // L starts as zero, the compact column as column 0 of A
    for (uint32_t r = blockIdx.x; r < d; r += gridDim.x) {
        for (uint32_t j = tid; j < d; j += LA_THREADS) L[(size_t)r * d + j] = 0.0;
        if (tid == 0) colbuf[r] = T[(size_t)r * d];
    }
    if constexpr (CG) grid.sync(); else grid_barrier(bar, epoch);
    for (uint32_t k = 0; k < d; ++k) {
        const double* col = colbuf + (size_t)(k & 1u) * d;            // T[i][k] as the previous steps left it (i >= k)
        double* coln = colbuf + (size_t)((k + 1u) & 1u) * d;
        const double ljj = __builtin_sqrt(col[k]);
        for (uint32_t i = k + tid; i < d; i += LA_THREADS) lcol[i] = (i == k) ? ljj : col[i] / ljj;
        __syncthreads();
        for (uint32_t i = k + blockIdx.x; i < d; i += gridDim.x) {          // rows i >= k, dealt cyclically from the current diagonal
            if (tid == 0) L[(size_t)i * d + k] = lcol[i];
            const double li = lcol[i];
            for (uint32_t j = k + 1u + tid; j <= i; j += LA_THREADS) {
                const double t = T[(size_t)i * d + j] - li * lcol[j];
                T[(size_t)i * d + j] = t;
                if (j == k + 1u) coln[i] = t;
            }
        }
        if constexpr (CG) grid.sync(); else grid_barrier(bar, epoch);
    }
}

int coop_grid(const void* kern, size_t lds, uint32_t want, uint32_t cap_arg, uint32_t& grid)
{
    int dev = 0, n_cu = 0, per_cu = 0, coop = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev));
    if (!coop) return host::fail(MI_ERR_UNSUPPORTED, "device linear algebra: the device does not support cooperative launches");
    HIP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, LA_THREADS, lds));
    if (n_cu < 1 || per_cu < 1) return host::fail(MI_ERR_HIP, "device linear algebra: no resident workgroup (%d CUs, %d per CU)", n_cu, per_cu);
    uint32_t cap = (uint32_t)n_cu;                                    // at most one workgroup per CU: every barrier is a round over the resident grid
    if (cap_arg && cap_arg < cap) cap = cap_arg;
    grid = want < cap ? want : cap;
    if (grid < 1) grid = 1;
    return MI_OK;
}

// The launch shape (measured, tools/linalg_bench.hip on MI355X): see LA_GRID_CAP below.
struct LaTiming { float kernel_ms = 0.f; };

// A, Ainv: HOST, row-major d x d.  Blocking.
template <bool CG>
int inverse_impl(const double* A, size_t d, double* Ainv, uint32_t grid_cap, LaTiming* tm)
{
    const size_t n = d * d, lds = 2 * d * sizeof(double);
    if (d == 0) return MI_OK;
    if (lds > 60 * 1024) return host::fail(MI_ERR_UNSUPPORTED, "device_inverse: d = %zu is beyond the LDS-staged pivot row (d <= 3840)", d);
    uint32_t grid = 1;
    const void* kern = reinterpret_cast<const void*>(gj_inverse_kernel<CG>);
    int rc = coop_grid(kern, lds, (uint32_t)d, grid_cap, grid); if (rc) return rc;
    host::DevBuf buf;                                   // ONE allocation: a0 | a1 | b0 | b1 | the two compact columns | the barrier counter
    HIP_TRY(buf.alloc((4 * n + 2 * d) * 8 + 64));
    double* base = buf.as<double>();
    HIP_TRY(hipMemcpyAsync(base, A, n * 8, hipMemcpyHostToDevice, nullptr));
    HIP_TRY(hipMemsetAsync(base + 4 * n + 2 * d, 0, 64, nullptr));
    double *pa0 = base, *pa1 = base + n, *pb0 = base + 2 * n, *pb1 = base + 3 * n, *pc = base + 4 * n;
    unsigned* pbar = reinterpret_cast<unsigned*>(base + 4 * n + 2 * d);
    uint32_t du = (uint32_t)d;
    void* args[] = {&pa0, &pa1, &pb0, &pb1, &pc, &pbar, &du};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (tm) { HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1)); HIP_TRY(hipEventRecord(e0, nullptr)); }
    HIP_TRY(hipLaunchCooperativeKernel(kern, dim3(grid), dim3(LA_THREADS), args, (unsigned)lds, nullptr));
    if (tm) { HIP_TRY(hipEventRecord(e1, nullptr)); HIP_TRY(hipEventSynchronize(e1)); HIP_TRY(hipEventElapsedTime(&tm->kernel_ms, e0, e1)); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); }
    HIP_TRY(hipMemcpy(Ainv, (d & 1u) ? pb1 : pb0, n * 8, hipMemcpyDeviceToHost));       // step c wrote buffer (c + 1) & 1
    return MI_OK;
}

// A: HOST, row-major d x d (its lower triangle is read); L: HOST, row-major d x d, zero above the diagonal.  Blocking.
template <bool CG>
int cholesky_impl(const double* A, size_t d, double* L, uint32_t grid_cap, LaTiming* tm)
{
    const size_t n = d * d, lds = d * sizeof(double);
    if (d == 0) return MI_OK;
    if (lds > 60 * 1024) return host::fail(MI_ERR_UNSUPPORTED, "device_cholesky_lower: d = %zu is beyond the LDS-staged column (d <= 7680)", d);
    uint32_t grid = 1;
    const void* kern = reinterpret_cast<const void*>(chol_lower_kernel<CG>);
    int rc = coop_grid(kern, lds, (uint32_t)d, grid_cap, grid); if (rc) return rc;
    host::DevBuf buf;                                   // ONE allocation: T | L | the two compact columns | the barrier counter
    HIP_TRY(buf.alloc((2 * n + 2 * d) * 8 + 64));
    double* base = buf.as<double>();
    HIP_TRY(hipMemcpyAsync(base, A, n * 8, hipMemcpyHostToDevice, nullptr));
    HIP_TRY(hipMemsetAsync(base + 2 * n + 2 * d, 0, 64, nullptr));
    double *pt = base, *pl = base + n, *pc = base + 2 * n;
    unsigned* pbar = reinterpret_cast<unsigned*>(base + 2 * n + 2 * d);
    uint32_t du = (uint32_t)d;
    void* args[] = {&pt, &pl, &pc, &pbar, &du};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (tm) { HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1)); HIP_TRY(hipEventRecord(e0, nullptr)); }
    HIP_TRY(hipLaunchCooperativeKernel(kern, dim3(grid), dim3(LA_THREADS), args, (unsigned)lds, nullptr));
    if (tm) { HIP_TRY(hipEventRecord(e1, nullptr)); HIP_TRY(hipEventSynchronize(e1)); HIP_TRY(hipEventElapsedTime(&tm->kernel_ms, e0, e1)); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); }
    HIP_TRY(hipMemcpy(L, pl, n * 8, hipMemcpyDeviceToHost));
    return MI_OK;
}

// The launch shape, measured (tools/linalg_bench.hip, MI355X, d = 512; kernel ms, INV / CHOL_LOWER):
//     grid <=            256            128           64           32           16
//     cg::grid.sync()    15.8 / 14.9    10.1 / 8.5    8.0 / 5.7    8.7 / 5.1    12.8 / 6.4
//     arrival counter     8.5 /  7.6     5.4 / 4.4    5.4 / 3.7    7.4 / 4.0    12.0 / 5.9
// A step is a few microseconds of work behind a barrier whose cost grows with the number of workgroups that arrive: 64 workgroups of 256
// threads (8 rows each at d = 512), the arrival-counter barrier.
#ifndef LA_GRID_CAP
#define LA_GRID_CAP 64
#endif
#ifndef LA_USE_CG
#define LA_USE_CG false
#endif

}  // namespace

namespace host {

int device_inverse(const double* A, size_t d, double* Ainv) { return inverse_impl<LA_USE_CG>(A, d, Ainv, LA_GRID_CAP, nullptr); }
int device_cholesky_lower(const double* A, size_t d, double* L) { return cholesky_impl<LA_USE_CG>(A, d, L, LA_GRID_CAP, nullptr); }

}  // namespace host
}  // namespace mi
