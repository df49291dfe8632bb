// valu_rate.hip -- issue rate of the fp64 VALU instructions the elementwise kernels are made of (gfx950).
// One wave per SIMD (grid = 1024 workgroups of 64) or more; 8 independent chains per lane; reports cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int OP>
__global__ void rate_kernel(double* out, int iters, double a, double b, unsigned long long* cyc)
{
    double x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = a + (double)(threadIdx.x + k) * 1e-9;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (OP == 0) x[k] = __builtin_fma(x[k], b, a);
            if (OP == 1) { double r; asm volatile("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(x[k]), "v"(b)); x[k] = r; }
            if (OP == 2) { double r; asm volatile("v_add_f64 %0, %1, %2" : "=v"(r) : "v"(x[k]), "v"(b)); x[k] = r; }
            if (OP == 3) { double r; asm volatile("v_mul_f64 %0, %1, 0.5" : "=v"(r) : "v"(x[k])); x[k] = r; }
            if (OP == 4) { double r; asm volatile("v_mul_f64 %0, %1, %2" : "=v"(r) : "s"(b), "v"(x[k])); x[k] = r; }
        }
    }
    const long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = (unsigned long long)(t1 - t0);
}

template <int OP>
void run(const char* name, int waves_per_simd)
{
    const int iters = 20000, block = 64 * (waves_per_simd > 4 ? 4 : 1), grid = 1024 * waves_per_simd / (block / 64) ;
    double* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)grid * block * 8); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<OP><<<grid, block>>>(out, 100, 1.0, 0.999999, cyc);
    hipEventRecord(e0);
    rate_kernel<OP><<<grid, block>>>(out, iters, 1.0, 0.999999, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double winstr = (double)iters * 8;
    const double total = winstr * grid * (block / 64);
    printf("%-22s waves/SIMD %d: %.2f shader-clk cycles per wave-instr (one wave), %.3e wave-instr/s chip-wide = %.2f cycles/instr/SIMD at 2.4 GHz\n",
           name, waves_per_simd, (double)h / winstr, total / (ms * 1e-3), 1024.0 * 2.4e9 / (total / (ms * 1e-3)));
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f64", w);
        run<1>("v_mul_f64 v,v", w);
        run<2>("v_add_f64 v,v", w);
        run<3>("v_mul_f64 v,0.5", w);
        run<4>("v_mul_f64 s,v", w);
    }
    return 0;
}
