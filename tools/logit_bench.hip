// Timing harness for the LDS-staged logistic kernel (experiments only; not part of the library).
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Iinclude/mi_mcmc_engine -DMI_RNG_NOINLINE -DMI_LOGIT_ABLATE=<bits> -o lb tools/logit_bench.hip
//   ./lb [chains] [draws] [algo 0=mala 1=hmc] [n_leap]
#ifndef MI_KC_MODE
#define MI_KC_MODE 2      // as logistic_lds.hip
#endif
#include "../mcmc_amd/csrc/logistic_lds.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv)
{
    constexpr int NTQ = 8;
    using G = mi::LogitGeo<NTQ>;
    const size_t C = argc > 1 ? atoll(argv[1]) : 65536;
    const uint32_t draws = argc > 2 ? atoi(argv[2]) : 20;
    const int algo = argc > 3 ? atoi(argv[3]) : 0;
    const uint32_t n_leap = argc > 4 ? atoi(argv[4]) : 4;
    const uint32_t d = 512, N = 1024, NB = N / 16;
    std::vector<double> X((size_t)N * d), y(N);
    uint64_t s = 12345;
    auto rnd = [&]() { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(s >> 11) * (1.0 / 9007199254740992.0); };
    for (auto& v : X) v = (rnd() - 0.5) * 0.2;
    for (auto& v : y) v = rnd() < 0.5 ? 0.0 : 1.0;
    double *Xd, *yd, *xp, *theta, *state;
    const size_t n_wg = (C + 31) / 32;
    CK(hipMalloc(&Xd, X.size() * 8)); CK(hipMalloc(&yd, y.size() * 8));
    CK(hipMalloc(&xp, (size_t)NB * G::XBUF_PAD * 8));
    CK(hipMalloc(&theta, (size_t)d * C * 8)); CK(hipMemset(theta, 0, (size_t)d * C * 8));
    CK(hipMalloc(&state, n_wg * 8 * 2 * G::NSQ * 64 * 8));
    CK(hipMemcpy(Xd, X.data(), X.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(yd, y.data(), y.size() * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mi::pack_logit_lds_kernel<NTQ>, dim3(NB), dim3(256), 0, 0, Xd, yd, d, N, xp);
    mi::LogitParams p{};
    p.Xp = xp; p.d = d; p.n_rows = N; p.NB = NB; p.C = C; p.chain0 = 0; p.theta = theta; p.state = state; p.draws = nullptr;
    uint64_t* clk; CK(hipMalloc(&clk, 8 * 128)); CK(hipMemset(clk, 0, 8 * 128));
    p.n_accept = (MI_LOGIT_ABLATE & (1024 | 2048)) ? clk : nullptr; p.seed = 1; p.n_burnin = draws; p.n_keep = 0; p.n_leap = n_leap;
    p.eps = 0.02; p.s2 = p.eps * p.eps; p.rs = 1.0 / p.s2; p.cons_term = -0.5 * d * 1.83787706640934548356; p.log_det = d * 2.0 * log(p.eps);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(theta, 0, (size_t)d * C * 8));
        CK(hipEventRecord(e0, 0));
        if (algo == 0) {
            auto k = mi::logit_lds_kernel<NTQ, mi::LOGIT_MALA>;
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES));
            hipLaunchKernelGGL(k, dim3((unsigned)n_wg), dim3(512), G::LDS_BYTES, 0, p);
        } else {
            auto k = mi::logit_lds_kernel<NTQ, mi::LOGIT_HMC>;
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES));
            hipLaunchKernelGGL(k, dim3((unsigned)n_wg), dim3(512), G::LDS_BYTES, 0, p);
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double evals = (double)C * (algo == 0 ? draws + 1 : draws * n_leap + 1);
        if (rep == 1) printf("ablate=%d algo=%d chains=%zu draws=%u ms=%.3f  TFLOP/s(4Nd per eval)=%.2f\n", MI_LOGIT_ABLATE, algo, C, draws, ms,
                             evals * 4.0 * N * d / (ms * 1e-3) / 1e12);
    }
    if (MI_LOGIT_ABLATE & 1024) { uint64_t h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost)); printf("wg0: %llu shader cycles, %llu wall ticks -> %.3f GHz\n", (unsigned long long)h[0], (unsigned long long)h[1], (double)h[0] / (double)h[1] * 0.1); }
    if (MI_LOGIT_ABLATE & 2048) { uint64_t h[128]; CK(hipMemcpy(h, clk, 8 * 128, hipMemcpyDeviceToHost));
        const char* nm[11] = {"eta-mfma", "part+barrierA", "row-terms", "barrierB", "grad-mfma", "dma-wait", "barrierC", "DRAW:rng+proposal", "eval-prologue", "eval-epilogue", "accept+store"};
        for (int w = 0; w < 8; ++w) { printf("wave %d:", w); for (int i = 0; i < 11; ++i) printf(" %s=%.0f", nm[i], (double)h[2 + w * 11 + i] / (i < 7 ? 64.0 * (draws + 1) : (double)draws)); printf("  (cycles per block; DRAW.. per draw)\n"); } }
    return 0;
}
