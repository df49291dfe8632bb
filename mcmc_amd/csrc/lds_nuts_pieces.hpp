// lds_nuts_pieces.hpp -- runs cut into PIECES on the persistent grids of nuts_lds.hpp: the launcher's side, shared by logistic_nuts_impl.hpp (identity / diagonal
// precond_mat, bounds) and logistic_nuts_dense_m.hip (a dense precond_mat).  The reasoning is in nuts_launch.hip (the plain kernel's cut), the protocol in nuts_lds.hpp.
#pragma once
#include "logistic_launch.hpp"

namespace mi {
namespace {

#ifndef MI_LDS_NUTS_PIECES
#define MI_LDS_NUTS_PIECES 4
#endif
constexpr uint32_t LDS_NUTS_PIECES = MI_LDS_NUTS_PIECES;
inline size_t lds_nuts_queue_bytes(uint64_t C) { return ((size_t)(LDS_NUTS_PIECES - 1u) * C * sizeof(uint32_t) + 255) & ~(size_t)255; }
// a chain that is flagged in a later piece is replayed from its INITIAL values, which its earlier pieces have overwritten in prm.theta: the launcher's copy comes back
__global__ void lds_nuts_restore_flagged_theta_kernel(const uint32_t* __restrict__ flag, const double* __restrict__ backup, double* __restrict__ theta, uint64_t C)
{
    if (flag[C] == 0u) return;                           // (the "any chain flagged" word)
    const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (c < C && flag[c] != 0u) theta[(size_t)blockIdx.y * C + c] = backup[(size_t)blockIdx.y * C + c];
}

// More chains than chain slots (32 per workgroup): every run is cut into LDS_NUTS_PIECES pieces that are handed out as separate work items.  Pieces of two draws
// and more: a hand-over costs one evaluation, a draw here tens of them.  With bounds the hand-over carries theta in the transformed space (nuts_lds.hpp).
// Sets prm.{n_pieces, piece_len, piece_q, piece_tail}, stand-ins for the outputs the hand-over goes through, and keeps a copy of prm.theta (*backup; nullptr:
// not cut) for the replay of chains flagged after their first piece.  Returns a hipError_t value.
template <int NTQ>
int lds_nuts_setup_pieces(LogitParams& prm, size_t n_wg, hipStream_t st, double** backup)
{
    prm.n_pieces = 1; prm.piece_len = 0; prm.piece_q = nullptr; prm.piece_tail = nullptr;
    *backup = nullptr;
    if constexpr (NTQ > 1) {                  // (NTQ = 1, d <= 64: measured 2 % SLOWER cut -- 65 536 chains of d = 20: 56.1 -> 57.1 ms; the wider tiles gain 4-9 %)
        const uint32_t n_total = prm.n_burnin + prm.n_keep;
        if (prm.split_ws != nullptr && prm.nf_flag != nullptr && prm.C > (uint64_t)n_wg * 32u && n_total >= 2u * LDS_NUTS_PIECES && prm.C < (1ull << 28)) {
            hipError_t e;
            prm.piece_len = (n_total + LDS_NUTS_PIECES - 1u) / LDS_NUTS_PIECES;
            prm.n_pieces = (n_total + prm.piece_len - 1u) / prm.piece_len;
            char* b = static_cast<char*>(prm.split_ws);
            prm.piece_tail = reinterpret_cast<uint32_t*>(b);
            prm.piece_q = reinterpret_cast<uint32_t*>(b + 256);
            const size_t q_bytes = lds_nuts_queue_bytes(prm.C);
            if ((e = hipMemsetAsync(prm.piece_tail, 0, 256, st)) != hipSuccess) return (int)e;
            if ((e = hipMemsetAsync(prm.piece_q, 0xff, q_bytes, st)) != hipSuccess) return (int)e;
            uint64_t* u = reinterpret_cast<uint64_t*>(b + 256 + q_bytes);       // stand-ins for what the hand-over goes through
            if (!prm.n_accept) prm.n_accept = u;
            if (!prm.n_leap_out) prm.n_leap_out = u + prm.C;
            if (!prm.n_exec_out) prm.n_exec_out = u + 2 * prm.C;
            double* dd = reinterpret_cast<double*>(u + 3 * prm.C);
            if (!prm.step_out) prm.step_out = dd;
            if (!prm.adapt_state) prm.adapt_state = dd + prm.C;
            *backup = dd + 4 * prm.C + 32;
            if ((e = hipMemcpyAsync(*backup, prm.theta, (size_t)prm.d * prm.C * sizeof(double), hipMemcpyDeviceToDevice, st)) != hipSuccess) return (int)e;
        }
    }
    return 0;
}
// behind the kernel: the flagged chains' initial values back into prm.theta, where the literal replay reads them
inline int lds_nuts_restore_flagged(const LogitParams& prm, double* backup, hipStream_t st)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (backup != nullptr)
        hipLaunchKernelGGL(lds_nuts_restore_flagged_theta_kernel, dim3((unsigned)((prm.C + 255) / 256), prm.d), dim3(256), 0, st, prm.nf_flag, backup, prm.theta, prm.C);
    return (int)hipGetLastError();
}
inline size_t lds_nuts_split_bytes(uint64_t C, uint32_t d) { return 256 + lds_nuts_queue_bytes(C) + (size_t)7 * C * 8 + 256 + (size_t)d * C * 8 + 256; }

}  // namespace
}  // namespace mi
