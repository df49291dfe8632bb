// launch_common.hpp -- shared by the *_launch.hip translation units
#pragma once
#include <hip/hip_runtime.h>

namespace mi {
// the kernel a call's time goes to, by the name rocprofv3 prints: read back through mi_mcmc_last_kernel() (thread-local;
// defined in mi_mcmc.hip).  Set by the launchers, so bench.py labels a measurement with what actually ran.
void note_kernel(const char* fmt, ...);
// TEST HOOK (mi_mcmc_test_set_grid_cap, mi_mcmc_probes.h): an upper limit on the workgroups of the PERSISTENT grids (nuts_memo.hpp,
// nuts_lds.hpp), 0 = none.  With a small cap a handful of chains exercises what only > 16 384 chains reach otherwise: the
// global counter, a slot taking its second and third chain, retire-on-leave.  Process-wide; results never depend on it.
uint64_t test_grid_cap();
inline uint64_t cap_grid(uint64_t n_wg) { const uint64_t c = test_grid_cap(); return (c != 0 && c < n_wg) ? c : n_wg; }
}  // namespace mi

#define MI_LAUNCH_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

// ---- runs cut into PIECES on the persistent grids of the memoised NUTS tick (nuts_memo_core.hpp, SPLIT; nuts_launch.hip has the reasoning): shared by the
// launchers of the built-in kernel, the bounded built-in route and the tile route.  `split_ws`: nuts_split_workspace_bytes(C, d) bytes --
// tails | queues | stand-ins for 3 counters, the step size and the dual-averaging state (7 C words) | a copy of the initial values (d C doubles)
#ifndef MI_MEMO_PIECES
#define MI_MEMO_PIECES 4
#endif
namespace mi {
constexpr uint32_t MEMO_PIECES = MI_MEMO_PIECES;
inline size_t memo_queue_bytes(uint64_t C) { return ((size_t)(MEMO_PIECES - 1u) * C * sizeof(uint32_t) + 255) & ~(size_t)255; }
inline size_t memo_split_bytes(uint64_t C, uint32_t d) { return 256 + memo_queue_bytes(C) + (size_t)7 * C * 8 + 256 + (size_t)d * C * 8 + 256; }
// PRM: NutsParams or TileParams.  Cuts when there are more chains than chain slots and the pieces hold two draws or more (a hand-over costs one tick, a draw tens of them); then also makes sure n_accept, n_leap, n_exec, step_out and
// adapt_state exist (the hand-over goes through them).  *backup: where the launcher may keep a copy of prm.theta (chains flagged after their first piece are
// replayed from their INITIAL values), or nullptr when the runs are not cut
template <class PRM>
int memo_setup_pieces(PRM& prm, void* split_ws, uint64_t n_slots, hipStream_t st, double** backup)
{
    prm.n_pieces = 1; prm.piece_len = 0; prm.piece_q = nullptr; prm.piece_tail = nullptr;
    *backup = nullptr;
    const uint32_t n_total = prm.n_burnin + prm.n_keep;
    if (split_ws == nullptr || prm.next_chain == nullptr || prm.C <= n_slots || n_total < 2u * MEMO_PIECES || prm.C >= (1ull << 28)) return 0;
    prm.piece_len = (n_total + MEMO_PIECES - 1u) / MEMO_PIECES;
    prm.n_pieces = (n_total + prm.piece_len - 1u) / prm.piece_len;
    char* b = static_cast<char*>(split_ws);
    prm.piece_tail = reinterpret_cast<uint32_t*>(b);
    prm.piece_q = reinterpret_cast<uint32_t*>(b + 256);
    const size_t q_bytes = memo_queue_bytes(prm.C);
    MI_LAUNCH_TRY(hipMemsetAsync(prm.piece_tail, 0, 256, st));
    MI_LAUNCH_TRY(hipMemsetAsync(prm.piece_q, 0xff, q_bytes, st));
    uint64_t* u = reinterpret_cast<uint64_t*>(b + 256 + q_bytes);       // stand-ins for what the hand-over goes through
    if (!prm.n_accept) prm.n_accept = u;
    if (!prm.n_leap) prm.n_leap = u + prm.C;
    if (!prm.n_exec) prm.n_exec = u + 2 * prm.C;
    double* dd = reinterpret_cast<double*>(u + 3 * prm.C);
    if (!prm.step_out) prm.step_out = dd;
    if (!prm.adapt_state) prm.adapt_state = dd + prm.C;
    *backup = dd + 4 * prm.C + 32;
    return 0;
}
}  // namespace mi

// dispatch on the tile count: F<NT>() for NT in {1, 2, 4, 8}
#define MI_DISPATCH_NT(nt, CALL1, CALL2, CALL4, CALL8) \
    ((nt) <= 1 ? (CALL1) : (nt) == 2 ? (CALL2) : (nt) <= 4 ? (CALL4) : (CALL8))
