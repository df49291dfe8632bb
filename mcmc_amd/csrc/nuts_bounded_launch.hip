// nuts_bounded_launch.hip -- mcmc::nuts on the built-in Gaussian targets WITH settings.vals_bound (and / or a diagonal precond_mat next to it): the
// memoised tick of nuts_memo_core.hpp with the policy of the tile route (include/mi_mcmc_engine/nuts_tile.hpp: TileMemoPolicy<., GEN = true> --
// the chain lives in the transformed space, the target is evaluated at x = inv_transform(theta), the kick uses [J^-1] grad, the drift Minv p,
// U = -(K(x) + log_jacobian(theta)); every identity / diagonal product carries the NaN rule of the reference's dense product, so no replay) and the
// built-in Gaussian written as a tile target.  Until round 5 these runs went to nuts_gauss_async_kernel<., GENERAL> (tick-local state, every leaf
// executed): 23.8 s against 6.5 s on configs[3]'s target with mixed bounds, 16 384 chains, same bits (tools/tile_nuts_general_time.py).
// Replaces mcmc::internal::nuts_impl (ref: src/nuts.cpp:30-332 with box_log_kernel :84-95, mntm_update_fn :108-135, leap_frog_fn :139-154).
#include "nuts_dense.hpp"
#include "nuts_tile.hpp"
#include "launchers.hpp"
#include "launch_common.hpp"

namespace mi {
namespace {

// grad = -(P theta), value = -theta . P theta / 2, P's MFMA A-fragments in LDS (the target of examples/user_tile_target.hip, inside the engine)
template <int NT_>
struct BuiltinGaussTile {
    static constexpr int NT = NT_;
    const double* P;        // device, d x d row-major
    uint32_t d;
    uint32_t sep;           // an ISO / DIAG target (P: its expanded diagonal): the gradient element-wise, as the reference's target function takes it (target_times)
    __device__ void stage(double* lds) const { stage_precision<NT>(P, d, lds); }
    __device__ void grad_tile(const double* lds, const double (&th)[4 * NT], double (&g)[4 * NT], double& value, bool want_value) const
    {
        double w[4 * NT];
        target_times<NT>(lds + (threadIdx.x & 63), P, d, sep != 0u, th, w);
#pragma unroll
        for (int s = 0; s < 4 * NT; ++s) g[s] = -w[s];
        if (want_value) value = -0.5 * dot4<4 * NT>(th, w);
    }
};

template <int NT>
int run(const TileParams& prm, const double* P, uint32_t sep, hipStream_t st)
{
    const size_t lds = (size_t)NT * 4 * NT * 64 * sizeof(double) + memo::lds_bytes() + TileGen<NT>::lds_doubles() * sizeof(double);
    auto kern = tile_nuts::nuts_tile_kernel<BuiltinGaussTile<NT>, true>;
    note_kernel("nuts_tile_kernel<built-in Gaussian %d, true>", NT);
    MI_LAUNCH_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (prm.next_chain) MI_LAUNCH_TRY(hipMemsetAsync(prm.next_chain, 0, sizeof(uint32_t), st));
    hipLaunchKernelGGL(kern, dim3(prm.nuts_grid ? (unsigned)prm.nuts_grid : (unsigned)((prm.C + 63) / 64)), dim3(256), lds, st, prm, BuiltinGaussTile<NT>{P, prm.d, sep});
    return (int)hipGetLastError();
}

}  // namespace

// The persistent grid of the memoised tick on the tile policy: one workgroup (64 chain slots) per CU at most -- the workspace (148 vectors per chain SLOT:
// 10.1 GB for 65 536 slots at d = 128) is sized by the grid, and chains beyond its slots are handed out as slots fall free (nuts_memo_core.hpp)
uint64_t nuts_tile_grid(uint64_t C)
{
    int dev = 0, n_cu = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (n_cu <= 0) n_cu = 256;
    const uint64_t need = (C + 63) / 64;
    return cap_grid(need < (uint64_t)n_cu ? need : (uint64_t)n_cu);
}
// the tile route (mi_mcmc_run_tile_target): the same cut for a user target's nuts_tile_kernel on its persistent grid
int nuts_tile_setup_pieces(TileParams& p, void* split_ws, hipStream_t st)
{
    double* unused = nullptr;
    return memo_setup_pieces(p, split_ws, (uint64_t)p.nuts_grid * 64u, st, &unused);
}
size_t nuts_bounded_workspace_bytes(uint64_t C, int nt) { return tile_nuts::ws_bytes_grid(nuts_tile_grid(C), nt <= 1 ? 1 : nt == 2 ? 2 : nt <= 4 ? 4 : 8); }

int launch_nuts_gauss_bounded(const NutsParams& q, int nt, hipStream_t st)
{
    TileParams p{};
    p.d = q.d; p.C = q.C; p.chain0 = q.chain0;
    p.theta = q.theta; p.draws = q.draws; p.n_accept = q.n_accept; p.n_leap = q.n_leap; p.n_exec = q.n_exec;
    p.seed = q.seed; p.n_burnin = q.n_burnin; p.n_keep = q.n_keep; p.draw0 = q.draw0;
    p.next_chain = q.next_chain; p.nuts_grid = (uint32_t)nuts_tile_grid(q.C);
    p.ws = q.ws; p.step_out = q.step_out; p.depth_trace = q.depth_trace; p.adapt_state = q.adapt_state;
    p.n_adapt = q.n_adapt; p.max_depth = q.max_depth;
    p.delta = q.delta; p.eps_bar0 = q.eps_bar0; p.gamma = q.gamma; p.t0 = q.t0; p.kappa = q.kappa;
    p.vals_bound = q.vals_bound; p.btype = q.btype; p.lb = q.lb; p.ub = q.ub; p.m_sqrt = q.m_sqrt; p.m_inv = q.m_inv;
    const int ntp = nt <= 1 ? 1 : nt == 2 ? 2 : nt <= 4 ? 4 : 8;
    p.lds_user_doubles = (uint32_t)(ntp * 4 * ntp * 64);
    // more chains than the grid's chain slots: the runs cut into pieces, as nuts_launch.hip cuts the plain kernel's (no copy of the initial values: on this
    // policy a chain is never flagged)
    double* unused = nullptr;
    MI_LAUNCH_TRY((hipError_t)memo_setup_pieces(p, q.split_ws, (uint64_t)p.nuts_grid * 64u, st, &unused));
    return MI_DISPATCH_NT(nt, (run<1>(p, q.P, q.sep_target, st)), (run<2>(p, q.P, q.sep_target, st)), (run<4>(p, q.P, q.sep_target, st)), (run<8>(p, q.P, q.sep_target, st)));
}

}  // namespace mi
