/*
 * mi_mcmc.h -- C ABI of the MI355X many-chain HMC / MALA / NUTS (and RWMH) engine (libmi_mcmc.so).
 *
 * This is the drop-in boundary for the one hot path of kthohr/mcmc that BASELINE.json names.
 * Each entry point replaces, for C independent chains at once, one reference function:
 *
 *   mi_mcmc_hmc_*   <->  mcmc::hmc  -> internal::hmc_impl   /root/reference/include/mcmc/hmc.hpp:42-48,65-72,78-85
 *                                                            /root/reference/src/hmc.cpp:30-227
 *   mi_mcmc_mala_*  <->  mcmc::mala -> internal::mala_impl  /root/reference/include/mcmc/mala.hpp:43-49,66-73,79-86
 *                                                            /root/reference/src/mala.cpp:30-208, include/mcmc/mala.ipp:30-70
 *   mi_mcmc_nuts_*  <->  mcmc::nuts -> internal::nuts_impl  /root/reference/include/mcmc/nuts.hpp:42-48,65-72,78-85
 *                                                            /root/reference/src/nuts.cpp:30-332, include/mcmc/nuts.ipp:30-241
 *   mi_mcmc_rwmh_*  <->  mcmc::rwmh -> internal::rwmh_impl  /root/reference/include/mcmc/rwmh.hpp:42-47,64-70,79-85
 *                                                            /root/reference/src/rwmh.cpp:30-175
 *   mi_mcmc_rmhmc_* <->  mcmc::rmhmc -> internal::rmhmc_impl /root/reference/include/mcmc/rmhmc.hpp, src/rmhmc.cpp:30-287
 *   mi_settings     <->  algo_settings_t + hmc_/mala_/nuts_/rwmh_settings_t
 *                                                            /root/reference/include/misc/mcmc_structs.hpp:66-101,123-134,151-184
 *
 * The reference's std::function callback cannot run on the GPU, so the target density is
 * selected by a tagged descriptor (mi_target).  Plain C types only: no torch, no Eigen.
 * All structs start with their own size so the ABI can grow.
 *
 * Layouts (fp64 everywhere, fp_t = double: /root/reference/include/misc/mcmc_options.hpp:80-81,99):
 *   state  theta  [d][C]          structure-of-arrays, chain index contiguous
 *   draws         [n_keep][d][C]  one kept draw of every chain per slab
 *   n_accept      [C] uint64      post-burn-in accepts per chain (src/hmc.cpp:196-199)
 * Chain c of this call is global chain (chain0 + c): the Philox counter uses the global id, so
 * results do not depend on how chains are sharded over GPUs.
 */
#ifndef MI_MCMC_H
#define MI_MCMC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_MCMC_VERSION 0x000600

typedef enum mi_status {
    MI_OK = 0,
    MI_ERR_BAD_ARG = 1,       /* NULL where forbidden, unsupported kind / dimension, struct_size mismatch */
    MI_ERR_HIP = 2,           /* a HIP runtime call failed; see mi_mcmc_last_error() */
    MI_ERR_UNSUPPORTED = 3,   /* valid request the device path does not implement (never falls back to CPU) */
    MI_ERR_OOM = 4,
    MI_ERR_NO_DEVICE = 5
} mi_status;

typedef enum mi_target_kind {
    MI_TARGET_GAUSS_ISO = 1,    /* log K = -1/2 |theta|^2 */
    MI_TARGET_GAUSS_DIAG = 2,   /* log K = -1/2 sum_i prec_i theta_i^2;       prec: d values */
    MI_TARGET_GAUSS_DENSE = 3,  /* log K = -1/2 theta^T P theta;              prec: d*d, symmetric, row-major */
    MI_TARGET_LOGISTIC = 4,     /* log K = sum_r [y_r eta_r - log(1+e^eta_r)] - 1/2 |beta|^2, eta = X beta.  d <= 512 on the LDS-staged MFMA
                                 * kernels: hmc / mala / rwmh / nuts; hmc and nuts also with vals_bound; a diagonal precond_mat (hmc, mala, nuts)
                                 * or, unbounded, a dense one (hmc, mala).  d <= 8: one chain per lane, every setting.  Everything else
                                 * (the remaining combinations, rwmh with a cov_mat or bounds beyond d = 8): the literal kernels -- same bits.  d > 512, hmc / mala / rwmh without
                                 * bounds / precond_mat (round 6): the state in HBM, two fp64 matrix products per gradient for all chains (gemm_samplers.hip);
                                 * the same for MI_TARGET_GAUSS_DENSE beyond d = 512 (one product per gradient) */
    MI_TARGET_NORMAL_MODEL = 5  /* d = 2, vals = (mu, sigma), observations x_1..x_n in y[0..n_rows): the model of the reference's
                                 * example programs (/root/reference/examples/eigen/rmhmc_normal.cpp:44-106),
                                 * log K = -n (log(2 pi)/2 + log sigma) - sum_r (x_r - mu)^2 / (2 sigma^2); its metric tensor for
                                 * rmhmc is the Fisher information diag(n / sigma^2, 2 n / sigma^2).  hmc, mala, nuts, rwmh, rmhmc (any
                                 * precond_mat / cov_mat, any bounds). */
} mi_target_kind;

typedef enum mi_mem { MI_MEM_HOST = 0, MI_MEM_DEVICE = 1 } mi_mem;

/* Explicit kernel choice (mi_target.kernel_hint).  Every kernel that can serve a request produces the SAME bits, so a hint
 * changes speed only; a hint the request cannot honour is ignored.  The library reads no environment variable. */
typedef enum mi_kernel_hint {
    MI_KERNEL_AUTO = 0,
    MI_KERNEL_ELEMENTWISE_1LANE = 1,  /* hmc, separable Gaussian targets: one lane per chain (default for d > 128, many chains) */
    MI_KERNEL_ELEMENTWISE_4LANE = 2,  /* hmc, separable Gaussian targets: four lanes per chain */
    MI_KERNEL_NUTS_LOCKSTEP = 3,      /* nuts, unbounded Gaussian targets: the lock-step predecessor of the asynchronous kernel -- compiled into the
                                       * A/B library (`make prof`) only; the shipped library runs the tick-local kernel for this hint */
    /* hmc, dense-gradient Gaussian target, 64 < d <= 128, unbounded, identity precond_mat: the launch shape.  AUTO picks it from
     * the number of chains and of compute units (the shapes below are what makes 65 536 chains strong-scale over 8 GPUs). */
    MI_KERNEL_HMC_TWO_WAVES_PER_SIMD = 4,  /* 8 waves per workgroup, one 16-chain tile per wave: enough chains to fill the chip */
    MI_KERNEL_HMC_ONE_WAVE_PER_SIMD = 5,   /* 4 waves per workgroup, one tile per wave */
    MI_KERNEL_HMC_SPLIT2 = 6,              /* two waves share a tile (row halves of the mat-vec, theta exchanged through LDS) */
    MI_KERNEL_HMC_SPLIT4 = 7,              /* four waves share a tile, one wave per SIMD (16 chains per workgroup) */
    MI_KERNEL_HMC_SPLIT4_TWO_WAVES = 8,    /* four waves share a tile, two waves per SIMD (32 chains per workgroup) */
    MI_KERNEL_NUTS_TICK_LOCAL = 9,         /* nuts, unbounded Gaussian targets, identity precond_mat: the asynchronous kernel that executes every leaf and
                                            * reloads its start record (what the bounded / preconditioned variants run) instead of the default
                                            * one on the memoised trajectory -- an independent implementation of the same bits, for A/B runs */
    MI_KERNEL_NUTS_REG = 10,               /* RETIRED (round 5), valid and ignored: the register-carried kernel of rounds 2-4 (one wave per 16-chain tile) */
    MI_KERNEL_NUTS_SPLIT = 11,             /* RETIRED (round 5): the kernel that split every 16-chain tile over two waves (64 < d <= 128, few chains) is gone --
                                            * the memoised kernel is faster at every chain count.  The value stays valid and is ignored (as any hint a request
                                            * cannot honour): the default kernel runs */
    MI_KERNEL_LITERAL = 12,                /* nuts (and hmc with bounds / a diagonal precond_mat; hmc, mala and nuts with a DENSE precond_mat) on the logistic target
                                            * (d <= 512) and on dense Gaussians with 128 < d <= 512: the literal kernel (one workgroup per chain) instead of the tiled kernel on the
                                            * LDS-streamed evaluation -- same bits, for A/B timing */
    MI_KERNEL_NUTS_DYN = 13,               /* RETIRED (round 5), valid and ignored: round 4's register-carried tick with dynamic chain hand-out */
    MI_KERNEL_NUTS_MEMO = 14,              /* nuts, same case: every doubling on a MEMOISED trajectory (nuts_memo.hpp) -- the 2^j leaves of a doubling visit only
                                            * 1 + j (j + 1) / 2 distinct states (the reference's crossed edge plumbing, nuts.ipp:195,207), each is computed
                                            * once, the tree is walked on scalars; same bits, ~40 % fewer leapfrogs executed on BASELINE configs[3]; chains
                                            * handed out dynamically -- THE kernel of the plain case (AUTO) */
    MI_KERNEL_NUTS_MEMO_INTICK = 15        /* the same kernel generating every draw's momentum INSIDE the tick (rounds 2-5).  AUTO instead fills a table of
                                            * all momenta of the run with a pre-pass kernel at full occupancy (16 NT + 2 doubles per chain and draw;
                                            * nuts_memo.hpp: nuts_momenta_kernel) and falls back to this form when the table would not fit.  Same bits */
} mi_kernel_hint;

typedef struct mi_target {
    uint32_t      struct_size;
    int32_t       kind;        /* mi_target_kind */
    uint64_t      d;           /* n_vals of one chain (BMO_MATOPS_SIZE(initial_vals), src/hmc.cpp:40) */
    const double* prec;        /* see mi_target_kind */
    const double* X;           /* LOGISTIC: n_rows x d row-major */
    const double* y;           /* LOGISTIC: n_rows */
    uint64_t      n_rows;
    int32_t       mem;         /* mi_mem: where prec / X / y live */
    int32_t       kernel_hint; /* mi_kernel_hint, 0 = automatic */
} mi_target;

/* POD mirror of algo_settings_t restricted to what hmc / mala / nuts read.
 * Field names and defaults follow mcmc_structs.hpp; mi_settings_default() fills them. */
typedef struct mi_settings {
    uint32_t struct_size;
    int32_t  vals_bound;          /* algo_settings_t::vals_bound (mcmc_structs.hpp:159) */
    uint64_t rng_seed_value;      /* algo_settings_t::rng_seed_value (:155) -> Philox key */
    const double* lower_bounds;   /* d values or NULL (host memory) */
    const double* upper_bounds;
    uint64_t n_burnin_draws;      /* default 1000 (:68) */
    uint64_t n_keep_draws;        /* default 1000 (:69) */
    uint64_t n_leap_steps;        /* hmc_settings_t::n_leap_steps, default 1 (:73) */
    double   step_size;           /* default 1.0 (:74, :94, :130); nuts: epsilon_bar_0; rwmh: par_scale (:145) */
    const double* precond_mat;    /* d*d (host) or NULL = identity (src/hmc.cpp:57); rwmh: cov_mat (:146, src/rwmh.cpp:58) */
    uint64_t n_adapt_draws;       /* nuts, default 1000 (:89) */
    double   target_accept_rate;  /* nuts, default 0.55 (:90) */
    uint64_t max_tree_depth;      /* nuts, default 10 (:92) */
    double   gamma_val;           /* 0.05 (:95) */
    double   t0_val;              /* 10 (:96) */
    double   kappa_val;           /* 0.75 (:97) */
    uint64_t n_fp_steps;          /* rmhmc_settings_t::n_fp_steps, default 5 (:116) */
} mi_settings;

/* One shard of chains. All pointers live in `mem` (host or device). */
typedef struct mi_chains {
    uint32_t  struct_size;
    int32_t   mem;            /* mi_mem of theta / draws / n_accept / step_size / n_leapfrogs */
    uint64_t  n_chains;       /* C: chains in this call */
    uint64_t  chain0;         /* global index of local chain 0 */
    double*   theta;          /* in: initial_vals [d][C]; out: last state of every chain */
    double*   draws;          /* out [n_keep][d][C], may be NULL (draws discarded) */
    uint64_t* n_accept;       /* out [C], may be NULL */
    double*   step_size;      /* nuts out [C]: adapted step size per chain (in: see draw0), may be NULL; other samplers leave it unchanged */
    uint64_t* n_leapfrogs;    /* out [C]: leapfrog steps per chain AS THE REFERENCE EXECUTES THEM (0 for mala / rwmh), may be NULL.  nuts: one per leaf of
                               * every tree (nuts.ipp:132) -- see n_leapfrogs_executed */
    uint32_t* nuts_depth;     /* nuts out [n_burnin+n_keep][C]: tree depth reached per draw, may be NULL */
    uint64_t  draw0;          /* index of this call's first draw in every chain's random stream: 0 for a fresh run; the
                               * n_burnin+n_keep of the call(s) before to CONTINUE them from their final theta -- the
                               * concatenation is then bit-identical to one long run (checkpoint / resume, chunked output).
                               * nuts: a continuation takes the step sizes back in through step_size; n_adapt_draws must be the
                               * same in every call of one run (it is the RUN's window); a continuation that starts at
                               * draw0 <= n_adapt_draws (draw n_adapt_draws itself still uses the last dual-averaging step, not
                               * epsilon_bar) also needs nuts_adapt_state below. */
    double*   nuts_adapt_state; /* nuts, may be NULL: the dual-averaging state [3][C] (h, epsilon_bar, mu: src/nuts.cpp:174-176,294-302), written
                               * at the end of every call.  With it (and step_size) a run can be cut ANYWHERE: a continuation that starts
                               * inside or at the end of the adaptation window (0 < draw0 <= n_adapt_draws) reads it back and continues the
                               * adaptation on the run's own schedule (draw indices are global: draw0 + i); the concatenation is
                               * bit-identical to one long run.  Without it a continuation must start after the window, as before. */
    const double* mass_diag;  /* hmc only, may be NULL: PER-CHAIN diagonal mass matrices [d][C] (same memory space as theta) -- NOT a
                               * reference mode (the reference has one precond_mat per call, i.e. per chain: this is C calls of
                               * mcmc::hmc with precond_mat = diag(mass_diag[:, c]) in one launch).  settings.precond_mat must be
                               * NULL.  Separable Gaussian targets without bounds run on the elementwise kernels (any d), everything
                               * else on the literal kernels.  See mi_mcmc_hmc_run_mass_adapted_per_chain. */
    uint64_t* n_leapfrogs_executed; /* out [C], may be NULL (needs n_leapfrogs next to it): the leapfrog steps the device really computed.
                               * Equal to n_leapfrogs except for nuts on the MEMOISED ticks -- the default of the plain and diagonal-mass
                               * Gaussian case, of the built-in Gaussian with vals_bound, of nuts on tile targets and (round 6) of nuts on the
                               * LDS-streamed evaluation (logistic d <= 512, dense Gaussians 128 < d <= 512) -- which compute every
                               * distinct state of a doubling once (same draws, fewer steps).  (MI_KERNEL_NUTS_MEMO is accepted and
                               * equivalent to MI_KERNEL_AUTO: the memoised tick is what runs unless MI_KERNEL_NUTS_TICK_LOCAL or
                               * MI_KERNEL_NUTS_LOCKSTEP asks for the kernel that executes every leaf.) */
} mi_chains;

void        mi_settings_default(mi_settings* s);
const char* mi_mcmc_last_error(void);
/* Name of the kernel the calling thread's last mi_mcmc_*_run spent its time in, as rocprofv3 prints it (e.g.
 * "hmc_gauss_mfma_kernel<8, 8, false, false, false>"); "" before the first call.  Kernel choice is automatic (mi_target.kernel_hint
 * overrides it): this is how a measurement names what actually ran. */
const char* mi_mcmc_last_kernel(void);
int         mi_mcmc_version(void);
int         mi_mcmc_device_count(void);

/* User-defined device targets (the reference's callback contract, hmc.hpp:42-48, as a compile-time __device__ functor; see
 * include/mi_mcmc_target.hpp): the generic host driver of the one-chain-per-lane engine.  A target library built with
 * MI_MCMC_DEFINE_TARGET calls this with its own `launch` function, which instantiates the kernels for its target type;
 * this validates, stages the chains (host or device memory) and packs the launch parameters.  algo: 0 hmc, 1 mala, 2 nuts,
 * 3 rwmh, 4 rmhmc; d <= 8; target_pod is handed through to `launch`. */
typedef int (*mi_small_launch_fn)(int algo, const void* small_params, const void* target_pod, void* stream);
int mi_mcmc_run_user_target(int algo, uint64_t d, mi_small_launch_fn launch, const void* target_pod, uint64_t small_params_bytes,
                            const mi_settings* settings, mi_chains* chains, void* stream);
/* What MI_MCMC_DEFINE_TARGET calls: the same with the MI_MCMC_VERSION of the headers the target library was compiled against.  A
 * target library instantiates engine kernels from those headers; one built against another version is refused (MI_ERR_BAD_ARG)
 * instead of running kernels whose parameter struct may mean something else at the same size. */
int mi_mcmc_run_user_target_v(int algo, uint64_t d, mi_small_launch_fn launch, const void* target_pod, uint64_t small_params_bytes,
                              int header_version, const mi_settings* settings, mi_chains* chains, void* stream);
/* The same for TILE targets (include/mi_mcmc_tile_target.hpp): value + gradient for 16 chains at once in the MFMA register layout,
 * d <= 16 nt, hmc (algo 0) and mala (1) with the identity precond_mat and no bounds.  wpb: waves per workgroup the target's kernels
 * are built for (4 or 8); lds_bytes: what the target stages; header_version: MI_MCMC_VERSION of the headers the target library was
 * compiled against -- it instantiates engine kernels from them, so a library built against other headers is refused (MI_ERR_BAD_ARG). */
typedef int (*mi_tile_launch_fn)(int algo, const void* tile_params, const void* target_pod, uint64_t lds_bytes, void* stream);
int mi_mcmc_run_tile_target(int algo, uint64_t d, int nt, int wpb, uint64_t lds_bytes, mi_tile_launch_fn launch, const void* target_pod,
                            uint64_t tile_params_bytes, int header_version, const mi_settings* settings, mi_chains* chains, void* stream);

/* Kernel workspaces are cached per (device, stream) and reused by later calls on that stream (NUTS at BASELINE configs[3]
 * holds 4 GiB).  This frees the cache of the CURRENT device for `stream` (all_streams != 0: for every stream, e.g. before
 * destroying streams); it synchronises first.  bytes_freed may be NULL.  Safe to call concurrently with runs. */
int mi_mcmc_release_workspace(void* stream, int all_streams, uint64_t* bytes_freed);

/* Blocking calls. `stream` is a hipStream_t (NULL = default stream); with mem == MI_MEM_DEVICE the
 * kernels are enqueued on it and the call returns after enqueueing (asynchronous), so inputs can be
 * resident in HBM and timed with events.  With MI_MEM_HOST buffers are staged and the call blocks.
 * Also blocking with MI_MEM_DEVICE: every configuration that owns temporary device tables for the call -- settings.vals_bound or
 * settings.precond_mat (bounds / preconditioner tables), a target given in host memory, and the runs that go to the literal kernels
 * with such tables -- ends in a stream synchronisation before those tables are freed.  The plain configurations (no bounds, no
 * precond_mat, target in device memory: every BASELINE config) do return after enqueueing. */
int mi_mcmc_hmc_run (const mi_target* target, const mi_settings* settings, mi_chains* chains, void* stream);
int mi_mcmc_mala_run(const mi_target* target, const mi_settings* settings, mi_chains* chains, void* stream);
int mi_mcmc_nuts_run(const mi_target* target, const mi_settings* settings, mi_chains* chains, void* stream);
/* mcmc::rwmh (/root/reference/include/mcmc/rwmh.hpp, src/rwmh.cpp:30-175) for many chains: settings->step_size carries
 * rwmh_settings_t::par_scale (mcmc_structs.hpp:145) and settings->precond_mat carries rwmh_settings_t::cov_mat (:146). */
int mi_mcmc_rwmh_run(const mi_target* target, const mi_settings* settings, mi_chains* chains, void* stream);

/* mcmc::rmhmc (/root/reference/include/mcmc/rmhmc.hpp, src/rmhmc.cpp:30-287) for many chains.  The reference takes the metric
 * tensor as a second callback (tensor_fn); here it is the one tied to the target kind: NORMAL_MODEL (d = 2) its Fisher
 * information; LOGISTIC with d <= 4 the Fisher information X^T diag(s(1-s)) X plus the prior precision I; a user-defined
 * target (mi_mcmc_target.hpp) its own tensor().  Reads
 * n_leap_steps, step_size, n_fp_steps, vals_bound / bounds from the settings; there is no precond_mat in rmhmc. */
int mi_mcmc_rmhmc_run(const mi_target* target, const mi_settings* settings, mi_chains* chains, void* stream);

/* mcmc::hmc with a DIAGONAL mass matrix adapted during burn-in -- NOT a reference mode (kthohr/mcmc has no mass adaptation;
 * SURVEY 8 f-2 asks for it because an ill-conditioned target does not mix with precond_mat = I at any step size).  Many chains
 * make the estimate cheap: the mass is POOLED over the chains, M = diag(1 / var_c(theta_i)), one matrix for all of them, taken
 * from the spread of the chains' current states -- first from initial_vals, then again after each of `n_windows` equal parts
 * of the burn-in.  Each part is an ordinary mi_mcmc_hmc_run call with that diagonal precond_mat (bit-exact against the oracle
 * given the mass, tests/test_gpu_mass_adapt.py) chained through mi_chains.draw0, so the whole run is reproducible; the kept
 * draws use the last estimate, returned in mass_diag_out[d] (host, may be NULL).  step_size is in the preconditioned metric
 * (every dimension near unit scale).  Gaussian targets (any d) and the logistic-regression target; settings->precond_mat
 * must be NULL; a dimension whose pooled variance is 0 or not finite keeps mass 1. */
int mi_mcmc_hmc_run_mass_adapted(const mi_target* target, const mi_settings* settings, mi_chains* chains, uint32_t n_windows,
                                 double* mass_diag_out, void* stream);
/* The per-chain form SURVEY 8 f-2 words ("per-chain diagonal mass adaptation"; what Stan does for each of its chains): every chain
 * estimates ITS OWN diagonal mass from ITS OWN draws.  The burn-in is cut into n_windows + 1 equal parts; part 0 runs with M = I;
 * the draws of part k are kept in a scratch slab and give, per chain and dimension, the variance over the part's draws (two passes,
 * draws ascending), regularised as Stan does, var' = (n var + 5e-3) / (n + 5), mass = 1 / var' (1 where that is not finite or
 * not positive); part k + 1 and finally the kept draws run with those masses.  Each part is an ordinary mi_mcmc_hmc_run with
 * mi_chains.mass_diag, chained through draw0: given the masses, chain c is bit-identical to mcmc::hmc with
 * precond_mat = diag(mass[:, c]) (tests/test_gpu_mass_adapt.py).  mass_diag_out: [d][C] in the memory space of `chains`, may be
 * NULL.  settings.step_size is in the preconditioned metric (every dimension near unit scale) and drives every part that has
 * masses; part 0 (M = I on the raw target) runs with first_step_size (0: settings.step_size) -- the reference's hmc has no step-size
 * adaptation, so the two scales are the caller's to give.  chains.mass_diag and settings.precond_mat must be NULL; n_windows >= 1;
 * every part needs at least 3 draws.  With >= 10^3
 * chains on one target the pooled estimate above is the better one (each chain sees few draws); this form is for chains that do
 * not share a scale (different data per chain are not expressible through mi_target, so: for parity with per-chain adaptation
 * elsewhere). */
int mi_mcmc_hmc_run_mass_adapted_per_chain(const mi_target* target, const mi_settings* settings, mi_chains* chains, uint32_t n_windows,
                                           double first_step_size, double* mass_diag_out, void* stream);

/* Host-callback form of mcmc::hmc for ONE chain: the reference's own target contract
 * (std::function<fp_t(const ColVec_t& vals_inp, ColVec_t* grad_out, void* target_data)>, hmc.hpp:42-48)
 * flattened to a C function pointer. The callback runs on the host, called exactly where the
 * reference calls it (2 gradient calls per leapfrog step, 1 value call per draw, 1 at setup);
 * everything else of the draw loop runs on the GPU. draws_out: n_keep x d column-major
 * (element (i,j) at i + j*n_keep, as Eigen's Mat_t stores draws_out). */
typedef double (*mi_log_kernel_cb)(const double* vals_inp, double* grad_out /* NULL = value only */, void* target_data);
int mi_mcmc_hmc_run_callback(const double* initial_vals, uint64_t d, mi_log_kernel_cb target_log_kernel,
                             void* target_data, const mi_settings* settings, double* draws_out,
                             uint64_t* n_accept_draws);

/* The same for mcmc::mala (mala.hpp:66-73: 3 gradient + 1 value callback per draw, src/mala.cpp:149-186 with
 * include/mcmc/mala.ipp:58-64) and mcmc::nuts (nuts.hpp:65-72: two gradient callbacks per leaf of the recursive
 * nuts_build_tree, one value callback per leaf and per accepted proposal; dual averaging; step_size_out = final step size,
 * may be NULL).  One chain (global chain id 0); bit-identical to the device kernels run on the same target.  With settings.vals_bound
 * and / or settings.precond_mat these calls (and mi_mcmc_hmc_run_callback, mi_mcmc_rwmh_run_callback) run the literal kernel of the sampler
 * with the callback as its target: one device kernel that asks the host for every evaluation (see mi_mcmc_rmhmc_run_callback). */
int mi_mcmc_mala_run_callback(const double* initial_vals, uint64_t d, mi_log_kernel_cb target_log_kernel,
                              void* target_data, const mi_settings* settings, double* draws_out,
                              uint64_t* n_accept_draws);
int mi_mcmc_nuts_run_callback(const double* initial_vals, uint64_t d, mi_log_kernel_cb target_log_kernel,
                              void* target_data, const mi_settings* settings, double* draws_out,
                              uint64_t* n_accept_draws, double* step_size_out);
/* mcmc::rwmh (ref: include/mcmc/rwmh.hpp:42-47, src/rwmh.cpp:105-151): the callback is asked for the value only (grad_out is
 * always NULL), once per draw and once for the initial point; settings.step_size carries rwmh_settings.par_scale; identity
 * cov_mat, unbounded, one chain. */
int mi_mcmc_rwmh_run_callback(const double* initial_vals, uint64_t d, mi_log_kernel_cb target_log_kernel,
                              void* target_data, const mi_settings* settings, double* draws_out,
                              uint64_t* n_accept_draws);

/* mcmc::rmhmc with HOST callbacks (ref: include/mcmc/rmhmc.hpp:44-63, src/rmhmc.cpp:30-287): target_log_kernel as above and the metric
 * tensor tensor_fn(vals_inp, tensor_deriv_out, tensor_data) flattened: tensor_out is d*d row-major, tensor_deriv_out (NULL when not asked
 * for) holds the d matrices dG/dvals_i at + i*d*d.  One chain (global chain id 0), d <= 64, bounds as in settings.  The sampler runs in ONE
 * device kernel (the literal rmhmc of mi_mcmc_rmhmc_run); each callback is a request the kernel posts to a mailbox in pinned memory and
 * this call serves on the calling thread until the kernel ends -- the callbacks are called exactly where the reference calls them
 * (per fixed-point step: 1 gradient; per position fixed-point step: 1 tensor; per leapfrog step 1 tensor + derivative and 1 gradient more;
 * 1 value per draw).  A request the host does not answer within 60 s makes the kernel give up (MI_ERR_HIP), never hang. */
typedef void (*mi_tensor_cb)(const double* vals_inp, double* tensor_out, double* tensor_deriv_out /* NULL = tensor only */, void* tensor_data);
int mi_mcmc_rmhmc_run_callback(const double* initial_vals, uint64_t d, mi_log_kernel_cb target_log_kernel, void* target_data,
                               mi_tensor_cb tensor_fn, void* tensor_data, const mi_settings* settings, double* draws_out,
                               uint64_t* n_accept_draws);

/* ---- the BaseMatrixOps shim's INV and CHOL_LOWER as the engine computes them for a dense precond_mat (ref: src/hmc.cpp:58-59,
 * src/mala.cpp:58, include/stats/dmvnorm.hpp:36-41 through include/mcmc/mala.ipp:63-64): Gauss-Jordan with partial pivoting (first largest
 * magnitude) and column Cholesky, in the operation order the oracle states (oracle/mcmc_oracle.c: orc_inv, orc_chol_lower) -- bit-identical to
 * it.  A, Ainv, L: HOST memory, d x d row-major; A's lower triangle is what CHOL_LOWER reads, L is zero above the diagonal.  d >= 64 runs on
 * the device (one cooperative launch: d pivot / column steps of a parallel update, mcmc_amd/csrc/linalg_device.hip), smaller matrices on the
 * calling thread; no input validation, as in the reference (a singular / non-SPD matrix gives inf / NaN entries, not an error).  Blocking.
 * (ABI 0x000600.  Every sampler call with a dense precond_mat goes through these.) */
int mi_mcmc_mat_inverse(const double* A, uint64_t d, double* Ainv);
int mi_mcmc_mat_cholesky_lower(const double* A, uint64_t d, double* L);

/* ---- multi-GPU: one process per GPU.  Chains are independent (the reference runs one per call), so the path shards with no
 * data-path collective: rank r of world_size runs the global chains [chain0, chain0 + n_local) in its own mi_mcmc_*_run call
 * (mi_chains.chain0 = chain0; the Philox counter uses the global id, so the union of the shards is bit-identical to one call).
 * The one exchange the path has is the collation of draws_out. */
void mi_mcmc_shard_bounds(uint64_t n_chains_total, uint32_t world_size, uint32_t rank, uint64_t* chain0, uint64_t* n_local);
/* All-gather of the kept draws over xGMI: every rank contributes its slab local_draws [n_keep][d][n_local] and receives
 * all_draws [n_keep][d][n_chains_total] (both DEVICE memory).  rccl_comm is the caller's ncclComm_t (one rank per GPU; RCCL is
 * loaded on first use, MI_ERR_UNSUPPORTED if librccl.so is absent).  Equal shards (n_chains_total divisible by world_size: every
 * BASELINE split) are ONE ncclAllGather into a rank-major staging buffer (`scratch`, n_keep * d * n_chains_total doubles on the
 * device); ragged shards need no padding: one grouped broadcast per rank into the same buffer (_ragged, which the first form
 * falls back to).  A merge kernel then interleaves the chain axis.  Enqueued on `stream`; returns after enqueueing. */
int mi_mcmc_allgather_draws(void* rccl_comm, uint32_t world_size, uint32_t rank, const double* local_draws, uint64_t n_keep,
                            uint64_t d, uint64_t n_chains_total, double* scratch, double* all_draws, void* stream);
int mi_mcmc_allgather_draws_ragged(void* rccl_comm, uint32_t world_size, uint32_t rank, const double* local_draws, uint64_t n_keep,
                                   uint64_t d, uint64_t n_chains_total, double* scratch, double* all_draws, void* stream);
/* the merge step alone: rank-major shards [r][n_keep][d][n_local(r)] (device) -> [n_keep][d][n_chains_total] (device) */
int mi_mcmc_merge_shards(const double* rank_major, uint32_t world_size, uint64_t n_keep, uint64_t d, uint64_t n_chains_total,
                         double* all_draws, void* stream);

/* The same collation in SURVEY 8(e)'s own receive layout -- rank-major, all_rank_major [G][n_keep][d][C / G] for equal shards (every
 * BASELINE split): ONE ncclAllGather straight into the caller's buffer, no staging buffer and no merge kernel (half the memory of the
 * form above: at configs[4], n_keep = 8, 64 GiB + the 8 GiB local slab per GPU).  Ragged shards arrive packed: shard r at offset
 * n_keep * d * chain0(r) with its own row length n_local(r) (what mi_mcmc_merge_shards reads), as one grouped broadcast per rank.
 * mi_mcmc_rank_major_index gives the offset of (kept draw i, dimension j, GLOBAL chain c) in either case (~0 if c is out of range). */
int mi_mcmc_allgather_draws_rank_major(void* rccl_comm, uint32_t world_size, uint32_t rank, const double* local_draws, uint64_t n_keep,
                                       uint64_t d, uint64_t n_chains_total, double* all_rank_major, void* stream);
uint64_t mi_mcmc_rank_major_index(uint64_t n_chains_total, uint32_t world_size, uint64_t n_keep, uint64_t d, uint64_t i, uint64_t j, uint64_t c);
/* Collation OVERLAPPED with sampling ("or per kept-draw slab, overlapped with the next trajectory").  _begin takes the
 * slab of kept draws [row0, row0 + n_keep) of a run that keeps n_keep_total (local_draws [n_keep][d][n_local], produced on
 * producer_stream), orders a gather into the run's rank-major buffer behind that stream's work so far, on the library's own
 * communication stream, and returns at once: the caller goes on to enqueue the next chunk of the run (mi_chains.draw0 continues
 * it bit-identically) on producer_stream while the slab travels.  _wait makes consumer_stream wait for the gather (block_host != 0:
 * the calling thread instead) and releases the handle.  Gathers of one communicator must be begun in the same order on every rank.
 * LIFETIME: between _begin and the completion of _wait the gather READS local_draws and WRITES rows [row0, row0 + n_keep) of
 * all_rank_major on the library's stream -- the caller must not overwrite local_draws (give every chunk in flight its own slab, or _wait
 * with block_host before re-using one) nor touch those rows of all_rank_major on any other stream until then.  Nothing enforces it. */
typedef struct mi_collation mi_collation;
int mi_mcmc_allgather_draws_begin(void* rccl_comm, uint32_t world_size, uint32_t rank, const double* local_draws, uint64_t n_keep,
                                  uint64_t d, uint64_t n_chains_total, uint64_t row0, uint64_t n_keep_total, double* all_rank_major,
                                  void* producer_stream, mi_collation** handle);
int mi_mcmc_allgather_draws_wait(mi_collation* handle, void* consumer_stream, int block_host);

/* Layout converters between the engine's [n_keep][d][C] slabs and the reference's per-chain
 * draws_out (n_keep x d, column-major as Eigen stores it: element (i,j) at i + j*n_keep;
 * src/hmc.cpp:138,197). Host memory. */
int mi_mcmc_draws_to_chain_major(const double* draws_kdc, uint64_t n_keep, uint64_t d, uint64_t n_chains,
                                 double* out_c_colmajor /* [C][d][n_keep] */);
/* The same for slabs in HBM: both pointers are device memory, the transpose runs as a kernel on `stream` (LDS-tiled, coalesced
 * on both sides) and the call returns after enqueueing it. */
int mi_mcmc_draws_to_chain_major_device(const double* draws_kdc_dev, uint64_t n_keep, uint64_t d, uint64_t n_chains,
                                        double* out_c_colmajor_dev /* [C][d][n_keep] */, void* stream);


/* Reducers over a draws_out slab [n_keep][d][C] (in `mem`), on the device (the reference has no ESS / R-hat
 * code, definitions as in mcmc_amd/ess.py).  Outputs are HOST arrays, any of them may be NULL:
 *   mean [d]          pooled mean over draws and chains
 *   acov [n_keep][d]  autocovariance at lag k, pooled over chains, unbiased per lag (divided by n_keep - k)
 *   rhat [d]          Gelman-Rubin potential scale reduction over the C chains
 *   ess  [d]          per-chain effective sample size (Geyer's initial positive sequence on acov); the many-chain ESS is
 *                     C times it.
 * With acov == NULL only as many lags are computed as Geyer's sum needs (blocks of 16, then 32 lags straight from HBM; a slowly
 * mixing series falls through to the full computation).  With acov: every lag for n_keep <= 160; for longer series (the
 * reference's default is 1 000 kept draws) lags 0..127, the acov rows beyond hold NaN and Geyer's sum runs over the computed
 * lags.  Blocking. */
int mi_mcmc_draw_stats(const double* draws_kdc, int32_t mem, uint64_t n_keep, uint64_t d, uint64_t n_chains,
                       double* mean, double* acov, double* rhat, double* ess, void* stream);

/* (The diagnostics the GPU tests and the measurement tools use -- mi_probe_* -- are not part of this library: they live in
 * libmi_mcmc_probes.so, declared in mcmc_amd/csrc/mi_mcmc_probes.h.) */

#ifdef __cplusplus
}
#endif
#endif /* MI_MCMC_H */
