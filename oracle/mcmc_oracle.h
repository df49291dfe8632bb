/*
 * oracle/mcmc_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C11) of the hot path of kthohr/mcmc (MCMCLib 2.1.0):
 *   mcmc::internal::hmc_impl   /root/reference/src/hmc.cpp:30-227
 *   mcmc::internal::mala_impl  /root/reference/src/mala.cpp:30-208
 *       + mala_prop_adjustment /root/reference/include/mcmc/mala.ipp:30-70
 *       + stats_mcmc::dmvnorm  /root/reference/include/stats/dmvnorm.hpp:28-54
 *   mcmc::internal::nuts_impl  /root/reference/src/nuts.cpp:30-332
 *       + nuts_find_initial_step_size / nuts_build_tree
 *                              /root/reference/include/mcmc/nuts.ipp:30-241
 *   mcmc::internal::rmhmc_impl /root/reference/src/rmhmc.cpp:30-287
 *   mcmc::internal::rwmh_impl  /root/reference/src/rwmh.cpp:30-175
 *   box-constraint helpers     /root/reference/include/misc/{determine_bounds_type,
 *                              transform_vals,log_jacobian,inv_jacobian_adjust}.hpp
 *
 * PARITY UNPINNED.  The reference ships no tests, golden vectors or known-answer
 * fixtures for this path, and it cannot be built in this image: its linear
 * algebra and RNG live in Eigen and in the un-vendored kthohr/BaseMatrixOps
 * submodule (include/BaseMatrixOps/ is empty; .gitmodules:1-3, pinned commit
 * unknown), neither of which is installed.  This restatement therefore pins the
 * reference's control flow, formulas, call order and RNG consumption order by
 * line-by-line reading only; the reduction orders and the random-number
 * generator are stated here (see orc_math.h, orc_dot) because the originals are
 * not recoverable.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything in this directory.
 */
#ifndef MCMC_ORACLE_H
#define MCMC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* user callback contract of the reference (hmc.hpp:42-48): returns the log
 * kernel; fills grad_out (length d) when it is not NULL. */
typedef double (*orc_kernel_fn)(const double* vals, double* grad_out, void* data);

/* built-in targets (ours; the reference has none).  data points to orc_target. */
enum { ORC_TARGET_GAUSS_ISO = 1, ORC_TARGET_GAUSS_DIAG = 2, ORC_TARGET_GAUSS_DENSE = 3,
       ORC_TARGET_LOGISTIC = 4,
       ORC_TARGET_NORMAL_MODEL = 5 /* d = 2, vals = (mu, sigma), data x_1..x_n in y[0..n_rows): the model of the reference's own
                                      example programs (ref: examples/eigen/{hmc,mala,nuts,rmhmc}_normal.cpp):
                                      log K = -n (log(2 pi)/2 + log sigma) - sum (x - mu)^2 / (2 sigma^2) */ };

typedef struct orc_target {
    int           kind;
    size_t        d;
    const double* prec;      /* DENSE: d*d row-major precision; DIAG: d precisions; ISO: NULL */
    const double* X;         /* LOGISTIC: n_rows*d row-major design matrix */
    const double* y;         /* LOGISTIC: n_rows labels in {0,1} */
    size_t        n_rows;
    int           reduce_width;  /* W of orc_dot used inside the target (see below) */
    int           reduce_blocks; /* >1: dimension-blocked reductions (see orc_dot_b), block size reduce_block_size */
    size_t        reduce_block_size;
    int           eta_chains;    /* LOGISTIC: the eta fma chain of every dimension block is cut into this many contiguous
                                    sub-chains of reduce_block_size/eta_chains dims, summed left to right (<=1: one chain) */
    uint64_t      n_grad_calls;  /* instrumentation */
    uint64_t      n_value_calls;
    const double* prec_t;        /* Mode B only (see orc_settings.work_mode): DENSE: the TRANSPOSE of prec, row-major; when set the
                                    mat-vec runs in axpy form (y_i = fma(prec_t[k][i], x_k, y_i), k ascending): per row the same
                                    sequential fma chain as orc_gemv, hence the same bits, but SIMD across rows */
} orc_target;

double orc_target_kernel(const double* vals, double* grad_out, void* data);

/* metric tensor contract of the reference (rmhmc.hpp: std::function<Mat_t (vals_inp, Cube_t* tensor_deriv_out, tensor_data)>):
 * writes the d x d tensor (row-major) and, when deriv_out is not NULL, the d matrices dG/dvals_i (deriv_out + i*d*d). */
typedef void (*orc_tensor_fn)(const double* vals, double* tensor_out, double* deriv_out, void* data);
/* built-in tensors (data points to orc_target): NORMAL_MODEL: the Fisher information diag(n/sigma^2, 2n/sigma^2) and its
 * derivative (ref: examples/eigen/rmhmc_normal.cpp:75-106); GAUSS_*: the constant precision, zero derivative; LOGISTIC: the
 * Fisher information X^T diag(s (1 - s)) X plus the prior precision I, and its derivative (ours: the reference has none). */
void orc_target_tensor(const double* vals, double* tensor_out, double* deriv_out, void* data);

/* POD mirror of algo_settings_t (mcmc_structs.hpp:151-184) restricted to the
 * fields hmc/mala/nuts read. */
typedef struct orc_settings {
    uint64_t rng_seed_value;
    int      vals_bound;
    const double* lower_bounds;   /* d, may be NULL if !vals_bound */
    const double* upper_bounds;
    size_t   n_burnin_draws;
    size_t   n_keep_draws;
    size_t   n_leap_steps;        /* hmc */
    double   step_size;           /* hmc, mala; nuts: epsilon_bar_0 */
    const double* precond_mat;    /* d*d column- or row-major (symmetric), NULL -> identity */
    /* nuts (mcmc_structs.hpp:82-101) */
    size_t   n_adapt_draws;
    double   target_accept_rate;
    size_t   max_tree_depth;
    double   gamma_val, t0_val, kappa_val;
    size_t   n_fp_steps;          /* rmhmc: fixed-point iterations (mcmc_structs.hpp:116, default 5) */
    /* oracle-only knobs */
    int      reduce_width;        /* W: number of strided partial sums in dot products (1,4,64,...) */
    int      reduce_blocks;       /* >1: dot products are ((B0+B1)+B2)+... over contiguous dimension blocks of
                                     reduce_block_size, each block an orc_dot of width W (the order of a kernel that
                                     splits the dimensions of a chain over several wavefronts) */
    size_t   reduce_block_size;
    int      hoist_factorizations;/* mala: 0 = factorise eps^2 M inside every dmvnorm call as the
                                     reference does (mala.ipp:63-64); 1 = once (same bits) */
    uint64_t chain_id;            /* Philox counter word: global chain index */
    int      work_mode;           /* CPU-baseline work profile, hmc only; the draws are the same bits:
                                     0 = Mode A "reference-faithful": two gradient callbacks per leapfrog step plus one
                                         value callback per draw (ref: src/hmc.cpp:167,175,178), dense M^-1 / chol(M)
                                         mat-vecs even for the identity (:57-59,158-160,171,184), buffers allocated per call;
                                     1 = Mode B "optimised CPU": the gradient at the end of a leapfrog step is reused as the
                                         start of the next, the value of the last gradient call is the draw's value call,
                                         no mat-vec with an identity precond_mat, no allocation inside the draw loop;
                                         unbounded runs only (a bounded run falls back to Mode A) */
} orc_settings;

typedef struct orc_stats {
    size_t   n_accept_draws;      /* post-burn-in accepts (hmc.cpp:198) */
    uint64_t n_leapfrogs;         /* executed leapfrog steps */
    double   final_step_size;     /* nuts */
    /* optional per-draw traces (n_burnin+n_keep entries) if non-NULL */
    uint8_t* accept_trace;
    uint32_t* depth_trace;        /* nuts: tree_depth reached */
    uint32_t* leap_trace;         /* nuts: leapfrogs executed in the draw */
    double*  eps_trace;           /* nuts: step size used by the draw */
} orc_stats;

/* draws_out: n_keep x d, element (i,j) at draws_out[i*d + j] (row per draw). */
int orc_hmc (const double* initial_vals, size_t d, orc_kernel_fn kernel, void* data,
             const orc_settings* s, double* draws_out, orc_stats* st);
double orc_mala_prop_adjustment_eval(orc_target* t, const orc_settings* s, const double* prop_vals, const double* prev_vals);   /* test hook */
int orc_mala(const double* initial_vals, size_t d, orc_kernel_fn kernel, void* data,
             const orc_settings* s, double* draws_out, orc_stats* st);
int orc_nuts(const double* initial_vals, size_t d, orc_kernel_fn kernel, void* data,
             const orc_settings* s, double* draws_out, orc_stats* st);
/* the same run with every doubling evaluated on a memoised trajectory (mcmc_oracle.c: nuts_doubling_memo): identical outputs, the reference's
 * leapfrog count in st->n_leapfrogs, the leap_frog calls really made in *n_exec_out (may be NULL).  max_tree_depth > 10: the recursion. */
int orc_nuts_memo(const double* initial_vals, size_t d, orc_kernel_fn kernel, void* data,
                  const orc_settings* s, double* draws_out, orc_stats* st, uint64_t* n_exec_out);
/* ... and across the doublings of a draw (same direction, no accepted proposal in between): the plain-case kernel's count since round 6 */
int orc_nuts_memo_xd(const double* initial_vals, size_t d, orc_kernel_fn kernel, void* data,
                  const orc_settings* s, double* draws_out, orc_stats* st, uint64_t* n_exec_out);
/* mcmc::rwmh (src/rwmh.cpp:30-175): step_size carries par_scale, precond_mat carries cov_mat */
int orc_rwmh(const double* initial_vals, size_t d, orc_kernel_fn kernel, void* data,
             const orc_settings* s, double* draws_out, orc_stats* st);

/* mcmc::rmhmc (src/rmhmc.cpp:30-287).  Reductions here are plain sequential chains (the reduce_* knobs are not used). */
int orc_rmhmc(const double* initial_vals, size_t d, orc_kernel_fn kernel, orc_tensor_fn tensor, void* data, void* tensor_data,
              const orc_settings* s, double* draws_out, orc_stats* st);

/* many independent chains of a built-in target, OpenMP over chains (the CPU
 * baseline of BASELINE.md section 3).  init: n_chains x d (row per chain).
 * draws_out: [n_keep][d][n_chains] (the engine's device layout) or NULL.
 * algo: 0 hmc, 1 mala, 2 nuts, 3 rwmh, 4 rmhmc, 5 nuts through orc_nuts_memo.  Chain c uses chain_id = chain0 + c. */
int orc_run_many(int algo, const orc_target* tgt, const orc_settings* s, size_t n_chains,
                 uint64_t chain0, const double* init, double* draws_out,
                 uint64_t* n_accept_out, uint64_t* n_leap_out, double* eps_out, int n_threads);

/* exported pieces for unit tests */
double orc_dot(const double* x, const double* y, size_t d, int W);
double orc_dot_b(const double* x, const double* y, size_t d, int W, int nblk, size_t bs);
void   orc_gemv(const double* A /*row-major d x d*/, const double* x, size_t d, double* y);
int    orc_inv(const double* A, size_t d, double* Ainv);
int    orc_chol_lower(const double* A, size_t d, double* L);
double orc_dmvnorm_log(const double* x, const double* mu, const double* Sigma, size_t d, int W);
void   orc_math_eval(int fn, const double* x, size_t n, double* out, double* out2);
void   orc_philox_eval(const uint32_t* ctr, const uint32_t* key, uint32_t* out);
void   orc_normal_vec(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t stream, size_t d, double* out);
double orc_uniform(uint64_t seed, uint64_t chain, uint32_t draw, uint32_t slot);
void   orc_transform(const double* vals, const int* btype, const double* lb, const double* ub, size_t d, double* out);
void   orc_inv_transform(const double* vals, const int* btype, const double* lb, const double* ub, size_t d, double* out);
double orc_log_jacobian(const double* vals, const int* btype, const double* lb, const double* ub, size_t d);
void   orc_determine_bounds_type(int vals_bound, size_t d, const double* lb, const double* ub, int* out);

#ifdef __cplusplus
}
#endif
#endif
