/*
 * krotov_hip.h -- C ABI of the MI355X Krotov engine (libkrotov_hip.so).
 *
 * The library replaces ONE path of qucontrol/krotov: the per-iteration
 * backward/forward time propagation and sequential pulse-update loop of
 * krotov.optimize_pulses (reference src/krotov/optimize.py:393-510), which the
 * reference dispatches per objective through parallel_map
 * (src/krotov/parallelization.py:233-495) to krotov.propagators.expm
 * (src/krotov/propagators.py:79-122).  The reference has no native FFI for
 * this path (it is pure Python); each entry point below names the Python call
 * site it stands in for.  INTEGRATION.md shows the ctypes stub a maintainer of
 * the reference would add.
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary: every call returns 0 on
 *     success or a negative kh_status; kh_last_error() gives the text.
 *   - complex128 = interleaved (re, im) doubles (numpy/torch layout).
 *   - operators: dense N x N, row-major.  States: length-N vectors; density
 *     matrices are column-stacked vec(rho) and operators Liouvillians
 *     (propagators.py:255-257, 306-307).
 *   - "dev" pointers are device (HBM) addresses owned by the caller (e.g.
 *     torch tensors); "host" pointers are ordinary host memory read during the
 *     call.  stream is a hipStream_t passed as void* (NULL = default stream).
 *     All sweeps are asynchronous on that stream.
 *   - not thread-safe per engine; one engine per device per stream.
 */
#ifndef KROTOV_HIP_H
#define KROTOV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kh_engine kh_engine;

typedef struct kh_cdouble {
    double re, im;
} kh_cdouble;

enum kh_status {
    KH_OK = 0,
    KH_ERR_INVALID = -1,     /* bad argument */
    KH_ERR_HIP = -2,         /* a HIP runtime call failed */
    KH_ERR_UNSUPPORTED = -3, /* size outside what the kernels handle */
    KH_ERR_TIMEOUT = -4,     /* in-kernel exchange gave up (see kh_check) */
    KH_ERR_NOMEM = -5
};

/* Problem description handed to kh_engine_create (all host memory, read once).
 *
 * ops[k*(1+L) + 0]   : drift operator of objective k   (sum of the non-list
 *                      entries of Objective.H)
 * ops[k*(1+L) + 1+l] : operator multiplying control l in objective k (sum over
 *                      every nested-list entry that carries control l,
 *                      mu.py:123-134), or NULL when the control does not occur
 *                      in objective k (mu.py:126-127).
 * Entries are DEVICE pointers; equal pointers mean a shared operator (the
 * engine stages each distinct operator, and its adjoint, once).
 */
typedef struct kh_problem {
    int32_t K;        /* objectives handled by this engine (this GPU's shard) */
    int32_t N;        /* state dimension */
    int32_t L;        /* controls: at most 32 (KH_ERR_UNSUPPORTED beyond); the register-resident kernel families
                         take up to 8 (4, 2 for some), with more the generic kernels run */
    int32_t nt;       /* len(tlist); nt-1 intervals */
    int32_t is_super; /* 0: Hilbert space, eqm factor -i (propagators.py:94);
                         1: Liouville space, factor 1 (propagators.py:96-98),
                            and mu = i * dL/d eps (mu.py:130-134) */
    int32_t reserved;
    const double *dt;              /* host [nt-1] interval lengths (may vary,
                                      optimize.py:450) */
    const kh_cdouble *const *ops;  /* host [K*(1+L)] of dev pointers */
    const double *op_norms;        /* host [K*(1+L)] upper bounds on the
                                      spectral norms, or NULL (engine then uses
                                      Frobenius norms: safe, slower) */
    double tol;       /* truncation tolerance of the exponential action per
                         step; 0 -> 2^-53 */
    double theta_max; /* largest ||A dt|| handled by one Taylor sub-step;
                         0 -> 1.0 */
} kh_problem;

/* Last error text of the calling thread ("" if none). */
const char *kh_last_error(void);

/* Library/kernels build info, e.g. "krotov_hip 0.1 gfx950". */
const char *kh_version(void);

/* Create an engine on the current HIP device.  Stages adjoint copies of every
 * distinct operator (the adjoint objectives of optimize.py:263,
 * objectives.py:240-258) and the exchange workspace, and reads the operators
 * once more to pick its algorithms: operators that equal their own adjoint bit
 * for bit (Hilbert space) get a shorter series for exp(-i H dt) than the Taylor
 * series every other generator gets, and control operators that equal +/- their
 * adjoint let the update sweep take <chi|dH/d eps|phi> on the co-state's side.
 * Results agree with the general path to rounding; the operator arrays must not
 * change while the engine lives. */
int kh_engine_create(const kh_problem *problem, kh_engine **out);
void kh_engine_destroy(kh_engine *engine);

/* Sparse operators (large Liouvillians with a few entries per row -- the regime of the
 * reference's DensityMatrixODEPropagator, propagators.py:162-327, which integrates
 * d/dt vec(rho) = L vec(rho) with a sparse L).  Same engine, same entry points below;
 * the exponential action is the same truncated Taylor series with CSR
 * matrix-vector products, so results agree with the dense path to round-off (the
 * reference's zvode integration is only accurate to its rtol = 1e-6).
 *   indptr [N+1], indices [nnz] int32, data [nnz] complex: device pointers;
 *   data == NULL: operator absent.  ops_adj[i] is the conjugate transpose of
 *   ops[i] (the adjoint objectives, objectives.py:240-258), supplied by the caller;
 *   op_norms (upper bounds on the spectral norms, e.g. sqrt(||A||_1 ||A||_inf))
 *   are required. */
typedef struct kh_csr {
    int64_t nnz;
    const int32_t *indptr;
    const int32_t *indices;
    const kh_cdouble *data;
} kh_csr;

typedef struct kh_problem_csr {
    int32_t K, N, L, nt, is_super, reserved;  /* as in kh_problem */
    const double *dt;                          /* host [nt-1] */
    const kh_csr *ops;                         /* host [K*(1+L)] */
    const kh_csr *ops_adj;                     /* host [K*(1+L)] */
    const double *op_norms;                    /* host [K*(1+L)] */
    double tol, theta_max;                     /* as in kh_problem */
} kh_problem_csr;

int kh_engine_create_csr(const kh_problem_csr *problem, kh_engine **out);

/* Which kernel family the engine selected: "tile64q2/512", "tile64/512", "tile64/256", "tile64/stream" (more objectives
 * than stay co-resident: one launch, the operators streamed; KH_NO_STREAM=1: "tile64/512 per interval"), "mini16/wave", "mini4/wave", "coop16/mfma", "tile128/512" (per-objective operators, 64 < N <= 128), "ell/csr"
 * (sparse operators with the matrix in registers), "generic" or "generic/csr". */
const char *kh_engine_kernel(const kh_engine *engine);

/* Forward propagation of K states over the whole grid under fixed pulses.
 * Replaces parallel_map[0](_forward_propagation, ...) (optimize.py:302-313,
 * 806-846).
 *   pulses_dev   [L][nt-1] doubles
 *   init_dev     [K][N]
 *   states_dev   [K][nt][N] all stored states (index 0 = init), or NULL to
 *                store nothing (first-order Krotov discards them, :329)
 *   psi_T_dev    [K][N] final states
 */
int kh_forward_store(kh_engine *engine, const double *pulses_dev,
                     const kh_cdouble *init_dev, kh_cdouble *states_dev,
                     kh_cdouble *psi_T_dev, void *stream);

/* Backward propagation of the (normalised) co-states under the guess pulses,
 * storing chi_k(t_n) for every n.  Replaces parallel_map[1](
 * _backward_propagation, ...) (optimize.py:413-425, 849-886): adjoint
 * operators, conjugated (real) pulse values, backwards=True.
 *   chi_T_dev    [K][N]
 *   chi_store_dev[K][nt][N]; [:, nt-1] = chi_T
 */
int kh_backward_store(kh_engine *engine, const kh_cdouble *chi_T_dev,
                      const double *pulses_dev, kh_cdouble *chi_store_dev,
                      void *stream);

/* Forward sweep with the sequential pulse update, whole grid, one launch.
 * Replaces the time loop optimize.py:444-501 (mu() / overlap() per
 * (step, pulse, objective), the cross-objective sum at :470, the update at
 * :471-477 and parallel_map[2](_forward_propagation_step, ...) at :479-491).
 *   chi_store_dev [K][nt][N] from kh_backward_store
 *   chi_norms_dev [K]       norms taken out of chi before the backward sweep
 *   init_dev      [K][N]
 *   guess_dev     [L][nt-1]
 *   shape_dev     [L][nt-1] update shapes S_l on the intervals
 *   lambda_dev    [L]
 *   opt_dev       [L][nt-1] OUT optimized pulses
 *   psi_T_dev     [K][N]    OUT
 *   g_a_dev       [L]       OUT integrals of g_a (optimize.py:475)
 * The cross-objective sum is evaluated in a fixed order (bitwise repeatable).
 */
int kh_forward_update(kh_engine *engine, const kh_cdouble *chi_store_dev,
                      const double *chi_norms_dev, const kh_cdouble *init_dev,
                      const double *guess_dev, const double *shape_dev,
                      const double *lambda_dev, double *opt_dev,
                      kh_cdouble *psi_T_dev, double *g_a_dev, void *stream);

/* The single-launch update sweep on at most `max_workgroups` workgroups (0: back to the engine's own choice).  What a
 * caller does after kh_check returned KH_ERR_TIMEOUT -- co-tenants held compute units, the sweep's workgroups were not
 * all resident at once -- before it falls back to one launch per interval: the register-tile families (N <= 64, 1..4
 * controls) then run kh_stream_forward_update (every workgroup walks through several objectives per interval, about
 * 2x the time at half the workgroups instead of 10x), ensembles run the matrix-core kernel with more objectives per
 * workgroup.  KH_ERR_UNSUPPORTED for the other families, for sharded engines and below the smallest grid the kernels
 * take (16 objectives per workgroup).  *chosen (may be NULL): the grid the next sweep will use (with 0: the engine's own).  No counterpart in the
 * reference (its objectives are separate processes: parallelization.py:233-311). */
int kh_set_update_workgroups(kh_engine *engine, int32_t max_workgroups, int32_t *chosen);

/* Second-order Krotov update (optimize.py:434-443, 468-469, 492-500): the
 * following update sweeps add 0.5 sigma_n <phi_k(t_n) - fw_prev[k][n] | mu |
 * phi_k(t_n)> to every summand of the pulse update and store the propagated
 * states.
 *   fw_prev_dev  [K][nt][N] states propagated under the guess pulses
 *                (forward_states0; iteration 0: kh_forward_store's states)
 *   fw_store_dev [K][nt][N] OUT states under the optimized pulses
 *   sigma_dev    [nt-1] sigma(t) at the interval mid-points (optimize.py:452)
 * Pass three NULLs to return to the first-order update. */
int kh_set_second_order(kh_engine *engine, const kh_cdouble *fw_prev_dev,
                        kh_cdouble *fw_store_dev, const double *sigma_dev);

/* The same sweep cut at the cross-objective sum, for objectives sharded over
 * several GPUs: the caller all-reduces `partial` (L doubles, Im parts) across
 * ranks between two calls (RCCL over xGMI).
 *
 *   kh_update_begin : phi <- init, opt <- guess, g_a <- 0, and the local
 *                     partial sums of interval 0 -> partial_dev[L]
 *   kh_update_step  : given the all-reduced sums D of interval n: update
 *                     eps[n] (optimize.py:471-477), propagate every local phi
 *                     over interval n (:479-491) and emit the local partial
 *                     sums of interval n+1 -> partial_dev[L] (untouched after
 *                     the last interval)
 *   kh_update_end   : psi_T <- phi
 */
int kh_update_begin(kh_engine *engine, const kh_cdouble *chi_store_dev,
                    const double *chi_norms_dev, const kh_cdouble *init_dev,
                    const double *guess_dev, double *opt_dev, double *g_a_dev,
                    double *partial_dev, void *stream);
int kh_update_step(kh_engine *engine, int32_t n, const double *D_dev,
                   const kh_cdouble *chi_store_dev, const double *chi_norms_dev,
                   const double *shape_dev, const double *lambda_dev,
                   double *opt_dev, double *g_a_dev, double *partial_dev,
                   void *stream);
/* kh_update_step with the interval index kept in device memory (*n_dev is read
 * by the kernels and incremented after each call): consecutive calls are then
 * byte-identical launches, so a block of intervals -- including the caller's
 * RCCL all-reduce between them -- can be captured once in a HIP graph and
 * replayed, which removes the per-interval host launch cost of the sharded
 * sweep.  Calls past the last interval are no-ops. */
int kh_update_step_dev(kh_engine *engine, int32_t *n_dev, const double *D_dev,
                       const kh_cdouble *chi_store_dev, const double *chi_norms_dev,
                       const double *shape_dev, const double *lambda_dev,
                       double *opt_dev, double *g_a_dev, double *partial_dev,
                       void *stream);
int kh_update_end(kh_engine *engine, kh_cdouble *psi_T_dev, void *stream);

/* Device-side exchange across the GPUs of one node (objectives sharded over
 * `world` ranks, one per GPU): with it, kh_forward_update stays ONE persistent
 * launch per rank -- after the in-GPU stage the per-GPU sums of every interval
 * are exchanged through peer-mapped windows (xGMI) with system-scope atomics,
 * summed in rank order (bit-identical on every GPU), instead of one RCCL
 * all-reduce plus kernel launches per interval.
 *
 *   kh_p2p_create_window : allocate this rank's window (fine-grained device
 *                          memory) and return its 64-byte IPC handle
 *   kh_p2p_open_peers    : map all ranks' windows from the all-gathered
 *                          handles ([world][64] bytes, rank order)
 *   kh_p2p_selftest      : collective; runs `rounds` in-kernel exchanges and
 *                          verifies the totals.  Only after it succeeded on
 *                          every rank (the caller checks that) does
 *                          kh_forward_update use the cross-GPU stage.
 *   kh_p2p_disable       : fall back to the per-interval path (kh_update_*)
 * Every rank must call kh_forward_update the same number of times (epochs
 * advance in lock step) and synchronise with the others between sweeps (the
 * per-iteration all-gather of tau does). */
int kh_p2p_create_window(kh_engine *engine, int32_t world, int32_t rank,
                         unsigned char *ipc_handle_out /* [64] */);
int kh_p2p_open_peers(kh_engine *engine, const unsigned char *all_handles);
int kh_p2p_selftest(kh_engine *engine, int32_t rounds, void *stream);
int kh_p2p_disable(kh_engine *engine);

/* Diagnostics of the cross-GPU exchange (no counterpart in the reference), for the first run on a real multi-GPU node:
 *   out[0]  us per interval workgroup 0 of this rank waited for its own GPU's workgroups in the last sharded
 *           single-launch update sweep (in-GPU gather),
 *   out[1]  us per interval between publishing this GPU's sum into the peers' windows and holding every rank's sum
 *           (the cross-GPU hop: xGMI stores + the slowest rank's lag),
 *   out[2]  us per round of the last kh_p2p_selftest (publish -> all ranks' values read back, first round excluded),
 *   out[3]  ranks.
 * Kernels that run their own cross-GPU stage (one-wave, N <= 128 register-generator and sparse families) report 0 in
 * out[0], out[1]. */
int kh_p2p_stats(kh_engine *engine, double out[4]);

/* tau_k = <target_k | psi_k(T)> (optimize.py:316-322, 502-508;
 * second_order.py:69-83).  targets_dev, psi_T_dev [K][N]; tau_dev [K]. */
int kh_tau(kh_engine *engine, const kh_cdouble *targets_dev,
           const kh_cdouble *psi_T_dev, kh_cdouble *tau_dev, void *stream);

/* Boundary co-states of the built-in functionals, normalised for the backward
 * sweep: v_k = c_k target_k + d_k psi_k(T), chi_T[k] = v_k / ||v_k||_2,
 * chi_norms[k] = ||v_k||_2.  Replaces the chi_constructor call and the
 * normalisation of optimize.py:396-410 for krotov.functionals.chis_re /
 * chis_ss / chis_sm / chis_hs (functionals.py:177-197, 225-253, 293-317,
 * 389-437), whose per-objective scalars (c_k, d_k) the host derives from
 * tau and the objective weights: re (w/2K, 0), ss (tau_k w/K, 0),
 * sm (w/K^2 sum_j w_j tau_j, 0), hs (w/2K, -w/2K).  All arrays on the device:
 * targets, psi_T, chi_T [K][N]; c, d [K] complex; chi_norms [K]. */
int kh_chi_boundary(kh_engine *engine, const kh_cdouble *targets_dev,
                    const kh_cdouble *psi_T_dev, const kh_cdouble *c_dev,
                    const kh_cdouble *d_dev, kh_cdouble *chi_T_dev,
                    double *chi_norms_dev, void *stream);

/* After synchronising the stream: returns KH_ERR_TIMEOUT if an in-kernel
 * exchange gave up since the last call (outputs are then invalid), else 0. */
int kh_check(kh_engine *engine);

/* Statistics of the last sweep launched (for the roofline accounting):
 * stats[0] = Taylor matrix-vector products issued per objective, summed over
 * the grid (host-side estimate from the pulses is not possible for the update
 * sweep, so kernels count them), stats[1] = intervals, stats[2] = workgroups. */
int kh_last_stats(kh_engine *engine, double stats[4]);

/* The coefficient tables the kernels evaluate exp(f A dt) v with (host only, no GPU needed; for inspection
 * and tests).  The series is sum_j c_j (f A dt)^j v with real c_j; for every degree m <= 64:
 *   theta[m]            largest ||A dt|| the degree serves at tolerance `tol` (tol <= 0: 2^-53),
 *   ratios[m*65 + 0]    c_0,   ratios[m*65 + 1] = c_1,   ratios[m*65 + j] = c_j / c_{j-1}  (j = 2..m).
 * real_spectrum = 0: Taylor (c_j = 1/j!, any generator; replaces SciPy's Pade expm behind
 * krotov.propagators.expm, propagators.py:100-117).  real_spectrum = 1: what an engine uses when every
 * operator equals its own adjoint and f = -+i: the truncated Chebyshev series of exp(-+i theta x) on [-1, 1]
 * in powers of (-+i theta x), even degrees only, theta <= 2 (Taylor beyond). */
int kh_series_tables(int32_t real_spectrum, double tol, double *theta /* [65] */, double *ratios /* [65*65] */);

/* The same tables of the Chebyshev form as an engine builds them for a generator f A dt that is anti-Hermitian only
 * up to a Hermitian part of norm <= `defect` (a weakly damped Liouvillian, a Hamiltonian with a small anti-Hermitian
 * part; the engine measures the defect of the drift, the controls must be exactly (anti-)self-adjoint), valid up to
 * theta <= `theta_cap` (2 for the register-tile kernels, 4 for the cooperative ones, 6 for the sparse ones; at most 8;
 * Taylor beyond).  The truncation
 * error is bounded on the numerical range, (1 + sqrt 2) 2 sum_{k>m} |J_k(theta)| cosh(k asinh(defect / theta)).
 * defect = 0 and theta_cap = 2 give kh_series_tables(1, ...).  Host only. */
int kh_series_tables_defect(double tol, double theta_cap, double defect, double *theta /* [65] */,
                            double *ratios /* [65*65] */);

/* The padded row form the sparse kernels keep in registers (host only, no GPU needed; for inspection and tests): the
 * union of the patterns of an operator list (drift + controls; data == NULL: absent; HOST arrays here), entries some
 * control touches in the first Ec slots of every row, every row padded to E entries (multiples of four).
 *   off  [E][S]          byte offset (16 x column) of every entry's vector element; padding: the row itself
 *   vals [n_ops][E][S]   values of every operator on the union pattern (0 where it has no entry)
 * with S = kh_ell_rows_of(N) padded rows: the threads x rows per lane of the kernel instantiation that serves N (512,
 * 768, 1024, 1536 or 2048; 0: N > 2048).  off / vals may be NULL (sizes only); E_cap: entry slots the caller's arrays can
 * hold.  KH_ERR_UNSUPPORTED when N > 2048 or a row is wider than 32 entries (16 for N > 512, 8 for N > 1024): such
 * engines run the STREAMED form of the same kernels (kh_engine_kernel: "ellstream/csr"; N <= 4096, rows up to 32 entries:
 * the same arrays with S = N rounded up to 64, read from memory in every term instead of living in registers), and
 * the generic CSR kernels beyond that (N <= 2540: their vectors must fit LDS). */
int32_t kh_ell_rows_of(int32_t N);
int kh_ell_layout(int32_t N, int32_t n_ops, const kh_csr *ops_host, int32_t *E, int32_t *Ec, int32_t *off,
                  kh_cdouble *vals, int32_t E_cap);

/* Test hook (no counterpart in the reference): keep `workgroups` CUs busy for `milliseconds` on `stream` with a
 * kernel that does nothing but hold a CU's LDS -- the situation the single-launch update sweep must survive
 * (another stream of the process holding compute units while its workgroups need to be resident all at once).
 * tests/test_hip_parity.py::test_update_sweep_next_to_a_busy_stream.  The kernel leaves all-ones bit patterns (NaN as
 * doubles) in the 128 KiB of LDS it held: test_kernels_do_not_read_uninitialised_lds runs the sweeps right behind it. */
int kh_debug_occupy(kh_engine *engine, int32_t workgroups, double milliseconds, void *stream);

/* Test hook (no counterpart in the reference): the sweep-kernel template instantiations of the library, one name per
 * line ("kh_q2_forward_update<false, true, true>").  which = 1: every instantiation some dispatch can select (host
 * only, no GPU needed); which = 0: those this process has launched so far; which = 2: forget that list (returns 0).  Writes at most cap - 1 characters and a
 * terminating 0 into buf (may be NULL); returns the buffer size the whole list needs.  With the environment variable
 * KH_LAUNCH_LOG=<file> every process also appends an instantiation's name to that file at its first launch:
 * tests/test_zz_kernel_coverage.py fails when an instantiation was never launched by an oracle-comparing test. */
int kh_debug_launched(int32_t which, char *buf, int32_t cap);

#ifdef __cplusplus
}
#endif
#endif /* KROTOV_HIP_H */
