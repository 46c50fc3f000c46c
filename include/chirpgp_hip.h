/*
 * chirpgp_hip.h -- C-ABI of libchirpgp_hip.so, the MI355X (gfx950) batched Kalman / RTS engine.
 *
 * Drop-in boundary for the hot path of spdes/chirpgp, chirpgp/filters_smoothers.py.  The reference is a
 * pure-Python/JAX module with no FFI of its own, so these entry points are what a binding
 * for that module would bind; each one names the reference functions it replaces:
 *
 *   cgp_filter   (method = CGP_F_EKF,     model = CGP_M_LINEAR)        kf          filters_smoothers.py:145-184
 *   cgp_filter   (method = CGP_F_EKF,     model = *_LCD)               ekf         filters_smoothers.py:222-264
 *   cgp_filter   (method = CGP_F_EKF_KPT, model = CGP_M_KPT)           ekf_for_kpt filters_smoothers.py:267-314
 *   cgp_filter   (method = CGP_F_SGP)                                  sgp_filter  filters_smoothers.py:446-490
 *   cgp_filter   (method = CGP_F_CD_EKF,  model = *_SDE)               cd_ekf      filters_smoothers.py:352-397
 *   cgp_filter   (method = CGP_F_CD_SGP,  model = *_SDE)               cd_sgp_filter   filters_smoothers.py:534-582
 *   cgp_smoother (method = CGP_S_EKS,     model = CGP_M_LINEAR)        rts         filters_smoothers.py:187-219
 *   cgp_smoother (method = CGP_S_EKS,     model = *_LCD)               eks         filters_smoothers.py:317-349
 *   cgp_smoother (method = CGP_S_SGP)                                  sgp_smoother    filters_smoothers.py:493-531
 *   cgp_smoother (method = CGP_S_CD_EKS)                               cd_eks      filters_smoothers.py:400-443
 *   cgp_smoother (method = CGP_S_CD_SGP)                               cd_sgp_smoother filters_smoothers.py:585-632
 *   cgp_smoother_select                                                the smoothers above + the marginal step behind them in every driver
 *                                                                      (demos/ekfs_mle.py:69-77: mss[:, k], Pss[:, k, k], gaussian_expectation), fused
 *   cgp_ekf_nll_grad                                                   value_and_grad of ekf(...)[-1][-1] through the scan, demos/ekfs_mle.py:43-51
 *   cgp_model_from_source, cgp_filter_custom, cgp_smoother_custom      ekf / eks / cd_ekf / cd_eks on ANY model (the reference traces any callable:
 *                                                                      filters_smoothers.py:255, 342, 382, 425; test/test_ekfs.py:11-62), compiled at run time
 *   cgp_gaussian_expectation                                           gaussian_expectation quadratures.py:234-274
 *   cgp_simulate                                                       simulate_sde tools.py:119-170 and the
 *                                                                      state + measurement simulation of
 *                                                                      tetralith/jobs/crlb_ekf.py:39-58
 *   cgp_add_noise                                                      y = chirp + sqrt(Xi) N(0, 1), demos/ekfs_mle.py:33-35
 *
 * The leading batch axis B is the reference's jax.vmap(..., in_axes=0) over ys (tetralith/jobs/crlb_ekf.py:68-72).
 *
 * Conventions
 *   - every array pointer is a DEVICE pointer owned by the caller (e.g. torch.Tensor.data_ptr()); float64,
 *     C-contiguous, batch-major: ys [B][T], mfs [B][T][d], Pfs [B][T][d][d], nll [B][T];
 *   - the library allocates nothing the caller sees and never frees caller memory; outputs are fully overwritten;
 *   - calls enqueue work on `stream` (a hipStream_t, NULL = default stream) and return without synchronising;
 *   - return value 0 = success, negative = CGP_E_* (message via cgp_last_error);
 *   - numerical breakdown (non-PD covariance in a Cholesky) is NOT an error: NaN is written where the
 *     reference (JAX) would produce NaN and the run continues;
 *   - model callables of the reference (cond_m_cov, a, b, h) are replaced by an enumerated model set with
 *     analytic Jacobians (cgp_model); "stride" fields are in doubles, 0 = shared by all trials.
 */
#ifndef CHIRPGP_HIP_H
#define CHIRPGP_HIP_H

#ifndef __HIPCC_RTC__       /* (hiprtc, compiling a custom model against the library's own headers, has these types built in) */
#include <stddef.h>
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define CGP_VERSION 160          /* 0.1.6: CGP_CUSTOM_MEASUREMENT (ekf_for_kpt's h as source), cgp_smoother_time_split for cd_eks, run-time models check their argument structs; 0.1.5: cgp_smoother_select, cgp_ekf_nll_grad, cgp_model_from_source / cgp_filter_custom / cgp_smoother_custom; 0.1.4: cgp_release_workspace, cgp_source_hash, pinned reserved workspaces, per-call launch lock; 0.1.3: per-stream workspace kept by the context (cgp_reserve_workspace); 0.1.2: cgp_debug_set / cgp_debug_counters, cgp_gaussian_expectation_fn, cgp_filter_time_split */
#define CGP_MAX_D   12           /* largest state dimension compiled in (9 .. 12: the harmonic LCD model with 4 or 5 harmonics only) */

typedef struct cgp_ctx cgp_ctx;

/* ---- filter methods ------------------------------------------------------------------------------------- */
enum {
    CGP_F_EKF     = 0,   /* discrete model: (J, mean, Sigma) per step; with CGP_M_LINEAR this is kf             */
    CGP_F_SGP     = 1,   /* discrete model through sigma points                                                 */
    CGP_F_CD_EKF  = 2,   /* SDE model, RK4 on the EKF moment ODEs                                               */
    CGP_F_CD_SGP  = 3,   /* SDE model, RK4 on the sigma-point moment ODEs                                       */
    CGP_F_EKF_KPT = 4    /* linear dynamics, nonlinear harmonic measurement h (CGP_M_KPT)                       */
};
/* ---- smoother methods ----------------------------------------------------------------------------------- */
enum {
    CGP_S_EKS    = 0,    /* with CGP_M_LINEAR this is rts */
    CGP_S_SGP    = 1,
    CGP_S_CD_EKS = 2,
    CGP_S_CD_SGP = 3
};

/* ---- models ---------------------------------------------------------------------------------------------
 * Discrete (cond_m_cov) models:
 *   CGP_M_LINEAR        params = [F (d*d, row-major), Sigma (d*d)]                        mean = F u
 *   CGP_M_HARMONIC_LCD  params = [lam, b, ell, sigma, freq_scale], n_harm >= 1, d = 2 n_harm + 2
 *                       models.py:332-386 (disc_harmonic_chirp_lcd); n_harm = 1, freq_scale = 1 is
 *                       disc_chirp_lcd (models.py:264-311)
 *   CGP_M_LASCALA_LCD   params = [ell, sigma], d = 4                                      models.py:419-434
 * SDE (drift a, constant dispersion b) models -- gamma = b b^T is passed densely in cgp_model.gamma:
 *   CGP_M_LINEAR_SDE    params = [A (d*d)]                                                a(u) = A u
 *   CGP_M_HARMONIC_SDE  params = [lam, ell, freq_scale], n_harm >= 1                       models.py:122-178
 *                       (n_harm = 1: model_chirp models.py:76-119; lam = 0: model_lascala models.py:181-261)
 * Measurement model for ekf_for_kpt:
 *   CGP_M_KPT           params = [F (d*d), Sigma (d*d)], d = n_harm + 2                   models.py:522-580
 */
enum {
    CGP_M_LINEAR       = 0,
    CGP_M_HARMONIC_LCD = 1,
    CGP_M_LASCALA_LCD  = 2,
    CGP_M_LINEAR_SDE   = 3,
    CGP_M_HARMONIC_SDE = 4,
    CGP_M_KPT          = 5
};

typedef struct cgp_model {
    int32_t       model_id;
    int32_t       d;             /* state dimension                                  */
    int32_t       n_harm;        /* harmonic / KPT models, else 0                     */
    int32_t       n_params;      /* doubles per parameter vector                      */
    const double* params;        /* [n_params] or [B][param_stride]                   */
    int64_t       param_stride;  /* 0 = one parameter vector shared by all trials     */
    const double* gamma;         /* SDE methods: b b^T, [d][d]; else NULL             */
    int64_t       gamma_stride;  /* 0 = shared                                        */
} cgp_model;

/* Sigma points (quadratures.py:84-231): chi_i = m + chol(P) xi_i, E[z] ~ sum_i w_i z(chi_i).
 * Optional grouping: the chirp models are nonlinear in ONE state coordinate v = d - 2 only, and with a lower-triangular
 * chol(P) the point's chi_v depends on xi[0..v] alone (27 distinct arguments among the 81 Gauss-Hermite
 * points of d = 4).  If the caller orders the points so that points with equal xi[0..v] are contiguous and passes the
 * n_groups + 1 range boundaries in group_start, the kernels evaluate the transcendental part once per group.
 * group_start == NULL: every point is its own group.  The order of the points only changes the summation order. */
typedef struct cgp_sigma {
    int32_t        s;             /* number of points                         */
    int32_t        d;
    const double*  xi;            /* [s][d]                                    */
    const double*  w;             /* [s]                                       */
    const int32_t* group_start;   /* [n_groups + 1] (device pointer) or NULL   */
    int32_t        n_groups;      /* ignored when group_start is NULL          */
    uint32_t       flags;         /* CGP_SIGMA_*                               */
} cgp_sigma;

/* The caller asserts (to ~1e-13) that the set is a standardised rule -- sum w = 1, sum w xi = 0, sum w xi xi^T = I (true
 * of SigmaPoints.cubature and SigmaPoints.gauss_hermite) -- and, when group_start is given, that the members of a group
 * differ in the LAST coordinate only with sum_{k in group} w_k xi_k[d-1] = 0.  For the chirp / harmonic models, whose
 * last two state components are linear, the kernels may then take the moments of the linear components in closed form
 * and run the quadrature over one representative per group with the group's total weight: an exact regrouping of the
 * reference's sums (filters_smoothers.py:88-137), different only in rounding.  The d = 4 kernels and, for sets of at
 * most 16 groups (every cubature rule), the d = 6 / 8 tile-layout kernels take this form; CGP_LITERAL_SIGMA_SUM forces
 * the literal sums. */
#define CGP_SIGMA_STANDARD    0x1u
/* cgp_sigma.flags bit: additionally, every point has at most ONE non-zero coordinate among xi_0 .. xi_{d-2} (every cubature rule:
 * xi = +- sqrt(d) e_k, quadratures.py:138-150).  The d = 6 / 8 tile-layout filter then takes one square root per lane and step
 * instead of one per pivot of chol(Pf).  Only read together with CGP_SIGMA_STANDARD. */
#define CGP_SIGMA_AXIAL       0x2u

/* Measurement model and initial condition of a filter. */
typedef struct cgp_init {
    const double* H;   int64_t H_stride;     /* [d]    (ignored by CGP_F_EKF_KPT)  */
    const double* Xi;  int64_t Xi_stride;    /* [1]                                 */
    const double* m0;  int64_t m0_stride;    /* [d]                                 */
    const double* P0;  int64_t P0_stride;    /* [d][d]                              */
} cgp_init;

/* ---- flags ---------------------------------------------------------------------------------------------- */
#define CGP_NLL_FINAL_ONLY    0x1u   /* nll is [B]: only the last cumulative value (the MLE objective ekf(...)[-1][-1]) */
#define CGP_WAVE_PER_TRIAL    0x2u   /* force one 64-lane wavefront per trial (small batches; default below a threshold)  */
#define CGP_THREAD_PER_TRIAL  0x4u   /* force one lane per trial (large batches)                                           */
#define CGP_SEQUENTIAL_SCAN   0x8u   /* smoothers: force the step-by-step reverse scan instead of the tile-parallel forms   */
#define CGP_GENERIC_KERNEL    0x10u  /* force the generic kernel where a lane-cooperative specialisation exists (filters; discrete
                                        smoothers at d >= 4: the lane-scan time-parallel kernel instead of the cooperative walk) */
#define CGP_LITERAL_SIGMA_SUM  0x40u  /* sigma-point methods: sum over every point even when the set is CGP_SIGMA_STANDARD     */
#define CGP_DPP_KERNEL        0x80u  /* d = 4 chirp / La Scala models (ekf, sgp_filter, cd_ekf, cd_eks, cd_sgp_filter, cd_sgp_smoother):
                                        the DPP / LDS-reduced cooperative kernels instead of the matrix-core (MFMA) ones       */
#define CGP_FOUR_TRIALS_PER_WAVE 0x200u /* d = 4 matrix-core EKF: four trials per wavefront whatever the batch (default above 1024) */
#define CGP_ONE_TRIAL_PER_WAVE   0x400u /* ... one trial per wavefront whatever the batch                                      */
#define CGP_TIME_SPLIT        0x800u  /* discrete smoothers (rts, eks, sgp_smoother), one wavefront per trial: cut every record into segments
                                        walked by different wavefronts (two passes: compose the segments' affine maps, then walk with the
                                        right carry) whatever the batch -- default when the batch leaves two thirds of the SIMDs idle and
                                        the model is one of the chirp family (eks, sgp_smoother); NOT default for CGP_M_LINEAR (rts): the
                                        composed maps are exact in exact arithmetic but only as accurate as products of the caller's gains
                                        are well conditioned (1e-11 of the one-wave walk for the chirp models, 1e-7 seen for a random F)   */
#define CGP_NO_TIME_SPLIT     0x1000u /* ... never                                                                                      */
#define CGP_SIM_FIXED_X0      0x20u  /* cgp_simulate: x_0 = m0 exactly, P0 unused (simulate_sde_init, simulate_lgssm)       */

/* ---- error codes ---------------------------------------------------------------------------------------- */
#define CGP_OK              0
#define CGP_E_ARG          -1
#define CGP_E_UNSUPPORTED  -2
#define CGP_E_HIP          -3

int         cgp_version(void);
int         cgp_create(cgp_ctx** out, int device);
void        cgp_destroy(cgp_ctx* ctx);
/* Message of the CALLING THREAD's most recent failed call on `ctx` ("" if none): kept per thread, so host threads that
 * share a context do not race; valid until that thread's next failing call.  Every entry point runs on the context's
 * device and restores the calling thread's current device before it returns. */
const char* cgp_last_error(const cgp_ctx* ctx);

/* Filters: reads ys, writes mfs / Pfs / nll (any of the three may be NULL = not wanted).
 *
 * Measurement records.  The reference's drivers run MANY filters over ONE record: the MLE objective under value_and_grad
 * (demos/ekfs_mle.py:43-48), a parameter-grid sweep (BASELINE config C5, demos/ghfs_harmonics_mle.py:50-64).  The trials of
 * a call therefore need not own a record each:
 *     trial b reads the T doubles at  ys + rec(b) * ys_stride,   rec(b) = ys_index ? ys_index[b / ys_repeat] : b / ys_repeat
 *   ys_stride   doubles between consecutive records: T for the dense [B][T] layout of jax.vmap over ys; 0 = one record
 *               shared by every trial; any value >= 0 is legal (records are only read, they may overlap)
 *   ys_repeat   >= 1; consecutive trials served by the same record (G grid points or 2 P + 1 gradient probes per record)
 *   ys_index    NULL, or a DEVICE array of ceil(B / ys_repeat) record numbers (a subset / permutation of the records)
 * The dense batched call is (ys_stride, ys_repeat, ys_index) = (T, 1, NULL). */
int cgp_filter(cgp_ctx* ctx, int method, const cgp_model* model, const cgp_sigma* sigma, const cgp_init* init,
               double dt, const double* ys, int64_t ys_stride, int64_t ys_repeat, const int32_t* ys_index,
               int64_t B, int64_t T, double* mfs, double* Pfs, double* nll, uint32_t flags, void* stream);

/* Time-split filter with burn-in (round 4) -- for batches that leave most SIMDs idle.  A filter cannot be cut in time exactly (step t
 * linearises at the filtered mean of step t - 1), but it FORGETS its initial condition: the record is cut into `segments`
 * pieces of whole 64-step chunks, and wavefront s > 0 starts `burn_in` steps BEFORE its piece from (m0, P0), writes nothing
 * until its piece begins, and carries on like the sequential filter from there.  junction_err[b] (DEVICE array, [B], required)
 * receives, per trial, the largest relative mismatch at a junction between the state a burn-in arrived at and the state the
 * previous segment ended with (inf if a NaN sits at a junction).  It is a HEURISTIC, not a bound: the mismatch is
 * max |difference| / max |reference| over the whole mean vector and, separately, over the covariance -- components of different
 * scale (chirp amplitude ~ 1, frequency state ~ 7) share one denominator, so a small component can be off by more, relative to
 * itself, than the figure says -- and segment s is compared with the end of segment s - 1, which is itself approximate, not with
 * the sequential filter.  On the chirp models the rows a segment wrote were within 5 x the reported mismatch of the sequential
 * filter's in every test (tests/test_gpu_filter_split.py compares them with the CPU oracle directly).  The caller picks burn_in for its model (the chirp models of the reference lose a decade per ~ 450 steps
 * after the first ~ 1000: 3000 steps for 1e-7) and checks junction_err against its own tolerance -- cgp_filter remains the
 * reference's sequential recursion.  Cumulative NLL rows are made continuous across segments by a fix-up pass.  Built for
 * sgp_filter / cd_sgp_filter / ekf on the d = 4 chirp / La Scala models and sgp_filter at d = 6 / 8 (matrix-core / tile-layout
 * kernels); CGP_E_UNSUPPORTED otherwise.  segments = 1: the same as cgp_filter (junction_err = 0). */
int cgp_filter_time_split(cgp_ctx* ctx, int method, const cgp_model* model, const cgp_sigma* sigma, const cgp_init* init,
                          double dt, const double* ys, int64_t ys_stride, int64_t ys_repeat, const int32_t* ys_index,
                          int64_t B, int64_t T, double* mfs, double* Pfs, double* nll, uint32_t flags,
                          int64_t segments, int64_t burn_in, double* junction_err, void* stream);

/* The EKF's final negative log-likelihood AND its exact gradient in one launch: forward tangents of (m, P, nll) carried through the
 * scan -- the reference differentiates its objective through the scan (demos/ekfs_mle.py:43-51: value_and_grad of
 * ekf(build_model(g(theta)), ys)[-1][-1]).  d = 4 chirp / La Scala LCD models (CGP_M_HARMONIC_LCD with n_harm = 1, CGP_M_LASCALA_LCD).
 * A direction is CGP_DIR_DOUBLES doubles: the derivative, along one coordinate of the caller's parametrisation, of the model's constants
 *     log rho (rho = exp(-lam dt)) | q (chirp noise variance, models.py:302-308) | M32_F (2 x 2, row-major) | M32_Sigma (00, 01, 11) | Xi |
 *     m0 (4) | P0 (packed lower triangle: 00 10 11 20 21 22 30 31 32 33)
 * computed by the caller from its model builder (chirpgp_amd/mle.py does it by complex step); the derivative through the state-
 * dependent part of the model (softplus frequency, rotation, Jacobian) is taken by the kernel.  dirs: [B][n_dir][CGP_DIR_DOUBLES]
 * (device), nll: [B], grad: [B][n_dir] = d nll[b] / d direction.  Records are addressed as in cgp_filter.  One lane per
 * (trial, direction). */
#define CGP_DIR_DOUBLES 24
int cgp_ekf_nll_grad(cgp_ctx* ctx, const cgp_model* model, const cgp_init* init, double dt,
                     const double* ys, int64_t ys_stride, int64_t ys_repeat, const int32_t* ys_index, int64_t B, int64_t T,
                     const double* dirs, int32_t n_dir, double* nll, double* grad, uint32_t flags, void* stream);

/* ---- models compiled at run time ------------------------------------------------------------------------------------------
 * The reference's filters take any JAX-traceable callable (filters_smoothers.py:255, 304, 382, 425); the enumerated models above are the
 * reference's own builders.  A model outside that set is handed over as device source and compiled by ROCm's runtime compiler (hiprtc)
 * into the generic one-lane-per-trial kernels -- ekf / eks / sgp_filter / sgp_smoother for a discrete model, cd_ekf / cd_eks /
 * cd_sgp_filter / cd_sgp_smoother for an SDE: every filter and smoother of the reference that takes a callable -- with its Jacobian taken
 * by forward-mode dual numbers in the kernel (the counterpart of jax.jacfwd; csrc/cgp_custom.hpp).  `body` defines, for a generic scalar
 * type T (double or a dual number; sin, cos, exp, log, sqrt, tanh, pow(x, const), softplus are overloaded for it),
 *     CGP_CUSTOM_DISCRETE:   template <class T> __device__ void cond_mean(const T* u, const double* p, double dt, T* mean);
 *                            __device__ void cond_cov(const double* u, const double* p, double dt, double* cov);      cov [d][d] row-major
 *     CGP_CUSTOM_SDE:        template <class T> __device__ void drift(const T* u, const double* p, T* a);              (b b^T is `gamma`)
 *     CGP_CUSTOM_MEASUREMENT: template <class T> __device__ T measure(const T* u, const double* q);                     (0.1.6)
 *         the scalar measurement function h of ekf_for_kpt (filters_smoothers.py:267-314: H = jacfwd(h)(mp), pred = h(mp)) over LINEAR dynamics:
 *         cgp_filter_custom then reads `params` as [F (d x d) | Sigma (d x d)] per trial, like a CGP_M_LINEAR model, and hands init->H -- d doubles,
 *         per trial with H_stride -- to the body as q (whatever the measurement function wants to be told; zeros if nothing).  Filter only.
 * p = the trial's parameter vector (`params` + trial * param_stride at launch; any length the body agrees on with its caller).
 * include_dir = the directory of this library's kernel headers (chirpgp_amd/csrc in the source tree; include/ is found beside it).
 * A body that does not compile gives CGP_E_ARG with the compiler's messages in cgp_last_error.  d <= 8. */
typedef struct cgp_custom_model cgp_custom_model;
#define CGP_CUSTOM_DISCRETE 0
#define CGP_CUSTOM_SDE      1
#define CGP_CUSTOM_MEASUREMENT 2
int  cgp_model_from_source(cgp_ctx* ctx, int kind, int32_t d, const char* body, const char* include_dir, cgp_custom_model** out);
void cgp_custom_model_destroy(cgp_custom_model* model);
/* ekf (filters_smoothers.py:222-264) / cd_ekf (:352-397) / ekf_for_kpt (:267-314, CGP_CUSTOM_MEASUREMENT) on a compiled model -- or, with a sigma-point set (a plain point list: xi, w, s, d;
 * groups and flags are not read), sgp_filter (:446-490) / cd_sgp_filter (:534-582); the other arguments as cgp_filter.  The discrete
 * model's covariance is evaluated at every sigma point (filters_smoothers.py:118-120 as written). */
int cgp_filter_custom(cgp_ctx* ctx, const cgp_custom_model* model, const cgp_sigma* sigma, const double* params, int64_t param_stride,
                      const double* gamma, int64_t gamma_stride, const cgp_init* init, double dt,
                      const double* ys, int64_t ys_stride, int64_t ys_repeat, const int32_t* ys_index, int64_t B, int64_t T,
                      double* mfs, double* Pfs, double* nll, uint32_t flags, void* stream);
/* eks (:317-349) / cd_eks (:400-443) -- with a sigma-point set sgp_smoother (:493-531) / cd_sgp_smoother (:585-632) -- on a compiled model */
int cgp_smoother_custom(cgp_ctx* ctx, const cgp_custom_model* model, const cgp_sigma* sigma, const double* params, int64_t param_stride,
                        const double* gamma, int64_t gamma_stride, double dt, const double* mfs, const double* Pfs,
                        int64_t B, int64_t T, double* mss, double* Pss, uint32_t flags, void* stream);

/* Scratch of the time-split launches (segment records of cgp_filter_time_split, composed maps of the time-split smoothers) lives in
 * ONE buffer per (context, stream), grown on demand and freed by cgp_destroy; the library allocates nothing else per call.  Growing
 * waits for the stream -- illegal while the stream is being captured into a graph -- so a caller that captures its launches sizes the
 * buffer first: `bytes` >= 8 * (2 (d + d^2) + 1) * B * segments for a time-split filter, 8 * 40 * B * segments (d = 4) or
 * 8 * 112 * B * segments (d <= 8) for a time-split smoother (a smoother that finds no workspace falls back to its one-wavefront-
 * per-trial form, which needs none).
 * Threads: the launches of one cgp_filter / cgp_smoother call are enqueued as a unit (a per-context lock held until the last kernel of
 * the call is queued), so host threads may share a context and even a stream; a buffer is only replaced between calls, after the
 * stream has drained.
 * Graphs: a buffer sized by cgp_reserve_workspace is PINNED -- a captured graph bakes its address in, so no later launch on that stream
 * frees or regrows it: a launch that would need more than was reserved gets none (the time-split filter fails with CGP_E_HIP, a
 * smoother takes its one-wavefront-per-trial form).  Reserve for the largest call the stream will see while a graph that used the
 * buffer is alive; cgp_reserve_workspace with a larger size regrows it (outside capture only: it waits for the stream), and
 * cgp_release_workspace frees it and forgets the stream (call it before destroying a stream whose handle may be reused). */
int cgp_reserve_workspace(cgp_ctx* ctx, size_t bytes, void* stream);
int cgp_release_workspace(cgp_ctx* ctx, void* stream);

/* sha256 (hex) of the sources this library was built from: every .hip and .hpp file of csrc/ (byte order of their names), then
 * csrc/Makefile and this header, concatenated.  The Python layer recomputes it from the checked-out tree and refuses a stale library. */
const char* cgp_source_hash(void);

/* Smoothers: reads mfs / Pfs, writes mss / Pss (row T-1 is the filtering row T-1). */
int cgp_smoother(cgp_ctx* ctx, int method, const cgp_model* model, const cgp_sigma* sigma,
                 double dt, const double* mfs, const double* Pfs, int64_t B, int64_t T,
                 double* mss, double* Pss, uint32_t flags, void* stream);

/* Time-split smoother with burn-in (round 6) -- the counterpart of cgp_filter_time_split for the continuous-discrete sigma-point smoother
 * (cd_sgp_smoother, filters_smoothers.py:585-632), whose backward moment ODE is not affine in its carry and so cannot be composed exactly like
 * the discrete smoothers (CGP_TIME_SPLIT).  It FORGETS its terminal condition instead: the record is cut into `segments` pieces of whole
 * 64-step chunks, and the wavefront of a piece starts `burn_in` steps LATER in time than its piece ends, from the FILTERING row there, writes
 * nothing until its piece begins, and carries on like the sequential smoother.  junction_err[b] (DEVICE, [B], required): the largest relative
 * mismatch, over trial b's junctions, between the state a burn-in arrived at and the row the piece later in time wrote there (max |difference| /
 * max |reference|, mean and covariance separately; inf for a NaN) -- a heuristic like the filters' (see there), checked by the caller against
 * its own tolerance.  Built for CGP_S_CD_SGP (standard sigma set) and CGP_S_CD_EKS on the d = 4 chirp / La Scala SDE (matrix-core kernels);
 * CGP_E_UNSUPPORTED otherwise.  segments = 1: cgp_smoother (junction_err = 0).  Scratch: 8 * 20 * B * segments bytes of the stream's workspace. */
int cgp_smoother_time_split(cgp_ctx* ctx, int method, const cgp_model* model, const cgp_sigma* sigma,
                            double dt, const double* mfs, const double* Pfs, int64_t B, int64_t T,
                            double* mss, double* Pss, uint32_t flags, int64_t segments, int64_t burn_in,
                            double* junction_err, void* stream);

/* Smoothers with selected outputs -- the step right behind the smoother in every driver of the reference, fused into it
 * (demos/ekfs_mle.py:69-77: gaussian_expectation(ms = mss[:, 2], chol_Ps = sqrt(Pss[:, 2, 2]), func = g), then rmse; quadratures.py:234-274).
 * Besides (or instead of) the full rows a launch writes, per trial and step, the smoothed mean and variance of ONE state component and
 * E[f(V)], V ~ N(mean, variance), by 1-D Gauss-Hermite -- [B][T] arrays, each optional.  With mss = Pss = NULL a d = 4 smoother writes
 * 8 - 24 bytes a step instead of 160.  The d = 4 discrete smoothers (rts, eks, sgp_smoother; both launch shapes of eks, cd_eks one lane
 * per trial) and the d = 5 .. 8 tile-layout smoothers write the selected outputs themselves; every other kernel needs mss and Pss as
 * well (CGP_E_UNSUPPORTED without them) and the selection is gathered from the full rows by a second launch.  Row T - 1 is the
 * filtering row's marginal.  A negative variance gives NaN in `expect`, like sqrt in the reference's call. */
typedef struct cgp_smooth_out {
    double*       mss;         /* [B][T][d] or NULL                                                             */
    double*       Pss;         /* [B][T][d][d] or NULL                                                          */
    int32_t       comp;        /* state component k of the three arrays below, 0 <= k < d                        */
    int32_t       func;        /* CGP_FN_* integrand of `expect`                                                 */
    double*       comp_mean;   /* [B][T]: mss[b][t][k], or NULL                                                  */
    double*       comp_var;    /* [B][T]: Pss[b][t][k][k], or NULL                                               */
    double*       expect;      /* [B][T]: E[f(V)], or NULL                                                       */
    const double* xi;          /* [order] nodes (DEVICE), the reference's scaling: sqrt(2) x the Hermite roots   */
    const double* w;           /* [order] weights (DEVICE), normalised                                           */
    int32_t       order;       /* 1 .. 32 (the reference's default: 10)                                          */
} cgp_smooth_out;
int cgp_smoother_select(cgp_ctx* ctx, int method, const cgp_model* model, const cgp_sigma* sigma,
                        double dt, const double* mfs, const double* Pfs, int64_t B, int64_t T,
                        const cgp_smooth_out* out, uint32_t flags, void* stream);

/* E[softplus(V)] for n scalar Gaussian marginals N(ms[i], sd[i]^2) by 1-D Gauss-Hermite of the given order
 * (nodes xi[order], weights w[order] already in the reference's scaling): quadratures.py:234-274 with func = g. */
int cgp_gaussian_expectation(cgp_ctx* ctx, const double* ms, const double* sd, int64_t n, int64_t in_stride,
                             const double* xi, const double* w, int32_t order, double* out, void* stream);
/* The same for an enumerated integrand (the reference takes any callable, quadratures.py:234-274; a callable cannot cross this
 * boundary): E[f(V)], f = softplus (= cgp_gaussian_expectation), exp (the reference's own test of this function,
 * test/test_utils.py:84-95), the identity, the square. */
#define CGP_FN_SOFTPLUS  0
#define CGP_FN_EXP       1
#define CGP_FN_IDENTITY  2
#define CGP_FN_SQUARE    3
int cgp_gaussian_expectation_fn(cgp_ctx* ctx, int func, const double* ms, const double* sd, int64_t n, int64_t in_stride,
                                const double* xi, const double* w, int32_t order, double* out, void* stream);

/* Per-step squared-error statistics over the trial axis -- what the reference's Monte-Carlo jobs reduce their runs to
 * (tetralith/jobs/crlb_ekf.py:82-89: mean and standard deviation over 10^6 trials of (mfs - xs)^2 per time step):
 *     e = (a[b][t][comps[c]] - r[b][t][comps[c]])^2,     sums[c][0][t] += sum_b e,     sums[c][1][t] += sum_b e^2
 * a, r: [B][T][d] (device); comps: n <= 8 component numbers (HOST array); sums: [n][2][T] (device), ACCUMULATED into -- zero it
 * before the first chunk, call once per chunk of trials (and all-reduce it over ranks: 2 n T doubles). */
int cgp_squared_error_sums(cgp_ctx* ctx, const double* a, const double* r, int64_t B, int64_t T, int32_t d,
                           const int32_t* comps, int32_t n, double* sums, void* stream);

/* ---- input side: Monte-Carlo data generated in HBM ------------------------------------
 * Random numbers are counter-based (Philox4x32-10 keyed by `seed`, counter = (global trial number, index, stream),
 * Box-Muller), so a trial's draws do not depend on B, on the launch shape or on how the batch is sharded over ranks:
 * pass the number of the shard's first trial in `trial0`.  The reference draws from jax.random, whose streams cannot
 * be reproduced; the streams here are defined in csrc/cgp_rng.hpp. */

/* x_0 = m0 + chol(P0) z;  x_k = mean(x_{k-1}) + chol(Sigma) dw_k;  y_k = H . x_k + sqrt(Xi) e_k,  k = 1..T
 * with (mean, Sigma) the discrete model's cond_m_cov (tools.py:148-167; crlb_ekf.py:39-58).
 * xs [B][T][d] and ys [B][T]: either may be NULL.  Only discrete models (CGP_M_LINEAR, *_LCD).
 * flags: CGP_SIM_FIXED_X0, CGP_WAVE_PER_TRIAL, CGP_THREAD_PER_TRIAL. */
int cgp_simulate(cgp_ctx* ctx, const cgp_model* model, const cgp_init* init, double dt, uint64_t seed, int64_t trial0,
                 int64_t B, int64_t T, double* xs, double* ys, uint32_t flags, void* stream);

/* ys[b][k] = clean[b * clean_stride + k] + sqrt(Xi[b * Xi_stride]) e_{b,k}   (clean_stride = 0: one clean record shared
 * by all trials).  e is the measurement-noise stream of cgp_simulate. */
int cgp_add_noise(cgp_ctx* ctx, const double* clean, int64_t clean_stride, const double* Xi, int64_t Xi_stride,
                  uint64_t seed, int64_t trial0, int64_t B, int64_t T, double* ys, void* stream);

/* Test hook: the raw generator.  out[4 i .. 4 i + 3] = Philox4x32-10(counter = ctr[4 i .. 4 i + 3], key = key[0..1]). */
int cgp_debug_philox(cgp_ctx* ctx, const uint32_t* ctr, const uint32_t* key, int64_t n, uint32_t* out, void* stream);

/* Test hook: evaluates one of the engine's in-kernel float64 elementary functions (csrc/cgp_fastmath.hpp) on n inputs.
 * op: 0 exp, 1 log on [1, inf] (softplus argument), 2 sincos (out0 = sin, out1 = cos), 3 reciprocal,
 *     4 softplus pair (out0 = log(exp(x) + 1), out1 = its derivative), 5 / 6 the wave-uniform variants of 4 / 2,
 *     7 the per-lane softplus pair in its wide common-regime form (with the naive fallback), 8 the speculative step's
 *     lean softplus pair (valid for 1.5 <= x < 700 only; ~1e-11), 9 reciprocal with one Newton step, 10 the branch-free softplus pair
 *     for any |x| < 700 (NaN where it reports "not valid"), 11 the polynomial softplus pair of the MID regime (valid for |x| <= 2).
 *     out1 may be NULL for one-output ops. */
int cgp_debug_math(cgp_ctx* ctx, int op, const double* x, int64_t n, double* out0, double* out1, void* stream);

/* Tuning / measurement hooks of ONE context (nothing process-wide, no environment variable is read by the library).
 *   CGP_DBG_WALK_SEGMENTS   cap on the number of segments of the time-split smoothers (CGP_TIME_SPLIT): 0 = chosen from
 *                           (B, T, CUs) (default), 1 = never split, n > 1 = at most n
 *   CGP_DBG_COUNT_REGIMES   1: the d = 4 matrix-core EKF (one trial per wavefront) adds, per launch, the number of 64-step chunks
 *                           it ran in each regime of its speculative step to the context's counters; 0 (default): it does not */
#define CGP_DBG_WALK_SEGMENTS  1
#define CGP_DBG_COUNT_REGIMES  2
#define CGP_DBG_LANE_BUFFERS   3   /* A/B aid: 3 = the large-batch smoothers (one lane per trial) request their covariance rows TWO steps ahead where the
                                      LDS allows a third buffer beside four workgroups a CU; 0 / 2 (default): one step ahead (measured: no difference) */
int cgp_debug_set(cgp_ctx* ctx, int key, int64_t value);
/* Waits for `stream`, copies the context's eight counters to the HOST array `out` and, with reset != 0, zeroes them.
 *   out[0] chunks kept from the HIGH regime (frequency state >= 5 throughout)      out[1] chunks kept from the common regime (>= 1.5)
 *   out[2] chunks repeated with the checked step (left the common regime)           out[3] chunks run on the checked step after such a repeat
 *   out[4] chunks that were tried in the HIGH regime, left it and were repeated in the common regime
 *   out[5] chunks run on the WIDE step after such a repeat (branch-free, any |state| < 700; round 5: out[3] counts only those that left even that)
 *   out[6] chunks kept from the LOW regime (round 5: frequency state <= -1.5 throughout, tried for chunks that start at or below -1.75)
 *   out[7] chunks kept from the MID regime (round 5: |frequency state| < 2 throughout, tried for chunks that start within +-1.75) */
int cgp_debug_counters(cgp_ctx* ctx, uint64_t* out, int reset, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CHIRPGP_HIP_H */
