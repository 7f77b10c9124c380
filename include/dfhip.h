/*
 * dfhip.h -- C-ABI of libdfhip.so, the MI355X (gfx950) GP-surrogate + acquisition engine
 * that sits under Dragonfly's GP hot path.
 *
 * The reference (dragonfly 0.1.7, /root/reference) is pure Python/NumPy/SciPy and has no
 * FFI of its own; the seams are Python class boundaries.  Every entry
 * point below names the reference function(s) it replaces (paths relative to the
 * reference root).  The Python host side in dragonfly_amd/ binds these with ctypes
 * (dragonfly_amd/_lib.py); INTEGRATION.md shows the stub a Dragonfly maintainer would add.
 *
 * Conventions
 *   - All matrices are row-major (NumPy C order) float64; indices are int64.
 *   - Every `const double*` / `double*` data argument may be a HOST pointer or a DEVICE
 *     pointer obtained from dfh_malloc(); the library detects which (hipPointerGetAttributes)
 *     and stages host buffers itself.  Scalars returned through pointers are host memory.
 *   - Return value: DFH_OK, or an error code; dfh_last_error() gives the text (thread local).
 *   - Calls on one dfh_ctx are serialised by the caller (the reference is single threaded).
 *   - No callbacks into the host language.  Not fork-safe after the first dfh_ctx_create().
 */
#ifndef DFHIP_H
#define DFHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFH_ABI_VERSION 2   /* 2: dfh_kernel_desc grew the additive-factor fields at its end */

/* ---- status codes --------------------------------------------------------------------- */
#define DFH_OK            0
#define DFH_ERR_NOT_PD    1  /* Cholesky met a non-positive pivot (numpy LinAlgError analogue;
                                dragonfly/utils/general_utils.py:178,190)                    */
#define DFH_ERR_BAD_ARG   2  /* ValueError analogue                                          */
#define DFH_ERR_HIP       3  /* HIP runtime error, see dfh_last_error()                      */
#define DFH_ERR_JITTER    4  /* jitter ladder exhausted (ValueError at general_utils.py:200) */

/* ---- kernel description ---------------------------------------------------------------- */
#define DFH_KERNEL_SE       0 /* dragonfly/gp/kernel.py:132 SEKernel                         */
#define DFH_KERNEL_MATERN   1 /* dragonfly/gp/kernel.py:225 MaternKernel, nu in {.5,1.5,2.5} */
#define DFH_KERNEL_ADDITIVE 2 /* dragonfly/gp/kernel.py:461 AdditiveKernel over SE/Matern    */
#define DFH_KERNEL_PRODUCT  3 /* dragonfly/gp/kernel.py:541 CoordinateProductKernel over SE/Matern
                               * (scale * prod_g k_g(X[:, coords_g]); same group fields as ADDITIVE) */
#define DFH_KERNEL_POLY     4 /* dragonfly/gp/kernel.py:331 PolyKernel: scale * ((x*s).(y*s) + 1)^order;
                               * `nu` = order (a non-negative integer), `bw` = dim_scalings s   */
#define DFH_KERNEL_EXPDECAY 5 /* dragonfly/gp/kernel.py:398 ExpDecayKernel (the multi-fidelity
                               * kernel of euclidean_gp.py:881-887): scale * prod_d (1 + x_d +
                               * y_d)^-powers_d + offset; `nu` = offset, `bw` = powers, dim <= 8.
                               * POLY and EXPDECAY are not stationary: k(x,x) depends on x.  They
                               * are accepted alone and as factors of a PRODUCT (sub_kind, with
                               * sub_nu = order / offset and sub_bw = scalings / powers); POLY also
                               * as a group of an ADDITIVE kernel (gp/euclidean_gp.py:870-879).    */

/* One Euclidean kernel.  For SE / MATERN: `dim`, `scale`, `nu`, `bw[dim]` (dim_bandwidths); POLY /
 * EXPDECAY reuse `nu` and `bw` as described above.
 * For ADDITIVE: `scale` is the outer scale, and the n_groups sub-kernels are described by the
 * flattened arrays: group g covers input columns group_dims[group_off[g] .. group_off[g+1])
 * with bandwidths sub_bw[group_off[g] .. group_off[g+1]) , kind sub_kind[g], scale
 * sub_scale[g], smoothness sub_nu[g].  All pointers are HOST pointers, read during the call. */
typedef struct dfh_kernel_desc {
  int32_t kind;
  int32_t dim;            /* input dimension d (number of columns of X)                      */
  double  scale;
  double  nu;             /* MATERN: nu ; POLY: order ; EXPDECAY: offset                     */
  const double*  bw;      /* [dim]   SE / MATERN: bandwidths ; POLY: scalings ; EXPDECAY: powers */
  int32_t n_groups;       /* ADDITIVE only                                                   */
  const int32_t* group_off;   /* [n_groups+1]                                                */
  const int32_t* group_dims;  /* [group_off[n_groups]] column indices                        */
  const int32_t* sub_kind;    /* [n_groups] SE | MATERN (| POLY | EXPDECAY in a PRODUCT)     */
  const double*  sub_scale;   /* [n_groups]                                                  */
  const double*  sub_nu;      /* [n_groups]                                                  */
  const double*  sub_bw;      /* [group_off[n_groups]]                                       */
  /* PRODUCT only, all three NULL when every group is a factor of its own.  Otherwise an
   * AdditiveKernel may stand among the product's kernels (the multi-fidelity GP with an additive
   * domain model, gp/euclidean_gp.py:696-707; kernel.py:461-501 inside kernel.py:578-589): its
   * groups are listed like any others, group_factor[g] (non-decreasing, 0 .. n_factors-1) says
   * which factor group g belongs to, and a factor f with factor_is_sum[f] != 0 is
   * factor_scale[f] * (k_g + k_g' + ...) over its groups (sub_scale[g] = the group kernels' own
   * scales); a factor with factor_is_sum[f] == 0 has exactly one group and is that kernel.        */
  const int32_t* group_factor;   /* [n_groups]                                               */
  const int32_t* factor_is_sum;  /* [n_factors]                                              */
  const double*  factor_scale;   /* [n_factors]                                              */
} dfh_kernel_desc;

/* ---- acquisitions ---------------------------------------------------------------------- */
#define DFH_ACQ_MEAN  0 /* mu                                                                */
#define DFH_ACQ_UCB   1 /* mu + p0*sd            p0 = beta_th   (gpb_acquisitions.py:211-223)*/
#define DFH_ACQ_EI    2 /* sd*(z*Phi(z)+phi(z)), z=(mu-p0)/sd, p0 = curr_best   (:247-261)   */
#define DFH_ACQ_PI    3 /* Phi((mu-p0)/sd)                                      (:230-239)   */
#define DFH_ACQ_TTEI  4 /* c*(z*Phi(z)+phi(z)), c=sqrt(p1^2+sd^2), z=(mu-p0)/c  (:269-280)   */
#define DFH_ACQ_STD   5 /* sd                                                                */

typedef struct dfh_ctx dfh_ctx;   /* one per device: stream, workspaces                      */
typedef struct dfh_gp  dfh_gp;    /* a fitted GP resident in HBM                              */

/* ---- context / memory ------------------------------------------------------------------ */
int  dfh_abi_version(void);
int  dfh_device_count(int* count);
int  dfh_ctx_create(int device, dfh_ctx** out);
void dfh_ctx_destroy(dfh_ctx* ctx);
int  dfh_sync(dfh_ctx* ctx);
const char* dfh_last_error(void);
int  dfh_device_name(dfh_ctx* ctx, char* buf, size_t buflen);
/* Free / total HBM of the context's device in bytes (hipMemGetInfo); plumbing for leak checks. */
int  dfh_mem_info(dfh_ctx* ctx, uint64_t* free_bytes, uint64_t* total_bytes);

int  dfh_malloc(dfh_ctx* ctx, size_t bytes, void** dptr);
int  dfh_free(dfh_ctx* ctx, void* dptr);
int  dfh_memcpy_h2d(dfh_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int  dfh_memcpy_d2h(dfh_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);

/* HIP-event timing on the context's stream (the stream every kernel is launched on).
 * dfh_timer_begin records an event; dfh_timer_end records a second one, waits for it and
 * returns the elapsed milliseconds.  Used by bench.py for the roofline numbers.            */
int  dfh_timer_begin(dfh_ctx* ctx);
int  dfh_timer_end(dfh_ctx* ctx, double* ms);

/* ---- linear-algebra building blocks (exposed for tests and for the drop-in utils) ------- */

/* K_out[n1 x n2] = kernel(X1[n1 x d], X2[n2 x d]).  X2 == NULL means X2 = X1.
 * Replaces Kernel.evaluate -> _child_evaluate (dragonfly/gp/kernel.py:76-89,171-181,292-299,
 * 484-494) including dist_squared's clip at 0 (dragonfly/utils/general_utils.py:58-70).
 * diag_add is added to K_out[i][i] (only meaningful when X2 == NULL; fuses
 * `K + noise_var*np.eye(n)`, dragonfly/gp/gp_core.py:843).                                  */
int dfh_kernel_matrix(dfh_ctx* ctx, const dfh_kernel_desc* k,
                      const double* X1, int64_t n1, const double* X2, int64_t n2,
                      double diag_add, double* K_out);

/* Squared Euclidean distances, dragonfly/utils/general_utils.py:58-70 (dist_squared).      */
int dfh_dist_squared(dfh_ctx* ctx, const double* X1, int64_t n1, const double* X2,
                     int64_t n2, int64_t d, double* D_out);

/* C[M x N] = beta*C + alpha * A[M x K] * op(B), op(B) = B^T with B[N x K] (transb = 0) or
 * B[K x N] (transb = 1); fp64 MFMA tiles.  The dense contraction np.dot / dgemm stands for
 * at general_utils.py:68, gp_core.py:174,181.  lower_only != 0 computes only tiles that
 * intersect the lower triangle (SYRK use).                                                   */
int dfh_gemm(dfh_ctx* ctx, int transb, int64_t M, int64_t N, int64_t K, double alpha,
             const double* A, int64_t lda, const double* B, int64_t ldb, double beta,
             double* C, int64_t ldc, int lower_only);

/* In-place lower Cholesky of the row-major n x n matrix A (lda = n); the strict upper
 * triangle is zeroed (numpy.linalg.cholesky semantics, general_utils.py:178).
 * Returns DFH_ERR_NOT_PD if a pivot is <= 0 or NaN; *info_pivot = 1-based index (0 if ok).   */
int dfh_cholesky(dfh_ctx* ctx, double* A, int64_t n, int64_t* info_pivot);

/* stable_cholesky (general_utils.py:166-204): factor M; on failure retry M + 10^p*max(diag M)
 * for p = -11..4.  M_in is preserved, L_out (n x n) receives the factor.
 * *jitter_power = INT32_MIN when no jitter was needed.  DFH_ERR_JITTER if p reaches 5.       */
int dfh_stable_cholesky(dfh_ctx* ctx, const double* M_in, int64_t n, double* L_out,
                        int32_t* jitter_power);

/* project_symmetric_to_psd_cone (utils/general_utils.py:150-163): out = V max(Lambda, epsilon) V^T
 * for the symmetric M = V Lambda V^T (n x n; out may equal M).  Used on the Gram matrix of kernels
 * that are not guaranteed PSD (gp_core.py:838-841, epsilon = 0) and on their posterior covariances
 * (get_post_covar_from_raw_covar, gp_core.py:849-857, epsilon = 0.05 noise_var).  No eigenvectors
 * are formed: epsilon I + (B + B sign(B)) / 2 with B = M - epsilon I and the matrix sign function
 * from 96 Newton-Schulz steps on the fp64 MFMA GEMM (csrc/psdproj.hip); agrees with the eigh route
 * to ~1e-13 |M|.                                                                                */
int dfh_project_psd(dfh_ctx* ctx, const double* M, int64_t n, double epsilon, double* out);

/* Solve L x = b (upper = 0) or L^T x = b (upper = 1) with L lower-triangular n x n row major;
 * b is n x nrhs row-major (nrhs >= 1).  Replaces solve_lower_triangular /
 * solve_upper_triangular (general_utils.py:208-221) as used at gp_core.py:162-163,180.       */
int dfh_solve_triangular(dfh_ctx* ctx, const double* L, int64_t n, int upper,
                         const double* b, int64_t nrhs, double* x_out);

/* ---- GP fit / posterior ----------------------------------------------------------------- */
#define DFH_FIT_NO_JITTER 1  /* report DFH_ERR_NOT_PD instead of running the jitter ladder    */
/* dfh_gp_fit_gram only -- kernels that are not guaranteed PSD (Cartesian-product / neural-network
 * GPs), _get_cholesky_decomp's other branches (gp/gp_core.py:827-840):                        */
#define DFH_FIT_PROJECT_FIRST      2  /* 'project_first': K -> its projection onto the PSD cone
                                         (dfh_project_psd), then K + noise I and the ladder     */
#define DFH_FIT_TRY_BEFORE_PROJECT 4  /* 'try_before_project': plain Cholesky of K + noise I; only
                                         if that is not positive definite, as project_first    */

/* GP.build_posterior (gp_core.py:155-163) + _get_cholesky_decomp 'guaranteed_psd'
 * (gp_core.py:841-844) + compute_log_marginal_likelihood (gp_core.py:222-227):
 *   K = kernel(X,X); L = stable_cholesky(K + noise_var I); alpha = L^T \ (L \ y_centred);
 *   lml = -1/2 y^T alpha - sum log L_ii - n/2 log 2 pi.
 * X[n x d], y_centred[n] = Y - mean_func(X) (the mean function is a host callable in the
 * reference, gp_core.py:161, so it is evaluated by the caller).                             */
int dfh_gp_fit(dfh_ctx* ctx, const dfh_kernel_desc* k, const double* X, int64_t n, int64_t d,
               const double* y_centred, double noise_var, int flags, dfh_gp** out,
               double* lml, int32_t* jitter_power);
int dfh_gp_free(dfh_gp* gp);

#define DFH_GET_L        0   /* n x n lower factor (GP.L)                                     */
#define DFH_GET_ALPHA    1   /* n (GP.alpha)                                                  */
#define DFH_GET_K        2   /* n x n kernel matrix without noise (GP.K_trtr_wo_noise),
                                recomputed on demand: the fit factors the Gram matrix in place  */
/* Copies one of the fitted quantities above into the caller's buffer `out` (host or device
 * memory, sized as the table says).  These are the attributes the reference reads off a GP from outside
 * (opt/gpb_acquisitions.py:169,171,367,369; gp/gp_core.py:203).  Handles from dfh_gp_fit_gram
 * keep no kernel: DFH_GET_K returns DFH_ERR_BAD_ARG for them.                                 */
int dfh_gp_get(dfh_gp* gp, int what, double* out);
int64_t dfh_gp_n(dfh_gp* gp);

/* Posterior for ANY positive semi-definite kernel the caller evaluates itself: GP.build_posterior
 * (gp/gp_core.py:155-163) with the Gram matrix from the documented override hook
 * GP._get_training_kernel_matrix (:149-153), and GP.eval (:165-190) with the caller's cross matrix.
 * K is n x n (kernel(X, X), without noise); the handle keeps L, alpha and the block inverses, no
 * kernel: it works with dfh_gp_predict_gram / dfh_gp_predict_covar_gram, dfh_gp_get (L, alpha),
 * dfh_gp_n, dfh_gp_free; the kernel-based entry points return DFH_ERR_BAD_ARG for it.            */
int dfh_gp_fit_gram(dfh_ctx* ctx, const double* K, int64_t n, const double* y_centred, double noise_var,
                    int flags, dfh_gp** out, double* lml, int32_t* jitter_power);
/* mu = Kcross alpha + mean, sd = sqrt(kss - rowsumsq(Kcross L^-T)) (no clipping: NaN as NumPy).
 * Kcross [m x n], row i = kernel(x*_i, X); kss [m] = kernel(x*_i, x*_i) (NULL with sd_out NULL).  */
int dfh_gp_predict_gram(dfh_gp* gp, const double* Kcross, int64_t m, const double* kss, double mean_const,
                        const double* mean_vals, double* mu_out, double* sd_out);
/* mu_out = Kcross alpha (no mean added), cov_out [m x m] = Ktete - V^T V  (gp_core.py:179-181).   */
int dfh_gp_predict_covar_gram(dfh_gp* gp, const double* Kcross, int64_t m, const double* Ktete,
                              double* mu_out, double* cov_out);

/* Incremental posterior update: the posterior of `gp` extended by q new observations, as a NEW
 * handle (`gp` stays valid and unchanged).  Replaces the rebuild of GP.add_data_multiple
 * (gp/gp_core.py:139-146: X.extend, Y.extend, build_posterior) with a block-row append of the
 * Cholesky factor, O(n^2 q) instead of O((n+q)^3); same kernel, noise variance and data order,
 * so L, alpha and lml equal the rebuilt ones up to rounding.  y_centred holds all n+q centred
 * labels (old ones first).  If the existing fit used the stable_cholesky ladder or the appended
 * block is not positive definite, the extended matrix is rebuilt and factored from scratch with
 * the ladder (utils/general_utils.py:166-204), as the reference's rebuild would.               */
int dfh_gp_append(dfh_gp* gp, const double* Xnew, int64_t q, const double* y_centred, int flags,
                  dfh_gp** out, double* lml, int32_t* jitter_power);

/* Hyper-parameter tuning inner loop: log marginal likelihoods of `nb` candidate settings on the
 * same data.  Replaces a loop of GPFitter._tuning_objective calls (gp/gp_core.py:551-564: build_gp
 * -> GP.build_posterior :155-163 -> compute_log_marginal_likelihood :222-227), as issued by
 * random_maximise / random_sample_cts_dscr for the 'rand' and 'rand_exp_sampling' tuners
 * (utils/oper_utils.py:70-80, 100-112; gp_core.py:435-445).  descs[c], mean_consts[c] (NULL = 0),
 * noise_vars[c] describe candidate c; y holds the raw labels (the constant mean is subtracted on
 * the device).  Gram matrices of a group of candidates are factored in lock-step; a candidate that
 * needs the stable_cholesky ladder gets it individually (jitter_powers[c], INT32_MIN = none).
 * Errors as dfh_gp_fit (DFH_ERR_NOT_PD with DFH_FIT_NO_JITTER, DFH_ERR_JITTER when the ladder is
 * exhausted, utils/general_utils.py:200).
 * A call with a handful of candidates -- what the slice sampler (sampling/slice.py:45-68) and the tree search
 * (utils/doo.py:112-121) issue a hundred thousand times per run -- is ONE kernel launch up to n = 128 with nothing
 * copied: descriptors are read from, and results written to, the context's mapped pinned buffer while the host polls
 * a status word (bounded; the call is still synchronous: lml_out is valid on return).                           */
/* Optional hints in `flags` of dfh_gp_lml_batch (a caller that knows where its buffers live saves the library one
 * pointer query each -- microseconds that matter when a slice sampler calls with one candidate at n = 50):          */
#define DFH_LML_X_IS_DEVICE 0x100   /* X is device memory (a dfh_alloc buffer)                                   */
#define DFH_LML_Y_IS_HOST   0x200   /* y is ordinary host memory                                                 */
int dfh_gp_lml_batch(dfh_ctx* ctx, const dfh_kernel_desc* descs, int32_t nb, const double* X, int64_t n,
                     int64_t d, const double* y, const double* mean_consts, const double* noise_vars,
                     int flags, double* lml_out /* [nb] */, int32_t* jitter_powers /* [nb] or NULL */);

/* Triangular solves with the factor (solve_lower/upper_triangular, utils/general_utils.py:208-221,
 * as used at gp_core.py:162-163,180) multiply by explicit inverses of the 512 x 512 diagonal
 * blocks of L; where such an inverse M is not good enough -- max|I - M L_bb| above 1e-13
 * (DFH_REFINE_TOL), i.e. an ill-conditioned block -- the solves add steps of iterative refinement
 * against L_bb so that their residual is that of a substitution (dtrtrs).  steps_out
 * [ceil(n/512)] receives the number of steps per block (0 everywhere for well-conditioned fits). */
int dfh_gp_refine_steps(dfh_gp* gp, int32_t* steps_out);

/* GP.eval(X_test, 'std') without the mean function (gp_core.py:165-190):
 *   mu_out[m]  = K(Xs, X) alpha           (caller adds mean_func(Xs))
 *   sd_out[m]  = sqrt(k(x,x) - ||L \ k(X,x)||^2), no clipping -> NaN for negative variance
 * sd_out may be NULL (uncert_form='none').  Only the diagonal of the posterior covariance
 * is formed (the reference forms the m x m matrix and takes its diagonal).
 * If Xh != NULL (q rows) the variance is that of the GP augmented with q hallucinated
 * observations at Xh (eval_with_hallucinated_observations, gp_core.py:192-220); the mean
 * is unchanged.                                                                             */
int dfh_gp_predict(dfh_gp* gp, const double* Xs, int64_t m, const double* Xh, int64_t q,
                   double* mu_out, double* sd_out);

/* Full posterior covariance GP.eval(X_test,'covar') (gp_core.py:179-184): cov_out[m x m].  */
int dfh_gp_predict_covar(dfh_gp* gp, const double* Xs, int64_t m, const double* Xh, int64_t q,
                         double* mu_out, double* cov_out);

/* Fused posterior + acquisition + arg-max over m candidates (gpb_acquisitions.py:215-280
 * closures + oper_utils.random_maximise's `obj_vals.argmax()`, oper_utils.py:73).
 *   mean_const : constant prior mean added to mu (fitter GPs use a constant mean,
 *                gp_core.py:527-530); if mean_vals (optional, [m]) is given it is used instead
 *                (arbitrary mean functions are host callables evaluated by the caller).
 *   params     : acquisition parameters p0, p1 (see DFH_ACQ_*).
 *   vals_out   : optional [m] acquisition values.
 *   best_val / best_idx : numpy argmax semantics -- first NaN wins, else first maximum.     */
int dfh_gp_acq_argmax(dfh_gp* gp, int acq, const double* params, const double* Xs, int64_t m,
                      const double* Xh, int64_t q, double mean_const, const double* mean_vals,
                      double* vals_out, double* best_val, int64_t* best_idx);

/* Blocked-joint Thompson sampling (asy_ts -> GP.draw_samples -> draw_gaussian_samples,
 * gpb_acquisitions.py:119-127, gp_core.py:250-254, general_utils.py:224-232).
 * Candidates are processed in blocks of `block` rows; inside a block the draw is the exact
 * joint draw  s = mu + stable_cholesky(Sigma_block) u ; blocks are independent.  With
 * block >= m this is literally gp.draw_samples(1, Xs).  U[m] are the standard normals
 * (np.random.normal in the reference) supplied by the caller.
 * samples_out optional [m]; jitter_powers_out optional [ceil(m/block)].                     */
int dfh_gp_ts(dfh_gp* gp, const double* Xs, int64_t m, int64_t block, const double* U,
              double mean_const, const double* mean_vals, double* samples_out,
              double* best_val, int64_t* best_idx, int32_t* jitter_powers_out);

/* add-UCB per-group posterior (gpb_acquisitions.py:139-189): for additive-kernel GPs,
 * group g's acquisition over its own candidate set Xg[m x |g|]:
 *   mu_g = scale*k_g(Xg, X[:,g]) alpha ; sd_g from the shared L ; val = mu_g + beta*sd_g.   */
int dfh_gp_add_ucb_group(dfh_gp* gp, int32_t group, double beta, const double* Xg, int64_t m,
                         double* vals_out, double* best_val, int64_t* best_idx);
/* All groups of the additive model in one call: group g's candidates Xg_all + sum_{h<g} m_h*|group h|
 * ([m_g x |group g|]), betas[g], results best_vals[g] / best_idx[g] (index within the group's
 * candidates), vals_out [sum m_g] or NULL.  The G triangular solves of the reference
 * (opt/gpb_acquisitions.py:161-176) become one solve with sum(m_g) right-hand sides.            */
int dfh_gp_add_ucb_all(dfh_gp* gp, const double* betas, const double* Xg_all, const int64_t* m_per_group,
                       double* vals_out, double* best_vals, int64_t* best_idx);

/* ---- candidate generation on the device ---------------------------------------------------
 * The m x d block of uniform candidates every random-search acquisition draws,
 *   np.random.random((max_evals, dim))                   (dragonfly/utils/oper_utils.py:62)
 *   pts * (bounds[:,1] - bounds[:,0]) + bounds[:,0]      (dragonfly/utils/general_utils.py:25-27)
 * produced in HBM, bit for bit what NumPy produces from the same generator state, so neither the
 * host generation nor the m x d host-to-device copy remains.  `bounds` is a HOST array [d][2]
 * (lo, hi) or NULL for the unit cube; `out` [m x d] may be a dfh_malloc() pointer (the candidates
 * then stay on the device for dfh_gp_acq_argmax / dfh_gp_ts / dfh_gp_predict) or a host pointer.
 *
 * dfh_rand_mt19937_uniform continues a legacy NumPy state -- `key`[624] and `pos` of
 * np.random.get_state() -- and updates both in place to the state NumPy would be left in, so the
 * caller hands them back with np.random.set_state() and later host draws continue the stream.
 * dfh_rand_philox_uniform does the same for numpy.random.Philox (Philox4x64-10): key[2], and in/out
 * counter[4], buffer[4], buffer_pos of the bit generator's state.
 * Only rows [row_begin, row_begin + row_count) of the m x d block are written (`out` is
 * [row_count x d]) while the state advances over the whole block: every rank of a multi-GPU run
 * starts from the same state, keeps its own shard of the candidates, and ends in the same state
 * (one process draws them all in the reference).  Philox computes the shard's blocks only; for
 * MT19937 the words before and after the shard are jumped over (polynomial jump-ahead on the host,
 * a few ms whatever the distance) once there are more than 2^23 of them, walked otherwise.
 *
 * dfh_mt19937_advance (host only, no device needed): key / pos after n_words further 32-bit words
 * of the stream, i.e. what np.random.get_state() shows after random_sample(n_words / 2).         */
int dfh_rand_mt19937_uniform(dfh_ctx* ctx, uint32_t* key, int32_t* pos, int64_t m, int64_t d,
                             int64_t row_begin, int64_t row_count, const double* bounds, double* out);
int dfh_mt19937_advance(uint32_t* key, int32_t* pos, int64_t n_words);
/* np.random.normal(size=m) from the same legacy state -- the standard normals draw_gaussian_samples
 * takes (dragonfly/utils/general_utils.py:230; Thompson sampling, gp_core.py:250-254) -- produced in
 * HBM bit for bit: NumPy's legacy_gauss (polar method with rejection on the MT19937 double stream).
 * has_gauss / gauss: in/out, the cached second normal of np.random.get_state()[3:5]; key / pos are
 * updated as by the uniform entry point.  `out` [m] may be a host or a device pointer.  log() is
 * evaluated in double-double on the device; the ~4 % of pairs whose log lies within 0.025 ulp of a
 * rounding boundary are decided by the host C library's log() (what NumPy itself calls).        */
int dfh_rand_mt19937_normal(dfh_ctx* ctx, uint32_t* key, int32_t* pos, int32_t* has_gauss, double* gauss,
                            int64_t m, double* out);
int dfh_rand_philox_uniform(dfh_ctx* ctx, const uint64_t* key, uint64_t* counter, uint64_t* buffer,
                            int32_t* buffer_pos, int64_t m, int64_t d, int64_t row_begin,
                            int64_t row_count, const double* bounds, double* out);

/* ---- multi-GPU: candidate shards across the GPUs of a node ------------
 * The reference has no collective: every candidate of a random-search acquisition sits in ONE
 * array in ONE process and the winner is obj_vals.argmax() (dragonfly/utils/oper_utils.py:59-80,
 * line 73).  Candidates are independent given the fitted GP, so they shard contiguously over the
 * devices; the fit is replicated (identical on every device; one n = 16384 factorisation does not
 * shard profitably over xGMI) and the ONLY exchange is an RCCL all-gather of one 16-byte
 * (value:f64, global row index:i64) pair per rank, followed by the same deterministic reduce on
 * every rank -- first NaN wins, else the largest value, ties to the lowest global index: exactly
 * np.argmax over the whole set (RCCL has no MAXLOC; all-reduce(max) alone would lose the index).
 * RCCL is dlopen'ed on first use; no PyTorch anywhere.                                         */

/* Host-only pieces of the contract (no GPU needed).
 * dfh_shard_bounds: rows [*lo, *hi) of rank's contiguous shard of m candidates, shard edges on
 * multiples of `align` (the Thompson block size, so the blocks -- hence the joint draws -- are
 * those of a single device doing it all).
 * dfh_reduce_argmax: the reduce above over `count` (value, global index) pairs; pairs with
 * idx < 0 (empty shards) are skipped; no pair at all gives (NaN, -1).                          */
int dfh_shard_bounds(int64_t m, int rank, int world, int64_t align, int64_t* lo, int64_t* hi);
int dfh_reduce_argmax(const double* vals, const int64_t* idxs, int count, double* best_val,
                      int64_t* best_idx);

/* (1) One process per GPU.  Rank 0 calls dfh_comm_unique_id (ncclGetUniqueId) and hands the
 * DFH_UNIQUE_ID_BYTES to the other processes by whatever means the host side has; every process
 * then calls dfh_comm_create on its own context (ncclCommInitRank: blocks until all arrived).  */
#define DFH_UNIQUE_ID_BYTES 128
typedef struct dfh_comm dfh_comm;
int  dfh_comm_unique_id(void* id_out /* [DFH_UNIQUE_ID_BYTES] */);
int  dfh_comm_create(dfh_ctx* ctx, int nranks, int rank, const void* id, dfh_comm** out);
void dfh_comm_destroy(dfh_comm* comm);
int  dfh_comm_rank(dfh_comm* comm);
int  dfh_comm_size(dfh_comm* comm);
/* The communicator as RCCL reports it: ranks it formed (ncclCommCount; 0 for the host-exchange test mode), this
 * rank (ncclCommUserRank), library version (ncclGetVersion, e.g. 22707).  No reference counterpart: the reference
 * is one process (dragonfly/utils/oper_utils.py:59-80 finds its arg-max in one array).                          */
int  dfh_comm_info(dfh_comm* comm, int32_t* ranks_formed, int32_t* rank, int32_t* rccl_version);
/* all-gather of (local_val, local_idx) on the context's stream + the reduce; identical result on
 * every rank.  local_idx is the GLOBAL row index (or < 0 for an empty shard).                   */
int  dfh_comm_allgather_argmax(dfh_comm* comm, double local_val, int64_t local_idx,
                               double* best_val, int64_t* best_idx);
/* recv[r*count + j] = rank r's send[j] (host buffers, count <= 4096): the winning candidate's
 * coordinates travelling from its owner to every rank.                                          */
int  dfh_comm_allgather_f64(dfh_comm* comm, const double* send, int count, double* recv);
/* plumbing for benchmarks: element-wise max of inout[count <= 64] over the ranks; barrier.      */
int  dfh_comm_allreduce_max(dfh_comm* comm, double* inout, int count);
int  dfh_comm_barrier(dfh_comm* comm);

/* (2) One process, N devices: a context per device, a host thread per device for the blocking
 * per-device work, one communicator clique (ncclCommInitAll).  device_ids NULL = devices 0..N-1.
 * Fails with DFH_ERR_BAD_ARG when fewer than n_devices are visible.                             */
typedef struct dfh_mgpu dfh_mgpu;
int      dfh_mgpu_create(int n_devices, const int* device_ids, dfh_mgpu** out);
void     dfh_mgpu_destroy(dfh_mgpu* mg);
int      dfh_mgpu_size(dfh_mgpu* mg);
dfh_ctx* dfh_mgpu_ctx(dfh_mgpu* mg, int rank);    /* rank's context (for dfh_malloc / uploads)  */
dfh_gp*  dfh_mgpu_gp(dfh_mgpu* mg, int rank);     /* rank's replica of the current fit, or NULL */
dfh_comm* dfh_mgpu_comm(dfh_mgpu* mg, int rank);
int      dfh_mgpu_sync(dfh_mgpu* mg);
/* Replicated dfh_gp_fit on every device (GP.build_posterior, gp/gp_core.py:155-163).  X[r],
 * y_centred[r]: rank r's copy -- a host pointer (the same one may be given for all ranks) or a
 * pointer into rank r's HBM.  lml, jitter_power: [n_devices] or NULL.  Replaces the previous fit. */
int dfh_mgpu_fit(dfh_mgpu* mg, const dfh_kernel_desc* k, const double* const* X, int64_t n, int64_t d,
                 const double* const* y_centred, double noise_var, int flags, double* lml,
                 int32_t* jitter_power);
int dfh_mgpu_free_fit(dfh_mgpu* mg);
/* dfh_gp_ts / dfh_gp_acq_argmax over contiguous shards: rank r holds rows [off_r, off_r + m[r]) of
 * the global candidate set, off_r = m[0] + ... + m[r-1]; Xs[r] (and U[r], the shard's standard
 * normals) are host pointers or pointers into rank r's HBM; m[r] == 0 is an empty shard.
 * best_idx is the GLOBAL row index, identical to the single-device call on the concatenated set
 * (for dfh_mgpu_ts when the shards are cut on multiples of `block`).  local_vals / local_idx:
 * optional [n_devices], each rank's own winner (global index) before the exchange.              */
int dfh_mgpu_ts(dfh_mgpu* mg, const double* const* Xs, const int64_t* m, int64_t block,
                const double* const* U, double mean_const, double* best_val, int64_t* best_idx,
                double* local_vals, int64_t* local_idx);
int dfh_mgpu_acq_argmax(dfh_mgpu* mg, int acq, const double* params, const double* const* Xs,
                        const int64_t* m, double mean_const, double* best_val, int64_t* best_idx,
                        double* local_vals, int64_t* local_idx);
/* the exchange alone: vals / idxs [n_devices] in, the reduced pair out                          */
int dfh_mgpu_allgather_argmax(dfh_mgpu* mg, const double* vals, const int64_t* idxs,
                              double* best_val, int64_t* best_idx);

/* ---- timing of the last call's dominant kernels (HIP events, ms) ------------------------- */
#define DFH_T_KERNMAT  0   /* training kernel-matrix build                                   */
#define DFH_T_CHOL     1   /* blocked Cholesky (all launches)                                */
#define DFH_T_SOLVE    2   /* alpha solves + lml                                             */
#define DFH_T_CROSS    3   /* cross kernel matrices                                          */
#define DFH_T_TRSM     4   /* posterior triangular solves                                    */
#define DFH_T_ACQ      5   /* variance + acquisition + arg-max                               */
#define DFH_T_TS       6   /* TS block covariance + cholesky + sample                        */
#define DFH_T_COUNT    8
int dfh_ctx_timings(dfh_ctx* ctx, int enable, double* ms_out /* [DFH_T_COUNT] or NULL */);

/* Counters of the context since it was created (diagnostics; nothing in the reference corresponds):
 * out[0] = factorisations repeated on the schedule without inter-workgroup hand-offs because a bounded
 * wait expired or a resident panel's block inverse was too poor (the result is the same, the call slower);
 * out[1] = how many of the next factorisations go straight to that schedule (set after two such repeats in
 * a row, e.g. on a device shared with other work); out[2..3] reserved (0).                          */
int dfh_ctx_counters(dfh_ctx* ctx, int64_t* out /* [4] */);

/* Per-launch HIP-event timing of the fp64 MFMA GEMM kernel (the dominant kernel of the path),
 * recorded on the stream each launch goes to.  Returns the totals since the last call in
 * stats_out[8][5] = per kernel variant {launches, sum of launch durations in ms, algorithmic
 * flop, busy ms = length of the union of the launch intervals (launches of different streams
 * overlap), algorithmic bytes = each operand once + the output tile written (and read when it
 * is updated)}; variant index = 4*(B is [K x N]) + 2*(edge path) + (64x64 tiles), so variant 0 is
 * the 128x128 NT throughput configuration.  stats_out may be NULL.  Then enables (1) / disables
 * (0) further recording.                                                                        */
int dfh_ctx_gemm_profile(dfh_ctx* ctx, int enable, double* stats_out /* [40] or NULL */);

#ifdef __cplusplus
}
#endif
#endif /* DFHIP_H */
