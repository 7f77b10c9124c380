// Projection of a symmetric matrix onto the cone of matrices with eigenvalues >= epsilon:
//     M -> V max(Lambda, epsilon) V^T,   M = V Lambda V^T
// Replaces project_symmetric_to_psd_cone (dragonfly/utils/general_utils.py:150-163: np.linalg.eigh,
// np.clip, (eigvecs * clipped).dot(eigvecs.T)) as used by _get_cholesky_decomp's 'project_first'
// / 'try_before_project' branches (gp/gp_core.py:827-840: the Gram matrix of a kernel that is not
// guaranteed PSD -- Cartesian-product and neural-network GPs) and by get_post_covar_from_raw_covar
// (gp_core.py:849-857: the posterior covariance of such a GP, epsilon = 0.05 noise_var).
//
// No eigen-decomposition is formed.  With B = M - epsilon I,
//     V max(Lambda, epsilon) V^T = epsilon I + (B + B sign(B)) / 2,
// and the matrix sign function comes from the Newton-Schulz iteration
//     X_0 = B / |B|_F,   X_{k+1} = X_k (3 I - X_k^2) / 2          (X_k -> sign(B), eigenvalue-wise)
// -- nothing but the fp64 MFMA GEMM this library already runs at 80 % of peak (two products per
// step, each computed on the lower triangle and mirrored), where a hand-written tridiagonalisation
// + QL would be a long chain of latency-bound kernels.  An eigenvalue lambda of B starts at
// lambda / |B|_F, grows by 3/2 per step until it is O(1) and then converges quadratically:
// PSD_ITERS = 96 steps settle every eigenvalue above 1e-16 |B|_F; smaller ones are left
// unconverged, which perturbs the result by at most 2 |lambda| -- below the rounding of the
// reference's own V Lambda V^T product.  The iteration stops earlier once a step no longer moves X
// (|X_{k+1} - X_k|_F <= 1e-15 sqrt(n), checked every eighth step from the sixteenth on): an
// eigenvalue still on its way then sits below 2e-15 |B|_F, the same class of perturbation; a
// spectrum bounded away from zero is done after ~25 steps instead of 96.  A non-finite norm (NaN /
// Inf in the input) is an error, as np.linalg.eigh's LinAlgError is in the reference.
#include "common.h"
#include <math.h>
#include <algorithm>
#include <utility>

namespace {

constexpr int PSD_ITERS = 96;

// sum of squares of a symmetric n x n matrix (full storage), two-stage, deterministic
__global__ __launch_bounds__(256) void k_sumsq_partial(const double* __restrict__ A, long n, long lda,
                                                        double* __restrict__ part) {
  __shared__ double sm[256];
  const long total = n * n;
  double s = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const double v = A[(i / n) * lda + (i % n)];
    s = fma(v, v, s);
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) sm[threadIdx.x] += sm[threadIdx.x + st]; __syncthreads(); }
  if (threadIdx.x == 0) part[blockIdx.x] = sm[0];
}
__global__ __launch_bounds__(256) void k_sum_final(const double* __restrict__ part, int nparts, double* __restrict__ out) {
  __shared__ double sm[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) sm[threadIdx.x] += sm[threadIdx.x + st]; __syncthreads(); }
  if (threadIdx.x == 0) out[0] = sm[0];
}

// partial sums of squares of X - Y (dense n x n, ld n): the size of a Newton-Schulz step
__global__ __launch_bounds__(256) void k_diff_sumsq_partial(const double* __restrict__ X, const double* __restrict__ Y,
                                                             long total, double* __restrict__ part) {
  __shared__ double sm[256];
  double s = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const double v = X[i] - Y[i];
    s = fma(v, v, s);
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) sm[threadIdx.x] += sm[threadIdx.x + st]; __syncthreads(); }
  if (threadIdx.x == 0) part[blockIdx.x] = sm[0];
}

// B = M - eps I   (dense n x n, ld n)
__global__ void k_shift(const double* __restrict__ M, long n, long ldm, double eps, double* __restrict__ B) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * n) return;
  const long r = i / n, c = i % n;
  B[i] = M[r * ldm + c] - (r == c ? eps : 0.0);
}
// X = B / |B|_F
__global__ void k_normalise(const double* __restrict__ B, long n, const double* __restrict__ sumsq, double* __restrict__ X) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * n) return;
  const double nrm = sqrt(sumsq[0]);
  X[i] = nrm > 0.0 ? B[i] / nrm : 0.0;
}

// lower triangle of T holds X^2: T <- 3 I - T on the lower triangle, mirrored to the upper
__global__ void k_three_minus_mirror(double* __restrict__ T, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * n) return;
  const long r = i / n, c = i % n;
  if (c > r) return;
  const double v = (r == c ? 3.0 : 0.0) - T[r * n + c];
  T[r * n + c] = v;
  T[c * n + r] = v;
}
// mirror the lower triangle to the upper
__global__ void k_mirror_lower(double* __restrict__ T, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * n) return;
  const long r = i / n, c = i % n;
  if (c < r) T[c * n + r] = T[r * n + c];
}
// out = eps I + (B + P) / 2 with P's lower triangle valid (P = B sign(B), symmetric): written symmetric
__global__ void k_combine(const double* __restrict__ B, const double* __restrict__ P, long n, double eps,
                          double* __restrict__ out, long ldo) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * n) return;
  const long r = i / n, c = i % n;
  if (c > r) return;
  const double v = 0.5 * (B[r * n + c] + P[r * n + c]) + (r == c ? eps : 0.0);
  out[r * ldo + c] = v;
  out[c * ldo + r] = v;
}

}  // namespace

// out (ldo) <- projection of the symmetric M (ldm) onto {eigenvalues >= eps}; out may alias M.
// Scratch: SCR_KCT2 (4 n^2 doubles), SCR_RED.  Asynchronous on ctx->stream.
int psd_project_device(dfh_ctx* ctx, const double* M, int64_t n, int64_t ldm, double eps, double* out, int64_t ldo) {
  if (n <= 0) return DFH_OK;
  double* W = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_KCT2, (size_t)4 * n * n * 8, (void**)&W));
  double* B = W; double* Xc = W + n * n; double* Xn = W + 2 * n * n; double* T = W + 3 * n * n;
  double* red = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_RED, (1024 + 8) * 8, (void**)&red));
  const unsigned g = (unsigned)((n * n + 255) / 256);
  hipLaunchKernelGGL(k_shift, dim3(g), dim3(256), 0, ctx->stream, M, (long)n, (long)ldm, eps, B);
  DFH_LAUNCH_CHECK();
  const int nparts = (int)std::min<int64_t>(1024, (n * n + 255) / 256);
  hipLaunchKernelGGL(k_sumsq_partial, dim3((unsigned)nparts), dim3(256), 0, ctx->stream, B, (long)n, (long)n, red);
  DFH_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(256), 0, ctx->stream, red, nparts, red + 1024);
  DFH_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_normalise, dim3(g), dim3(256), 0, ctx->stream, B, (long)n, red + 1024, Xc);
  DFH_LAUNCH_CHECK();
  {
    double h_sumsq = 0.0;
    DFH_HIP(hipMemcpyAsync(&h_sumsq, red + 1024, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DFH_HIP(hipStreamSynchronize(ctx->stream));
    if (!(h_sumsq == h_sumsq) || h_sumsq > 1.7e308) {     // NaN or Inf: eigh would not converge either
      dfh_set_error("PSD projection: the matrix has non-finite entries");
      return DFH_ERR_NOT_PD;
    }
  }
  static const bool early_stop = []() { const char* e = getenv("DFH_PSD_EARLY_STOP"); return e ? atoi(e) != 0 : true; }();
  for (int it = 0; it < PSD_ITERS; ++it) {
    // T = X X^T (= X^2, X symmetric) on the lower triangle; T <- 3 I - T, mirrored
    DFH_TRY(gemm_f64(ctx, GEMM_LOWER, n, n, n, 1.0, Xc, n, Xc, n, 0.0, nullptr, 0, T, n));
    hipLaunchKernelGGL(k_three_minus_mirror, dim3(g), dim3(256), 0, ctx->stream, T, (long)n);
    DFH_LAUNCH_CHECK();
    // X <- X (3 I - X^2) / 2: symmetric, lower triangle computed and mirrored
    DFH_TRY(gemm_f64(ctx, GEMM_LOWER, n, n, n, 0.5, Xc, n, T, n, 0.0, nullptr, 0, Xn, n));
    hipLaunchKernelGGL(k_mirror_lower, dim3(g), dim3(256), 0, ctx->stream, Xn, (long)n);
    DFH_LAUNCH_CHECK();
    std::swap(Xc, Xn);
    if (early_stop && it + 1 >= 16 && (it + 1) % 8 == 0 && it + 1 < PSD_ITERS) {
      hipLaunchKernelGGL(k_diff_sumsq_partial, dim3((unsigned)nparts), dim3(256), 0, ctx->stream, Xc, Xn, (long)(n * n), red);
      DFH_LAUNCH_CHECK();
      hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(256), 0, ctx->stream, red, nparts, red + 1025);
      DFH_LAUNCH_CHECK();
      double step2 = 1.0;
      DFH_HIP(hipMemcpyAsync(&step2, red + 1025, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      DFH_HIP(hipStreamSynchronize(ctx->stream));
      if (step2 <= 1e-30 * (double)n) break;
    }
  }
  // P = B sign(B) (symmetric; lower triangle), out = eps I + (B + P) / 2
  DFH_TRY(gemm_f64(ctx, GEMM_LOWER, n, n, n, 1.0, B, n, Xc, n, 0.0, nullptr, 0, T, n));
  hipLaunchKernelGGL(k_combine, dim3(g), dim3(256), 0, ctx->stream, B, T, (long)n, eps, out, (long)ldo);
  DFH_LAUNCH_CHECK();
  return DFH_OK;
}

// project_symmetric_to_psd_cone (general_utils.py:150-163) for a caller-held matrix
extern "C" int dfh_project_psd(dfh_ctx* ctx, const double* M, int64_t n, double epsilon, double* out) {
  DFH_ARG(ctx && n >= 0);
  if (n == 0) return DFH_OK;
  DFH_ARG(M && out);
  DFH_HIP(hipSetDevice(ctx->device));
  const double* dM = nullptr;
  DFH_TRY(to_device(ctx, M, (size_t)n * n * 8, SCR_TSK, &dM));
  const bool dev_out = is_device_ptr(out);
  double* dO = out;
  if (!dev_out) DFH_TRY(scratch_get(ctx, SCR_TSL, (size_t)n * n * 8, (void**)&dO));
  DFH_TRY(psd_project_device(ctx, dM, n, n, epsilon, dO, n));
  if (!dev_out) DFH_TRY(from_device(ctx, out, dO, (size_t)n * n * 8));
  DFH_HIP(hipStreamSynchronize(ctx->stream));
  return DFH_OK;
}
