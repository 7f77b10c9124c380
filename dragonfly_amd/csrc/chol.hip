// Blocked right-looking Cholesky + the triangular solves built on its diagonal-block inverses.
//
// Replaces np.linalg.cholesky (LAPACK dpotrf) at dragonfly/utils/general_utils.py:178,190 and
// scipy.linalg.solve_triangular (dtrtrs) at general_utils.py:213 as used by
// GP.build_posterior / GP.eval (dragonfly/gp/gp_core.py:159-163,180).
//
// Structure (row-major lower, n x n):
//   outer panels of CHOL_NB = 512 columns; inside a panel the 512 x 512 diagonal block is
//   factored with 64-wide steps: a single-workgroup LDS kernel factors the 64 x 64 pivot block
//   and inverts it (potf2_inv64), the 64-wide column below it is solved by multiplying with
//   that inverse (MFMA GEMM) and the rest of the diagonal block is updated by an MFMA SYRK.
//   The eight 64-block inverses are then merged into the inverse of the 512 block by three
//   levels of batched GEMMs ([[A,0],[B,C]]^-1 = [[A^-1,0],[-C^-1 B A^-1, C^-1]]), so the panel
//   solve  L21 = A21 L11^-T  and the trailing update  A22 -= L21 L21^T  are two large MFMA_F64
//   GEMMs -- where n^3/3 of the flops are.  The 512-block inverses are kept: the posterior
//   solve (trsm_rows) and the alpha solves (trsv_*) reuse them, which turns every triangular
//   solve on the hot path into GEMM / GEMV work.
#include "common.h"
#include <math.h>

namespace {

constexpr int PB = 64;       // pivot block
constexpr int PBP = 65;      // LDS row stride

// One workgroup: factor the nb x nb (nb <= 64) block at A in place (upper part zeroed) and
// write its inverse (row-major, upper part zero) to inv.  info[0] <- pivot_base + j + 1 for the
// first non-positive / NaN pivot (only the first failure of a factorisation is recorded).
__global__ __launch_bounds__(256) void potf2_inv64_kernel(double* __restrict__ A, long lda, int nb,
                                                          double* __restrict__ inv, long ldinv,
                                                          long pivot_base, long long* info,
                                                          int do_factor) {
  __shared__ double S[PB * PBP];
  const int tid = threadIdx.x;
  for (int idx = tid; idx < PB * PB; idx += 256) {
    const int i = idx >> 6, k = idx & 63;
    double v = (i == k) ? 1.0 : 0.0;                  // identity padding beyond nb
    if (i < nb && k < nb) v = (k <= i) ? A[i * lda + k] : 0.0;
    S[i * PBP + k] = v;
  }
  __syncthreads();

  for (int j = 0; do_factor && j < nb; ++j) {
    const double d = S[j * PBP + j];
    if (!(d > 0.0)) {                                  // uniform: every thread reads the same d
      if (tid == 0) {
        if (info[0] == 0) info[0] = pivot_base + j + 1;
      }
      return;
    }
    const double sd = sqrt(d);
    __syncthreads();                                   // everyone has read S[j][j]
    if (tid > j && tid < nb) S[tid * PBP + j] = S[tid * PBP + j] / sd;
    if (tid == 0) S[j * PBP + j] = sd;
    __syncthreads();
    // rank-1 update of the trailing lower triangle
    const int k = j + 1 + (tid & 63);
    for (int i = j + 1 + (tid >> 6); i < nb; i += 4) {
      if (k <= i) S[i * PBP + k] = fma(-S[i * PBP + j], S[k * PBP + j], S[i * PBP + k]);
    }
    __syncthreads();
  }

  if (do_factor) {
    for (int idx = tid; idx < nb * nb; idx += 256) {
      const int i = idx / nb, k = idx - i * nb;
      A[i * lda + k] = (k <= i) ? S[i * PBP + k] : 0.0;
    }
  }

  // inverse by back substitution on rows:  x_r L = e_r ; lane r owns row r, L entries are
  // wave-uniform LDS broadcasts.  Padded rows/cols are identity so the full 64-loop is safe.
  if (tid < 64) {
    const int r = tid;
    double x[PB];
#pragma unroll
    for (int c = PB - 1; c >= 0; --c) {
      double s = (r == c) ? 1.0 : 0.0;
#pragma unroll
      for (int kk = c + 1; kk < PB; ++kk) s = fma(-x[kk], S[kk * PBP + c], s);
      x[c] = s / S[c * PBP + c];
    }
    if (r < nb) {
#pragma unroll
      for (int c = 0; c < PB; ++c)
        if (c < nb) inv[r * ldinv + c] = (c <= r) ? x[c] : 0.0;
    }
  }
}

// Merge the 64-block inverses on the diagonal of Linv (ld = CHOL_NB) into the inverse of the
// nbk x nbk lower-triangular block D (ld = lda) by recursive doubling:
//   [[A,0],[B,C]]^-1 = [[A^-1,0],[-C^-1 B A^-1, C^-1]]
int assemble_block_inverse(dfh_ctx* ctx, const double* D, int64_t lda, int64_t nbk, double* Linv,
                           double* T) {
  const int64_t NB = CHOL_NB;
  for (int64_t s = PB; s < nbk; s *= 2) {
    const int64_t full_pairs = nbk / (2 * s);
    if (full_pairs > 0) {
      GemmBatch b1, b2;
      b1.count = b2.count = (int)full_pairs;
      // T_q = B_q * A_q^-1 : B_q = D[hi rows, lo cols], A_q^-1 = Linv[lo, lo]
      b1.sA = 2 * s * (lda + 1); b1.sB = 2 * s * (NB + 1); b1.sCout = s * s;
      DFH_TRY(gemm_f64(ctx, GEMM_TRANSB, s, s, s, 1.0, D + s * lda, lda, Linv, NB, 0.0, nullptr, 0,
                       T, s, &b1));
      // X_q = -C_q^-1 * T_q -> Linv[hi rows, lo cols]
      b2.sA = 2 * s * (NB + 1); b2.sB = s * s; b2.sCout = 2 * s * (NB + 1);
      DFH_TRY(gemm_f64(ctx, GEMM_TRANSB, s, s, s, -1.0, Linv + s * (NB + 1), NB, T, s, 0.0, nullptr,
                       0, Linv + s * NB, NB, &b2));
    }
    const int64_t lo = full_pairs * 2 * s, hi = lo + s;
    if (hi < nbk) {                                   // trailing partial pair
      const int64_t hs = nbk - hi;
      DFH_TRY(gemm_f64(ctx, GEMM_TRANSB, hs, s, s, 1.0, D + hi * lda + lo, lda, Linv + lo * (NB + 1),
                       NB, 0.0, nullptr, 0, T, s));
      DFH_TRY(gemm_f64(ctx, GEMM_TRANSB, hs, s, hs, -1.0, Linv + hi * (NB + 1), NB, T, s, 0.0,
                       nullptr, 0, Linv + hi * NB + lo, NB));
    }
  }

  return DFH_OK;
}

}  // namespace

int cholesky_device(dfh_ctx* ctx, double* A, int64_t n, int64_t lda, double* keep_inv,
                    int64_t* info_pivot) {
  if (info_pivot) *info_pivot = 0;
  if (n <= 0) return DFH_OK;
  const int64_t NB = CHOL_NB;
  long long* d_info = reinterpret_cast<long long*>(ctx->d_info);
  DFH_HIP(hipMemsetAsync(d_info, 0, 8, ctx->stream));

  double* inv_scratch = nullptr;
  if (!keep_inv) DFH_TRY(scratch_get(ctx, SCR_CHOLINV, (size_t)NB * NB * 8, (void**)&inv_scratch));
  double* T = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_CHOLT, (size_t)NB * NB * 8, (void**)&T));
  double* W = nullptr;
  if (n > NB) DFH_TRY(scratch_get(ctx, SCR_CHOLW, (size_t)(n - NB) * NB * 8, (void**)&W));

  for (int64_t k0 = 0; k0 < n; k0 += NB) {
    const int64_t nbk = (n - k0 < NB) ? n - k0 : NB;
    double* Linv = keep_inv ? keep_inv + (k0 / NB) * NB * NB : inv_scratch;
    DFH_HIP(hipMemsetAsync(Linv, 0, (size_t)NB * NB * 8, ctx->stream));
    double* D = A + k0 * lda + k0;

    // ---- factor the diagonal block with 64-wide steps -------------------------------------
    for (int64_t j0 = 0; j0 < nbk; j0 += PB) {
      const int w = (int)((nbk - j0 < PB) ? nbk - j0 : PB);
      double* Djj = D + j0 * lda + j0;
      double* Ijj = Linv + j0 * NB + j0;
      hipLaunchKernelGGL(potf2_inv64_kernel, dim3(1), dim3(256), 0, ctx->stream, Djj, (long)lda, w,
                         Ijj, (long)NB, (long)(k0 + j0), d_info, 1);
      DFH_LAUNCH_CHECK();
      const int64_t rows = nbk - j0 - w;
      if (rows > 0) {
        double* P = D + (j0 + w) * lda + j0;                       // rows x w
        // P <- P * Ljj^-T   (single column tile: in-place safe)
        DFH_TRY(gemm_f64(ctx, 0, rows, w, w, 1.0, P, lda, Ijj, NB, 0.0, nullptr, 0, P, lda));
        // D22 <- D22 - P P^T (lower)
        double* D22 = D + (j0 + w) * lda + (j0 + w);
        DFH_TRY(gemm_f64(ctx, GEMM_LOWER, rows, rows, w, -1.0, P, lda, P, lda, 1.0, D22, lda, D22, lda));
      }
    }

    DFH_TRY(assemble_block_inverse(ctx, D, lda, nbk, Linv, T));

    // ---- panel solve and trailing update --------------------------------------------------
    const int64_t rem = n - k0 - nbk;
    if (rem > 0) {
      double* A21 = A + (k0 + nbk) * lda + k0;          // rem x nbk
      DFH_TRY(copy_matrix(ctx, A21, lda, W, NB, rem, nbk));
      DFH_TRY(gemm_f64(ctx, GEMM_KTRI_B, rem, nbk, nbk, 1.0, W, NB, Linv, NB, 0.0, nullptr, 0, A21, lda));
      double* A22 = A + (k0 + nbk) * lda + (k0 + nbk);
      DFH_TRY(gemm_f64(ctx, GEMM_LOWER, rem, rem, nbk, -1.0, A21, lda, A21, lda, 1.0, A22, lda, A22, lda));
    }
  }

  DFH_HIP(hipMemcpyAsync(ctx->h_info, d_info, 8, hipMemcpyDeviceToHost, ctx->stream));
  DFH_HIP(hipStreamSynchronize(ctx->stream));
  const int64_t piv = ctx->h_info[0];
  if (info_pivot) *info_pivot = piv;
  if (piv != 0) {
    dfh_set_error("Matrix is not positive definite (pivot %lld)", (long long)piv);
    return DFH_ERR_NOT_PD;
  }
  return DFH_OK;
}

int trsv_forward(dfh_ctx* ctx, const double* L, int64_t n, int64_t ldl, const double* inv,
                 double* x) {
  const int64_t NB = CHOL_NB;
  double* tmp = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_VEC3, (size_t)NB * 8, (void**)&tmp));
  for (int64_t c0 = 0; c0 < n; c0 += NB) {
    const int64_t w = (n - c0 < NB) ? n - c0 : NB;
    // x_i <- x_i - L[i, 0:c0] x[0:c0]
    if (c0 > 0) DFH_TRY(gemv_rows(ctx, L + c0 * ldl, w, c0, ldl, x, -1.0, x + c0, 1.0, x + c0));
    // x_i <- Linv_ii x_i
    DFH_TRY(gemv_rows(ctx, inv + (c0 / NB) * NB * NB, w, w, NB, x + c0, 1.0, nullptr, 0.0, tmp));
    DFH_HIP(hipMemcpyAsync(x + c0, tmp, (size_t)w * 8, hipMemcpyDeviceToDevice, ctx->stream));
  }
  return DFH_OK;
}

int trsv_backward(dfh_ctx* ctx, const double* L, int64_t n, int64_t ldl, const double* inv,
                  double* x) {
  const int64_t NB = CHOL_NB;
  double* tmp = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_VEC3, (size_t)NB * 8, (void**)&tmp));
  const int64_t nblk = (n + NB - 1) / NB;
  for (int64_t b = nblk - 1; b >= 0; --b) {
    const int64_t c0 = b * NB;
    const int64_t w = (n - c0 < NB) ? n - c0 : NB;
    const int64_t below = n - c0 - w;
    // x_i <- x_i - L[i+1:, i]^T x[i+1:]
    if (below > 0)
      DFH_TRY(gemv_cols(ctx, L + (c0 + w) * ldl + c0, below, w, ldl, x + c0 + w, -1.0, x + c0, 1.0, x + c0));
    // x_i <- Linv_ii^T x_i
    DFH_TRY(gemv_cols(ctx, inv + b * NB * NB, w, w, NB, x + c0, 1.0, nullptr, 0.0, tmp));
    DFH_HIP(hipMemcpyAsync(x + c0, tmp, (size_t)w * 8, hipMemcpyDeviceToDevice, ctx->stream));
  }
  return DFH_OK;
}

int trsm_rows(dfh_ctx* ctx, const double* L, int64_t n, int64_t ldl, const double* inv,
              double* Kct, int64_t m, int64_t ldk) {
  if (m <= 0 || n <= 0) return DFH_OK;
  const int64_t NB = CHOL_NB;
  double* T = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_TMP, (size_t)m * NB * 8, (void**)&T));
  for (int64_t c0 = 0; c0 < n; c0 += NB) {
    const int64_t w = (n - c0 < NB) ? n - c0 : NB;
    // T = Kct[:, c0:c0+w] - Vt[:, 0:c0] * L[c0:c0+w, 0:c0]^T      (K = 0 degenerates to a copy)
    DFH_TRY(gemm_f64(ctx, 0, m, w, c0, -1.0, Kct, ldk, L + c0 * ldl, ldl, 1.0, Kct + c0, ldk, T, NB));
    // Vt[:, c0:c0+w] = T * Linv_ii^T
    DFH_TRY(gemm_f64(ctx, GEMM_KTRI_B, m, w, w, 1.0, T, NB, inv + (c0 / NB) * NB * NB, NB, 0.0,
                     nullptr, 0, Kct + c0, ldk));
  }
  return DFH_OK;
}

int tri_block_inverses(dfh_ctx* ctx, const double* L, int64_t n, int64_t ldl, double* inv) {
  const int64_t NB = CHOL_NB;
  double* T = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_CHOLT, (size_t)NB * NB * 8, (void**)&T));
  long long* d_info = reinterpret_cast<long long*>(ctx->d_info);
  for (int64_t k0 = 0; k0 < n; k0 += NB) {
    const int64_t nbk = (n - k0 < NB) ? n - k0 : NB;
    double* Linv = inv + (k0 / NB) * NB * NB;
    DFH_HIP(hipMemsetAsync(Linv, 0, (size_t)NB * NB * 8, ctx->stream));
    const double* D = L + k0 * ldl + k0;
    for (int64_t j0 = 0; j0 < nbk; j0 += PB) {
      const int w = (int)((nbk - j0 < PB) ? nbk - j0 : PB);
      hipLaunchKernelGGL(potf2_inv64_kernel, dim3(1), dim3(256), 0, ctx->stream,
                         const_cast<double*>(D + j0 * ldl + j0), (long)ldl, w, Linv + j0 * NB + j0,
                         (long)NB, (long)(k0 + j0), d_info, 0);
      DFH_LAUNCH_CHECK();
    }
    DFH_TRY(assemble_block_inverse(ctx, D, ldl, nbk, Linv, T));
  }
  return DFH_OK;
}

int trsm_rows_backward(dfh_ctx* ctx, const double* L, int64_t n, int64_t ldl, const double* inv,
                       double* Bt, int64_t m, int64_t ldb) {
  if (m <= 0 || n <= 0) return DFH_OK;
  const int64_t NB = CHOL_NB;
  double* T = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_TMP, (size_t)m * NB * 8, (void**)&T));
  const int64_t nblk = (n + NB - 1) / NB;
  for (int64_t b = nblk - 1; b >= 0; --b) {
    const int64_t c0 = b * NB;
    const int64_t w = (n - c0 < NB) ? n - c0 : NB;
    const int64_t below = n - c0 - w;
    // T = Bt[:, c0:c0+w] - Xt[:, c0+w:] * L[c0+w:, c0:c0+w]
    DFH_TRY(gemm_f64(ctx, GEMM_TRANSB, m, w, below, -1.0, Bt + c0 + w, ldb, L + (c0 + w) * ldl + c0, ldl,
                     1.0, Bt + c0, ldb, T, NB));
    // Xt[:, c0:c0+w] = T * Linv_ii
    DFH_TRY(gemm_f64(ctx, GEMM_TRANSB, m, w, w, 1.0, T, NB, inv + b * NB * NB, NB, 0.0, nullptr, 0,
                     Bt + c0, ldb));
  }
  return DFH_OK;
}
