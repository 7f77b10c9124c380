/*
 * ld_truth.c -- extended-precision "truth" for Dragonfly's GP posterior (x87 long double: 64-bit
 * mantissa, eps = 1.1e-19), the third party of every conditioning test (SURVEY.md section 7,
 * step 1 / hard part 1c: "always report both implementations against the long-double oracle").
 *
 * TEST INFRASTRUCTURE ONLY: nothing in dragonfly_amd/ links or loads this; tests/ and bench.py's
 * checks do.  Built by oracle/build_truth.py (gcc -O2 -fopenmp -shared) into oracle/_build/.
 *
 * It follows the MATHEMATICS of the reference path, not its floating-point evaluation order:
 *   kernel      SE  k = scale exp(-d2/2);  Matern(nu = p + 1/2)
 *               k = scale [sum_i c_i (sqrt(8 nu) r)^(p-i)] Gamma(p+1)/Gamma(2p+1) exp(-sqrt(2 nu) r)
 *               (dragonfly/gp/kernel.py:171-181, 242-299), with d2 = sum_j ((a_j - b_j)/bw_j)^2 by
 *               direct differences -- dist_squared's expansion (utils/general_utils.py:58-70) is
 *               the reference's way of evaluating the same number in double
 *   fit         L L^T = K + (noise + jitter) I;  alpha = L^-T L^-1 (y - m)    (gp/gp_core.py:155-163)
 *   lml         -1/2 (y-m)^T alpha - sum log L_ii - n/2 log 2 pi                (gp_core.py:222-227)
 *   posterior   mu = m + K(X*,X) alpha;  sd = sqrt(k(x,x) - |L^-1 k(X,x)|^2)    (gp_core.py:165-190)
 *   EI          sd (z Phi(z) + phi(z)), z = (mu - best)/sd        (opt/gpb_acquisitions.py:247-261)
 * Inputs are the doubles the other two implementations receive; results are rounded to double once,
 * at the end.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef long double ld;

/* OpenMP team size (hundreds of host threads on a GPU box do not help an n = 2048 problem) */
void ld_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

static ld kern_value(int kind, double nu, ld scale, ld d2) {
  if (kind == 0) return scale * expl(-d2 / 2);
  const int p = (int)nu;
  const ld r = sqrtl(d2);
  const ld t = sqrtl((ld)8 * (ld)nu) * r;
  ld fact[32];
  fact[0] = 1;
  for (int i = 1; i < 32; ++i) fact[i] = fact[i - 1] * i;
  ld poly = 0;
  for (int i = 0; i <= p; ++i) {
    ld term = fact[p + i] / (fact[i] * fact[p - i]);
    for (int e = 0; e < p - i; ++e) term *= t;
    poly += term;
  }
  return scale * poly * (fact[p] / fact[2 * p]) * expl(-sqrtl((ld)2 * (ld)nu) * r);
}

static ld dist2(const double* a, const double* b, const double* bw, int d) {
  ld s = 0;
  for (int j = 0; j < d; ++j) {
    const ld t = ((ld)a[j] - (ld)b[j]) / (ld)bw[j];
    s += t * t;
  }
  return s;
}

/* in-place lower Cholesky of the row-major n x n matrix A (long double); returns 0 or the 1-based
 * index of the first non-positive pivot */
static int64_t chol_ld(ld* A, int64_t n) {
  const int64_t NB = 64;
  for (int64_t k0 = 0; k0 < n; k0 += NB) {
    const int64_t kb = (n - k0 < NB) ? n - k0 : NB;
    for (int64_t k = k0; k < k0 + kb; ++k) {          /* diagonal block, unblocked */
      ld s = A[k * n + k];
      for (int64_t p = k0; p < k; ++p) s -= A[k * n + p] * A[k * n + p];
      if (!(s > 0)) return k + 1;
      const ld piv = sqrtl(s);
      A[k * n + k] = piv;
      for (int64_t i = k + 1; i < k0 + kb; ++i) {
        ld t = A[i * n + k];
        for (int64_t p = k0; p < k; ++p) t -= A[i * n + p] * A[k * n + p];
        A[i * n + k] = t / piv;
      }
    }
    const int64_t r0 = k0 + kb;
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t i = r0; i < n; ++i) {                /* panel rows: X L11^T = A21 */
      for (int64_t k = k0; k < k0 + kb; ++k) {
        ld t = A[i * n + k];
        for (int64_t p = k0; p < k; ++p) t -= A[i * n + p] * A[k * n + p];
        A[i * n + k] = t / A[k * n + k];
      }
    }
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t i = r0; i < n; ++i) {                /* trailing update, lower triangle */
      const ld* li = A + i * n + k0;
      for (int64_t j = r0; j <= i; ++j) {
        const ld* lj = A + j * n + k0;
        ld s = 0;
        for (int64_t p = 0; p < kb; ++p) s += li[p] * lj[p];
        A[i * n + j] -= s;
      }
    }
  }
  return 0;
}

/*
 * kind: 0 SE, 1 Matern.  X [n x d], yc [n] = y - mean, Xs [m x d] (m may be 0).
 * diag_add = noise variance (+ the jitter the reference's stable_cholesky ended up adding).
 * Outputs (any may be NULL): alpha [n], lml [1], mu [m] (mean_const added), sd [m], ei [m] with
 * incumbent `best`, L [n x n] (lower, upper zero), K [n x n] (without diag_add).
 * With u [m] and draw_out [m]: the joint Thompson draw mu + chol(Sigma + ts_jitter I) u over the m
 * test points, Sigma = K(X*,X*) - V^T V kept in extended precision (gp_core.py:250-254,
 * utils/general_utils.py:224-232); cov_out [m x m] optionally receives Sigma rounded to double.
 * Returns 0, or the 1-based failing pivot, or -1 on allocation failure.
 */
int64_t ld_gp_truth(int kind, double nu, double scale, const double* bw, const double* X, const double* yc,
                    int64_t n, int d, double diag_add, const double* Xs, int64_t m, double mean_const,
                    double best, double* alpha_out, double* lml_out, double* mu_out, double* sd_out,
                    double* ei_out, double* L_out, double* K_out, const double* u, double ts_jitter,
                    double* draw_out, double* cov_out) {
  ld* A = (ld*)malloc(sizeof(ld) * (size_t)n * (size_t)n);
  ld* z = (ld*)malloc(sizeof(ld) * (size_t)n);
  ld* al = (ld*)malloc(sizeof(ld) * (size_t)n);
  if (!A || !z || !al) { free(A); free(z); free(al); return -1; }
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t i = 0; i < n; ++i) {
    for (int64_t j = 0; j <= i; ++j) {
      const ld k = kern_value(kind, nu, (ld)scale, dist2(X + i * d, X + j * d, bw, d));
      A[i * n + j] = k;
      A[j * n + i] = k;
    }
  }
  if (K_out)
    for (int64_t i = 0; i < n * n; ++i) K_out[i] = (double)A[i];
  for (int64_t i = 0; i < n; ++i) A[i * n + i] += (ld)diag_add;
  const int64_t piv = chol_ld(A, n);
  if (piv != 0) { free(A); free(z); free(al); return piv; }
  if (L_out) {
    for (int64_t i = 0; i < n; ++i)
      for (int64_t j = 0; j < n; ++j) L_out[i * n + j] = (j <= i) ? (double)A[i * n + j] : 0.0;
  }
  /* z = L^-1 yc ; alpha = L^-T z */
  for (int64_t i = 0; i < n; ++i) {
    ld s = (ld)yc[i];
    for (int64_t p = 0; p < i; ++p) s -= A[i * n + p] * z[p];
    z[i] = s / A[i * n + i];
  }
  for (int64_t i = n - 1; i >= 0; --i) {
    ld s = z[i];
    for (int64_t p = i + 1; p < n; ++p) s -= A[p * n + i] * al[p];
    al[i] = s / A[i * n + i];
  }
  if (alpha_out)
    for (int64_t i = 0; i < n; ++i) alpha_out[i] = (double)al[i];
  if (lml_out) {
    ld dot = 0, logdet = 0;
    for (int64_t i = 0; i < n; ++i) { dot += (ld)yc[i] * al[i]; logdet += logl(A[i * n + i]); }
    *lml_out = (double)(-dot / 2 - logdet - (ld)n / 2 * logl(2 * acosl((ld)-1)));
  }
  if (m > 0 && Xs) {
    const ld kxx = kern_value(kind, nu, (ld)scale, 0);
    const ld sqrt2 = sqrtl((ld)2), sqrt2pi = sqrtl(2 * acosl((ld)-1));
    const int want_draw = (u && draw_out) || cov_out;
    ld* V = want_draw ? (ld*)malloc(sizeof(ld) * (size_t)m * (size_t)n) : NULL;
    ld* mus = want_draw ? (ld*)malloc(sizeof(ld) * (size_t)m) : NULL;
    if (want_draw && (!V || !mus)) { free(V); free(mus); free(A); free(z); free(al); return -1; }
#pragma omp parallel
    {
      ld* vbuf = V ? NULL : (ld*)malloc(sizeof(ld) * (size_t)n);
#pragma omp for schedule(dynamic, 4)
      for (int64_t c = 0; c < m; ++c) {
        ld* v = V ? V + c * n : vbuf;
        ld mu = (ld)mean_const, ss = 0;
        for (int64_t i = 0; i < n; ++i) {
          const ld k = kern_value(kind, nu, (ld)scale, dist2(Xs + c * d, X + i * d, bw, d));
          mu += k * al[i];
          ld s = k;
          for (int64_t p = 0; p < i; ++p) s -= A[i * n + p] * v[p];
          v[i] = s / A[i * n + i];
          ss += v[i] * v[i];
        }
        const ld sd = sqrtl(kxx - ss);
        if (mus) mus[c] = mu;
        if (mu_out) mu_out[c] = (double)mu;
        if (sd_out) sd_out[c] = (double)sd;
        if (ei_out) {
          const ld zz = (mu - (ld)best) / sd;
          const ld Phi = erfcl(-zz / sqrt2) / 2, phi = expl(-zz * zz / 2) / sqrt2pi;
          ei_out[c] = (double)(sd * (zz * Phi + phi));
        }
      }
      free(vbuf);
    }
    if (want_draw) {
      ld* S = (ld*)malloc(sizeof(ld) * (size_t)m * (size_t)m);
      if (!S) { free(V); free(mus); free(A); free(z); free(al); return -1; }
#pragma omp parallel for schedule(dynamic, 4)
      for (int64_t a = 0; a < m; ++a) {
        for (int64_t b = 0; b <= a; ++b) {
          ld s = kern_value(kind, nu, (ld)scale, dist2(Xs + a * d, Xs + b * d, bw, d));
          const ld *va = V + a * n, *vb = V + b * n;
          for (int64_t p = 0; p < n; ++p) s -= va[p] * vb[p];
          S[a * m + b] = s;
          S[b * m + a] = s;
        }
      }
      if (cov_out)
        for (int64_t i = 0; i < m * m; ++i) cov_out[i] = (double)S[i];
      if (u && draw_out) {
        for (int64_t i = 0; i < m; ++i) S[i * m + i] += (ld)ts_jitter;
        const int64_t pv = chol_ld(S, m);
        if (pv != 0) { free(S); free(V); free(mus); free(A); free(z); free(al); return -(1000000 + pv); }
        for (int64_t i = 0; i < m; ++i) {
          ld s = mus[i];
          for (int64_t p = 0; p <= i; ++p) s += S[i * m + p] * (ld)u[p];
          draw_out[i] = (double)s;
        }
      }
      free(S);
    }
    free(V); free(mus);
  }
  free(A); free(z); free(al);
  return 0;
}

/* K [n1 x n2] in extended precision, rounded to double (kernel-matrix parity adjudication) */
int ld_kernel_matrix(int kind, double nu, double scale, const double* bw, const double* X1, int64_t n1,
                     const double* X2, int64_t n2, int d, double* K_out) {
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t i = 0; i < n1; ++i)
    for (int64_t j = 0; j < n2; ++j)
      K_out[i * n2 + j] = (double)kern_value(kind, nu, (ld)scale, dist2(X1 + i * d, X2 + j * d, bw, d));
  return 0;
}

/* s = mu + chol(C + jitter I) u for a given covariance C [b x b] (double), in extended precision:
 * the joint Thompson draw of one block (utils/general_utils.py:224-232) given its inputs */
int64_t ld_gaussian_draw(const double* C, int64_t b, double jitter, const double* mu, const double* u,
                         double* s_out) {
  ld* A = (ld*)malloc(sizeof(ld) * (size_t)b * (size_t)b);
  if (!A) return -1;
  for (int64_t i = 0; i < b * b; ++i) A[i] = (ld)C[i];
  for (int64_t i = 0; i < b; ++i) A[i * b + i] += (ld)jitter;
  const int64_t piv = chol_ld(A, b);
  if (piv != 0) { free(A); return piv; }
  for (int64_t i = 0; i < b; ++i) {
    ld s = (ld)mu[i];
    for (int64_t p = 0; p <= i; ++p) s += A[i * b + p] * (ld)u[p];
    s_out[i] = (double)s;
  }
  free(A);
  return 0;
}

/* The same posterior for ANY kernel, given its Gram matrices in double (additive / polynomial /
 * product kernels, host-evaluated kernels: everything ld_gp_truth has no formula for): the LINEAR
 * ALGEBRA of gp_core.py:155-190, 222-227 in extended precision on the inputs the other two
 * implementations' linear algebra receives.  K [n x n] without noise, Kx [m x n] = K(X*, X),
 * kxx [m] = k(x*, x*), Kss [m x m] = K(X*, X*) (for cov_out; may be NULL).  Returns 0, the 1-based
 * failing pivot, or -1 (allocation). */
int64_t ld_gram_truth(const double* K, int64_t n, double diag_add, const double* yc, const double* Kx,
                      const double* kxx, const double* Kss, int64_t m, double mean_const, double* alpha_out,
                      double* lml_out, double* mu_out, double* sd_out, double* cov_out) {
  ld* A = (ld*)malloc(sizeof(ld) * (size_t)n * (size_t)n);
  ld* z = (ld*)malloc(sizeof(ld) * (size_t)n);
  ld* al = (ld*)malloc(sizeof(ld) * (size_t)n);
  if (!A || !z || !al) { free(A); free(z); free(al); return -1; }
  for (int64_t i = 0; i < n * n; ++i) A[i] = (ld)K[i];
  for (int64_t i = 0; i < n; ++i) A[i * n + i] += (ld)diag_add;
  const int64_t piv = chol_ld(A, n);
  if (piv != 0) { free(A); free(z); free(al); return piv; }
  for (int64_t i = 0; i < n; ++i) {
    ld s = (ld)yc[i];
    for (int64_t p = 0; p < i; ++p) s -= A[i * n + p] * z[p];
    z[i] = s / A[i * n + i];
  }
  for (int64_t i = n - 1; i >= 0; --i) {
    ld s = z[i];
    for (int64_t p = i + 1; p < n; ++p) s -= A[p * n + i] * al[p];
    al[i] = s / A[i * n + i];
  }
  if (alpha_out)
    for (int64_t i = 0; i < n; ++i) alpha_out[i] = (double)al[i];
  if (lml_out) {
    ld dot = 0, logdet = 0;
    for (int64_t i = 0; i < n; ++i) { dot += (ld)yc[i] * al[i]; logdet += logl(A[i * n + i]); }
    *lml_out = (double)(-dot / 2 - logdet - (ld)n / 2 * logl(2 * acosl((ld)-1)));
  }
  if (m > 0 && Kx) {
    ld* V = (ld*)malloc(sizeof(ld) * (size_t)m * (size_t)n);
    if (!V) { free(A); free(z); free(al); return -1; }
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t c = 0; c < m; ++c) {
      ld* v = V + c * n;
      ld mu = (ld)mean_const, ss = 0;
      for (int64_t i = 0; i < n; ++i) {
        const ld k = (ld)Kx[c * n + i];
        mu += k * al[i];
        ld s = k;
        for (int64_t p = 0; p < i; ++p) s -= A[i * n + p] * v[p];
        v[i] = s / A[i * n + i];
        ss += v[i] * v[i];
      }
      if (mu_out) mu_out[c] = (double)mu;
      if (sd_out && kxx) sd_out[c] = (double)sqrtl((ld)kxx[c] - ss);
    }
    if (cov_out && Kss) {
#pragma omp parallel for schedule(dynamic, 4)
      for (int64_t a = 0; a < m; ++a)
        for (int64_t b = 0; b <= a; ++b) {
          ld s = (ld)Kss[a * m + b];
          const ld *va = V + a * n, *vb = V + b * n;
          for (int64_t p = 0; p < n; ++p) s -= va[p] * vb[p];
          cov_out[a * m + b] = (double)s;
          cov_out[b * m + a] = (double)s;
        }
    }
    free(V);
  }
  free(A); free(z); free(al);
  return 0;
}
