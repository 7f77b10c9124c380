// extern "C" entry points of libdfhip.so (see include/dfhip.h for the contract and the
// reference functions each one replaces) and the GP object that lives in HBM.
#include "common.h"
#include <cstring>
#include <math.h>
#include <limits.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <stdlib.h>

struct dfh_gp {
  dfh_ctx* ctx = nullptr;
  KernDev kd;
  int64_t n = 0, d = 0, nblk = 0;
  double noise_var = 0.0;
  double diag_jitter = 0.0;      // what the ladder added on top of noise_var (0 if none)
  double* Xp = nullptr;          // [n][P] packed scaled training inputs
  double* Np = nullptr;          // [n][n_parts]
  double* L = nullptr;           // [n][n] lower factor (strict upper part unspecified)
  double* inv = nullptr;         // [2][nblk][NB][NB]: inverses of the diagonal blocks of L, then clean copies of the blocks
  std::vector<int> refine;       // [nblk] refinement steps the solves take with each block (chol.hip: refine_steps)
  double* alpha = nullptr;       // [n]
  bool upper_zeroed = false;
  bool gram = false;             // built from a host-evaluated Gram matrix: no kernel, no packed inputs
};

namespace {

int64_t pick_chunk(dfh_ctx* ctx, int64_t n, int64_t m) {
  // candidate rows per posterior chunk: the m_c x n cross matrix (solved in place into V^T) is sized
  // for a 288 GB part -- DFH_CHUNK_GIB (32) GiB, two of them alive in the pipelined Thompson
  // sampling.  Measured on the bench step (n = 16384, 262144 candidates, TS blocks factored in
  // lock-step batches of DFH_TS_BATCH): 4 GiB / 8 blocks 1434 ms, 8 GiB / 16 1403-1412, 16 GiB / 32
  // 1397, 32 GiB / 64 1389 -- bigger chunks mean taller TRSM products (1048 -> 1021 ms) and more
  // Thompson blocks per latency-bound factorisation chain (332 -> 310 ms).
  // The cap is per CONTEXT (its device, and whoever else is on it): an eighth of what was free on the
  // context's device when the context first asked, plus what the context's own scratch pool already
  // held then -- not an eighth of the first device's total memory for the whole process.
  static const double chunk_gib = []() { const char* e = getenv("DFH_CHUNK_GIB"); double v = e ? atof(e) : 32.0; return v > 0.0 ? v : 32.0; }();
  if (ctx->chunk_cap_gib <= 0.0) {
    size_t f = 0, t = 0;
    double cap = 36.0;
    if (hipMemGetInfo(&f, &t) == hipSuccess) {
      size_t own = 0;
      for (const DevBuf& b : ctx->scratch) own += b.bytes;
      cap = (double)(f + own) / 8.0 / 1073741824.0;
    } else {
      (void)hipGetLastError();
    }
    ctx->chunk_cap_gib = cap > 0.25 ? cap : 0.25;
  }
  const double gib = chunk_gib < ctx->chunk_cap_gib ? chunk_gib : ctx->chunk_cap_gib;
  int64_t mc = (int64_t)(gib * (double)(1LL << 27)) / (n > 0 ? n : 1);
  mc = std::max<int64_t>(512, std::min<int64_t>(mc, 262144));
  mc = (mc / 512) * 512;
  if (mc > m) mc = m;
  return mc;
}

// numpy argmax ordering: a NaN beats everything, earlier index wins ties
__device__ __forceinline__ bool better(double va, long ia, double vb, long ib) {
  const bool na = va != va, nb = vb != vb;
  if (na || nb) {
    if (na && nb) return ia < ib;
    return na;
  }
  if (va > vb) return true;
  if (va < vb) return false;
  return ia < ib;
}

__global__ void k_argmax_stage1(const double* __restrict__ v, long m, long idx_base,
                                double* __restrict__ pv, long* __restrict__ pi) {
  __shared__ double sv[256];
  __shared__ long si[256];
  double bv = -INFINITY;
  long bi = LONG_MAX;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long)gridDim.x * blockDim.x) {
    const double x = v[i];
    if (bi == LONG_MAX || better(x, idx_base + i, bv, bi)) { bv = x; bi = idx_base + i; }
  }
  sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const double ov = sv[threadIdx.x + s]; const long oi = si[threadIdx.x + s];
      if (oi != LONG_MAX && (si[threadIdx.x] == LONG_MAX || better(ov, oi, sv[threadIdx.x], si[threadIdx.x]))) {
        sv[threadIdx.x] = ov; si[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { pv[blockIdx.x] = sv[0]; pi[blockIdx.x] = si[0]; }
}

__global__ void k_argmax_stage2(const double* pv, const long* pi, int nparts, double* out_v, long* out_i) {
  __shared__ double sv[256];
  __shared__ long si[256];
  double bv = -INFINITY; long bi = LONG_MAX;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
    if (pi[i] != LONG_MAX && (bi == LONG_MAX || better(pv[i], pi[i], bv, bi))) { bv = pv[i]; bi = pi[i]; }
  }
  sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const double ov = sv[threadIdx.x + s]; const long oi = si[threadIdx.x + s];
      if (oi != LONG_MAX && (si[threadIdx.x] == LONG_MAX || better(ov, oi, sv[threadIdx.x], si[threadIdx.x]))) {
        sv[threadIdx.x] = ov; si[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out_v[0] = sv[0]; out_i[0] = si[0]; }
}

// One workgroup per segment [off[g], off[g+1]) of v: its arg-max (np.argmax rule, index local to the segment)
__global__ void k_argmax_segments(const double* __restrict__ v, const long* __restrict__ off,
                                  double* __restrict__ out_v, long* __restrict__ out_i) {
  __shared__ double sv[256];
  __shared__ long si[256];
  const long lo = off[blockIdx.x], hi = off[blockIdx.x + 1];
  double bv = -INFINITY;
  long bi = LONG_MAX;
  for (long i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const double x = v[i];
    if (bi == LONG_MAX || better(x, i - lo, bv, bi)) { bv = x; bi = i - lo; }
  }
  sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const double ov = sv[threadIdx.x + s]; const long oi = si[threadIdx.x + s];
      if (oi != LONG_MAX && (si[threadIdx.x] == LONG_MAX || better(ov, oi, sv[threadIdx.x], si[threadIdx.x]))) {
        sv[threadIdx.x] = ov; si[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out_v[blockIdx.x] = sv[0]; out_i[blockIdx.x] = si[0]; }
}

bool host_better(double va, int64_t ia, double vb, int64_t ib) {
  const bool na = va != va, nb = vb != vb;
  if (na || nb) { if (na && nb) return ia < ib; return na; }
  if (va > vb) return true;
  if (va < vb) return false;
  return ia < ib;
}

// arg-max of v[0..m) (device), indices offset by idx_base; merges into host running best
int argmax_update(dfh_ctx* ctx, const double* v, int64_t m, int64_t idx_base, bool* have,
                  double* best_v, int64_t* best_i) {
  if (m <= 0) return DFH_OK;
  const int nblocks = (int)std::min<int64_t>(1024, (m + 255) / 256);
  char* buf = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_RED, (size_t)(nblocks + 1) * 16 + 64, (void**)&buf));
  double* pv = reinterpret_cast<double*>(buf);
  long* pi = reinterpret_cast<long*>(buf + (size_t)(nblocks + 1) * 8);
  hipLaunchKernelGGL(k_argmax_stage1, dim3(nblocks), dim3(256), 0, ctx->stream, v, (long)m, (long)idx_base, pv, pi);
  DFH_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_argmax_stage2, dim3(1), dim3(256), 0, ctx->stream, pv, pi, nblocks, pv + nblocks, pi + nblocks);
  DFH_LAUNCH_CHECK();
  double hv; long hi;
  DFH_HIP(hipMemcpyAsync(&hv, pv + nblocks, 8, hipMemcpyDeviceToHost, ctx->stream));
  DFH_HIP(hipMemcpyAsync(&hi, pi + nblocks, 8, hipMemcpyDeviceToHost, ctx->stream));
  DFH_HIP(hipStreamSynchronize(ctx->stream));
  if (!*have || host_better(hv, (int64_t)hi, *best_v, *best_i)) { *best_v = hv; *best_i = (int64_t)hi; *have = true; }
  return DFH_OK;
}

// Phi(x): scipy.special.ndtr structure (xsf/cephes/ndtr.h) on the device erf/erfc
__device__ __forceinline__ double ndtr_dev(double a) {
  if (a != a) return a;
  const double x = a * 0.70710678118654752440;   // M_SQRT1_2
  const double z = fabs(x);
  double y;
  if (z < 1.0) {
    y = 0.5 + 0.5 * erf(x);
  } else {
    y = 0.5 * erfc(z);
    if (x > 0) y = 1.0 - y;
  }
  return y;
}
__device__ __forceinline__ double norm_pdf_dev(double x) {
  return exp(-(x * x) / 2.0) / 2.5066282746310002;   // scipy _norm_pdf: exp(-x**2/2.0)/sqrt(2*pi)
}
__device__ __forceinline__ double ei_norm_diff(double nd) {
  return nd * ndtr_dev(nd) + norm_pdf_dev(nd);       // gpb_acquisitions.py:247-249
}

// mu/sd/acquisition for one chunk.
//   mu_raw = K(Xs,X) alpha ; ss = ||L^-1 k||^2 ; ss2 = extra hallucination term (or null)
//   kss = prior variances k(x_i, x_i) of a kernel that is not stationary (else null: kxx)
__global__ void k_posterior_acq(int acq, double p0, double p1, double kxx, const double* __restrict__ kss,
                                double mean_const, const double* __restrict__ mean_vals, const double* __restrict__ mu_raw,
                                const double* __restrict__ ss, const double* __restrict__ ss2, long m,
                                double* __restrict__ mu_out, double* __restrict__ sd_out,
                                double* __restrict__ val_out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const double mu = (mean_vals ? mean_vals[i] : mean_const) + mu_raw[i];   // gp_core.py:173-175
  double sd = 0.0;
  if (ss) {
    if (kss) kxx = kss[i];
    double var = kxx - ss[i];                               // diag(K_tete - V^T V), gp_core.py:181
    if (ss2) var = kxx - (ss[i] + ss2[i]);
    sd = sqrt(var);                                         // gp_core.py:187 (NaN if var < 0)
  }
  if (mu_out) mu_out[i] = mu;
  if (sd_out) sd_out[i] = sd;
  if (!val_out) return;
  double v;
  switch (acq) {
    case DFH_ACQ_MEAN: v = mu; break;
    case DFH_ACQ_STD: v = sd; break;
    case DFH_ACQ_UCB: v = mu + p0 * sd; break;              // gpb_acquisitions.py:222
    case DFH_ACQ_EI: {                                      // :256-260
      const double nd = (mu - p0) / sd;
      v = sd * ei_norm_diff(nd);
      break;
    }
    case DFH_ACQ_PI: v = ndtr_dev((mu - p0) / sd); break;   // :238
    case DFH_ACQ_TTEI: {                                    // :275-279
      const double comb = sqrt(p1 * p1 + sd * sd);
      const double nd = (mu - p0) / comb;
      v = comb * ei_norm_diff(nd);
      break;
    }
    default: v = mu;
  }
  val_out[i] = v;
}

// hallucination tail: T[m x q] holds k(x, Xh) - V1 W^T ; solve rows with Lh (q x q lower) and
// return the squared norms.
__global__ void k_halluc_rows(double* __restrict__ T, long m, int q, const double* __restrict__ Lh,
                              double* __restrict__ ss2) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  double* t = T + i * q;
  double acc = 0.0;
  for (int c = 0; c < q; ++c) {
    double s = t[c];
    for (int k = 0; k < c; ++k) s = fma(-Lh[c * q + k], t[k], s);
    s = s / Lh[c * q + c];
    t[c] = s;
    acc = fma(s, s, acc);
  }
  ss2[i] = acc;
}

__global__ void k_add_vec(double* __restrict__ y, const double* __restrict__ a, double c, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = (a ? a[i] : 0.0) + c + y[i];
}

// state of the hallucinated augmentation (gp_core.py:192-220)
struct Halluc {
  int64_t q = 0;
  double* Xhp = nullptr; double* Nhp = nullptr;   // packed Xh
  double* Wt = nullptr;                           // [q][n] = K(Xh,X) L^-T
  double* Lh = nullptr;                           // [q][q] chol(K_hh + noise I - Wt Wt^T)
};

// The reference factors the whole augmented matrix with stable_cholesky (gp_core.py:199-206).  The
// block form below gives the same factor as long as that factorisation needs no jitter; when the
// Schur complement is not positive definite, or the base fit itself needed the ladder (its jitter
// was chosen for the n x n matrix, the reference would re-run the ladder on the (n+q) x (n+q) one),
// DFH_ERR_NOT_PD is returned and the callers fall back to halluc_augmented_gp.
int halluc_prepare(dfh_gp* gp, const double* Xh_user, int64_t q, Halluc* h) {
  dfh_ctx* ctx = gp->ctx;
  h->q = q;
  if (q <= 0) return DFH_OK;
  DFH_ARG(q <= 4096);
  if (gp->diag_jitter != 0.0) return DFH_ERR_NOT_PD;
  const KernDev& kd = gp->kd;
  const double* Xh = nullptr;
  DFH_TRY(to_device(ctx, Xh_user, (size_t)q * gp->d * 8, SCR_STAGE_C, &Xh));
  char* buf = nullptr;
  const size_t b_xhp = (size_t)q * kd.P * 8, b_nhp = (size_t)q * kd.n_parts * 8;
  const size_t b_wt = (size_t)q * gp->n * 8, b_lh = (size_t)q * q * 8;
  DFH_TRY(scratch_get(ctx, SCR_AUG, b_xhp + b_nhp + b_wt + b_lh + 1024, (void**)&buf));
  h->Xhp = reinterpret_cast<double*>(buf);
  h->Nhp = reinterpret_cast<double*>(buf + ((b_xhp + 255) / 256) * 256);
  h->Wt = reinterpret_cast<double*>(reinterpret_cast<char*>(h->Nhp) + ((b_nhp + 255) / 256) * 256);
  h->Lh = reinterpret_cast<double*>(reinterpret_cast<char*>(h->Wt) + ((b_wt + 255) / 256) * 256);
  DFH_TRY(pack_scaled(ctx, kd, 0, kd.n_parts, false, Xh, q, gp->d, h->Xhp, h->Nhp));
  // Wt = K(Xh, X) L^-T
  DFH_TRY(kernmat_packed(ctx, kd, 0, kd.n_parts, true, h->Xhp, h->Nhp, q, gp->Xp, gp->Np, gp->n, false, 0.0, h->Wt, gp->n));
  DFH_TRY(trsm_rows(ctx, gp->L, gp->n, gp->n, gp->inv, h->Wt, q, gp->n, gp->refine.data()));
  // S = K(Xh,Xh) + (noise + jitter) I - Wt Wt^T ; Lh = chol(S)
  const std::function<int()> build_S = [&]() -> int {
    DFH_TRY(kernmat_packed(ctx, kd, 0, kd.n_parts, true, h->Xhp, h->Nhp, q, h->Xhp, h->Nhp, q, true, gp->noise_var, h->Lh, q));
    return gemm_f64(ctx, 0, q, q, gp->n, -1.0, h->Wt, gp->n, h->Wt, gp->n, 1.0, h->Lh, q, h->Lh, q);
  };
  DFH_TRY(build_S());
  int64_t piv = 0;
  int rc = cholesky_device(ctx, h->Lh, q, q, nullptr, &piv, 1, 0, 0, nullptr, false, &build_S);
  if (rc == DFH_ERR_NOT_PD)
    dfh_set_error("augmented (hallucinated) kernel matrix is not positive definite at pivot %lld",
                  (long long)(gp->n + piv));
  return rc;
}

// One chunk of candidates (device pointer Xs_dev, mc rows): fills mu_raw, ss (and ss2).
// pre_gathered/part range select the add-UCB group path.
int posterior_chunk(dfh_gp* gp, const double* Xs_dev, int64_t mc, int64_t ldxs, int part_lo, int part_hi,
                    bool pre_gathered, bool want_var, const Halluc* h, double** Kct_out,
                    double* mu_raw, double* ss, double* ss2, int parity = 0, double** Xsp_out = nullptr,
                    double** Nsp_out = nullptr) {
  dfh_ctx* ctx = gp->ctx;
  const KernDev& kd = gp->kd;
  double* Xsp = nullptr; double* Nsp = nullptr; double* Kct = nullptr;
  char* xs = nullptr;
  const size_t b_xsp = ((size_t)mc * kd.P * 8 + 255) / 256 * 256;
  DFH_TRY(scratch_get(ctx, parity ? SCR_XS2 : SCR_XS, b_xsp + (size_t)mc * kd.n_parts * 8, (void**)&xs));
  Xsp = reinterpret_cast<double*>(xs);
  Nsp = reinterpret_cast<double*>(xs + b_xsp);
  DFH_TRY(scratch_get(ctx, parity ? SCR_KCT2 : SCR_KCT, (size_t)mc * gp->n * 8, (void**)&Kct));
  if (Xsp_out) *Xsp_out = Xsp;
  if (Nsp_out) *Nsp_out = Nsp;
  {
    SectionTimer t(ctx, DFH_T_CROSS);
    DFH_TRY(pack_scaled(ctx, kd, part_lo, part_hi, pre_gathered, Xs_dev, mc, ldxs, Xsp, Nsp));
    bool mu_done = false;      // gp_core.py:174, from the same pass where the kernel can
    DFH_TRY(kernmat_packed(ctx, kd, part_lo, part_hi, true, Xsp, Nsp, mc, gp->Xp, gp->Np, gp->n, false, 0.0, Kct, gp->n,
                           gp->alpha, mu_raw, &mu_done));
    if (!mu_done) DFH_TRY(gemv_rows(ctx, Kct, mc, gp->n, gp->n, gp->alpha, 1.0, nullptr, 0.0, mu_raw));
  }
  if (want_var) {
    {
      SectionTimer t(ctx, DFH_T_TRSM);
      DFH_TRY(trsm_rows(ctx, gp->L, gp->n, gp->n, gp->inv, Kct, mc, gp->n, gp->refine.data()));                // gp_core.py:180
    }
    SectionTimer t(ctx, DFH_T_ACQ);
    if (ss) DFH_TRY(row_sumsq(ctx, Kct, mc, gp->n, gp->n, ss));
    if (h && h->q > 0) {
      const int64_t q = h->q;
      double* T = nullptr;
      DFH_TRY(scratch_get(ctx, SCR_AUG2, (size_t)mc * q * 8, (void**)&T));
      // T = k(Xs, Xh) - V1t Wt^T
      DFH_TRY(kernmat_packed(ctx, kd, 0, kd.n_parts, true, Xsp, Nsp, mc, h->Xhp, h->Nhp, q, false, 0.0, T, q));
      DFH_TRY(gemm_f64(ctx, 0, mc, q, gp->n, -1.0, Kct, gp->n, h->Wt, gp->n, 1.0, T, q, T, q));
      hipLaunchKernelGGL(k_halluc_rows, dim3((unsigned)((mc + 255) / 256)), dim3(256), 0, ctx->stream, T, (long)mc, (int)q, h->Lh, ss2);
      DFH_LAUNCH_CHECK();
    }
  }
  if (Kct_out) *Kct_out = Kct;
  return DFH_OK;
}

// Fallback of the hallucinated posterior: the GP over (X, Xh) factored from scratch with the
// stable_cholesky ladder -- literally gp_core.py:196-206; only its variance is used (the labels
// are irrelevant: zeros).  The caller frees *aug.
int halluc_augmented_gp(dfh_gp* gp, const double* Xh, int64_t q, dfh_gp** aug) {
  std::vector<double> y0((size_t)(gp->n + q), 0.0);
  return dfh_gp_append(gp, Xh, q, y0.data(), 0, aug, nullptr, nullptr);
}

int ladder_pow(int p, double max_M, double* out) {
  *out = pow(10.0, (double)p) * max_M;      // (10 ** diag_noise_power) * max_M, general_utils.py:189
  return DFH_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
extern "C" int dfh_kernel_matrix(dfh_ctx* ctx, const dfh_kernel_desc* k, const double* X1, int64_t n1,
                                 const double* X2, int64_t n2, double diag_add, double* K_out) {
  DFH_ARG(ctx && k && K_out);
  DFH_ARG(n1 >= 0 && (X2 == nullptr || n2 >= 0));
  const bool sym = (X2 == nullptr);
  if (sym) n2 = n1;
  if (n1 == 0 || n2 == 0) return DFH_OK;     // kernel.py:81-82: empty result
  DFH_ARG(X1 != nullptr);
  DFH_HIP(hipSetDevice(ctx->device));
  KernDev kd;
  int rc = kerndev_build(ctx, k, &kd);
  if (rc != DFH_OK) { kerndev_free(&kd); return rc; }
  auto body = [&]() -> int {
    const int64_t d = k->dim;
    const double *dX1 = nullptr, *dX2 = nullptr;
    DFH_TRY(to_device(ctx, X1, (size_t)n1 * d * 8, SCR_STAGE_A, &dX1));
    if (!sym) DFH_TRY(to_device(ctx, X2, (size_t)n2 * d * 8, SCR_STAGE_B, &dX2));
    char* buf = nullptr;
    const size_t b1 = ((size_t)n1 * kd.P * 8 + 255) / 256 * 256, bn1 = ((size_t)n1 * kd.n_parts * 8 + 255) / 256 * 256;
    const size_t b2 = sym ? 0 : ((size_t)n2 * kd.P * 8 + 255) / 256 * 256, bn2 = sym ? 0 : (size_t)n2 * kd.n_parts * 8;
    DFH_TRY(scratch_get(ctx, SCR_XS, b1 + bn1 + b2 + bn2 + 256, (void**)&buf));
    double* Xp1 = reinterpret_cast<double*>(buf);
    double* Np1 = reinterpret_cast<double*>(buf + b1);
    double* Xp2 = sym ? Xp1 : reinterpret_cast<double*>(buf + b1 + bn1);
    double* Np2 = sym ? Np1 : reinterpret_cast<double*>(buf + b1 + bn1 + b2);
    DFH_TRY(pack_scaled(ctx, kd, 0, kd.n_parts, false, dX1, n1, d, Xp1, Np1));
    if (!sym) DFH_TRY(pack_scaled(ctx, kd, 0, kd.n_parts, false, dX2, n2, d, Xp2, Np2));
    const bool dev_out = is_device_ptr(K_out);
    double* Kd = K_out;
    if (!dev_out) DFH_TRY(scratch_get(ctx, SCR_KCT, (size_t)n1 * n2 * 8, (void**)&Kd));
    {
      SectionTimer t(ctx, sym ? DFH_T_KERNMAT : DFH_T_CROSS);
      DFH_TRY(kernmat_packed(ctx, kd, 0, kd.n_parts, true, Xp1, Np1, n1, Xp2, Np2, n2, sym, diag_add, Kd, n2));
    }
    if (!dev_out) DFH_TRY(from_device(ctx, K_out, Kd, (size_t)n1 * n2 * 8));
    return DFH_OK;
  };
  rc = body();
  (void)hipStreamSynchronize(ctx->stream);
  kerndev_free(&kd);
  return rc;
}

extern "C" int dfh_dist_squared(dfh_ctx* ctx, const double* X1, int64_t n1, const double* X2, int64_t n2,
                                int64_t d, double* D_out) {
  DFH_ARG(ctx && D_out && d >= 1 && n1 >= 0 && n2 >= 0);
  if (n1 == 0 || n2 == 0) return DFH_OK;
  DFH_ARG(X1 && X2);
  DFH_HIP(hipSetDevice(ctx->device));
  KernDev kd;
  int rc = kerndev_build_dist(ctx, (int)d, &kd);
  if (rc != DFH_OK) { kerndev_free(&kd); return rc; }
  auto body = [&]() -> int {
    const double *dX1 = nullptr, *dX2 = nullptr;
    DFH_TRY(to_device(ctx, X1, (size_t)n1 * d * 8, SCR_STAGE_A, &dX1));
    DFH_TRY(to_device(ctx, X2, (size_t)n2 * d * 8, SCR_STAGE_B, &dX2));
    char* buf = nullptr;
    const size_t b1 = ((size_t)n1 * kd.P * 8 + 255) / 256 * 256, bn1 = ((size_t)n1 * 8 + 255) / 256 * 256;
    const size_t b2 = ((size_t)n2 * kd.P * 8 + 255) / 256 * 256, bn2 = (size_t)n2 * 8;
    DFH_TRY(scratch_get(ctx, SCR_XS, b1 + bn1 + b2 + bn2 + 256, (void**)&buf));
    double* Xp1 = reinterpret_cast<double*>(buf);
    double* Np1 = reinterpret_cast<double*>(buf + b1);
    double* Xp2 = reinterpret_cast<double*>(buf + b1 + bn1);
    double* Np2 = reinterpret_cast<double*>(buf + b1 + bn1 + b2);
    DFH_TRY(pack_scaled(ctx, kd, 0, 1, false, dX1, n1, d, Xp1, Np1));
    DFH_TRY(pack_scaled(ctx, kd, 0, 1, false, dX2, n2, d, Xp2, Np2));
    const bool dev_out = is_device_ptr(D_out);
    double* Kd = D_out;
    if (!dev_out) DFH_TRY(scratch_get(ctx, SCR_KCT, (size_t)n1 * n2 * 8, (void**)&Kd));
    DFH_TRY(kernmat_packed(ctx, kd, 0, 1, false, Xp1, Np1, n1, Xp2, Np2, n2, false, 0.0, Kd, n2));
    if (!dev_out) DFH_TRY(from_device(ctx, D_out, Kd, (size_t)n1 * n2 * 8));
    return DFH_OK;
  };
  rc = body();
  (void)hipStreamSynchronize(ctx->stream);
  kerndev_free(&kd);
  return rc;
}

extern "C" int dfh_gemm(dfh_ctx* ctx, int transb, int64_t M, int64_t N, int64_t K, double alpha,
                        const double* A, int64_t lda, const double* B, int64_t ldb, double beta,
                        double* C, int64_t ldc, int lower_only) {
  DFH_ARG(ctx && C && M >= 0 && N >= 0 && K >= 0);
  if (M == 0 || N == 0) return DFH_OK;
  DFH_ARG((K == 0 || (A && B)) && lda >= K && ldc >= N && ldb >= (transb ? N : K));
  DFH_HIP(hipSetDevice(ctx->device));
  const double *dA = nullptr, *dB = nullptr, *dCin = nullptr;
  DFH_TRY(to_device(ctx, A, (size_t)M * lda * 8, SCR_STAGE_A, &dA));
  DFH_TRY(to_device(ctx, B, (size_t)(transb ? K : N) * ldb * 8, SCR_STAGE_B, &dB));
  const bool dev_out = is_device_ptr(C);
  double* dC = C;
  if (!dev_out) {
    DFH_TRY(scratch_get(ctx, SCR_STAGE_C, (size_t)M * ldc * 8, (void**)&dC));
    DFH_HIP(hipMemcpyAsync(dC, C, (size_t)M * ldc * 8, hipMemcpyHostToDevice, ctx->stream));
  }
  dCin = dC;
  int flags = (transb ? GEMM_TRANSB : 0) | (lower_only ? GEMM_LOWER : 0);
  DFH_TRY(gemm_f64(ctx, flags, M, N, K, alpha, dA, lda, dB, ldb, beta, dCin, ldc, dC, ldc));
  if (!dev_out) DFH_TRY(from_device(ctx, C, dC, (size_t)M * ldc * 8));
  DFH_HIP(hipStreamSynchronize(ctx->stream));
  return DFH_OK;
}

extern "C" int dfh_cholesky(dfh_ctx* ctx, double* A, int64_t n, int64_t* info_pivot) {
  DFH_ARG(ctx && n >= 0);
  if (info_pivot) *info_pivot = 0;
  if (n == 0) return DFH_OK;
  DFH_ARG(A != nullptr);
  DFH_HIP(hipSetDevice(ctx->device));
  const bool dev = is_device_ptr(A);
  double* dA = A;
  if (!dev) {
    DFH_TRY(scratch_get(ctx, SCR_TSL, (size_t)n * n * 8, (void**)&dA));
    DFH_HIP(hipMemcpyAsync(dA, A, (size_t)n * n * 8, hipMemcpyHostToDevice, ctx->stream));
  }
  int rc;
  {
    SectionTimer t(ctx, DFH_T_CHOL);
    // a host matrix can be uploaded again: the schedules that may need a second attempt are open to it
    const std::function<int()> reupload = [&]() -> int {
      DFH_HIP(hipMemcpyAsync(dA, A, (size_t)n * n * 8, hipMemcpyHostToDevice, ctx->stream));
      return DFH_OK;
    };
    rc = cholesky_device(ctx, dA, n, n, nullptr, info_pivot, 1, 0, 0, nullptr, false, dev ? nullptr : &reupload);
  }
  if (rc != DFH_OK) return rc;
  DFH_TRY(zero_upper(ctx, dA, n, n));
  if (!dev) DFH_TRY(from_device(ctx, A, dA, (size_t)n * n * 8));
  DFH_HIP(hipStreamSynchronize(ctx->stream));
  return DFH_OK;
}

// factor dL (holding M) in place with the stable_cholesky ladder; M is re-created by `rebuild`
// when a retry is needed (the failed factorisation destroys it).
template <typename Rebuild>
static int stable_cholesky_device(dfh_ctx* ctx, double* dL, int64_t n, double* keep_inv, bool allow_jitter,
                                  Rebuild rebuild, int32_t* jitter_power, double* jitter_added,
                                  int64_t ld = 0, int* refine_out = nullptr) {
  if (ld == 0) ld = n;
  if (jitter_power) *jitter_power = INT32_MIN;
  if (jitter_added) *jitter_added = 0.0;
  int64_t piv = 0;
  const std::function<int()> rebuild_fn = [&]() -> int { return rebuild(); };
  int rc = cholesky_device(ctx, dL, n, ld, keep_inv, &piv, 1, 0, 0, refine_out, false, &rebuild_fn);
  if (rc != DFH_ERR_NOT_PD || !allow_jitter) return rc;
  // general_utils.py:183-203
  DFH_TRY(rebuild());
  double max_M = 0.0;
  DFH_TRY(diag_max(ctx, dL, n, ld, &max_M));
  bool first = true;
  for (int p = -11; p < 5; ++p) {
    double diag_noise;
    ladder_pow(p, max_M, &diag_noise);
    if (!first) DFH_TRY(rebuild());
    first = false;
    DFH_TRY(add_diag(ctx, dL, n, ld, diag_noise));      // M + diag_noise * np.eye(n)
    const std::function<int()> rebuild_jit = [&]() -> int { DFH_TRY(rebuild()); return add_diag(ctx, dL, n, ld, diag_noise); };
    rc = cholesky_device(ctx, dL, n, ld, keep_inv, &piv, 1, 0, 0, refine_out, false, &rebuild_jit);
    if (rc == DFH_OK) {
      if (jitter_power) *jitter_power = p;
      if (jitter_added) *jitter_added = diag_noise;
      return DFH_OK;
    }
    if (rc != DFH_ERR_NOT_PD) return rc;
    if (p + 1 >= 5) {
      dfh_set_error("Could not compute Cholesky decomposition despite adding %0.4f to the diagonal. "
                    "This is likely because the M is not positive semi-definite or has infinities/nans.",
                    diag_noise);
      return DFH_ERR_JITTER;
    }
  }
  return DFH_ERR_JITTER;
}

extern "C" int dfh_stable_cholesky(dfh_ctx* ctx, const double* M_in, int64_t n, double* L_out,
                                   int32_t* jitter_power) {
  DFH_ARG(ctx && n >= 0);
  if (jitter_power) *jitter_power = INT32_MIN;
  if (n == 0) return DFH_OK;                         // general_utils.py:174-175
  DFH_ARG(M_in && L_out);
  DFH_HIP(hipSetDevice(ctx->device));
  const double* dM = nullptr;
  DFH_TRY(to_device(ctx, M_in, (size_t)n * n * 8, SCR_TSK, &dM));
  const bool dev_out = is_device_ptr(L_out);
  double* dL = L_out;
  if (!dev_out) DFH_TRY(scratch_get(ctx, SCR_TSL, (size_t)n * n * 8, (void**)&dL));
  auto rebuild = [&]() -> int { return copy_matrix(ctx, dM, n, dL, n, n, n); };
  DFH_TRY(rebuild());
  int rc;
  {
    SectionTimer t(ctx, DFH_T_CHOL);
    rc = stable_cholesky_device(ctx, dL, n, nullptr, true, rebuild, jitter_power, nullptr);
  }
  if (rc != DFH_OK) return rc;
  DFH_TRY(zero_upper(ctx, dL, n, n));
  if (!dev_out) DFH_TRY(from_device(ctx, L_out, dL, (size_t)n * n * 8));
  DFH_HIP(hipStreamSynchronize(ctx->stream));
  return DFH_OK;
}

extern "C" int dfh_solve_triangular(dfh_ctx* ctx, const double* L, int64_t n, int upper, const double* b,
                                    int64_t nrhs, double* x_out) {
  DFH_ARG(ctx && n >= 0 && nrhs >= 1);
  if (n == 0) return DFH_OK;
  DFH_ARG(L && b && x_out);
  DFH_HIP(hipSetDevice(ctx->device));
  const double *dL = nullptr, *dB = nullptr;
  DFH_TRY(to_device(ctx, L, (size_t)n * n * 8, SCR_TSL, &dL));
  DFH_TRY(to_device(ctx, b, (size_t)n * nrhs * 8, SCR_STAGE_B, &dB));
  const int64_t nblk = (n + CHOL_NB - 1) / CHOL_NB;
  double* inv = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_CHOLINV, (size_t)inv_buffer_doubles(n) * 8, (void**)&inv));
  std::vector<int> refine((size_t)nblk, 0);
  DFH_TRY(tri_block_inverses(ctx, dL, n, n, inv, refine.data()));
  double* dX = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_OUT, (size_t)n * nrhs * 8, (void**)&dX));
  if (nrhs == 1) {
    DFH_HIP(hipMemcpyAsync(dX, dB, (size_t)n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    if (!upper) DFH_TRY(trsv_forward(ctx, dL, n, n, inv, dX, refine.data()));
    else DFH_TRY(trsv_backward(ctx, dL, n, n, inv, dX, refine.data()));
  } else {
    double* Bt = nullptr;
    DFH_TRY(scratch_get(ctx, SCR_KCT, (size_t)n * nrhs * 8, (void**)&Bt));
    DFH_TRY(transpose_matrix(ctx, dB, nrhs, Bt, n, n, nrhs));       // Bt[nrhs][n]
    {
      SectionTimer t(ctx, DFH_T_TRSM);
      if (!upper) DFH_TRY(trsm_rows(ctx, dL, n, n, inv, Bt, nrhs, n, refine.data()));
      else DFH_TRY(trsm_rows_backward(ctx, dL, n, n, inv, Bt, nrhs, n, refine.data()));
    }
    DFH_TRY(transpose_matrix(ctx, Bt, n, dX, nrhs, nrhs, n));
  }
  DFH_TRY(from_device(ctx, x_out, dX, (size_t)n * nrhs * 8));
  DFH_HIP(hipStreamSynchronize(ctx->stream));
  return DFH_OK;
}

// ---------------------------------------------------------------------------------------------
extern "C" int dfh_gp_free(dfh_gp* gp) {
  if (!gp) return DFH_OK;
  if (gp->ctx && ctx_is_live(gp->ctx)) {      // the context may already be gone (teardown order)
    (void)hipSetDevice(gp->ctx->device);
    (void)hipStreamSynchronize(gp->ctx->stream);
  }
  kerndev_free(&gp->kd);
  dev_release(gp->ctx, gp->Xp);
  dev_release(gp->ctx, gp->Np);
  dev_release(gp->ctx, gp->L);
  dev_release(gp->ctx, gp->inv);
  dev_release(gp->ctx, gp->alpha);
  delete gp;
  return DFH_OK;
}

extern "C" int64_t dfh_gp_n(dfh_gp* gp) { return gp ? gp->n : -1; }

extern "C" int dfh_gp_refine_steps(dfh_gp* gp, int32_t* steps_out) {
  DFH_ARG(gp && steps_out);
  for (int64_t b = 0; b < gp->nblk; ++b) steps_out[b] = b < (int64_t)gp->refine.size() ? gp->refine[b] : 0;
  return DFH_OK;
}

// alpha = L^T \ (L \ y_centred) (gp_core.py:161-163) and the log marginal likelihood (:224-226)
static int gp_alpha_and_lml(dfh_gp* gp, const double* dy, double* lml) {
  dfh_ctx* ctx = gp->ctx;
  const int64_t n = gp->n;
  SectionTimer t(ctx, DFH_T_SOLVE);
  DFH_HIP(hipMemcpyAsync(gp->alpha, dy, (size_t)n * 8, hipMemcpyDeviceToDevice, ctx->stream));
  DFH_TRY(trsv_both(ctx, gp->L, n, n, gp->inv, gp->alpha, gp->refine.data()));
  double logdet = 0.0, dot = 0.0;
  DFH_TRY(logdet_and_dot(ctx, gp->L, n, n, dy, gp->alpha, &logdet, &dot));
  if (lml) *lml = -0.5 * dot - logdet - 0.5 * (double)n * log(2.0 * M_PI);
  return DFH_OK;
}

extern "C" int dfh_gp_fit(dfh_ctx* ctx, const dfh_kernel_desc* k, const double* X, int64_t n, int64_t d,
                          const double* y_centred, double noise_var, int flags, dfh_gp** out,
                          double* lml, int32_t* jitter_power) {
  DFH_ARG(ctx && k && out && n >= 1 && d >= 1 && X && y_centred);
  DFH_ARG(k->dim == d);
  *out = nullptr;
  if (jitter_power) *jitter_power = INT32_MIN;
  DFH_HIP(hipSetDevice(ctx->device));
  dfh_gp* gp = new dfh_gp();
  gp->ctx = ctx; gp->n = n; gp->d = d; gp->noise_var = noise_var;
  gp->nblk = (n + CHOL_NB - 1) / CHOL_NB;
  auto body = [&]() -> int {
    DFH_TRY(kerndev_build(ctx, k, &gp->kd));
    const KernDev& kd = gp->kd;
    DFH_TRY(dev_alloc(ctx, (size_t)n * kd.P * 8, (void**)&gp->Xp));
    DFH_TRY(dev_alloc(ctx, (size_t)n * kd.n_parts * 8, (void**)&gp->Np));
    DFH_TRY(dev_alloc(ctx, (size_t)n * n * 8, (void**)&gp->L));
    DFH_TRY(dev_alloc(ctx, (size_t)inv_buffer_doubles(n) * 8, (void**)&gp->inv));
    gp->refine.assign((size_t)gp->nblk, 0);
    DFH_TRY(dev_alloc(ctx, (size_t)n * 8, (void**)&gp->alpha));
    const double *dX = nullptr, *dy = nullptr;
    DFH_TRY(to_device(ctx, X, (size_t)n * d * 8, SCR_STAGE_A, &dX));
    DFH_TRY(to_device(ctx, y_centred, (size_t)n * 8, SCR_STAGE_B, &dy));
    // From n = 2048 on only the lower triangle of the Gram matrix is written (the factorisation, in place in this buffer,
    // reads nothing else; whoever asks for GP.L gets a zeroed upper part, dfh_gp_get): half the bytes of the
    // HBM-write-bound build.  DFH_KM_LOWER_ONLY=0: the full symmetric matrix as before.
    static const bool lower_env = []() { const char* e = getenv("DFH_KM_LOWER_ONLY"); return e ? atoi(e) != 0 : true; }();
    auto build_M = [&]() -> int {     // K + noise_var * I     (gp_core.py:843)
      SectionTimer t(ctx, DFH_T_KERNMAT);
      ctx->km_lower_only = lower_env && n >= 2048;
      // test hook (tests/test_gpu_upper_triangle_unread.py): the buffer comes recycled from the pool, and with the
      // lower-triangle-only build the tiles above the diagonal keep whatever it held -- correctness rests on no schedule
      // of the factorisation or the solves ever reading them.  DFH_TEST_POISON_L=1 fills the buffer with NaN first.
      if (const char* e = getenv("DFH_TEST_POISON_L"); e && atoi(e) != 0)
        DFH_HIP(hipMemsetAsync(gp->L, 0xFF, (size_t)n * n * 8, ctx->stream));
      const int rc = kernmat_packed(ctx, kd, 0, kd.n_parts, true, gp->Xp, gp->Np, n, gp->Xp, gp->Np, n, true, noise_var, gp->L, n);
      ctx->km_lower_only = false;
      return rc;
    };
    {
      SectionTimer t(ctx, DFH_T_KERNMAT);
      DFH_TRY(pack_scaled(ctx, kd, 0, kd.n_parts, false, dX, n, d, gp->Xp, gp->Np));
    }
    DFH_TRY(build_M());
    {
      SectionTimer t(ctx, DFH_T_CHOL);
      DFH_TRY(stable_cholesky_device(ctx, gp->L, n, gp->inv, !(flags & DFH_FIT_NO_JITTER), build_M,
                                     jitter_power, &gp->diag_jitter, 0, gp->refine.data()));
    }
    return gp_alpha_and_lml(gp, dy, lml);
  };
  int rc = body();
  if (rc != DFH_OK) { dfh_gp_free(gp); return rc; }
  DFH_HIP(hipStreamSynchronize(ctx->stream));
  *out = gp;
  return DFH_OK;
}

// ---------------------------------------------------------------------------------------------
// Posterior for an arbitrary positive semi-definite kernel evaluated by the caller:
// GP.build_posterior with the Gram matrix coming from the documented override hook
// GP._get_training_kernel_matrix (gp_core.py:149-163), and GP.eval with the caller's K(X*, X)
// (gp_core.py:165-190).  The O(n^3) / O(n^2 m) linear algebra is the same device path as for the
// built-in kernels; only the kernel evaluations stay with the caller.
__global__ void k_sd_from_prior(const double* __restrict__ kss, const double* __restrict__ ss,
                                double* __restrict__ sd, long m) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) sd[i] = sqrt(kss[i] - ss[i]);        // no clipping: NaN as in np.sqrt(np.diag(.)), gp_core.py:187
}

extern "C" int dfh_gp_fit_gram(dfh_ctx* ctx, const double* K, int64_t n, const double* y_centred,
                               double noise_var, int flags, dfh_gp** out, double* lml, int32_t* jitter_power) {
  DFH_ARG(ctx && K && out && n >= 1 && y_centred);
  *out = nullptr;
  if (jitter_power) *jitter_power = INT32_MIN;
  DFH_HIP(hipSetDevice(ctx->device));
  dfh_gp* gp = new dfh_gp();
  gp->ctx = ctx; gp->n = n; gp->d = 0; gp->noise_var = noise_var; gp->gram = true;
  gp->nblk = (n + CHOL_NB - 1) / CHOL_NB;
  auto body = [&]() -> int {
    DFH_TRY(dev_alloc(ctx, (size_t)n * n * 8, (void**)&gp->L));
    DFH_TRY(dev_alloc(ctx, (size_t)inv_buffer_doubles(n) * 8, (void**)&gp->inv));
    gp->refine.assign((size_t)gp->nblk, 0);
    DFH_TRY(dev_alloc(ctx, (size_t)n * 8, (void**)&gp->alpha));
    const double *dK = nullptr, *dy = nullptr;
    DFH_TRY(to_device(ctx, K, (size_t)n * n * 8, SCR_KCT, &dK));
    DFH_TRY(to_device(ctx, y_centred, (size_t)n * 8, SCR_STAGE_B, &dy));
    auto build_M = [&]() -> int {     // K + noise_var * I     (gp_core.py:843)
      DFH_TRY(copy_matrix(ctx, dK, n, gp->L, n, n, n));
      return add_diag(ctx, gp->L, n, n, noise_var);
    };
    bool project = (flags & DFH_FIT_PROJECT_FIRST) != 0;
    if (flags & DFH_FIT_TRY_BEFORE_PROJECT) {
      // gp_core.py:829-837: plain Cholesky of K + noise I (no ladder); only if that fails, project
      DFH_TRY(build_M());
      SectionTimer t(ctx, DFH_T_CHOL);
      int64_t piv = 0;
      const std::function<int()> rebuild_M = build_M;      // (a hand-off time-out repeats on the safe schedule)
      const int rc = cholesky_device(ctx, gp->L, n, n, gp->inv, &piv, 1, 0, 0, gp->refine.data(), false, &rebuild_M);
      if (rc == DFH_OK) return gp_alpha_and_lml(gp, dy, lml);
      if (rc != DFH_ERR_NOT_PD) return rc;
      project = true;
    }
    if (project) {
      // gp_core.py:838-841: the kernel matrix (without noise) goes to the PSD cone first
      double* Kp = nullptr;
      DFH_TRY(scratch_get(ctx, SCR_TSK, (size_t)n * n * 8, (void**)&Kp));
      DFH_TRY(psd_project_device(ctx, dK, n, n, 0.0, Kp, n));
      dK = Kp;
    }
    DFH_TRY(build_M());
    {
      SectionTimer t(ctx, DFH_T_CHOL);
      DFH_TRY(stable_cholesky_device(ctx, gp->L, n, gp->inv, !(flags & DFH_FIT_NO_JITTER), build_M,
                                     jitter_power, &gp->diag_jitter, 0, gp->refine.data()));
    }
    return gp_alpha_and_lml(gp, dy, lml);
  };
  int rc = body();
  if (rc != DFH_OK) { dfh_gp_free(gp); return rc; }
  DFH_HIP(hipStreamSynchronize(ctx->stream));
  *out = gp;
  return DFH_OK;
}

// mu = Kcross alpha (+ mean), sd = sqrt(kss - rowsumsq(Kcross L^-T)); Kcross is m x n, row i = k(x*_i, X)
extern "C" int dfh_gp_predict_gram(dfh_gp* gp, const double* Kcross, int64_t m, const double* kss,
                                   double mean_const, const double* mean_vals, double* mu_out, double* sd_out) {
  DFH_ARG(gp && m >= 0 && (sd_out == nullptr || kss != nullptr));
  if (m == 0) return DFH_OK;
  DFH_ARG(Kcross && mu_out);
  dfh_ctx* ctx = gp->ctx;
  DFH_HIP(hipSetDevice(ctx->device));
  const int64_t n = gp->n;
  const int64_t mc_max = pick_chunk(ctx, n, m);
  const bool k_dev = is_device_ptr(Kcross), s_dev = kss ? is_device_ptr(kss) : true;
  const bool mv_dev = mean_vals ? is_device_ptr(mean_vals) : true;
  double *vec = nullptr, *Kct = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_VEC, (size_t)mc_max * 8 * 4, (void**)&vec));
  DFH_TRY(scratch_get(ctx, SCR_KCT, (size_t)mc_max * n * 8, (void**)&Kct));
  double* mu = vec; double* ss = vec + mc_max; double* sd = vec + 2 * mc_max; double* ks = vec + 3 * mc_max;
  for (int64_t i0 = 0; i0 < m; i0 += mc_max) {
    const int64_t mc = std::min(mc_max, m - i0);
    // the chunk of K(X*, X) is solved in place, so it always goes through the workspace
    DFH_HIP(hipMemcpyAsync(Kct, Kcross + i0 * n, (size_t)mc * n * 8,
                           k_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    {
      SectionTimer t(ctx, DFH_T_CROSS);
      DFH_TRY(gemv_rows(ctx, Kct, mc, n, n, gp->alpha, 1.0, nullptr, 0.0, mu));     // gp_core.py:174
    }
    const double* mv_c = nullptr;
    if (mean_vals) {
      if (mv_dev) mv_c = mean_vals + i0;
      else DFH_TRY(to_device(ctx, mean_vals + i0, (size_t)mc * 8, SCR_STAGE_D, &mv_c));
    }
    hipLaunchKernelGGL(k_add_vec, dim3((unsigned)((mc + 255) / 256)), dim3(256), 0, ctx->stream, mu, mv_c,
                       mv_c ? 0.0 : mean_const, (long)mc);
    DFH_LAUNCH_CHECK();
    DFH_TRY(from_device(ctx, mu_out + i0, mu, (size_t)mc * 8));
    if (sd_out) {
      {
        SectionTimer t(ctx, DFH_T_TRSM);
        DFH_TRY(trsm_rows(ctx, gp->L, n, n, gp->inv, Kct, mc, n, gp->refine.data()));                  // gp_core.py:180
      }
      SectionTimer t(ctx, DFH_T_ACQ);
      DFH_TRY(row_sumsq(ctx, Kct, mc, n, n, ss));
      const double* ks_c = kss + i0;
      if (!s_dev) {
        DFH_HIP(hipMemcpyAsync(ks, kss + i0, (size_t)mc * 8, hipMemcpyHostToDevice, ctx->stream));
        ks_c = ks;
      }
      hipLaunchKernelGGL(k_sd_from_prior, dim3((unsigned)((mc + 255) / 256)), dim3(256), 0, ctx->stream, ks_c, ss, sd, (long)mc);
      DFH_LAUNCH_CHECK();
      DFH_TRY(from_device(ctx, sd_out + i0, sd, (size_t)mc * 8));
    }
    DFH_HIP(hipStreamSynchronize(ctx->stream));      // host source buffers may be reused by the caller
  }
  return DFH_OK;
}

// mu_out = Kcross alpha (raw, no mean), cov_out = Ktete - V^T V with V^T = Kcross L^-T  (gp_core.py:179-181)
extern "C" int dfh_gp_predict_covar_gram(dfh_gp* gp, const double* Kcross, int64_t m, const double* Ktete,
                                         double* mu_out, double* cov_out) {
  DFH_ARG(gp && m >= 0);
  if (m == 0) return DFH_OK;
  DFH_ARG(Kcross && Ktete && mu_out && cov_out);
  DFH_ARG((double)m * (double)gp->n * 8.0 < 64e9 && (double)m * (double)m * 8.0 < 64e9);
  dfh_ctx* ctx = gp->ctx;
  DFH_HIP(hipSetDevice(ctx->device));
  const int64_t n = gp->n;
  double *vec = nullptr, *Kct = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_VEC, (size_t)m * 8, (void**)&vec));
  DFH_TRY(scratch_get(ctx, SCR_KCT, (size_t)m * n * 8, (void**)&Kct));
  DFH_HIP(hipMemcpyAsync(Kct, Kcross, (size_t)m * n * 8,
                         is_device_ptr(Kcross) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
  DFH_TRY(gemv_rows(ctx, Kct, m, n, n, gp->alpha, 1.0, nullptr, 0.0, vec));
  DFH_TRY(from_device(ctx, mu_out, vec, (size_t)m * 8));
  DFH_TRY(trsm_rows(ctx, gp->L, n, n, gp->inv, Kct, m, n, gp->refine.data()));
  const bool dev_out = is_device_ptr(cov_out);
  double* C = cov_out;
  if (!dev_out) DFH_TRY(scratch_get(ctx, SCR_TSK, (size_t)m * m * 8, (void**)&C));
  DFH_HIP(hipMemcpyAsync(C, Ktete, (size_t)m * m * 8,
                         is_device_ptr(Ktete) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
  DFH_TRY(gemm_f64(ctx, 0, m, m, n, -1.0, Kct, n, Kct, n, 1.0, C, m, C, m));
  if (!dev_out) DFH_TRY(from_device(ctx, cov_out, C, (size_t)m * m * 8));
  DFH_HIP(hipStreamSynchronize(ctx->stream));
  return DFH_OK;
}

// ---------------------------------------------------------------------------------------------
// Incremental posterior update.  GP.add_data_multiple (gp_core.py:139-146)
// extends X, Y and rebuilds the posterior from scratch -- O((n+q)^3).  With the same kernel,
// noise and data order the factor of the extended matrix is
//     L' = [ L  0 ; B  Ls ],  B = K(Xnew, X) L^-T,  Ls = chol(K(Xnew,Xnew) + noise I - B B^T)
// (the Cholesky factor is unique), which costs O(n^2 q).  A NEW handle is returned; `gp` is left
// untouched (shallow copies of a GP share the handle).  When the reference's rebuild would leave
// the plain-Cholesky branch -- the existing fit needed the stable_cholesky ladder, or the Schur
// complement is not positive definite -- the extended matrix is rebuilt and factored from
// scratch with the ladder, exactly what build_posterior would do.
extern "C" int dfh_gp_append(dfh_gp* gp, const double* Xnew, int64_t q, const double* y_centred, int flags,
                             dfh_gp** out, double* lml, int32_t* jitter_power) {
  DFH_ARG(gp && out && q >= 1 && Xnew && y_centred);
  DFH_ARG(!gp->gram);      // needs the kernel: this posterior was built from a Gram matrix
  *out = nullptr;
  if (jitter_power) *jitter_power = INT32_MIN;
  dfh_ctx* ctx = gp->ctx;
  DFH_HIP(hipSetDevice(ctx->device));
  const int64_t n = gp->n, n2 = gp->n + q, d = gp->d, NB = CHOL_NB;
  dfh_gp* g2 = new dfh_gp();
  g2->ctx = ctx; g2->n = n2; g2->d = d; g2->noise_var = gp->noise_var;
  g2->nblk = (n2 + NB - 1) / NB;
  auto body = [&]() -> int {
    DFH_TRY(kerndev_clone(ctx, gp->kd, &g2->kd));
    const KernDev& kd = g2->kd;
    const int64_t P = kd.P, parts = kd.n_parts;
    DFH_TRY(dev_alloc(ctx, (size_t)n2 * P * 8, (void**)&g2->Xp));
    DFH_TRY(dev_alloc(ctx, (size_t)n2 * parts * 8, (void**)&g2->Np));
    DFH_TRY(dev_alloc(ctx, (size_t)n2 * n2 * 8, (void**)&g2->L));
    DFH_TRY(dev_alloc(ctx, (size_t)inv_buffer_doubles(n2) * 8, (void**)&g2->inv));
    g2->refine.assign((size_t)g2->nblk, 0);
    DFH_TRY(dev_alloc(ctx, (size_t)n2 * 8, (void**)&g2->alpha));
    const double *dXn = nullptr, *dy = nullptr;
    DFH_TRY(to_device(ctx, Xnew, (size_t)q * d * 8, SCR_STAGE_A, &dXn));
    DFH_TRY(to_device(ctx, y_centred, (size_t)n2 * 8, SCR_STAGE_B, &dy));
    double* Xpn = g2->Xp + n * P; double* Npn = g2->Np + n * parts;
    {
      SectionTimer t(ctx, DFH_T_KERNMAT);
      DFH_HIP(hipMemcpyAsync(g2->Xp, gp->Xp, (size_t)n * P * 8, hipMemcpyDeviceToDevice, ctx->stream));
      DFH_HIP(hipMemcpyAsync(g2->Np, gp->Np, (size_t)n * parts * 8, hipMemcpyDeviceToDevice, ctx->stream));
      DFH_TRY(pack_scaled(ctx, kd, 0, parts, false, dXn, q, d, Xpn, Npn));
    }
    auto full_refit = [&]() -> int {             // what build_posterior does: K' + noise I, ladder
      auto build_M = [&]() -> int {
        SectionTimer t(ctx, DFH_T_KERNMAT);
        return kernmat_packed(ctx, kd, 0, parts, true, g2->Xp, g2->Np, n2, g2->Xp, g2->Np, n2, true,
                              g2->noise_var, g2->L, n2);
      };
      DFH_TRY(build_M());
      SectionTimer t(ctx, DFH_T_CHOL);
      return stable_cholesky_device(ctx, g2->L, n2, g2->inv, !(flags & DFH_FIT_NO_JITTER), build_M,
                                    jitter_power, &g2->diag_jitter, 0, g2->refine.data());
    };
    bool appended = false;
    if (gp->diag_jitter == 0.0) {
      double* Bm = g2->L + n * n2;               // rows n.., columns 0..n-1
      double* S = g2->L + n * n2 + n;            // the new diagonal block (ld n2)
      DFH_TRY(copy_matrix(ctx, gp->L, n, g2->L, n2, n, n));
      {
        SectionTimer t(ctx, DFH_T_CROSS);
        DFH_TRY(kernmat_packed(ctx, kd, 0, parts, true, Xpn, Npn, q, g2->Xp, g2->Np, n, false, 0.0, Bm, n2));
      }
      {
        SectionTimer t(ctx, DFH_T_TRSM);
        DFH_TRY(trsm_rows(ctx, gp->L, n, n, gp->inv, Bm, q, n2, gp->refine.data()));
      }
      {
        SectionTimer t(ctx, DFH_T_CHOL);
        const std::function<int()> build_S = [&]() -> int {
          DFH_TRY(kernmat_packed(ctx, kd, 0, parts, true, Xpn, Npn, q, Xpn, Npn, q, true, g2->noise_var, S, n2));
          return gemm_f64(ctx, GEMM_LOWER, q, q, n, -1.0, Bm, n2, Bm, n2, 1.0, S, n2, S, n2);
        };
        DFH_TRY(build_S());
        int64_t piv = 0;
        const int rc = cholesky_device(ctx, S, q, n2, nullptr, &piv, 1, 0, 0, nullptr, false, &build_S);
        if (rc == DFH_OK) {
          // inverses of the 512-blocks: untouched blocks are copied, the rest recomputed from L'
          const int64_t kb0 = n / NB;            // first diagonal block that contains a new row
          double* diag2 = g2->inv + g2->nblk * NB * NB;
          if (kb0 > 0) {
            DFH_HIP(hipMemcpyAsync(g2->inv, gp->inv, (size_t)kb0 * NB * NB * 8, hipMemcpyDeviceToDevice, ctx->stream));
            DFH_HIP(hipMemcpyAsync(diag2, gp->inv + gp->nblk * NB * NB, (size_t)kb0 * NB * NB * 8,
                                   hipMemcpyDeviceToDevice, ctx->stream));
            std::copy(gp->refine.begin(), gp->refine.begin() + kb0, g2->refine.begin());
          }
          DFH_TRY(tri_block_inverses(ctx, g2->L + kb0 * NB * (n2 + 1), n2 - kb0 * NB, n2, g2->inv + kb0 * NB * NB,
                                     g2->refine.data() + kb0, diag2 + kb0 * NB * NB));
          appended = true;
        } else if (rc != DFH_ERR_NOT_PD) {
          return rc;
        }
      }
    }
    if (!appended) DFH_TRY(full_refit());
    return gp_alpha_and_lml(g2, dy, lml);
  };
  int rc = body();
  if (rc != DFH_OK) { dfh_gp_free(g2); return rc; }
  DFH_HIP(hipStreamSynchronize(ctx->stream));
  *out = g2;
  return DFH_OK;
}

// ---------------------------------------------------------------------------------------------
// Hyper-parameter tuning inner loop: the log marginal likelihoods of `nb`
// candidate hyper-parameter settings on the same data, i.e. GPFitter._tuning_objective
// (gp_core.py:551-564 -> build_gp -> build_posterior -> compute_log_marginal_likelihood, :222-227)
// for the list of candidates random_maximise / random_sample_cts_dscr evaluate one by one
// (oper_utils.py:70-80, 100-112).  Candidates are processed in groups whose Gram matrices are
// factored in lock-step by one batched launch sequence; a candidate whose matrix is not positive
// definite falls back to the stable_cholesky ladder on its own, exactly as a single fit would.
__global__ void k_centre(const double* __restrict__ y, double c, double* __restrict__ out,
                         double* __restrict__ out2, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const double v = y[i] - c; out[i] = v; out2[i] = v; }
}

// ---- the solve stage of a lock-step group, all candidates per launch (n > CHOL_NB) --------------------
// The log marginal likelihood needs sum(log L_ii) and (y - m)^T alpha = ||L^-1 (y - m)||^2: one FORWARD
// solve per candidate, no backward solve (what k_lml_tiny does in LDS).  Right-looking block
// substitution as in trsv_forward, but every launch carries all candidates (blockIdx.y): 3 launches
// per 512-block for the whole group instead of ~50 per candidate.
__global__ void k_centre_batch(const double* __restrict__ y, const double* __restrict__ means, double* __restrict__ vecs,
                               long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) vecs[(long)blockIdx.y * 2 * n + i] = y[i] - means[blockIdx.y];     // r = y - m  (z is written block by block)
}

// yout[c][row] = beta * yin[c][row] + alpha * A[c][row, 0:n] . x[c][0:n]; one wave per row (n <= 1024), four rows per
// workgroup, blockIdx.y = candidate c; operands sA / sx / sy doubles apart between candidates
__global__ void k_gemv_rows_wave_batch(const double* __restrict__ A, long sA, long m, long n, long lda,
                                       const double* __restrict__ x, long sx, double alpha, const double* yin, double beta,
                                       double* yout, long sy) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= m) return;
  const int lane = threadIdx.x & 63;
  const double* a = A + (long)blockIdx.y * sA + row * lda;
  const double* xv = x + (long)blockIdx.y * sx;
  double s0 = 0.0, s1 = 0.0;
  const bool vec = ((lda & 1) == 0) && ((sA & 1) == 0) && ((sx & 1) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  if (vec) {
    const long n2 = n >> 1;
    for (long j = lane; j < n2; j += 64) {
      const double2_t av = reinterpret_cast<const double2_t*>(a)[j];
      const double2_t xx = reinterpret_cast<const double2_t*>(xv)[j];
      s0 = fma(av.x, xx.x, s0);
      s1 = fma(av.y, xx.y, s1);
    }
    if ((n & 1) && lane == 0) s0 = fma(a[n - 1], xv[n - 1], s0);
  } else {
    for (long j = lane; j < n; j += 64) s0 = fma(a[j], xv[j], s0);
  }
  double sum = s0 + s1;
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
  if (lane == 0) {
    double v = alpha * sum;
    if (beta != 0.0) v += beta * yin[(long)blockIdx.y * sy + row];
    yout[(long)blockIdx.y * sy + row] = v;
  }
}

// out[2c] = sum(log L_c[i][i]), out[2c+1] = z_c . z_c      (fixed summation order: deterministic)
__global__ __launch_bounds__(256) void k_logdet_sumsq_batch(const double* __restrict__ L, long sL, long n, long ldl,
                                                            const double* __restrict__ z, long sz, double* __restrict__ out) {
  __shared__ double s1[256], s2[256];
  const double* Lc = L + (long)blockIdx.x * sL;
  const double* zc = z + (long)blockIdx.x * sz;
  double ld = 0.0, dt = 0.0;
  for (long i = threadIdx.x; i < n; i += blockDim.x) {
    ld += log(Lc[i * ldl + i]);
    dt = fma(zc[i], zc[i], dt);
  }
  s1[threadIdx.x] = ld;
  s2[threadIdx.x] = dt;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) { s1[threadIdx.x] += s1[threadIdx.x + st]; s2[threadIdx.x] += s2[threadIdx.x + st]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = s1[0]; out[2 * blockIdx.x + 1] = s2[0]; }
}

// n <= CHOL_NB (one diagonal block): the whole solve stage of a candidate in one workgroup --
// yc = y - m, z = L^-1 yc through the explicit block inverse M (followed by steps[c] steps of
// iterative refinement against the clean copy Ld of the block, chol.hip: refine_steps), then
// out = {sum log L_ii, z . z}  (= yc . alpha: the backward solve is not needed).  blockIdx.x = candidate.
__global__ __launch_bounds__(256) void k_lml_finish_small(const double* __restrict__ inv, long sInv,
                                                          const double* __restrict__ y,
                                                          const double* __restrict__ means, int n,
                                                          const int* __restrict__ steps,
                                                          double* __restrict__ out2) {
  __shared__ double yc[CHOL_NB], z[CHOL_NB], r[CHOL_NB], red[8];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  inv += (long)c * sInv;
  const double* Ld = inv + CHOL_NB * CHOL_NB;                // clean copy of the factor (one block: nblk = 1)
  const double mean = means[c];
  const int nsteps = steps[c];
  for (int i = tid; i < n; i += 256) yc[i] = y[i] - mean;
  __syncthreads();
  // dst_i (+)= sum_{j <= i} A[i][j] src[j], a wave per row
  auto lower_mv = [&](const double* A, const double* src, double* dst, double sign, const double* base) {
    for (int i = wave; i < n; i += 4) {
      const double* row = A + (long)i * CHOL_NB;
      double s = 0.0;
      for (int j = lane; j <= i; j += 64) s = fma(row[j], src[j], s);
      for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
      if (lane == 0) dst[i] = (base ? base[i] : 0.0) + sign * s;
    }
    __syncthreads();
  };
  lower_mv(inv, yc, z, 1.0, nullptr);                       // z = M yc
  for (int s = 0; s < nsteps; ++s) {
    lower_mv(Ld, z, r, -1.0, yc);                           // r = yc - L z
    lower_mv(inv, r, z, 1.0, z);                            // z += M r
  }
  // (y - m)^T alpha = ||L^-1 (y - m)||^2 = z . z: the backward solve is not needed for the likelihood
  double ld = 0.0, dt = 0.0;
  for (int j = tid; j < n; j += 256) {
    dt = fma(z[j], z[j], dt);
    ld += log(Ld[(long)j * CHOL_NB + j]);
  }
  for (int off = 32; off > 0; off >>= 1) { ld += __shfl_down(ld, off, 64); dt += __shfl_down(dt, off, 64); }
  if (lane == 0) { red[wave] = ld; red[4 + wave] = dt; }
  __syncthreads();
  if (tid == 0) {
    out2[2 * c] = (red[0] + red[1]) + (red[2] + red[3]);
    out2[2 * c + 1] = (red[4] + red[5]) + (red[6] + red[7]);
  }
}

// The same stage without the 512-block inverse: forward substitution over 64-blocks with the factor
// itself and the inverses of its 64 x 64 diagonal blocks (what trtri64_kernel leaves on the diagonal
// of the inverse buffer) -- z_b = Linv_bb (r_b - sum_{i<b} L_bi z_i).  Saves the inverse assembly
// (six GEMM launches) and its quality measurement per call; a 64-block inverse needs no refinement.
__global__ __launch_bounds__(256) void k_lml_finish_small64(const double* __restrict__ L, long sL, long ldl,
                                                            const double* __restrict__ inv, long sInv,
                                                            const double* __restrict__ y,
                                                            const double* __restrict__ means, int n,
                                                            double* __restrict__ out2) {
  __shared__ double r[CHOL_NB], z[CHOL_NB], t[64], red[8];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  L += (long)c * sL;
  inv += (long)c * sInv;
  const double mean = means[c];
  for (int i = tid; i < n; i += 256) r[i] = y[i] - mean;
  __syncthreads();
  for (int b0 = 0; b0 < n; b0 += 64) {
    const int w = min(64, n - b0);
    // t = r_b - L[b, 0:b0] z[0:b0], a wave per row
    for (int i = wave; i < w; i += 4) {
      const double* row = L + (long)(b0 + i) * ldl;
      double s = 0.0;
      for (int j = lane; j < b0; j += 64) s = fma(row[j], z[j], s);
      for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
      if (lane == 0) t[i] = r[b0 + i] - s;
    }
    __syncthreads();
    // z_b = Linv_bb t (lower triangular 64 x 64, row stride CHOL_NB)
    for (int i = wave; i < w; i += 4) {
      const double* row = inv + (long)(b0 + i) * CHOL_NB + b0;
      double s = (lane <= i) ? row[lane] * t[lane] : 0.0;
      for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
      if (lane == 0) z[b0 + i] = s;
    }
    __syncthreads();
  }
  double ld = 0.0, dt = 0.0;
  for (int j = tid; j < n; j += 256) {
    dt = fma(z[j], z[j], dt);
    ld += log(L[(long)j * ldl + j]);
  }
  for (int off = 32; off > 0; off >>= 1) { ld += __shfl_down(ld, off, 64); dt += __shfl_down(dt, off, 64); }
  if (lane == 0) { red[wave] = ld; red[4 + wave] = dt; }
  __syncthreads();
  if (tid == 0) {
    out2[2 * c] = (red[0] + red[1]) + (red[2] + red[3]);
    out2[2 * c + 1] = (red[4] + red[5]) + (red[6] + red[7]);
  }
}

// The lock-step schedule: groups of up to CHOL_MAX_BATCH candidates through the batched cholesky_device (any n);
// cand_base: index of descs[0] in the caller's list (error messages).
static int lml_batch_lockstep(dfh_ctx* ctx, const dfh_kernel_desc* descs, int32_t nb, const double* dX,
                              int64_t n, int64_t d, const double* y, const double* mean_consts,
                              const double* noise_vars, int flags, double* lml_out,
                              int32_t* jitter_powers, int cand_base) {
  const int64_t NB = CHOL_NB;
  const int64_t nblk = (n + NB - 1) / NB;
  const int64_t ldK = (n + 1) & ~(int64_t)1;                 // even leading dimension: 16-byte row starts
  const int64_t strideK = n * ldK, strideInv = inv_buffer_doubles(n);
  // group size: up to CHOL_MAX_BATCH matrices and (DFH_LML_GROUP_GIB, default 8) GiB of Gram
  // matrices at a time.  Measured ms per candidate at 2 / 8 GiB: n=4096 1.55 / 1.07, n=16384
  // 42.6 (one at a time) / 30.0 (four in lock-step: the panel chains of the four interleave).
  static const double group_gib = []() { const char* e = getenv("DFH_LML_GROUP_GIB"); double v = e ? atof(e) : 8.0; return v > 0.0 ? v : 8.0; }();
  const int64_t by_mem = std::max<int64_t>(1, (int64_t)(group_gib * 1073741824.0 / ((double)strideK * 8.0)));
  const int G = (int)std::min<int64_t>(std::min<int64_t>(nb, CHOL_MAX_BATCH), by_mem);
  std::vector<KernDev> kds((size_t)G);       // device images live in one scratch blob: nothing to free
  const double* dy = nullptr;
  DFH_TRY(to_device(ctx, y, (size_t)n * 8, SCR_STAGE_B, &dy));
  double *Kb = nullptr, *invb = nullptr, *vecs = nullptr, *red = nullptr, *dpar = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_KCT, (size_t)G * strideK * 8, (void**)&Kb));
  DFH_TRY(scratch_get(ctx, SCR_TSK, (size_t)G * strideInv * 8, (void**)&invb));
  DFH_TRY(scratch_get(ctx, SCR_VEC, (size_t)G * n * 8 * 2, (void**)&vecs));
  DFH_TRY(scratch_get(ctx, SCR_OUT2, (size_t)std::max(256, G * 16), (void**)&red));   // SCR_RED belongs to the gemv partials
  DFH_TRY(scratch_get(ctx, SCR_OUT, (size_t)std::max(256, G * 24), (void**)&dpar));   // per candidate {noise, mean}, then int steps
  std::vector<double> hred((size_t)G * 2), hpar((size_t)G * 2);
  std::vector<int> refine((size_t)G * nblk, 0);           // refinement steps per candidate and diagonal block
  for (int c0 = 0; c0 < nb; c0 += G) {
    const int g = std::min(G, nb - c0);
    // packed inputs of the group's candidates (pad-to-4 columns per kernel part)
    int64_t Pmax = 0, parts_max = 0;
    size_t blob_bytes = 0;
    bool uniform = true;                     // structurally identical single-part kernels
    for (int c = 0; c < g; ++c) {
      kds[c] = KernDev();
      DFH_TRY(kerndev_build_host(&descs[c0 + c], &kds[c]));
      Pmax = std::max<int64_t>(Pmax, kds[c].P);
      parts_max = std::max<int64_t>(parts_max, kds[c].n_parts);
      blob_bytes += kerndev_blob_bytes(kds[c]);
      uniform = uniform && !kds[c].multi && kds[c].n_parts == 1 && kds[c].P == kds[0].P &&
                kerndev_blob_bytes(kds[c]) == kerndev_blob_bytes(kds[0]);
    }
    void* blob = nullptr;
    DFH_TRY(scratch_get(ctx, SCR_AUG2, blob_bytes, &blob));
    DFH_TRY(kerndev_upload_many(ctx, kds.data(), g, blob, blob_bytes));
    double *Xpb = nullptr, *Npb = nullptr;
    DFH_TRY(scratch_get(ctx, SCR_XS, (size_t)g * n * Pmax * 8, (void**)&Xpb));
    DFH_TRY(scratch_get(ctx, SCR_XS2, (size_t)g * n * parts_max * 8, (void**)&Npb));
    const int64_t sXp = n * Pmax, sNp = n * parts_max;
    for (int c = 0; c < g; ++c) {
      hpar[c] = noise_vars[c0 + c];
      hpar[g + c] = mean_consts ? mean_consts[c0 + c] : 0.0;
    }
    DFH_HIP(hipMemcpyAsync(dpar, hpar.data(), (size_t)g * 16, hipMemcpyHostToDevice, ctx->stream));
    auto build_M = [&](int c) -> int {           // K + noise_var * I     (gp_core.py:843)
      double* Xp = Xpb + c * sXp; double* Np = Npb + c * sNp;
      return kernmat_packed(ctx, kds[c], 0, kds[c].n_parts, true, Xp, Np, n, Xp, Np, n, true,
                            noise_vars[c0 + c], Kb + c * strideK, ldK);
    };
    {
      SectionTimer t(ctx, DFH_T_KERNMAT);
      if (uniform) {
        // one launch each for the whole group: pack, norms, Gram matrices
        const int64_t sBlob = (int64_t)kerndev_blob_bytes(kds[0]);
        DFH_TRY(pack_scaled(ctx, kds[0], 0, 1, false, dX, n, d, Xpb, Npb, g, sBlob, sXp, sNp));
        DFH_TRY(kernmat_sym_batch(ctx, kds[0], g, sBlob, Xpb, sXp, Npb, sNp, n, dpar, Kb, strideK, ldK));
      } else {
        for (int c = 0; c < g; ++c) {
          DFH_TRY(pack_scaled(ctx, kds[c], 0, kds[c].n_parts, false, dX, n, d, Xpb + c * sXp, Npb + c * sNp));
          DFH_TRY(build_M(c));
        }
      }
    }
    {
      SectionTimer t(ctx, DFH_T_CHOL);
      int64_t piv[CHOL_MAX_BATCH] = {0};
      // n <= 512: the finish kernel substitutes with 64-blocks, so the 512-block inverse is not built
      static const bool small64 = []() { const char* e = getenv("DFH_LML_SMALL64"); return e ? atoi(e) != 0 : true; }();
      const bool inv64_only = small64 && n <= NB;
      const std::function<int()> rebuild_all = [&]() -> int {
        for (int c = 0; c < g; ++c) DFH_TRY(build_M(c));
        return DFH_OK;
      };
      int rc = cholesky_device(ctx, Kb, n, ldK, invb, piv, g, strideK, strideInv, refine.data(), inv64_only, &rebuild_all);
      if (rc != DFH_OK && rc != DFH_ERR_NOT_PD) return rc;
      for (int c = 0; c < g; ++c) {
        if (jitter_powers) jitter_powers[c0 + c] = INT32_MIN;
        if (piv[c] == 0) continue;
        if (flags & DFH_FIT_NO_JITTER) {
          dfh_set_error("Matrix is not positive definite (candidate %d, pivot %lld)", cand_base + c0 + c, (long long)piv[c]);
          return DFH_ERR_NOT_PD;
        }
        auto rebuild = [&]() -> int { return build_M(c); };
        DFH_TRY(rebuild());
        int32_t jp = INT32_MIN;
        DFH_TRY(stable_cholesky_device(ctx, Kb + c * strideK, n, invb + c * strideInv, true, rebuild, &jp, nullptr, ldK,
                                       refine.data() + (size_t)c * nblk));
        if (jitter_powers) jitter_powers[c0 + c] = jp;
      }
    }
    {
      SectionTimer t(ctx, DFH_T_SOLVE);
      static const bool small64b = []() { const char* e = getenv("DFH_LML_SMALL64"); return e ? atoi(e) != 0 : true; }();
      if (n <= NB && small64b) {
        // (a candidate that went through the jitter ladder has the full inverse in its slot: its diagonal
        //  64-blocks are the inverses of the factor's diagonal blocks all the same)
        hipLaunchKernelGGL(k_lml_finish_small64, dim3((unsigned)g), dim3(256), 0, ctx->stream, Kb, (long)strideK, (long)ldK,
                           invb, (long)strideInv, dy, dpar + g, (int)n, red);
        DFH_LAUNCH_CHECK();
      } else if (n <= NB) {
        // one block per candidate (nblk = 1): its refinement steps ride behind {noise, mean} in dpar
        int* dsteps = reinterpret_cast<int*>(dpar + 2 * g);
        DFH_HIP(hipMemcpyAsync(dsteps, refine.data(), (size_t)g * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_lml_finish_small, dim3((unsigned)g), dim3(256), 0, ctx->stream, invb, (long)strideInv,
                           dy, dpar + g, (int)n, dsteps, red);
        DFH_LAUNCH_CHECK();
      } else {
        bool any_refine = false;
        for (size_t i = 0; i < (size_t)g * nblk; ++i) any_refine = any_refine || refine[i] > 0;
        static const bool batch_solve = []() { const char* e = getenv("DFH_LML_BATCH_SOLVE"); return e ? atoi(e) != 0 : true; }();
        if (!any_refine && batch_solve) {
          // all candidates per launch: r = y - m; for each 512-block z_b = M_b r_b, r_below -= L[below, b] z_b
          const long sv = 2 * (long)n;                 // candidate c: r at vecs + c*sv, z behind it
          hipLaunchKernelGGL(k_centre_batch, dim3((unsigned)((n + 255) / 256), (unsigned)g), dim3(256), 0, ctx->stream, dy,
                             dpar + g, vecs, (long)n);
          DFH_LAUNCH_CHECK();
          for (int64_t b0 = 0; b0 < n; b0 += NB) {
            const int64_t w = std::min<int64_t>(NB, n - b0), below = n - b0 - w;
            hipLaunchKernelGGL(k_gemv_rows_wave_batch, dim3((unsigned)((w + 3) / 4), (unsigned)g), dim3(256), 0, ctx->stream,
                               invb + (b0 / NB) * NB * NB, (long)strideInv, (long)w, (long)w, (long)NB, vecs + b0, sv, 1.0,
                               (const double*)nullptr, 0.0, vecs + n + b0, sv);
            DFH_LAUNCH_CHECK();
            if (below > 0) {
              hipLaunchKernelGGL(k_gemv_rows_wave_batch, dim3((unsigned)((below + 3) / 4), (unsigned)g), dim3(256), 0,
                                 ctx->stream, Kb + (b0 + w) * ldK + b0, (long)strideK, (long)below, (long)w, (long)ldK,
                                 vecs + n + b0, sv, -1.0, vecs + b0 + w, 1.0, vecs + b0 + w, sv);
              DFH_LAUNCH_CHECK();
            }
          }
          hipLaunchKernelGGL(k_logdet_sumsq_batch, dim3((unsigned)g), dim3(256), 0, ctx->stream, Kb, (long)strideK, (long)n,
                             (long)ldK, vecs + n, sv, red);
          DFH_LAUNCH_CHECK();
        } else {
          for (int c = 0; c < g; ++c) {
            double* yc = vecs + (int64_t)c * 2 * n;
            double* alpha = yc + n;
            hipLaunchKernelGGL(k_centre, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, dy, hpar[g + c], yc, alpha, (long)n);
            DFH_LAUNCH_CHECK();
            // alpha = L^T \ (L \ (y - m))      (gp_core.py:161-163)
            DFH_TRY(trsv_both(ctx, Kb + c * strideK, n, ldK, invb + c * strideInv, alpha, refine.data() + (size_t)c * nblk));
            DFH_TRY(logdet_and_dot_device(ctx, Kb + c * strideK, n, ldK, yc, alpha, red + 2 * c));
          }
        }
      }
      DFH_HIP(hipMemcpyAsync(hred.data(), red, (size_t)g * 16, hipMemcpyDeviceToHost, ctx->stream));
      DFH_HIP(hipStreamSynchronize(ctx->stream));
      for (int c = 0; c < g; ++c)     // gp_core.py:224-226
        lml_out[c0 + c] = -0.5 * hred[2 * c + 1] - hred[2 * c] - 0.5 * (double)n * log(2.0 * M_PI);
    }
  }
  return DFH_OK;
}

// One workgroup per candidate (chol.hip: lml_wg_kernel), 128 < n <= LMLWG_MAX_N: per group of up to one candidate
// per CU three launches -- pack, Gram matrices, factor + forward solve + reductions -- and one copy back.  A
// candidate whose matrix does not factor as it stands (or whose augmented pivot fails) is handed to the
// lock-step schedule on its own, which runs the stable_cholesky ladder exactly as before.
static int lml_batch_wg(dfh_ctx* ctx, const dfh_kernel_desc* descs, int32_t nb, const double* dX,
                        int64_t n, int64_t d, const double* y, const double* mean_consts,
                        const double* noise_vars, int flags, double* lml_out, int32_t* jitter_powers) {
  const int64_t nbt = (n + 1 + 63) / 64, NP = 64 * nbt, sK = NP * NP;
  static const double group_gib = []() { const char* e = getenv("DFH_LML_GROUP_GIB"); double v = e ? atof(e) : 8.0; return v > 0.0 ? v : 8.0; }();
  static const int group_max = []() { const char* e = getenv("DFH_LML_WG_GROUP"); int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();
  const int64_t by_mem = std::max<int64_t>(1, (int64_t)(group_gib * 1073741824.0 / ((double)sK * 8.0)));
  const int64_t by_cu = group_max > 0 ? group_max : std::max(1, ctx->n_cu);
  const int G = (int)std::min<int64_t>(std::min<int64_t>(nb, by_cu), by_mem);
  // The labels stay resident between calls (round 6): a fitter asks thousands of times with the same y, and staging
  // 16 KB of pageable memory per call -- copy, synchronise -- was a sixth of a small group's call.  Host labels are
  // compared with the copy of the last call (memcmp: exact); device labels are used where they are.
  const double* dy = nullptr;
  double sum_y = 0.0, sum_y2 = 0.0;
  if (!(flags & DFH_LML_Y_IS_HOST) && is_device_ptr(y)) {
    dy = y;
    std::vector<double> y_host((size_t)n);
    DFH_HIP(hipMemcpyAsync(y_host.data(), y, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    DFH_HIP(hipStreamSynchronize(ctx->stream));
    for (int64_t i = 0; i < n; ++i) { sum_y += y_host[(size_t)i]; sum_y2 = fma(y_host[(size_t)i], y_host[(size_t)i], sum_y2); }
  } else {
    double* ybuf = nullptr;
    DFH_TRY(scratch_get(ctx, SCR_YCACHE, (size_t)std::max<int64_t>(2048, n) * 8, (void**)&ybuf));
    if (ybuf != ctx->ycache_dev || ctx->ycache_host.size() != (size_t)n ||
        std::memcmp(ctx->ycache_host.data(), y, (size_t)n * 8) != 0) {
      ctx->ycache_host.assign(y, y + n);
      DFH_HIP(hipMemcpyAsync(ybuf, ctx->ycache_host.data(), (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
      DFH_HIP(hipStreamSynchronize(ctx->stream));
      ctx->ycache_dev = ybuf;
      double s1 = 0.0, s2 = 0.0;
      for (int64_t i = 0; i < n; ++i) { s1 += y[i]; s2 = fma(y[i], y[i], s2); }
      ctx->ycache_sum = s1; ctx->ycache_sum2 = s2;
    }
    dy = ybuf;
    sum_y = ctx->ycache_sum; sum_y2 = ctx->ycache_sum2;
  }
  // One control block per group on the device -- results [2 g] | failed pivots [g] | status [1] | team flags -- zeroed by
  // ONE memset and copied back by ONE copy into the pinned buffer; descriptors and {aug. diagonal, mean, noise} go up
  // from the pinned buffer in ONE copy.  (Round 5: three pageable copies up, three memsets, three pageable copies
  // back and three synchronisations per group -- 160 of a small group's 210 us, profiles/r06_small_calls.txt.)
  double *Kb = nullptr, *ctl = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_KCT, (size_t)G * sK * 8, (void**)&Kb));
  const size_t ctl_bytes = (size_t)(3 * G + 8) * 8 + (size_t)G * LMLT_SYNC_INTS_PER_CANDIDATE * sizeof(int);
  DFH_TRY(scratch_get(ctx, SCR_LMLCTL, ctl_bytes, (void**)&ctl));
  std::vector<KernDev> kds((size_t)G);
  std::vector<char> skip((size_t)G, 0);
  // DFH_LML_TEAM: 0 = never a team, N = teams of up to N workgroups (default: up to 8)
  static const int team_env = []() { const char* e = getenv("DFH_LML_TEAM"); return e ? atoi(e) : -1; }();
  std::vector<int> redo;                       // candidates for the lock-step schedule
  for (int c0 = 0; c0 < nb; c0 += G) {
    const int g = std::min(G, nb - c0);
    int64_t Pmax = 0, parts_max = 0;
    size_t blob_bytes = 0;
    bool uniform = true;                       // structurally identical single-part kernels
    for (int c = 0; c < g; ++c) {
      kds[c] = KernDev();
      DFH_TRY(kerndev_build_host(&descs[c0 + c], &kds[c]));
      Pmax = std::max<int64_t>(Pmax, kds[c].P);
      parts_max = std::max<int64_t>(parts_max, kds[c].n_parts);
      blob_bytes += kerndev_blob_bytes(kds[c]);
      uniform = uniform && !kds[c].multi && kds[c].n_parts == 1 && kds[c].P == kds[0].P &&
                kerndev_blob_bytes(kds[c]) == kerndev_blob_bytes(kds[0]);
    }
    // pinned: descriptors | {aug. diagonal, mean, noise} [3 g] | (64-byte aligned) what comes back [3 g + 1]
    const size_t up_par = (blob_bytes + 15) & ~size_t(15), up_bytes = up_par + (size_t)g * 24;
    const size_t back_off = (up_bytes + 63) & ~size_t(63), back_bytes = (size_t)(3 * g + 1) * 8;
    void* pinned = nullptr;
    DFH_TRY(pinned_get(ctx, back_off + back_bytes, &pinned));
    char* hup = static_cast<char*>(pinned);
    double* hpar = reinterpret_cast<double*>(hup + up_par);
    const double* hred = reinterpret_cast<const double*>(hup + back_off);
    const long long* hinfo = reinterpret_cast<const long long*>(hup + back_off) + 2 * g;
    const unsigned long long* hstatus_p = reinterpret_cast<const unsigned long long*>(hup + back_off) + 3 * g;
    void* blob = nullptr;
    DFH_TRY(scratch_get(ctx, SCR_AUG2, up_bytes, &blob));
    DFH_TRY(kerndev_stage_many(kds.data(), g, hup, blob, blob_bytes));
    double* dpar = reinterpret_cast<double*>(static_cast<char*>(blob) + up_par);
    double* red = ctl;
    long long* dinfo = reinterpret_cast<long long*>(ctl + 2 * g);
    unsigned long long* d_status = reinterpret_cast<unsigned long long*>(ctl + 3 * g);
    int* d_sync = reinterpret_cast<int*>(ctl + 3 * g + 1);
    double *Xpb = nullptr, *Npb = nullptr;
    DFH_TRY(scratch_get(ctx, SCR_XS, (size_t)g * n * Pmax * 8, (void**)&Xpb));
    DFH_TRY(scratch_get(ctx, SCR_XS2, (size_t)g * n * parts_max * 8, (void**)&Npb));
    const int64_t sXp = n * Pmax, sNp = n * parts_max;
    for (int c = 0; c < g; ++c) {
      // the augmented row's diagonal entry: c = 1 + |y - m|^2 / s2 > z.z (the eigenvalues of K + s2 I are >= s2)
      // (|y - m|^2 = sum y^2 - 2 m sum y + n m^2: a bound needs no more than that, with a hair of slack for its rounding)
      const double m = mean_consts ? mean_consts[c0 + c] : 0.0, s2 = noise_vars[c0 + c];
      const double r2 = std::max(0.0, (sum_y2 - 2.0 * m * sum_y + (double)n * m * m)) * (1.0 + 1e-6) + 1e-6 * sum_y2;
      hpar[c] = 1.0 + r2 / s2;
      hpar[g + c] = m;
      hpar[2 * g + c] = s2;
      // (no noise, or a ratio beyond the double range: nothing bounds z.z -- such a candidate takes the lock-step schedule)
      skip[c] = !(s2 > 0.0) || !std::isfinite(hpar[c]);
      if (skip[c]) hpar[c] = 1.0;
    }
    DFH_HIP(hipMemcpyAsync(blob, hup, up_bytes, hipMemcpyHostToDevice, ctx->stream));
    // a group that leaves most of the device idle gets a TEAM of workgroups per candidate (chol.hip: lml_team_kernel)
    int team = 1;
    // (a timed-out hand-off costs ~0.1 s of polling plus the rebuilt group, and a slice sampler calls a hundred thousand
    //  times: after one, the context's next 32 groups take one workgroup per candidate -- a shared device does not pay
    //  the stall on every call; advisor, round 5)
    const bool team_cooling = ctx->lml_team_cooldown > 0;
    if (team_cooling) --ctx->lml_team_cooldown;
    if (team_env != 0 && !team_cooling) {
      const int cap = team_env > 0 ? team_env : 8;
      while (team * 2 <= cap && (int64_t)team * 2 * g <= ctx->n_cu && team * 2 <= nbt) team *= 2;
    }
    auto run_group = [&](int tm) -> int {
      {
        SectionTimer t(ctx, DFH_T_KERNMAT);
        if (uniform) {
          const int64_t sBlob = (int64_t)kerndev_blob_bytes(kds[0]);
          DFH_TRY(pack_scaled(ctx, kds[0], 0, 1, false, dX, n, d, Xpb, Npb, g, sBlob, sXp, sNp));
          DFH_TRY(kernmat_sym_batch(ctx, kds[0], g, sBlob, Xpb, sXp, Npb, sNp, n, dpar + 2 * g, Kb, sK, NP));
        } else {
          for (int c = 0; c < g; ++c) {
            double* Xp = Xpb + c * sXp; double* Np = Npb + c * sNp;
            DFH_TRY(pack_scaled(ctx, kds[c], 0, kds[c].n_parts, false, dX, n, d, Xp, Np));
            DFH_TRY(kernmat_packed(ctx, kds[c], 0, kds[c].n_parts, true, Xp, Np, n, Xp, Np, n, true, noise_vars[c0 + c],
                                   Kb + c * sK, NP));
          }
        }
      }
      {
        SectionTimer t(ctx, DFH_T_CHOL);
        // failed pivots, status and the team's flags: one memset (the results in front of them are always written)
        DFH_HIP(hipMemsetAsync(dinfo, 0, (size_t)(g + 1) * 8 + (tm > 1 ? (size_t)g * LMLT_SYNC_INTS_PER_CANDIDATE * sizeof(int) : 0),
                               ctx->stream));
        DFH_TRY(lml_wg_batch(ctx, Kb, sK, NP, n, g, dy, dpar, red, dinfo, tm, d_status, d_sync));
      }
      DFH_HIP(hipMemcpyAsync(hup + back_off, ctl, back_bytes, hipMemcpyDeviceToHost, ctx->stream));
      DFH_HIP(hipStreamSynchronize(ctx->stream));
      return DFH_OK;
    };
    DFH_TRY(run_group(team));
    if (team > 1 && *hstatus_p != 0) {
      // a hand-off between the members of a team timed out (the device is shared, or not all of them were
      // resident): the matrices are rebuilt and every candidate gets ONE workgroup, which waits for nobody
      ++ctx->chol_fallbacks;
      ctx->lml_team_cooldown = 32;
      DFH_TRY(run_group(1));
    }
    for (int c = 0; c < g; ++c) {
      if (skip[c] || hinfo[c] != 0 || !std::isfinite(hred[2 * c]) || !std::isfinite(hred[2 * c + 1])) { redo.push_back(c0 + c); continue; }
      if (jitter_powers) jitter_powers[c0 + c] = INT32_MIN;
      lml_out[c0 + c] = -0.5 * hred[2 * c + 1] - hred[2 * c] - 0.5 * (double)n * log(2.0 * M_PI);     // gp_core.py:224-226
    }
  }
  for (int c : redo)
    DFH_TRY(lml_batch_lockstep(ctx, descs + c, 1, dX, n, d, y, mean_consts ? mean_consts + c : nullptr, noise_vars + c, flags,
                               lml_out + c, jitter_powers ? jitter_powers + c : nullptr, c));
  return DFH_OK;
}

extern "C" int dfh_gp_lml_batch(dfh_ctx* ctx, const dfh_kernel_desc* descs, int32_t nb, const double* X,
                                int64_t n, int64_t d, const double* y, const double* mean_consts,
                                const double* noise_vars, int flags, double* lml_out,
                                int32_t* jitter_powers) {
  DFH_ARG(ctx && descs && nb >= 0 && X && y && noise_vars && lml_out && n >= 1 && d >= 1);
  if (nb == 0) return DFH_OK;
  for (int c = 0; c < nb; ++c) DFH_ARG(descs[c].dim == d);
  DFH_HIP(hipSetDevice(ctx->device));
  const double* dX = nullptr;
  if (flags & DFH_LML_X_IS_DEVICE) dX = X;
  else DFH_TRY(to_device(ctx, X, (size_t)n * d * 8, SCR_STAGE_A, &dX));
  static const bool tiny_enabled = []() { const char* e = getenv("DFH_LML_TINY"); return e ? atoi(e) != 0 : true; }();
  if (tiny_enabled && n > TINY64_MAX_N && n <= 255 && nb <= 64) {
    // a handful of mid-sized candidates (a slice sampler's call at 64 <= n <= 128): Gram matrix, factorisation and
    // forward solve of each in ONE launch by one workgroup, nothing copied (chol.hip: lml_wgf_kernel)
    std::vector<KernDev> all((size_t)nb);
    for (int c = 0; c < nb; ++c) DFH_TRY(kerndev_build_host(&descs[c], &all[c]));
    if (lml_wg_fused_applies(all.data(), nb, n)) {
      std::vector<double> y_dl, ld_dot((size_t)nb * 2);
      std::vector<long long> info((size_t)nb);
      const double* y_host = y;
      if (!(flags & DFH_LML_Y_IS_HOST) && is_device_ptr(y)) {
        y_dl.resize((size_t)n);
        DFH_HIP(hipMemcpyAsync(y_dl.data(), y, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
        DFH_HIP(hipStreamSynchronize(ctx->stream));
        y_host = y_dl.data();
      }
      {
        SectionTimer t(ctx, DFH_T_CHOL);
        DFH_TRY(lml_wg_fused_batch(ctx, all.data(), nb, dX, n, d, y_host, noise_vars, mean_consts, ld_dot.data(), info.data()));
      }
      for (int c = 0; c < nb; ++c) {
        if (info[c] != 0) {        // a failed pivot (the ladder) or no bound on the augmented pivot: the lock-step schedule, alone
          DFH_TRY(lml_batch_lockstep(ctx, descs + c, 1, dX, n, d, y, mean_consts ? mean_consts + c : nullptr, noise_vars + c,
                                     flags, lml_out + c, jitter_powers ? jitter_powers + c : nullptr, c));
          continue;
        }
        if (jitter_powers) jitter_powers[c] = INT32_MIN;
        lml_out[c] = -0.5 * ld_dot[2 * c + 1] - ld_dot[2 * c] - 0.5 * (double)n * log(2.0 * M_PI);     // gp_core.py:224-226
      }
      return DFH_OK;
    }
  }
  if (tiny_enabled && n <= TINY_MAX_N) {
    // small problems: pack, Gram matrix, stable_cholesky and the solve of every candidate in ONE
    // launch (kernmat.hip: k_lml_tiny)
    std::vector<KernDev> all((size_t)nb);
    for (int c = 0; c < nb; ++c) DFH_TRY(kerndev_build_host(&descs[c], &all[c]));
    if (lml_tiny_applies(all.data(), nb, n)) {
      std::vector<double> y_dl, ld_dot((size_t)nb * 2);
      const double* y_host = y;
      if (!(flags & DFH_LML_Y_IS_HOST) && is_device_ptr(y)) {
        y_dl.resize((size_t)n);
        DFH_HIP(hipMemcpyAsync(y_dl.data(), y, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
        DFH_HIP(hipStreamSynchronize(ctx->stream));
        y_host = y_dl.data();
      }
      SectionTimer t(ctx, DFH_T_CHOL);
      DFH_TRY(lml_tiny_batch(ctx, all.data(), nb, dX, n, d, y_host, noise_vars, mean_consts,
                             !(flags & DFH_FIT_NO_JITTER), ld_dot.data(), jitter_powers));
      for (int c = 0; c < nb; ++c)     // gp_core.py:224-226
        lml_out[c] = -0.5 * ld_dot[2 * c + 1] - ld_dot[2 * c] - 0.5 * (double)n * log(2.0 * M_PI);
      return DFH_OK;
    }
  }
  // one workgroup per candidate up to n = 2047 (DFH_LML_WG=0: the lock-step schedule for every n)
  static const int wg_min_batch = []() { const char* e = getenv("DFH_LML_WG_MIN_BATCH"); return e ? atoi(e) : 1; }();
  static const bool wg_enabled = []() { const char* e = getenv("DFH_LML_WG"); return e ? atoi(e) != 0 : true; }();
  if (wg_enabled && n <= LMLWG_MAX_N && nb >= wg_min_batch)
    return lml_batch_wg(ctx, descs, nb, dX, n, d, y, mean_consts, noise_vars, flags, lml_out, jitter_powers);
  return lml_batch_lockstep(ctx, descs, nb, dX, n, d, y, mean_consts, noise_vars, flags, lml_out, jitter_powers, 0);
}

extern "C" int dfh_gp_get(dfh_gp* gp, int what, double* out) {
  DFH_ARG(gp && out);
  dfh_ctx* ctx = gp->ctx;
  DFH_HIP(hipSetDevice(ctx->device));
  const int64_t n = gp->n;
  if (what == DFH_GET_ALPHA) return from_device(ctx, out, gp->alpha, (size_t)n * 8);
  if (what == DFH_GET_L) {
    if (!gp->upper_zeroed) { DFH_TRY(zero_upper(ctx, gp->L, n, n)); gp->upper_zeroed = true; }
    return from_device(ctx, out, gp->L, (size_t)n * n * 8);
  }
  if (what == DFH_GET_K) {
    DFH_ARG(!gp->gram);      // the caller evaluated the Gram matrix and still has it
    const bool dev_out = is_device_ptr(out);
    double* Kd = out;
    if (!dev_out) DFH_TRY(scratch_get(ctx, SCR_KCT, (size_t)n * n * 8, (void**)&Kd));
    DFH_TRY(kernmat_packed(ctx, gp->kd, 0, gp->kd.n_parts, true, gp->Xp, gp->Np, n, gp->Xp, gp->Np, n, true, 0.0, Kd, n));
    if (!dev_out) DFH_TRY(from_device(ctx, out, Kd, (size_t)n * n * 8));
    return DFH_OK;
  }
  dfh_set_error("dfh_gp_get: unknown selector %d", what);
  return DFH_ERR_BAD_ARG;
}

// shared driver for predict / acquisition arg-max
static int gp_eval_driver(dfh_gp* gp, int acq, const double* params, const double* Xs, int64_t m, int64_t ldxs,
                          int part_lo, int part_hi, bool pre_gathered, double kxx, const double* Xh, int64_t q,
                          double mean_const, const double* mean_vals, bool want_var, double* mu_out,
                          double* sd_out, double* vals_out, double* best_val, int64_t* best_idx) {
  dfh_ctx* ctx = gp->ctx;
  DFH_HIP(hipSetDevice(ctx->device));
  Halluc h;
  dfh_gp* aug = nullptr;                    // set when the variance comes from the re-factored augmented GP
  if (q > 0 && want_var) {
    int rc = halluc_prepare(gp, Xh, q, &h);
    if (rc == DFH_ERR_NOT_PD) {
      h.q = 0;
      rc = halluc_augmented_gp(gp, Xh, q, &aug);
    }
    DFH_TRY(rc);
  }
  struct AugGuard { dfh_gp* g; ~AugGuard() { if (g) dfh_gp_free(g); } } aug_guard{nullptr};
  aug_guard.g = aug;
  const int64_t mc_max = pick_chunk(gp->ctx, gp->n + (aug ? q : 0), m);
  const bool xs_dev = is_device_ptr(Xs);
  const bool mv_dev = mean_vals ? is_device_ptr(mean_vals) : true;
  double* vec = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_VEC, (size_t)mc_max * 8 * 8, (void**)&vec));
  // a kernel with a polynomial / exponential-decay factor: k(x, x) per candidate; on the add-UCB group path
  // (pre_gathered: one group of an additive kernel) the group's own prior variance, if it is such a group
  bool range_stationary = gp->kd.stationary;
  if (pre_gathered) {
    range_stationary = true;
    for (int g = part_lo; g < part_hi; ++g)
      range_stationary = range_stationary && (gp->kd.parts[g].kind == DFH_KERNEL_SE || gp->kd.parts[g].kind == DFH_KERNEL_MATERN);
  }
  double* kss = (want_var && !range_stationary) ? vec + 7 * mc_max : nullptr;
  double* mu_raw = vec; double* ss = vec + mc_max; double* ss2 = vec + 2 * mc_max;
  double* mu_c = vec + 3 * mc_max; double* sd_c = vec + 4 * mc_max; double* val_c = vec + 5 * mc_max;
  bool have = false; double bv = 0.0; int64_t bi = -1;
  const double p0 = params ? params[0] : 0.0, p1 = params ? params[1] : 0.0;
  for (int64_t i0 = 0; i0 < m; i0 += mc_max) {
    const int64_t mc = std::min(mc_max, m - i0);
    const double* xs_c = nullptr;
    if (xs_dev) xs_c = Xs + i0 * ldxs;
    else DFH_TRY(to_device(ctx, Xs + i0 * ldxs, (size_t)mc * ldxs * 8, SCR_STAGE_A, &xs_c));
    const double* mv_c = nullptr;
    if (mean_vals) {
      if (mv_dev) mv_c = mean_vals + i0;
      else DFH_TRY(to_device(ctx, mean_vals + i0, (size_t)mc * 8, SCR_STAGE_D, &mv_c));
    }
    double* xsp = nullptr; double* nsp = nullptr;
    if (aug) {
      // mean from the real data, variance from the augmented factor (gp_core.py:195, 207-213)
      DFH_TRY(posterior_chunk(aug, xs_c, mc, ldxs, part_lo, part_hi, pre_gathered, true, nullptr, nullptr, vec + 6 * mc_max, ss, ss2));
      DFH_TRY(posterior_chunk(gp, xs_c, mc, ldxs, part_lo, part_hi, pre_gathered, false, nullptr, nullptr, mu_raw, nullptr, nullptr,
                              0, &xsp, &nsp));
    } else {
      DFH_TRY(posterior_chunk(gp, xs_c, mc, ldxs, part_lo, part_hi, pre_gathered, want_var, &h, nullptr, mu_raw, ss, ss2,
                              0, &xsp, &nsp));
    }
    if (kss) DFH_TRY(pre_gathered ? prior_diag(ctx, gp->kd, xsp, nsp, mc, kss, part_lo, part_hi) : prior_diag(ctx, gp->kd, xsp, nsp, mc, kss));
    {
      SectionTimer t(ctx, DFH_T_ACQ);
      const bool need_val = vals_out || best_val || best_idx;
      hipLaunchKernelGGL(k_posterior_acq, dim3((unsigned)((mc + 255) / 256)), dim3(256), 0, ctx->stream, acq, p0, p1,
                         kxx, kss, mean_const, mv_c, mu_raw, want_var ? ss : nullptr,
                         (want_var && h.q > 0) ? ss2 : nullptr, (long)mc, mu_out ? mu_c : nullptr,
                         sd_out ? sd_c : nullptr, need_val ? val_c : nullptr);
      DFH_LAUNCH_CHECK();
      if (need_val && (best_val || best_idx)) DFH_TRY(argmax_update(ctx, val_c, mc, i0, &have, &bv, &bi));
    }
    if (mu_out) DFH_TRY(from_device(ctx, mu_out + i0, mu_c, (size_t)mc * 8));
    if (sd_out) DFH_TRY(from_device(ctx, sd_out + i0, sd_c, (size_t)mc * 8));
    if (vals_out) DFH_TRY(from_device(ctx, vals_out + i0, val_c, (size_t)mc * 8));
  }
  DFH_HIP(hipStreamSynchronize(ctx->stream));
  if (best_val) *best_val = bv;
  if (best_idx) *best_idx = bi;
  return DFH_OK;
}

extern "C" int dfh_gp_predict(dfh_gp* gp, const double* Xs, int64_t m, const double* Xh, int64_t q,
                              double* mu_out, double* sd_out) {
  DFH_ARG(gp && m >= 0 && q >= 0);
  DFH_ARG(!gp->gram);      // needs the kernel: this posterior was built from a Gram matrix
  if (m == 0) return DFH_OK;
  DFH_ARG(Xs && mu_out && (q == 0 || Xh));
  return gp_eval_driver(gp, DFH_ACQ_MEAN, nullptr, Xs, m, gp->d, 0, gp->kd.n_parts, false, gp->kd.kxx, Xh, q,
                        0.0, nullptr, sd_out != nullptr, mu_out, sd_out, nullptr, nullptr, nullptr);
}

extern "C" int dfh_gp_acq_argmax(dfh_gp* gp, int acq, const double* params, const double* Xs, int64_t m,
                                 const double* Xh, int64_t q, double mean_const, const double* mean_vals,
                                 double* vals_out, double* best_val, int64_t* best_idx) {
  DFH_ARG(gp && m >= 1 && Xs && q >= 0 && (q == 0 || Xh));
  DFH_ARG(!gp->gram);      // needs the kernel: this posterior was built from a Gram matrix
  DFH_ARG(acq >= DFH_ACQ_MEAN && acq <= DFH_ACQ_STD);
  DFH_ARG(params || acq == DFH_ACQ_MEAN || acq == DFH_ACQ_STD);
  const bool want_var = acq != DFH_ACQ_MEAN;
  return gp_eval_driver(gp, acq, params, Xs, m, gp->d, 0, gp->kd.n_parts, false, gp->kd.kxx, Xh, q, mean_const,
                        mean_vals, want_var, nullptr, nullptr, vals_out, best_val, best_idx);
}

extern "C" int dfh_gp_add_ucb_group(dfh_gp* gp, int32_t group, double beta, const double* Xg, int64_t m,
                                    double* vals_out, double* best_val, int64_t* best_idx) {
  DFH_ARG(gp && Xg && m >= 1);
  DFH_ARG(!gp->gram);      // needs the kernel: this posterior was built from a Gram matrix
  DFH_ARG(gp->kd.multi && !gp->kd.product && group >= 0 && group < gp->kd.n_parts);   // additive kernels only
  const PartDev& pd = gp->kd.parts[group];
  int gdim = 0;
  for (int c = 0; c < pd.kc; ++c) gdim += gp->kd.cols[pd.poff + c] >= 0;
  const double kxx = gp->kd.outer_scale * kerndev_part_kxx(gp->kd, group);   // kern_scale * kernel_j(x,x)
  const double params[2] = {beta, 0.0};
  return gp_eval_driver(gp, DFH_ACQ_UCB, params, Xg, m, gdim, group, group + 1, true, kxx, nullptr, 0, 0.0,
                        nullptr, true, nullptr, nullptr, vals_out, best_val, best_idx);
}

// All additive groups at once: the per-group cross matrices are stacked into one (sum m_g) x n
// matrix so that the posterior solve is ONE triangular solve with sum(m_g) right-hand sides instead
// of G small ones (the reference issues G solve_lower_triangular calls, gpb_acquisitions.py:161-176).
// Xg_all: group g's candidates [m_g x |group g|], back to back.  Falls back to the per-group
// route when the stack does not fit one posterior chunk.
extern "C" int dfh_gp_add_ucb_all(dfh_gp* gp, const double* betas, const double* Xg_all, const int64_t* m_per_group,
                                  double* vals_out, double* best_vals, int64_t* best_idx) {
  DFH_ARG(gp && betas && Xg_all && m_per_group && best_vals && best_idx);
  DFH_ARG(!gp->gram);
  DFH_ARG(gp->kd.multi && !gp->kd.product);
  dfh_ctx* ctx = gp->ctx;
  DFH_HIP(hipSetDevice(ctx->device));
  const KernDev& kd = gp->kd;
  const int G = kd.n_parts;
  const int64_t n = gp->n;
  std::vector<int64_t> off(G + 1, 0), xoff(G + 1, 0);
  std::vector<int> gdim(G, 0);
  for (int g = 0; g < G; ++g) {
    DFH_ARG(m_per_group[g] >= 1);
    for (int c = 0; c < kd.parts[g].kc; ++c) gdim[g] += kd.cols[kd.parts[g].poff + c] >= 0;
    off[g + 1] = off[g] + m_per_group[g];
    xoff[g + 1] = xoff[g] + m_per_group[g] * gdim[g];
  }
  const int64_t M = off[G];
  if (M > pick_chunk(ctx, n, M)) {
    for (int g = 0; g < G; ++g)
      DFH_TRY(dfh_gp_add_ucb_group(gp, g, betas[g], Xg_all + xoff[g], m_per_group[g],
                                   vals_out ? vals_out + off[g] : nullptr, &best_vals[g], &best_idx[g]));
    return DFH_OK;
  }
  const double* dXg = nullptr;
  DFH_TRY(to_device(ctx, Xg_all, (size_t)xoff[G] * 8, SCR_STAGE_A, &dXg));
  static_assert(sizeof(long) == sizeof(int64_t), "offsets travel as int64");
  const double* d_off = nullptr;                     // segment offsets for the per-group arg-max
  DFH_TRY(to_device(ctx, reinterpret_cast<const double*>(off.data()), (size_t)(G + 1) * 8, SCR_STAGE_B, &d_off));
  char* xs = nullptr;
  const size_t b_xsp = ((size_t)M * kd.P * 8 + 255) / 256 * 256;
  DFH_TRY(scratch_get(ctx, SCR_XS, b_xsp + (size_t)M * kd.n_parts * 8, (void**)&xs));
  double* Xsp = reinterpret_cast<double*>(xs);
  double* Nsp = reinterpret_cast<double*>(xs + b_xsp);
  double *Kct = nullptr, *vec = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_KCT, (size_t)M * n * 8, (void**)&Kct));
  DFH_TRY(scratch_get(ctx, SCR_VEC, (size_t)M * 8 * 4, (void**)&vec));
  double* mu_raw = vec; double* ss = vec + M; double* val = vec + 2 * M; double* kss_w = vec + 3 * M;
  {
    SectionTimer t(ctx, DFH_T_CROSS);
    bool mu_all = true;          // posterior means from the cross-matrix pass itself where the kernel can
    for (int g = 0; g < G; ++g) {
      double* Xsp_g = Xsp + off[g] * kd.P; double* Nsp_g = Nsp + off[g] * kd.n_parts;
      DFH_TRY(pack_scaled(ctx, kd, g, g + 1, true, dXg + xoff[g], m_per_group[g], gdim[g], Xsp_g, Nsp_g));
      // K_j(X*_j, X[:, group j]) with the outer scale        (gpb_acquisitions.py:166-168)
      bool mu_done = false;
      DFH_TRY(kernmat_packed(ctx, kd, g, g + 1, true, Xsp_g, Nsp_g, m_per_group[g], gp->Xp, gp->Np, n, false,
                             0.0, Kct + off[g] * n, n, gp->alpha, mu_raw + off[g], &mu_done));
      mu_all = mu_all && mu_done;
    }
    if (!mu_all) DFH_TRY(gemv_rows(ctx, Kct, M, n, n, gp->alpha, 1.0, nullptr, 0.0, mu_raw));
  }
  {
    SectionTimer t(ctx, DFH_T_TRSM);
    DFH_TRY(trsm_rows(ctx, gp->L, n, n, gp->inv, Kct, M, n, gp->refine.data()));
  }
  SectionTimer t(ctx, DFH_T_ACQ);
  DFH_TRY(row_sumsq(ctx, Kct, M, n, n, ss));
  for (int g = 0; g < G; ++g) {
    const double kxx = kd.outer_scale * kerndev_part_kxx(kd, g);        // kern_scale * kernel_j(x, x)
    const int64_t mg = m_per_group[g];
    const double* kss_g = nullptr;
    if (kd.parts[g].kind != DFH_KERNEL_SE && kd.parts[g].kind != DFH_KERNEL_MATERN) {
      // a polynomial group: its prior variance depends on the point
      DFH_TRY(prior_diag(ctx, kd, Xsp + off[g] * kd.P, Nsp + off[g] * kd.n_parts, mg, kss_w + off[g], g, g + 1));
      kss_g = kss_w + off[g];
    }
    hipLaunchKernelGGL(k_posterior_acq, dim3((unsigned)((mg + 255) / 256)), dim3(256), 0, ctx->stream, (int)DFH_ACQ_UCB,
                       betas[g], 0.0, kxx, kss_g, 0.0, (const double*)nullptr, mu_raw + off[g], ss + off[g],
                       (const double*)nullptr, (long)mg, (double*)nullptr, (double*)nullptr, val + off[g]);
    DFH_LAUNCH_CHECK();
  }
  {   // the G arg-maxes in one launch and one copy back (each used to cost a stream synchronisation)
    char* red = nullptr;
    DFH_TRY(scratch_get(ctx, SCR_RED, (size_t)G * 16, (void**)&red));
    double* d_bv = reinterpret_cast<double*>(red);
    long* d_bi = reinterpret_cast<long*>(red + (size_t)G * 8);
    hipLaunchKernelGGL(k_argmax_segments, dim3((unsigned)G), dim3(256), 0, ctx->stream, val,
                       reinterpret_cast<const long*>(d_off), d_bv, d_bi);
    DFH_LAUNCH_CHECK();
    DFH_HIP(hipMemcpyAsync(best_vals, d_bv, (size_t)G * 8, hipMemcpyDeviceToHost, ctx->stream));
    DFH_HIP(hipMemcpyAsync(best_idx, d_bi, (size_t)G * 8, hipMemcpyDeviceToHost, ctx->stream));
  }
  if (vals_out) DFH_TRY(from_device(ctx, vals_out, val, (size_t)M * 8));
  DFH_HIP(hipStreamSynchronize(ctx->stream));
  return DFH_OK;
}

extern "C" int dfh_gp_predict_covar(dfh_gp* gp, const double* Xs, int64_t m, const double* Xh, int64_t q,
                                    double* mu_out, double* cov_out) {
  DFH_ARG(gp && m >= 0 && q >= 0);
  DFH_ARG(!gp->gram);      // needs the kernel: this posterior was built from a Gram matrix
  if (m == 0) return DFH_OK;
  DFH_ARG(Xs && mu_out && cov_out && (q == 0 || Xh));
  DFH_ARG((double)m * (double)gp->n * 8.0 < 64e9);
  dfh_ctx* ctx = gp->ctx;
  DFH_HIP(hipSetDevice(ctx->device));
  const KernDev& kd = gp->kd;
  Halluc h;
  if (q > 0) {
    const int rc = halluc_prepare(gp, Xh, q, &h);
    if (rc == DFH_ERR_NOT_PD) {
      // covariance from the augmented GP factored from scratch, mean from the real data
      dfh_gp* aug = nullptr;
      DFH_TRY(halluc_augmented_gp(gp, Xh, q, &aug));
      int rc2 = dfh_gp_predict_covar(aug, Xs, m, nullptr, 0, mu_out, cov_out);
      dfh_gp_free(aug);
      DFH_TRY(rc2);
      return dfh_gp_predict(gp, Xs, m, nullptr, 0, mu_out, nullptr);
    }
    DFH_TRY(rc);
  }
  const double* dXs = nullptr;
  DFH_TRY(to_device(ctx, Xs, (size_t)m * gp->d * 8, SCR_STAGE_A, &dXs));
  double* vec = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_VEC, (size_t)m * 8 * 3, (void**)&vec));
  double* Kct = nullptr;
  DFH_TRY(posterior_chunk(gp, dXs, m, gp->d, 0, kd.n_parts, false, true, nullptr, &Kct, vec, vec + m, vec + 2 * m));
  DFH_TRY(from_device(ctx, mu_out, vec, (size_t)m * 8));
  // cov = K(Xs,Xs) - V^T V     (gp_core.py:179-181)
  const bool dev_out = is_device_ptr(cov_out);
  double* C = cov_out;
  if (!dev_out) DFH_TRY(scratch_get(ctx, SCR_TSK, (size_t)m * m * 8, (void**)&C));
  char* xs = reinterpret_cast<char*>(ctx->scratch[SCR_XS].p);      // packed Xs left by posterior_chunk
  double* Xsp = reinterpret_cast<double*>(xs);
  double* Nsp = reinterpret_cast<double*>(xs + ((size_t)m * kd.P * 8 + 255) / 256 * 256);
  DFH_TRY(kernmat_packed(ctx, kd, 0, kd.n_parts, true, Xsp, Nsp, m, Xsp, Nsp, m, true, 0.0, C, m));
  DFH_TRY(gemm_f64(ctx, 0, m, m, gp->n, -1.0, Kct, gp->n, Kct, gp->n, 1.0, C, m, C, m));
  if (q > 0) {
    // second block row of the augmented solve: V2t = (k(Xs,Xh) - V1t Wt^T) Lh^-T ; cov -= V2t V2t^T
    double* T = nullptr;
    DFH_TRY(scratch_get(ctx, SCR_AUG2, (size_t)m * q * 8, (void**)&T));
    DFH_TRY(kernmat_packed(ctx, kd, 0, kd.n_parts, true, Xsp, Nsp, m, h.Xhp, h.Nhp, q, false, 0.0, T, q));
    DFH_TRY(gemm_f64(ctx, 0, m, q, gp->n, -1.0, Kct, gp->n, h.Wt, gp->n, 1.0, T, q, T, q));
    hipLaunchKernelGGL(k_halluc_rows, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, T, (long)m, (int)q, h.Lh, vec + 2 * m);
    DFH_LAUNCH_CHECK();
    DFH_TRY(gemm_f64(ctx, 0, m, m, q, -1.0, T, q, T, q, 1.0, C, m, C, m));
  }
  if (!dev_out) DFH_TRY(from_device(ctx, cov_out, C, (size_t)m * m * 8));
  DFH_HIP(hipStreamSynchronize(ctx->stream));
  return DFH_OK;
}

extern "C" int dfh_gp_ts(dfh_gp* gp, const double* Xs, int64_t m, int64_t block, const double* U,
                         double mean_const, const double* mean_vals, double* samples_out, double* best_val,
                         int64_t* best_idx, int32_t* jitter_powers_out) {
  DFH_ARG(gp && Xs && U && m >= 1 && block >= 1);
  DFH_ARG(!gp->gram);      // needs the kernel: this posterior was built from a Gram matrix
  dfh_ctx* ctx = gp->ctx;
  DFH_HIP(hipSetDevice(ctx->device));
  const KernDev& kd = gp->kd;
  const int64_t n = gp->n;
  if (block > m) block = m;
  DFH_ARG((double)block * (double)block * 8.0 < 32e9);
  // several TS blocks share one posterior chunk so the TRSM runs on big GEMMs
  int64_t bpc = std::max<int64_t>(1, pick_chunk(ctx, n, m) / block);
  const int64_t mc_max = std::min(m, bpc * block);
  const int64_t nchunks = (m + mc_max - 1) / mc_max;
  const bool xs_dev = is_device_ptr(Xs), u_dev = is_device_ptr(U);
  const bool mv_dev = mean_vals ? is_device_ptr(mean_vals) : true;
  double* vec[2] = {nullptr, nullptr};
  DFH_TRY(scratch_get(ctx, SCR_VEC, (size_t)mc_max * 8 * 2, (void**)&vec[0]));
  DFH_TRY(scratch_get(ctx, SCR_VECB, (size_t)mc_max * 8 * 2, (void**)&vec[1]));
  // up to DFH_TS_BATCH (64) blocks of a chunk are factored as one lock-step batch
  static const int ts_batch = []() { const char* e = getenv("DFH_TS_BATCH"); int v = e ? atoi(e) : 64; return v < 1 ? 1 : (v > CHOL_MAX_BATCH ? CHOL_MAX_BATCH : v); }();
  const int64_t lb_slots = std::max<int64_t>(1, std::min<int64_t>(ts_batch, mc_max / block));
  double* Lb = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_TSL, (size_t)lb_slots * block * block * 8, (void**)&Lb));
  // Two-stage software pipeline over chunks.  Stage 1 (low-priority `bulk` stream): cross kernel
  // matrix, mu, and the posterior TRSM of chunk c+1 -- large MFMA GEMMs.  Stage 2 (main + panel
  // streams): per TS block of chunk c the covariance SYRK, its stable_cholesky (latency-bound
  // look-ahead factorisation, host-synchronous because of the jitter ladder) and the draw.
  // The factorisations hide behind the next chunk's TRSM instead of idling the GPU.
  static const bool ts_half_occ = []() { const char* e = getenv("DFH_TS_HALF_OCC"); return e ? atoi(e) != 0 : false; }();
  hipStream_t mainS = ctx->stream, bulkS = ctx->bulk;
  struct Stage1 { double* Kct; double* Xsp; double* Nsp; double* mu; };
  Stage1 st[2];
  hipEvent_t ev_in, ev_ready[2], ev_free[2];
  DFH_TRY(ctx_event(ctx, EV_TS_BASE, &ev_in));
  for (int p = 0; p < 2; ++p) {
    DFH_TRY(ctx_event(ctx, EV_TS_BASE + 1 + p, &ev_ready[p]));
    DFH_TRY(ctx_event(ctx, EV_TS_BASE + 3 + p, &ev_free[p]));
  }
  DFH_HIP(hipEventRecord(ev_in, mainS));
  DFH_HIP(hipStreamWaitEvent(bulkS, ev_in, 0));        // inputs produced on the main stream are ready

  auto stage1 = [&](int64_t c) -> int {
    const int p = (int)(c & 1);
    const int64_t i0 = c * mc_max;
    const int64_t mc = std::min(mc_max, m - i0);
    StreamSwap on_bulk(ctx, bulkS);
    // DFH_TS_HALF_OCC=1: one workgroup per CU for the bulk GEMMs so the other half of each CU stays
    // free for the latency-bound factorisation kernels of stage 2.  Off by default: with the blocks
    // of a chunk factored as one batch the latency-bound share is small and the ~14% the bulk GEMMs
    // lose at half occupancy costs more than the overlap returns (measured 1602 vs 1495 ms/step).
    struct HalfOcc { dfh_ctx* c; bool old; HalfOcc(dfh_ctx* x, bool v) : c(x), old(x->gemm_half_occupancy) { c->gemm_half_occupancy = v; }
                     ~HalfOcc() { c->gemm_half_occupancy = old; } } half(ctx, nchunks > 1 && ts_half_occ);
    if (c >= 2) DFH_HIP(hipStreamWaitEvent(bulkS, ev_free[p], 0));   // parity buffers released by stage 2
    const double* xs_c = nullptr;
    if (xs_dev) xs_c = Xs + i0 * gp->d;
    else DFH_TRY(to_device(ctx, Xs + i0 * gp->d, (size_t)mc * gp->d * 8, p ? SCR_STAGE_A2 : SCR_STAGE_A, &xs_c));
    st[p].mu = vec[p];
    DFH_TRY(posterior_chunk(gp, xs_c, mc, gp->d, 0, kd.n_parts, false, true, nullptr, &st[p].Kct, st[p].mu,
                            nullptr, nullptr, p, &st[p].Xsp, &st[p].Nsp));
    DFH_HIP(hipEventRecord(ev_ready[p], bulkS));
    return DFH_OK;
  };

  bool have = false; double bv = 0.0; int64_t bi = -1;
  int64_t blk_idx = 0;
  DFH_TRY(stage1(0));
  for (int64_t c = 0; c < nchunks; ++c) {
    const int p = (int)(c & 1);
    const int64_t i0 = c * mc_max;
    const int64_t mc = std::min(mc_max, m - i0);
    if (c + 1 < nchunks) DFH_TRY(stage1(c + 1));        // enqueue ahead: overlaps with the blocks below
    DFH_HIP(hipStreamWaitEvent(mainS, ev_ready[p], 0));
    const double* u_c = nullptr;
    if (u_dev) u_c = U + i0;
    else DFH_TRY(to_device(ctx, U + i0, (size_t)mc * 8, SCR_STAGE_C, &u_c));
    const double* mv_c = nullptr;
    if (mean_vals) {
      if (mv_dev) mv_c = mean_vals + i0;
      else DFH_TRY(to_device(ctx, mean_vals + i0, (size_t)mc * 8, SCR_STAGE_D, &mv_c));
    }
    double* mu_raw = st[p].mu;
    double* samp = vec[p] + mc_max;
    double* Kct = st[p].Kct;
    // mean_vals = test_mean + K_tetr alpha
    hipLaunchKernelGGL(k_add_vec, dim3((unsigned)((mc + 255) / 256)), dim3(256), 0, ctx->stream, mu_raw, mv_c,
                       mv_c ? 0.0 : mean_const, (long)mc);
    DFH_LAUNCH_CHECK();
    // one TS block: Sigma = K(Xb,Xb) - V^T V (gp_core.py:179-181; chol reads the lower triangle),
    // stable_cholesky (general_utils.py:229), s = L u + mu (general_utils.py:231)
    auto sigma_kernel = [&](int64_t b0, int64_t B, double* dst) -> int {
      const double* Xbp = st[p].Xsp + b0 * kd.P;
      const double* Nbp = st[p].Nsp + b0 * kd.n_parts;
      return kernmat_packed(ctx, kd, 0, kd.n_parts, true, Xbp, Nbp, B, Xbp, Nbp, B, true, 0.0, dst, B);
    };
    auto single_block = [&](int64_t b0, int64_t B, double* dst, int64_t bidx) -> int {
      const double* Vt = Kct + b0 * n;
      auto build_sigma = [&]() -> int {
        DFH_TRY(sigma_kernel(b0, B, dst));
        return gemm_f64(ctx, GEMM_LOWER, B, B, n, -1.0, Vt, n, Vt, n, 1.0, dst, B, dst, B);
      };
      DFH_TRY(build_sigma());
      int32_t jp = INT32_MIN;
      DFH_TRY(stable_cholesky_device(ctx, dst, B, nullptr, true, build_sigma, &jp, nullptr));
      if (jitter_powers_out) jitter_powers_out[bidx] = jp;
      return DFH_OK;
    };
    const int64_t nfull = mc / block;
    for (int64_t g0 = 0; g0 < nfull; g0 += lb_slots) {
      // the equal-sized blocks of the chunk are factored in lock-step: one batched launch sequence
      // instead of `nb` latency-bound ones
      const int nb = (int)std::min<int64_t>(lb_slots, nfull - g0);
      const int64_t B = block;
      SectionTimer t(ctx, DFH_T_TS);
      if (nb == 1) {
        DFH_TRY(single_block(g0 * B, B, Lb, blk_idx + g0));
      } else {
        // (also the rebuild closure of the factorisation: small groups take the one-launch panels, whose
        //  hand-offs are bounded waits -- on expiry, e.g. with other contexts crowding the device, the group
        //  is rebuilt and factored on the schedule without inter-workgroup waits)
        const std::function<int()> build_group = [&]() -> int {
          for (int b = 0; b < nb; ++b) DFH_TRY(sigma_kernel((g0 + b) * B, B, Lb + b * B * B));
          GemmBatch bs;
          bs.count = nb; bs.sA = bs.sB = B * n; bs.sCin = bs.sCout = B * B;
          const double* Vt = Kct + g0 * B * n;
          return gemm_f64(ctx, GEMM_LOWER, B, B, n, -1.0, Vt, n, Vt, n, 1.0, Lb, B, Lb, B, &bs);
        };
        DFH_TRY(build_group());
        int64_t piv[CHOL_MAX_BATCH] = {0};
        int rc = cholesky_device(ctx, Lb, B, B, nullptr, piv, nb, B * B, 0, nullptr, false, &build_group);
        if (rc != DFH_OK && rc != DFH_ERR_NOT_PD) return rc;
        for (int b = 0; b < nb; ++b) {
          if (piv[b] == 0) { if (jitter_powers_out) jitter_powers_out[blk_idx + g0 + b] = INT32_MIN; continue; }
          // this block needs the jitter ladder: redo it alone (rebuilds Sigma first)
          DFH_TRY(single_block((g0 + b) * B, B, Lb + b * B * B, blk_idx + g0 + b));
        }
      }
      for (int b = 0; b < nb; ++b) {
        const int64_t b0 = (g0 + b) * B;
        DFH_TRY(gemv_rows(ctx, Lb + b * B * B, B, B, B, u_c + b0, 1.0, mu_raw + b0, 1.0, samp + b0, true));
      }
    }
    if (nfull * block < mc) {           // ragged last block
      const int64_t b0 = nfull * block, B = mc - b0;
      SectionTimer t(ctx, DFH_T_TS);
      DFH_TRY(single_block(b0, B, Lb, blk_idx + nfull));
      DFH_TRY(gemv_rows(ctx, Lb, B, B, B, u_c + b0, 1.0, mu_raw + b0, 1.0, samp + b0, true));
    }
    blk_idx += (mc + block - 1) / block;
    DFH_TRY(argmax_update(ctx, samp, mc, i0, &have, &bv, &bi));
    if (samples_out) DFH_TRY(from_device(ctx, samples_out + i0, samp, (size_t)mc * 8));
    DFH_HIP(hipEventRecord(ev_free[p], mainS));
  }
  DFH_HIP(hipStreamSynchronize(mainS));
  DFH_HIP(hipStreamSynchronize(bulkS));
  if (best_val) *best_val = bv;
  if (best_idx) *best_idx = bi;
  return DFH_OK;
}

#ifdef DFH_DEBUG_HOOKS      // diagnostics: built only with `python -m dragonfly_amd.build --debug-hooks` (include/dfhip_debug.h)
// Diagnostics hook (not part of the product path): do kernels on the bulk stream and on the main /
// panel streams actually run concurrently?  Enqueues `n_big` large GEMMs on stream A and `n_small`
// tiny kernels on stream B and reports the time of each alone and together.
// which: 0 = A is bulk, B is main; 1 = A is main, B is side; 2 = A is bulk, B is side
extern "C" int dfh_debug_overlap(dfh_ctx* ctx, int which, int n_big, int n_small, double* out_ms /*[4]*/) {
  DFH_ARG(ctx && out_ms);
  hipStream_t A = (which == 1) ? ctx->main_stream : ctx->bulk;
  hipStream_t B = (which == 0) ? ctx->main_stream : ctx->side;
  const int64_t M = 16384, N = 512, K = 4096;
  double *a = nullptr, *b = nullptr, *c = nullptr, *v = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_KCT, (size_t)M * K * 8, (void**)&a));
  DFH_TRY(scratch_get(ctx, SCR_TSK, (size_t)N * K * 8, (void**)&b));
  DFH_TRY(scratch_get(ctx, SCR_TMP, (size_t)M * N * 8, (void**)&c));
  DFH_TRY(scratch_get(ctx, SCR_VEC, 1 << 20, (void**)&v));
  DFH_TRY(fill_f64(ctx, a, M * K, 0.5));
  DFH_TRY(fill_f64(ctx, b, N * K, 0.25));
  DFH_HIP(hipDeviceSynchronize());
  int least = 0, greatest = 0;
  DFH_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
  out_ms[3] = least * 100.0 + greatest;
  auto run = [&](bool big, bool small, double* ms) -> int {
    DFH_HIP(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    if (big) {
      StreamSwap sw(ctx, A);
      for (int i = 0; i < n_big; ++i) DFH_TRY(gemm_f64(ctx, 0, M, N, K, 1.0, a, K, b, K, 0.0, nullptr, 0, c, N));
    }
    if (small) {
      StreamSwap sw(ctx, B);
      for (int i = 0; i < n_small; ++i) {
        if (which >= 10) DFH_TRY(fill_f64(ctx, v, 4096, 1.0));
        else DFH_TRY(gemm_f64(ctx, 0, 64, 64, 64, 1.0, a, K, b, K, 0.0, nullptr, 0, v, 64));   // 38 KB LDS, 1 workgroup
      }
    }
    DFH_HIP(hipStreamSynchronize(B));
    auto t1 = std::chrono::steady_clock::now();
    DFH_HIP(hipDeviceSynchronize());
    auto t2 = std::chrono::steady_clock::now();
    ms[0] = std::chrono::duration<double, std::milli>(t1 - t0).count();
    ms[1] = std::chrono::duration<double, std::milli>(t2 - t0).count();
    return DFH_OK;
  };
  double m[2];
  DFH_TRY(run(true, false, m));  out_ms[0] = m[1];            // big alone
  DFH_TRY(run(false, true, m));  out_ms[1] = m[1];            // small alone
  DFH_TRY(run(true, true, m));   out_ms[2] = m[0];            // small-stream completion time when both run
  out_ms[3] += m[1] * 1e6;                                    // total together (packed: ms*1e6 + prio)
  return DFH_OK;
}

// Diagnostics hook: achievable pure-write HBM bandwidth (the kernel-matrix build is write-only).
__global__ void k_dbg_fill16(double2_t* p, long n2, double v) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  const double2_t vv = (double2_t){v, v};
  for (; i < n2; i += stride) p[i] = vv;
}
__global__ void k_dbg_copy16(const double2_t* __restrict__ s, double2_t* __restrict__ d, long n2) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n2; i += stride) d[i] = s[i];
}
extern "C" int dfh_debug_write_bw(dfh_ctx* ctx, double gbytes, double* out /*[3]: fill16 TB/s, memset TB/s, copy TB/s (r+w)*/) {
  DFH_ARG(ctx && out && gbytes > 0.01 && gbytes < 16);
  const long n2 = (long)(gbytes * 1e9 / 16);
  double2_t *a = nullptr, *b = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_KCT, (size_t)n2 * 16, (void**)&a));
  DFH_TRY(scratch_get(ctx, SCR_KCT2, (size_t)n2 * 16, (void**)&b));
  hipEvent_t e0, e1;
  DFH_HIP(hipEventCreate(&e0)); DFH_HIP(hipEventCreate(&e1));
  float ms;
  for (int grid : {2048, 8192}) {
    hipLaunchKernelGGL(k_dbg_fill16, dim3(grid), dim3(256), 0, ctx->stream, a, n2, 1.0);
  }
  DFH_HIP(hipEventRecord(e0, ctx->stream));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_dbg_fill16, dim3(4096), dim3(256), 0, ctx->stream, a, n2, 2.0);
  DFH_HIP(hipEventRecord(e1, ctx->stream)); DFH_HIP(hipEventSynchronize(e1));
  DFH_HIP(hipEventElapsedTime(&ms, e0, e1)); out[0] = 5.0 * n2 * 16 / (ms * 1e-3) / 1e12;
  DFH_HIP(hipEventRecord(e0, ctx->stream));
  for (int r = 0; r < 5; ++r) DFH_HIP(hipMemsetAsync(a, 0, (size_t)n2 * 16, ctx->stream));
  DFH_HIP(hipEventRecord(e1, ctx->stream)); DFH_HIP(hipEventSynchronize(e1));
  DFH_HIP(hipEventElapsedTime(&ms, e0, e1)); out[1] = 5.0 * n2 * 16 / (ms * 1e-3) / 1e12;
  DFH_HIP(hipEventRecord(e0, ctx->stream));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_dbg_copy16, dim3(4096), dim3(256), 0, ctx->stream, a, b, n2);
  DFH_HIP(hipEventRecord(e1, ctx->stream)); DFH_HIP(hipEventSynchronize(e1));
  DFH_HIP(hipEventElapsedTime(&ms, e0, e1)); out[2] = 5.0 * 2.0 * n2 * 16 / (ms * 1e-3) / 1e12;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return DFH_OK;
}
#endif  // DFH_DEBUG_HOOKS
