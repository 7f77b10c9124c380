// Kernel evaluation on the device: exp / sqrt for the arguments a stationary kernel produces, one part's value from a
// squared distance (kern_eval), the polynomial / exponential-decay parts, the combination rule of a product kernel with
// additive factors, NumPy's pairwise row sum of squares.  Shared by kernmat.hip (the kernel-matrix kernels, the tuning
// objective of small problems) and chol.hip (the one-workgroup tuning objective with its Gram matrix built in the same
// launch).  Include inside the translation unit's anonymous namespace.
#pragma once

// exp_fast's constants travel as kernel arguments (scalar loads into SGPR pairs, unknown to the
// compiler): with literal coefficients the compiler emits v_fmac_f64 and re-materialises every
// 64-bit constant with two v_mov_b32 per Horner step -- 18 extra VALU instructions per element,
// over a third of the VALU-bound epilogue.
struct ExpConsts {
  double log2e, ln2_hi, ln2_lo;
  double c[12];
};
static const ExpConsts kExpConsts = {
    1.4426950408889634, 6.93147180369123816490e-01, 1.90821492927058770002e-10,
    {0x1.af631d0059becp-26, 0x1.28b4057f44145p-22, 0x1.71ddf5749d126p-19, 0x1.a01991ac8730ap-16,
     0x1.a01a01b14378fp-13, 0x1.6c16c187fbe02p-10, 0x1.111111110f225p-7, 0x1.555555554f0cfp-5,
     0x1.555555555555ap-3, 0x1.0000000000011p-1, 1.0, 1.0}};


__device__ __forceinline__ double ipow(double m, int k) {
  double r = 1.0;                      // 0**0 == 1 as in numpy (kernel.py:266)
  for (int i = 0; i < k; ++i) r *= m;
  return r;
}

// exp(x) for the arguments a stationary kernel produces (x <= 0; also fine for moderate x > 0):
// x = n ln2 + r, |r| <= ln2/2, degree-11 polynomial (Chebyshev-node interpolant of exp on that
// interval: 4e-18 approximation error, ~0.7 ulp after Horner in fp64), scaled by v_ldexp_f64.
// 17 fp64 VALU ops -- the epilogue of the kernel-matrix build is VALU-bound, so this is the lever.
__device__ __forceinline__ double exp_fast(double x, const ExpConsts& ec) {
  const double n = rint(x * ec.log2e);
  double r = fma(-n, ec.ln2_hi, x);
  r = fma(-n, ec.ln2_lo, r);
  double p = ec.c[0];
#pragma unroll
  for (int i = 1; i < 12; ++i) p = fma(p, r, ec.c[i]);
  // |x| beyond the double exponent range: n saturates, ldexp returns 0 / inf; NaN propagates
  return ldexp(p, (int)fmax(fmin(n, 4000.0), -4000.0));
}

// exp(x) for x <= 0 without the exponent clamp of exp_fast: v_cvt_i32_f64 saturates, and v_ldexp_f64
// with a hugely negative exponent returns 0 (what exp of such an x rounds to); NaN propagates
__device__ __forceinline__ double exp_fast_neg(double x, const ExpConsts& ec) {
  const double n = rint(x * ec.log2e);
  double r = fma(-n, ec.ln2_hi, x);
  r = fma(-n, ec.ln2_lo, r);
  double p = ec.c[0];
#pragma unroll
  for (int i = 1; i < 12; ++i) p = fma(p, r, ec.c[i]);
  return ldexp(p, (int)n);
}

__device__ __forceinline__ double sqrt_fast(double d) {
  // rsq + two Goldschmidt steps: <= 1 ulp on the range a clipped squared distance has; sqrt(0) = 0
  const double y = __builtin_amdgcn_rsq(d);
  double g = d * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  return d > 0.0 ? g : d;               // d == 0 -> 0 ; NaN -> NaN
}

__device__ __forceinline__ double kern_eval(const PartDev& pd, double dsq, const ExpConsts& ec) {
  if (pd.kind == DFH_KERNEL_SE) {
    return pd.scale_c * exp_fast(-dsq / 2, ec);                // kernel.py:176
  } else if (pd.kind == DFH_KERNEL_MATERN) {
    const double dist = sqrt_fast(dsq);                    // kernel.py:296 (<= 1 ulp)
    const double mult = pd.s8 * dist;                      // kernel.py:265
    double u;                                              // sum_i coeff_i mult^(p-i), kernel.py:266 (Horner)
    if (pd.p == 0) u = pd.coeff[0];
    else if (pd.p == 1) u = fma(pd.coeff[0], mult, pd.coeff[1]);
    else if (pd.p == 2) u = fma(fma(pd.coeff[0], mult, pd.coeff[1]), mult, pd.coeff[2]);
    else {
      u = 0.0;
      for (int i = 0; i <= pd.p; ++i) u += pd.coeff[i] * ipow(mult, pd.p - i);
    }
    u *= (pd.gfac * exp_fast_neg(-pd.s2 * dist, ec));      // kernel.py:268-269
    return pd.scale_c * u;                                 // kernel.py:298
  }
  return dsq;                                              // DFH_KERNEL_DIST
}

// x ** order as NumPy evaluates it for a scalar integer exponent: its fast paths for 0, 1 and 2
// (ones, copy, square), libm pow otherwise.
__device__ __forceinline__ double pow_order(double x, int order) {
  if (order == 0) return 1.0;
  if (order == 1) return x;
  if (order == 2) return x * x;
  return pow(x, (double)order);
}

// Polynomial kernel from the dot product of the scaled points (kernel.py:381-386)
__device__ __forceinline__ double poly_eval(const PartDev& pd, double dot) {
  return pd.scale_c * pow_order(dot + 1.0, pd.p);
}

// Exponential-decay kernel from the two (unscaled) points (kernel.py:418-432): the product runs
// over the dimensions in order, starting from the scale, and the offset is added last.
__device__ __forceinline__ double expdecay_eval(const PartDev& pd, const double* x, const double* y) {
  double r = pd.scale_c;
  for (int c = 0; c < pd.p; ++c) r *= 1.0 / pow(1.0 + (x[c] + y[c]), pd.coeff[c]);
  return r + pd.gfac;
}

// One part's value into the running result of a product kernel whose factors may be sums of parts
// (PartDev::fmode): a plain factor multiplies; inside an additive factor the parts are added up from
// zero (np.zeros + k_1 + k_2 ..., kernel.py:490-493) and the scaled sum multiplies at its last part.
__device__ __forceinline__ void combine_nested(const PartDev& pd, double kv, double& res, double& fsum) {
  if (pd.fmode == 0) { res = res * kv; return; }
  fsum = (pd.fmode & FM_BEGIN) ? 0.0 + kv : fsum + kv;
  if (pd.fmode & FM_END) res = res * (pd.fscale * fsum);
}


// (X**2).sum(axis=1) with numpy's pairwise-sum order for rows of <= 128 elements
__device__ double np_sumsq(const double* a, int n) {
  if (n < 8) {
    double res = 0.0;
    for (int i = 0; i < n; ++i) res += a[i] * a[i];
    return res;
  }
  double r[8];
  for (int j = 0; j < 8; ++j) r[j] = a[j] * a[j];
  int i = 8;
  for (; i < n - (n % 8); i += 8)
    for (int j = 0; j < 8; ++j) r[j] += a[i + j] * a[i + j];
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res += a[i] * a[i];
  return res;
}

// one candidate of a small-problem tuning call: where its kernel image sits in the blob, and its scalars
struct TinyCand {
  long image;              // byte offset of the kernel image (blob_layout) in the blob
  int P, n_parts, multi, product;
  double outer, noise, mean;
};

// Results of a candidate.  direct: `out` is pinned host memory and the host is polling out[3] -- the three values go out
// as system-scope (write-through) stores, are drained, and only then the status word follows.
__device__ __forceinline__ void tiny_publish(double* out, bool direct, double v0, double v1, double v2, double status) {
  if (direct) {
    __hip_atomic_store(out + 0, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(out + 1, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(out + 2, v2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(out + 3, status, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  } else {
    out[0] = v0; out[1] = v1; out[2] = v2; out[3] = status;
  }
}

// The lower triangle of a small Gram matrix K (the caller's store adds what belongs on the diagonal) from scaled inputs in LDS (Xp [n][P], Np [n][n_parts]), by the
// 256 threads of a workgroup; store(i, j, value) for j <= i < n.  The triangle is walked as a rectangle: row p and row
// n - 1 - p together hold n + 1 entries.  (k_lml_tiny64 and the fused tuning objective of chol.hip, round 6.)
// A single SE or Matern (nu <= 2.5) part -- the commonest candidates -- takes four entries per thread at a time: such a workgroup runs one wave
// per SIMD, nothing hides the latency of an entry's chain (index division -> LDS -> dot product -> twelve dependent FMAs
// of the exponential, ~850 cycles), so four independent chains in flight are worth a factor of two to three.  The
// arithmetic of an entry is the same in both paths, operation for operation.
__device__ __forceinline__ void tri_rect_index(int idx, int n, int& i, int& j, bool& valid) {
  const int p = idx / (n + 1), q = idx - p * (n + 1);
  i = q <= p ? p : n - 1 - p;
  j = q <= p ? q : q - p - 1;
  valid = !(q > p && n - 1 - p == p);                  // (odd n: the middle row is its own partner)
}

template <typename Store>
__device__ __forceinline__ void tiny_gram_lower(const TinyCand& cand, const PartDev* parts, int n_parts, const double* Xp, int P,
                                                const double* Np, int n, const ExpConsts& ec, Store store) {
  const int tid = threadIdx.x;
  const int total = ((n + 1) >> 1) * (n + 1);
  // KIND 0: SE; 1 + p: Matern with nu = p + 1/2, p <= 2 (Dragonfly's default kernel is Matern-2.5)
  const int fast_kind = (!cand.multi && n_parts == 1 && parts[0].poff == 0)
                            ? (parts[0].kind == DFH_KERNEL_SE ? 0 : (parts[0].kind == DFH_KERNEL_MATERN && parts[0].p <= 2 ? 1 + parts[0].p : -1))
                            : -1;
  if (fast_kind >= 0) {                                // (uniform)
    const PartDev pd = parts[0];
    const int kc = pd.kc;
    const double scale_c = pd.scale_c;
    for (int base = tid; base < total; base += 4 * 256) {
      int i[4], j[4];
      bool ok[4];
      double dot[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = base + 256 * u;
        tri_rect_index(idx < total ? idx : base, n, i[u], j[u], ok[u]);
        ok[u] = ok[u] && idx < total;
        dot[u] = 0.0;
      }
      for (int q = 0; q < kc; ++q) {
#pragma unroll
        for (int u = 0; u < 4; ++u) dot[u] = fma(Xp[i[u] * P + q], Xp[j[u] * P + q], dot[u]);
      }
      double res[4], dsq[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        dsq[u] = (Np[j[u]] + Np[i[u]]) - 2.0 * dot[u];                        // general_utils.py:66-68
        dsq[u] = dsq[u] < 0.0 ? 0.0 : dsq[u];
      }
      if (fast_kind == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) res[u] = scale_c * exp_fast(-dsq[u] / 2, ec);              // kernel.py:176 (kern_eval's SE branch)
      } else {
        // kern_eval's Matern branch, operation for operation, the branch on p taken once for the four entries
        double dist[4], mult[4], uu[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { dist[u] = sqrt_fast(dsq[u]); mult[u] = pd.s8 * dist[u]; }      // kernel.py:296, 265
        if (fast_kind == 1) {
#pragma unroll
          for (int u = 0; u < 4; ++u) uu[u] = pd.coeff[0];
        } else if (fast_kind == 2) {
#pragma unroll
          for (int u = 0; u < 4; ++u) uu[u] = fma(pd.coeff[0], mult[u], pd.coeff[1]);
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) uu[u] = fma(fma(pd.coeff[0], mult[u], pd.coeff[1]), mult[u], pd.coeff[2]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          uu[u] *= (pd.gfac * exp_fast_neg(-pd.s2 * dist[u], ec));             // kernel.py:268-269
          res[u] = scale_c * uu[u];                                           // kernel.py:298
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (ok[u]) store(i[u], j[u], res[u]);
    }
    return;
  }
  for (int idx = tid; idx < total; idx += 256) {
    int i, j;
    bool ok;
    tri_rect_index(idx, n, i, j, ok);
    if (!ok) continue;
    double res = cand.multi ? (cand.product ? cand.outer : 0.0) : 0.0;
    double fsum = 0.0;
    for (int part = 0; part < n_parts; ++part) {
      const PartDev& pd = parts[part];
      const double* xi = Xp + i * P + pd.poff;
      const double* xj = Xp + j * P + pd.poff;
      double dot = 0.0;
      for (int q = 0; q < pd.kc; ++q) dot = fma(xi[q], xj[q], dot);
      double dsq = (Np[j * n_parts + part] + Np[i * n_parts + part]) - 2.0 * dot;   // general_utils.py:66-68
      dsq = dsq < 0.0 ? 0.0 : dsq;
      const double kv = kern_eval(pd, dsq, ec);
      if (!cand.multi) res = kv;
      else if (!cand.product) res = res + kv;
      else combine_nested(pd, kv, res, fsum);
    }
    if (cand.multi && !cand.product) res = cand.outer * res;
    store(i, j, res);
  }
}
