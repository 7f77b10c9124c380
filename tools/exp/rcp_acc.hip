// accuracy of v_rcp_f64 / v_rsq_f64 and of ONE Newton step on top (round 6: is the second step of fast_rcp needed?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double* d, double* r0, double* r1, double* r2, double* r3, long n) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  double x = d[i];
  double r = __builtin_amdgcn_rcp(x);
  r0[i] = r;
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  r1[i] = r;
  e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  r2[i] = r;
  // one cubic step instead: r0 (1 + e + e^2), three dependent operations after the reciprocal instead of four
  const double q = __builtin_amdgcn_rcp(x);
  const double e3 = fma(-x, q, 1.0);
  const double t = fma(e3, e3, e3);
  r3[i] = fma(q, t, q);
}
int main() {
  const long n = 1 << 24;
  std::vector<double> h(n);
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> u(-300.0, 300.0);
  for (long i = 0; i < n; ++i) h[i] = (i & 1 ? 1.0 : 1.0) * std::exp2(u(g) / 10.0) * (1.0 + (double)(g() >> 11) * 0x1p-53);
  double *d, *r0, *r1, *r2, *r3;
  hipMalloc(&d, n * 8); hipMalloc(&r0, n * 8); hipMalloc(&r1, n * 8); hipMalloc(&r2, n * 8); hipMalloc(&r3, n * 8);
  hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
  k<<<(n + 255) / 256, 256>>>(d, r0, r1, r2, r3, n);
  std::vector<double> a(n), b(n), c(n), c3(n);
  hipMemcpy(a.data(), r0, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), r1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), r2, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(c3.data(), r3, n * 8, hipMemcpyDeviceToHost);
  double m0 = 0, m1 = 0, m2 = 0, m3 = 0;
  for (long i = 0; i < n; ++i) {
    long double t = 1.0L / (long double)h[i];
    m0 = fmax(m0, (double)fabsl(((long double)a[i] - t) / t));
    m1 = fmax(m1, (double)fabsl(((long double)b[i] - t) / t));
    m2 = fmax(m2, (double)fabsl(((long double)c[i] - t) / t));
    m3 = fmax(m3, (double)fabsl(((long double)c3[i] - t) / t));
  }
  printf("one cubic step: %.3e (%.2f ulp)\n", m3, m3 / 0x1p-53);
  printf("max rel err: rcp %.3e (2^%.1f)  +1 Newton %.3e (%.2f ulp)  +2 Newton %.3e (%.2f ulp)\n", m0, log2(m0), m1, m1 / 0x1p-53, m2, m2 / 0x1p-53);
  return 0;
}
