// Development benchmark for the cross kernel-matrix kernel (not part of the product build):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/km_bench.hip -o tools/km_bench
//   tools/km_bench [n1 n2 d]
// Variants of K[n1 x n2] = scale * exp(-|a_i - b_j|^2 / 2) (SE) on packed inputs, timed with HIP events.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef double double2_t __attribute__((ext_vector_type(2)));
typedef double double4_t __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

struct ExpC { double log2e, ln2_hi, ln2_lo; double c[12]; };
static ExpC make_expc(double scale) {
  ExpC e = {1.4426950408889634, 6.93147180369123816490e-01, 1.90821492927058770002e-10,
            {0x1.af631d0059becp-26, 0x1.28b4057f44145p-22, 0x1.71ddf5749d126p-19, 0x1.a01991ac8730ap-16,
             0x1.a01a01b14378fp-13, 0x1.6c16c187fbe02p-10, 0x1.111111110f225p-7, 0x1.555555554f0cfp-5,
             0x1.555555555555ap-3, 0x1.0000000000011p-1, 1.0, 1.0}};
  for (int i = 0; i < 12; ++i) e.c[i] *= scale;      // scale folded into the polynomial
  return e;
}

// scale * exp(x), x <= 0 (scale already inside ec.c)
__device__ __forceinline__ double sexp(double x, const ExpC& ec) {
  const double n = rint(x * ec.log2e);
  double r = fma(-n, ec.ln2_hi, x);
  r = fma(-n, ec.ln2_lo, r);
  double p = ec.c[0];
#pragma unroll
  for (int i = 1; i < 12; ++i) p = fma(p, r, ec.c[i]);
  return ldexp(p, (int)n);        // v_cvt_i32_f64 saturates; ldexp underflows to 0
}

struct Args {
  ExpC ec;
  const double* A; const double* nA;   // [n1][P], half squared norms [n1]
  const double* B; const double* nB;   // [n2][P], [n2]
  double* K; long ldk; int n1, n2, P;
};

// Register-operand kernel: no LDS.  A wave owns a (16*WI) x (16*WJ) tile; lane (l15, l4) of an
// MFMA holds, for row/col l15 of each 16-tile, the packed columns [l4*P/4, (l4+1)*P/4) -- a
// contiguous run (the k index assignment inside a dot product is free as long as A and B agree),
// so operands come straight from L2 with 16-byte loads.  MODE bits: 1 no epilogue (store raw dot),
// 2 no MFMA, 4 no store (one conditional store keeps the compiler honest), 8 no operand loads.
template <int WI, int WJ, int C, int MODE>
__global__ __launch_bounds__(256) void km_reg(Args p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  // 4 waves as 2 x 2
  const long m0 = ((long)blockIdx.y * 2 + (wave >> 1)) * (16 * WI);
  const long n0 = ((long)blockIdx.x * 2 + (wave & 1)) * (16 * WJ);
  double4_t acc[WI][WJ];
#pragma unroll
  for (int i = 0; i < WI; ++i)
#pragma unroll
    for (int j = 0; j < WJ; ++j) acc[i][j] = (double4_t){0, 0, 0, 0};
  const int per_lane = p.P >> 2;                    // packed columns per lane quarter
  for (int c0 = 0; c0 < per_lane; c0 += C) {
    double a[WI][C], b[WJ][C];
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const double* src = p.A + (m0 + i * 16 + l15) * p.P + l4 * per_lane + c0;
#pragma unroll
      for (int c = 0; c < C; c += 2) {
        double2_t v = (double2_t){1.0 + lane, 0.5};
        if (!(MODE & 8)) v = *reinterpret_cast<const double2_t*>(src + c);
        a[i][c] = v.x; a[i][c + 1] = v.y;
      }
    }
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const double* src = p.B + (n0 + j * 16 + l15) * p.P + l4 * per_lane + c0;
#pragma unroll
      for (int c = 0; c < C; c += 2) {
        double2_t v = (double2_t){0.25, 2.0 - lane};
        if (!(MODE & 8)) v = *reinterpret_cast<const double2_t*>(src + c);
        b[j][c] = v.x; b[j][c + 1] = v.y;
      }
    }
    if (!(MODE & 2)) {
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int i = 0; i < WI; ++i)
#pragma unroll
          for (int j = 0; j < WJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][c], b[j][c], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int j = 0; j < WJ; ++j) acc[i][j][0] += a[i][0] * b[j][0];
    }
  }
  double nbh[WJ];
#pragma unroll
  for (int j = 0; j < WJ; ++j) nbh[j] = p.nB[n0 + j * 16 + l15];
#pragma unroll
  for (int i = 0; i < WI; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long row = m0 + i * 16 + l4 + 4 * r;
      const double nah = p.nA[row];
#pragma unroll
      for (int j = 0; j < WJ; ++j) {
        double v = acc[i][j][r];
        if (!(MODE & 1)) {
          double t = v - (nbh[j] + nah);
          t = t > 0.0 ? 0.0 : t;
          v = sexp(t, p.ec);
        }
        const long col = n0 + j * 16 + l15;
        if (!(MODE & 4)) p.K[row * p.ldk + col] = v;
        else if (v == 123.456) p.K[row * p.ldk + col] = v;
      }
    }
  }
}


// Strip kernel: a wave keeps the A fragments of 64 rows in registers and walks along the columns
// in tiles of 16*WJ, prefetching the next tile's B fragments (register double buffer) while the
// current tile runs its MFMAs and epilogue.  fp64 MFMA and fp64 VALU share the SIMD's fp64 pipe
// on this part (measured: mfma 0.56 ms + valu 0.43 ms -> 0.89 ms together), so the job is to keep
// that pipe busy: no barriers, no LDS, loads a full tile ahead, stores fire-and-forget.
template <int WJ, int C, int NOSTORE = 0, int WI = 4>
__global__ __launch_bounds__(256, (WI <= 2 ? 2 : 1)) void km_strip(Args p, int tiles_per_seg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const long m0 = ((long)blockIdx.y * 4 + wave) * (16 * WI);
  if (m0 >= p.n1) return;
  const long ntile = (p.n2 + 16 * WJ - 1) / (16 * WJ);
  const long t0 = (long)blockIdx.x * tiles_per_seg;
  const long t1 = t0 + tiles_per_seg < ntile ? t0 + tiles_per_seg : ntile;
  if (t0 >= t1) return;
  double a[WI][C];
  double nah[WI][4];
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    long row = m0 + i * 16 + l15;
    row = row < p.n1 ? row : p.n1 - 1;
    const double* src = p.A + row * p.P + l4 * C;
#pragma unroll
    for (int c = 0; c < C; c += 2) {
      const double2_t v = *reinterpret_cast<const double2_t*>(src + c);
      a[i][c] = v.x; a[i][c + 1] = v.y;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      long rr = m0 + i * 16 + l4 + 4 * r;
      nah[i][r] = p.nA[rr < p.n1 ? rr : p.n1 - 1];
    }
  }
  double b[WJ][C], nbh[WJ];
  auto load_b = [&](long t, double (&bb)[WJ][C], double (&nn)[WJ]) {
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      long col = t * (16 * WJ) + j * 16 + l15;
      col = col < p.n2 ? col : p.n2 - 1;
      const double* src = p.B + col * p.P + l4 * C;
#pragma unroll
      for (int c = 0; c < C; c += 2) {
        const double2_t v = *reinterpret_cast<const double2_t*>(src + c);
        bb[j][c] = v.x; bb[j][c + 1] = v.y;
      }
      nn[j] = p.nB[col];
    }
  };
  load_b(t0, b, nbh);
  for (long t = t0; t < t1; ++t) {
    double bn[WJ][C], nbn[WJ];
    load_b(t + 1 < t1 ? t + 1 : t, bn, nbn);
    double4_t acc[WI][WJ];
#pragma unroll
    for (int i = 0; i < WI; ++i)
#pragma unroll
      for (int j = 0; j < WJ; ++j) acc[i][j] = (double4_t){0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int j = 0; j < WJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][c], b[j][c], acc[i][j], 0, 0, 0);
    const long n0 = t * (16 * WJ);
#pragma unroll
    for (int i = 0; i < WI; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = m0 + i * 16 + l4 + 4 * r;
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
          double tt = acc[i][j][r] - (nbh[j] + nah[i][r]);
          tt = tt > 0.0 ? 0.0 : tt;
          const double v = sexp(tt, p.ec);
          const long col = n0 + j * 16 + l15;
          if (NOSTORE) { if (v == 123.456) p.K[row * p.ldk + col] = v; }
          else if (row < p.n1 && col < p.n2) p.K[row * p.ldk + col] = v;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      nbh[j] = nbn[j];
#pragma unroll
      for (int c = 0; c < C; ++c) b[j][c] = bn[j][c];
    }
  }
}

template <int WJ, int C, int NOSTORE = 0, int WI = 4>
static float run_strip(const Args& a, int reps, int segs, const char* name) {
  const long ntile = (a.n2 + 16 * WJ - 1) / (16 * WJ);
  const int tps = (int)((ntile + segs - 1) / segs);
  dim3 grid((unsigned)segs, (unsigned)((a.n1 + 64 * WI - 1) / (64 * WI)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((km_strip<WJ, C, NOSTORE, WI>), grid, dim3(256), 0, 0, a, tps);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((km_strip<WJ, C, NOSTORE, WI>), grid, dim3(256), 0, 0, a, tps);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double bytes = 8.0 * ((double)a.n1 * a.n2 + ((double)a.n1 + a.n2) * a.P);
  printf("%-34s %8.3f ms  %6.2f TB/s (algorithmic)\n", name, ms, bytes / (ms * 1e-3) / 1e12);
  return ms;
}

template <int WI, int WJ, int C, int MODE>
static float run(const Args& a, int reps, const char* name) {
  dim3 grid((unsigned)(a.n2 / (32 * WJ)), (unsigned)(a.n1 / (32 * WI)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((km_reg<WI, WJ, C, MODE>), grid, dim3(256), 0, 0, a);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((km_reg<WI, WJ, C, MODE>), grid, dim3(256), 0, 0, a);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double bytes = 8.0 * ((double)a.n1 * a.n2 + ((double)a.n1 + a.n2) * a.P);
  printf("%-34s %8.3f ms  %6.2f TB/s (algorithmic)\n", name, ms, bytes / (ms * 1e-3) / 1e12);
  return ms;
}

int main(int argc, char** argv) {
  const int n1 = argc > 1 ? atoi(argv[1]) : 32768, n2 = argc > 2 ? atoi(argv[2]) : 16384, d = argc > 3 ? atoi(argv[3]) : 32;
  const int P = (d + 3) / 4 * 4;
  std::vector<double> hA((size_t)n1 * P, 0.0), hB((size_t)n2 * P, 0.0), hnA(n1), hnB(n2);
  srand(1);
  auto fill = [&](std::vector<double>& X, std::vector<double>& nh, int n) {
    for (int i = 0; i < n; ++i) {
      double s = 0;
      for (int k = 0; k < d; ++k) { double v = (rand() / (double)RAND_MAX) / (0.6 + 0.03 * k); X[(size_t)i * P + k] = v; s += v * v; }
      nh[i] = 0.5 * s;
    }
  };
  fill(hA, hnA, n1); fill(hB, hnB, n2);
  double *dA, *dB, *dnA, *dnB, *dK;
  CK(hipMalloc(&dA, hA.size() * 8)); CK(hipMalloc(&dB, hB.size() * 8));
  CK(hipMalloc(&dnA, n1 * 8)); CK(hipMalloc(&dnB, n2 * 8));
  CK(hipMalloc(&dK, (size_t)n1 * n2 * 8));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, hB.data(), hB.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dnA, hnA.data(), n1 * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dnB, hnB.data(), n2 * 8, hipMemcpyHostToDevice));
  Args a;
  const double scale = 0.37;
  a.ec = make_expc(scale);
  a.A = dA; a.nA = dnA; a.B = dB; a.nB = dnB; a.K = dK; a.ldk = n2; a.n1 = n1; a.n2 = n2; a.P = P;
  printf("cross kernel matrix %d x %d, d = %d (P = %d), SE\n", n1, n2, d, P);
  const int reps = 10;
  auto check = [&](const char* what) {
    std::vector<double> row((size_t)n2);
    double maxrel = 0;
    for (int i : {0, 17, n1 - 1}) {
      CK(hipMemcpy(row.data(), dK + (size_t)i * n2, (size_t)n2 * 8, hipMemcpyDeviceToHost));
      for (int j = 0; j < n2; j += 97) {
        long double s2 = 0;
        for (int k = 0; k < d; ++k) { long double t = (long double)hA[(size_t)i * P + k] - hB[(size_t)j * P + k]; s2 += t * t; }
        const double ref = (double)(scale * expl(-s2 / 2));
        maxrel = fmax(maxrel, fabs(row[j] - ref) / ref);
      }
    }
    printf("  %s: max rel. error vs long double: %.2e\n", what, maxrel);
  };
  if (P == 32) {
    CK(hipMemset(dK, 0, (size_t)n1 * n2 * 8));
    run_strip<2, 8>(a, reps, 4, "strip WJ=2 segs=4");
    check("strip");
    run_strip<2, 8, 1>(a, reps, 4, "strip WJ=2 segs=4 NO STORES");
    run_strip<2, 8, 0, 2>(a, reps, 2, "strip WI=2 WJ=2 segs=2");
    run_strip<2, 8, 0, 2>(a, reps, 4, "strip WI=2 WJ=2 segs=4");
    run_strip<4, 8, 0, 2>(a, reps, 2, "strip WI=2 WJ=4 segs=2");
    run_strip<4, 8, 0, 2>(a, reps, 4, "strip WI=2 WJ=4 segs=4");
    run_strip<2, 8, 1, 2>(a, reps, 2, "strip WI=2 WJ=2 segs=2 NO STORES");
    run_strip<2, 8>(a, reps, 8, "strip WJ=2 segs=8");
    run_strip<2, 8>(a, reps, 16, "strip WJ=2 segs=16");
    run_strip<2, 8>(a, reps, 2, "strip WJ=2 segs=2");
    run_strip<1, 8>(a, reps, 4, "strip WJ=1 segs=4");
    run_strip<1, 8>(a, reps, 8, "strip WJ=1 segs=8");
    run_strip<3, 8>(a, reps, 4, "strip WJ=3 segs=4");
  }
  if (P == 8) {
    CK(hipMemset(dK, 0, (size_t)n1 * n2 * 8));
    run_strip<2, 2>(a, reps, 1, "strip WJ=2 C=2 segs=1");
    check("strip");
    run_strip<2, 2>(a, reps, 2, "strip WJ=2 C=2 segs=2");
    run_strip<4, 2>(a, reps, 1, "strip WJ=4 C=2 segs=1");
    run_strip<4, 2>(a, reps, 2, "strip WJ=4 C=2 segs=2");
  }
  if (P % 32 == 0) {
    run<4, 2, 8, 0>(a, reps, "reg 64x32/wave C=8 full");
    // correctness of the full variant
    std::vector<double> row((size_t)n2);
    double maxrel = 0;
    for (int i : {0, 17, n1 - 1}) {
      CK(hipMemcpy(row.data(), dK + (size_t)i * n2, (size_t)n2 * 8, hipMemcpyDeviceToHost));
      for (int j = 0; j < n2; j += 97) {
        long double s = 0;
        for (int k = 0; k < d; ++k) { long double t = (long double)hA[(size_t)i * P + k] - hB[(size_t)j * P + k]; s += t * t; }
        const double ref = (double)(scale * expl(-s / 2));
        maxrel = fmax(maxrel, fabs(row[j] - ref) / ref);
      }
    }
    printf("max rel. error vs long double: %.2e\n", maxrel);
    run<4, 2, 8, 1>(a, reps, "no-epilogue");
    run<4, 2, 8, 2>(a, reps, "no-mfma");
    run<4, 2, 8, 4>(a, reps, "no-store");
    run<4, 2, 8, 4 + 8>(a, reps, "no-store no-load (mfma+valu)");
    run<4, 2, 8, 4 + 8 + 1>(a, reps, "mfma only");
    run<4, 2, 8, 4 + 8 + 2>(a, reps, "valu only");
    run<4, 2, 8, 4 + 1>(a, reps, "load+mfma only");
    run<4, 2, 8, 4 + 2 + 1>(a, reps, "load only");
    run<2, 2, 8, 4 + 8>(a, reps, "32x32: mfma+valu");
    run<2, 2, 8, 4 + 8 + 1>(a, reps, "32x32: mfma only");
    run<2, 2, 8, 4 + 8 + 2>(a, reps, "32x32: valu only");
    run<2, 2, 8, 0>(a, reps, "reg 32x32/wave C=8 full");
    run<4, 4, 8, 0>(a, reps, "reg 64x64/wave C=8 full");
    run<4, 2, 4, 0>(a, reps, "reg 64x32/wave C=4 full");
    run<2, 4, 8, 0>(a, reps, "reg 32x64/wave C=8 full");
    run<4, 1, 8, 0>(a, reps, "reg 64x16/wave C=8 full");
  } else {
    run<4, 2, 2, 0>(a, reps, "reg 64x32/wave C=2 full");
    run<2, 2, 2, 0>(a, reps, "reg 32x32/wave C=2 full");
    run<4, 4, 2, 0>(a, reps, "reg 64x64/wave C=2 full");
  }
  return 0;
}
