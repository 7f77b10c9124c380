// Does a PENDING high-priority workgroup that cannot be placed (needs a whole CU) slow down the
// dispatch of a running lower-priority grid?  busy: 2 workgroups per CU (74 KB LDS each), fixed work
// per workgroup; big: 8 workgroups of 140 KB LDS on a high-priority stream.
//   hipcc --offload-arch=gfx950 -O3 tools/prio_exp.hip -o gpurun_out/prio_exp && gpurun_out/prio_exp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256, 2) void busy(double* out, int iters) {
  extern __shared__ double sm[];
  double a = threadIdx.x * 1e-3, b = 1.000001, c = 0.5;
  for (int i = 0; i < iters; ++i) { a = fma(a, b, c); c = fma(c, b, a * 1e-9); }
  sm[threadIdx.x] = a + c;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = sm[(blockIdx.x * 7) & 255];
}
__global__ __launch_bounds__(256, 1) void big(double* out, int iters, int prio) {
  extern __shared__ double sm[];
  if (prio) __builtin_amdgcn_s_setprio(3);
  double a = threadIdx.x * 1e-3, b = 1.000001, c = 0.5;
  for (int i = 0; i < iters; ++i) { a = fma(a, b, c); c = fma(c, b, a * 1e-9); }
  sm[threadIdx.x] = a + c;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = sm[(blockIdx.x * 7) & 255];
}
__global__ void tiny(double* out) { if (threadIdx.x == 0) out[blockIdx.x] = 1.0; }

int main() {
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t M, P, Q;
  CK(hipStreamCreateWithFlags(&M, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&P, hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithFlags(&Q, hipStreamNonBlocking));
  double* out; CK(hipMalloc(&out, 1 << 20));
  CK(hipFuncSetAttribute((const void*)busy, hipFuncAttributeMaxDynamicSharedMemorySize, 74 * 1024));
  CK(hipFuncSetAttribute((const void*)big, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  hipEvent_t e0, e1, p0, p1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&p0)); CK(hipEventCreate(&p1));
  const int grid = 7750, iters = 20000;
  for (int mode = 0; mode < 6; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, M));
      hipLaunchKernelGGL(busy, dim3(grid), dim3(256), 74 * 1024, M, out, iters);
      CK(hipEventRecord(e1, M));
      hipStream_t S = (mode == 3) ? Q : P;
      CK(hipEventRecord(p0, S));
      if (mode == 1 || mode == 3) hipLaunchKernelGGL(big, dim3(8), dim3(256), 140 * 1024, S, out + 8192, 2000, 0);   // pending: cannot be placed
      if (mode == 2) for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(tiny, dim3(64), dim3(64), 0, S, out + 16384);  // many small launches
      if (mode == 4) for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(tiny, dim3(1024), dim3(256), 0, S, out + 16384);
      if (mode == 5) hipLaunchKernelGGL(busy, dim3(500), dim3(256), 74 * 1024, S, out + 32768, iters);            // a priority GEMM-like launch
      CK(hipEventRecord(p1, S));
      CK(hipDeviceSynchronize());
      float ms = 0, pms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventElapsedTime(&pms, p0, p1));
      const char* names[] = {"alone", "pending whole-CU kernel, high priority", "40 tiny launches, high priority",
                             "pending whole-CU kernel, normal priority", "40 launches of 1024 small WGs, high priority", "500-WG busy launch, high priority"};
      printf("mode %d (%s): busy grid %.3f ms ; other stream %.3f ms\n", mode, names[mode], ms, pms);
    }
  }
  return 0;
}
