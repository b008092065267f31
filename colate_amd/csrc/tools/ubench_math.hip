// tools/ubench_math.hip -- latency / throughput of the kernel's math helpers for ONE wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../em_math.hpp"
__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define N 64
__global__ void k(double* out, unsigned long long* cyc) {
  double x = -1e-3 * (threadIdx.x + 1), y = -2e-3 * (threadIdx.x + 1), z = 0.5 + 1e-3 * threadIdx.x;
  unsigned long long t0, t1;
  t0 = now();
#pragma unroll 1
  for (int i = 0; i < N; i++) x = -em::em_exp(x);
  asm volatile("" ::"v"(x));
  t1 = now();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  t0 = now();
#pragma unroll 1
  for (int i = 0; i < N; i++) {
    x = -em::em_exp(x);
    y = -em::em_exp(y);
  }
  asm volatile("" ::"v"(x), "v"(y));
  t1 = now();
  if (threadIdx.x == 0) cyc[1] = t1 - t0;
  t0 = now();
#pragma unroll 1
  for (int i = 0; i < N; i++) z = 1.0 + em::em_log(z) * 0.1;
  asm volatile("" ::"v"(z));
  t1 = now();
  if (threadIdx.x == 0) cyc[2] = t1 - t0;
  t0 = now();
#pragma unroll 1
  for (int i = 0; i < N; i++) z = 1.0 + em::em_rcp(z) * 0.1;
  asm volatile("" ::"v"(z));
  t1 = now();
  if (threadIdx.x == 0) cyc[3] = t1 - t0;
  t0 = now();
#pragma unroll 1
  for (int i = 0; i < N; i++) z = 1.0 + 0.1 / z;
  asm volatile("" ::"v"(z));
  t1 = now();
  if (threadIdx.x == 0) cyc[4] = t1 - t0;
  double om;
  t0 = now();
#pragma unroll 1
  for (int i = 0; i < N; i++) {
    x = -em::em_exp_om(x, &om);
    x = x - om * 1e-9;
  }
  asm volatile("" ::"v"(x));
  t1 = now();
  if (threadIdx.x == 0) cyc[5] = t1 - t0;
  t0 = now();
#pragma unroll 1
  for (int i = 0; i < N; i++) x = -__builtin_ldexp(x, -1) - 1e-3;
  asm volatile("" ::"v"(x));
  t1 = now();
  if (threadIdx.x == 0) cyc[6] = t1 - t0;
  t0 = now();
#pragma unroll 1
  for (int i = 0; i < N; i++) x = __builtin_rint(x * 1.5) * 0.3 - 1e-3;
  asm volatile("" ::"v"(x));
  t1 = now();
  if (threadIdx.x == 0) cyc[7] = t1 - t0;
  out[threadIdx.x] = x + y + z;
}
int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 64);
  for (int rep = 0; rep < 2; rep++) {
    k<<<1, 256>>>(out, cyc);
    hipDeviceSynchronize();
  }
  unsigned long long h[8];
  hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("per call: em_exp dep %.0f | 2 indep em_exp %.0f (per pair) | em_log dep %.0f | em_rcp dep %.0f | IEEE div dep %.0f | em_exp_om dep %.0f | ldexp+add %.0f | mul+rint+fma %.0f\n",
         (double)h[0] / N, (double)h[1] / N, (double)h[2] / N, (double)h[3] / N, (double)h[4] / N, (double)h[5] / N, (double)h[6] / N, (double)h[7] / N);
  return 0;
}
