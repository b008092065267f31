// tools/ubench.hip -- micro-benchmarks of the instruction costs that bound the EM kernel when ONE
// wave runs per SIMD (the B <= 256 regime): f64 FMA issue/latency, DPP move, LDS round trip, barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 512
__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
__global__ void k(double* out, unsigned long long* cyc, int nthreads_active) {
  __shared__ double sh[1024];
  double a = threadIdx.x * 1e-9 + 1.0, b = 0.999999, c = 1e-7;
  double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
  unsigned long long t0, t1;
  // dependent fma chain
  t0 = now();
#pragma unroll
  for (int i = 0; i < N; i++) x0 = __builtin_fma(x0, b, c);
  asm volatile("" ::"v"(x0));
  t1 = now();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  // 4 independent chains
  t0 = now();
#pragma unroll
  for (int i = 0; i < N / 4; i++) {
    x0 = __builtin_fma(x0, b, c);
    x1 = __builtin_fma(x1, b, c);
    x2 = __builtin_fma(x2, b, c);
    x3 = __builtin_fma(x3, b, c);
  }
  asm volatile("" ::"v"(x0), "v"(x1), "v"(x2), "v"(x3));
  t1 = now();
  if (threadIdx.x == 0) cyc[1] = t1 - t0;
  // dependent add chain
  t0 = now();
#pragma unroll
  for (int i = 0; i < N; i++) x1 = x1 + c;
  asm volatile("" ::"v"(x1));
  t1 = now();
  if (threadIdx.x == 0) cyc[2] = t1 - t0;
  // dpp mov chain (32-bit)
  int v = threadIdx.x;
  t0 = now();
#pragma unroll
  for (int i = 0; i < N; i++) v = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true) + 1;
  asm volatile("" ::"v"(v));
  t1 = now();
  if (threadIdx.x == 0) cyc[3] = t1 - t0;
  // LDS write -> read round trip, dependent
  sh[threadIdx.x] = x2;
  int idx = (threadIdx.x * 7) & 255;
  t0 = now();
#pragma unroll
  for (int i = 0; i < 64; i++) {
    double r = sh[idx];
    idx = ((int)r + idx * 5 + 1) & 255;
  }
  t1 = now();
  if (threadIdx.x == 0) cyc[4] = t1 - t0;
  // barrier
  t0 = now();
#pragma unroll
  for (int i = 0; i < 64; i++) __syncthreads();
  t1 = now();
  if (threadIdx.x == 0) cyc[5] = t1 - t0;
  // division chain
  t0 = now();
#pragma unroll
  for (int i = 0; i < 64; i++) x3 = 1.0 / (x3 + 1.5);
  asm volatile("" ::"v"(x3));
  t1 = now();
  if (threadIdx.x == 0) cyc[6] = t1 - t0;
  // v_cndmask pair chain on doubles
  t0 = now();
#pragma unroll
  for (int i = 0; i < N; i++) x2 = (x2 > 0.5) ? x2 + c : c;
  asm volatile("" ::"v"(x2));
  t1 = now();
  if (threadIdx.x == 0) cyc[7] = t1 - t0;
  out[threadIdx.x] = x0 + x1 + x2 + x3 + v + idx;
}
int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 64);
  for (int threads : {256, 64, 128, 192, 256, 320, 384, 448, 512, 768, 1024}) {
    hipMemset(cyc, 0, 64);
    k<<<1, threads>>>(out, cyc, threads);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("threads=%d: dep fma %.1f | 4-indep fma %.1f per instr | dep add %.1f | dpp+add(i32) %.1f | lds dep read %.1f | barrier %.1f | div chain %.1f | cmp+cndmask+add %.1f\n",
           threads, (double)h[0] / N, (double)h[1] / N, (double)h[2] / N, (double)h[3] / N, (double)h[4] / 64, (double)h[5] / 64, (double)h[6] / 64, (double)h[7] / N);
  }
  return 0;
}
