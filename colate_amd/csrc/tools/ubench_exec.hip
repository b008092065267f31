// tools/ubench_exec.hip -- does a VALU instruction of a wave with few active lanes issue faster?  (No: gfx950 runs all four
// 16-lane passes whatever EXEC holds -- profiles/r03/ubench_exec.txt -- so packing the 23 epochs into fewer lanes buys nothing.)
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
__global__ void k(double* out, unsigned long long* cyc, int nact) {
  double a = threadIdx.x * 1e-9 + 1.0, b = 0.999999, c = 1e-7;
  double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
  unsigned long long t0, t1, r0 = 0, r1 = 0, r2 = 0;
  if ((int)threadIdx.x < nact) {
    t0 = now();
#pragma unroll
    for (int i = 0; i < N; i++) x0 = __builtin_fma(x0, b, c);
    asm volatile("" ::"v"(x0));
    t1 = now();
    r0 = t1 - t0;
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
    r1 = t1 - t0;
    int v = threadIdx.x;
    t0 = now();
#pragma unroll
    for (int i = 0; i < N; i++) v = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true) + 1;
    asm volatile("" ::"v"(v));
    t1 = now();
    r2 = t1 - t0;
    x1 += v;
  }
  if (threadIdx.x == 0) cyc[0] = r0, cyc[1] = r1, cyc[2] = r2;
  out[threadIdx.x] = x0 + x1 + x2 + x3;
}
int main() {
  double* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 1024 * 8); (void)hipMalloc(&cyc, 64);
  for (int nact : {64, 64, 48, 32, 23, 16, 8, 1}) {
    hipMemset(cyc, 0, 64);
    k<<<1, 64>>>(out, cyc, nact);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("active lanes=%d: dep fma %.2f | 4-indep fma %.2f | dpp+add(i32) %.2f cycles per instr\n", nact, (double)h[0] / N, (double)h[1] / N, (double)h[2] / N);
  }
  return 0;
}
