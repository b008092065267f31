// tools/ubench_mfma.hip -- FP64 matrix-core latencies on a lone wave (the EM kernel's regime: one wave per SIMD, every
// result needed by the next instruction of the same wave).  North-star's MFMA clause ("MFMA only if the epoch x age
// contraction proves genuinely dense") is priced with these numbers in profiles/r03_mfma_experiment.txt:
//   v_mfma_f64_16x16x4_f64     D[16x16] += A[16x4] B[4x16]   (4 VGPR pairs of accumulator per lane)
//   v_mfma_f64_4x4x4_4b_f64    4 blocks of D[4x4] += A[4x4] B[4x4]  (1 accumulator pair per lane)
// measured as (a) a dependent chain through the accumulator (issue -> result usable by the next MFMA), (b) a dependent
// chain MFMA -> v_add_f64 on the result -> MFMA operand (issue -> result usable by the VALU and back), (c) independent
// back-to-back issue, next to the dependent v_fma_f64 and the DPP row-reduction step they would replace.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 256
typedef double double4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
__device__ __forceinline__ double dpp_shr1(double v) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x111, 0xf, 0xf, true);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x111, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__global__ void k(double* out, unsigned long long* cyc) {
  const double a = 1.0 + 1e-9 * threadIdx.x, b = (threadIdx.x & 3) == 0 ? 1.0 : 0.0;  // a 0/1 selection matrix on the B side
  double4_t c16 = {0.0, 0.0, 0.0, 0.0}, d16 = {1.0, 1.0, 1.0, 1.0}, e16 = {2.0, 2.0, 2.0, 2.0}, f16 = {3.0, 3.0, 3.0, 3.0};
  double c4 = 0.0, d4 = 1.0, e4 = 2.0, f4 = 3.0, x = a;
  unsigned long long t0, t1;
  // (a) 16x16x4, dependent through the accumulator
  t0 = now();
#pragma unroll
  for (int i = 0; i < N; i++) c16 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c16, 0, 0, 0);
  asm volatile("" ::"v"(c16));
  t1 = now();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  // (b) 16x16x4 -> v_add_f64 on the result -> back into the A operand
  t0 = now();
#pragma unroll
  for (int i = 0; i < N; i++) {
    c16 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, b, d16, 0, 0, 0);
    x = c16[0] + 1e-9;
  }
  asm volatile("" ::"v"(x));
  t1 = now();
  if (threadIdx.x == 0) cyc[1] = t1 - t0;
  // (c) 16x16x4, four independent accumulators back to back
  t0 = now();
#pragma unroll
  for (int i = 0; i < N / 4; i++) {
    c16 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c16, 0, 0, 0);
    d16 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d16, 0, 0, 0);
    e16 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, e16, 0, 0, 0);
    f16 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, f16, 0, 0, 0);
  }
  asm volatile("" ::"v"(c16), "v"(d16), "v"(e16), "v"(f16));
  t1 = now();
  if (threadIdx.x == 0) cyc[2] = t1 - t0;
  // (a') 4x4x4 (4 blocks), dependent through the accumulator
  t0 = now();
#pragma unroll
  for (int i = 0; i < N; i++) c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
  asm volatile("" ::"v"(c4));
  t1 = now();
  if (threadIdx.x == 0) cyc[3] = t1 - t0;
  // (b') 4x4x4 -> v_add_f64 -> operand
  t0 = now();
#pragma unroll
  for (int i = 0; i < N; i++) {
    c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, b, d4, 0, 0, 0);
    x = c4 + 1e-9;
  }
  asm volatile("" ::"v"(x));
  t1 = now();
  if (threadIdx.x == 0) cyc[4] = t1 - t0;
  // (c') 4x4x4 independent
  t0 = now();
#pragma unroll
  for (int i = 0; i < N / 4; i++) {
    c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
    d4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d4, 0, 0, 0);
    e4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, e4, 0, 0, 0);
    f4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, f4, 0, 0, 0);
  }
  asm volatile("" ::"v"(c4), "v"(d4), "v"(e4), "v"(f4));
  t1 = now();
  if (threadIdx.x == 0) cyc[5] = t1 - t0;
  // reference points: dependent v_fma_f64; one step of the row-segmented DPP reduce (2 DPP moves + 1 fma, dependent)
  double y = a;
  t0 = now();
#pragma unroll
  for (int i = 0; i < N; i++) y = __builtin_fma(y, 0.999999, 1e-7);
  asm volatile("" ::"v"(y));
  t1 = now();
  if (threadIdx.x == 0) cyc[6] = t1 - t0;
  double z = a;
  t0 = now();
#pragma unroll
  for (int i = 0; i < N; i++) z = __builtin_fma(dpp_shr1(z), b, z);
  asm volatile("" ::"v"(z));
  t1 = now();
  if (threadIdx.x == 0) cyc[7] = t1 - t0;
  out[threadIdx.x] = c16[0] + c16[1] + d16[2] + e16[3] + f16[0] + c4 + d4 + e4 + f4 + x + y + z;
}
int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 64);
  for (int threads : {64, 64, 256}) {
    hipMemset(cyc, 0, 64);
    k<<<1, threads>>>(out, cyc);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("threads=%d (cycles per instruction, s_memtime): mfma_f64_16x16x4 dep-acc %.1f | ->v_add->operand %.1f | 4 independent %.1f || "
           "mfma_f64_4x4x4_4b dep-acc %.1f | ->v_add->operand %.1f | 4 independent %.1f || dep v_fma_f64 %.1f | dpp(2 moves)+fma step %.1f\n",
           threads, (double)h[0] / N, (double)h[1] / N, (double)h[2] / N, (double)h[3] / N, (double)h[4] / N, (double)h[5] / N,
           (double)h[6] / N, (double)h[7] / N);
  }
  return 0;
}
