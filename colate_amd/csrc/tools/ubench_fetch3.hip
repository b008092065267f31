// tools/ubench_fetch3.hip -- run-length form of ubench_fetch2: one 4-byte instruction, then n 8-byte ones, repeated (every other group of
// 8-byte instructions starts 4 bytes behind an 8-byte boundary); and the all-misaligned stream with a 4-byte PAIR every n instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define A8 "v_add_f64 %0, %0, %4\n\t"
#define B8 "v_add_f64 %1, %1, %4\n\t"
#define V4 "v_add_u32_e32 %2, %5, %2\n\t"
#define S4 "s_mov_b32 s20, s21\n\t"
template <int N, int KIND>
__global__ void k(double* out, unsigned long long* cyc) {
  double a = threadIdx.x, b = a + 1, e = 1e-9;
  int i = threadIdx.x, j = i + 1, kk = 3;
  unsigned long long t0 = now();
#define BODY(G) asm volatile(".p2align 6\n\t.rept 128\n\t" G ".endr" : "+v"(a), "+v"(b), "+v"(i), "+v"(j) : "v"(e), "v"(kk));
  if (KIND == 0) {  // V4 + N x 8
    if (N == 1) BODY(V4 A8) if (N == 2) BODY(V4 A8 B8) if (N == 3) BODY(V4 A8 B8 A8) if (N == 4) BODY(V4 A8 B8 A8 B8)
    if (N == 6) BODY(V4 A8 B8 A8 B8 A8 B8) if (N == 8) BODY(V4 A8 B8 A8 B8 A8 B8 A8 B8) if (N == 12) BODY(V4 A8 B8 A8 B8 A8 B8 A8 B8 A8 B8 A8 B8)
  } else if (KIND == 1) {  // S4 + N x 8
    if (N == 1) BODY(S4 A8) if (N == 2) BODY(S4 A8 B8) if (N == 3) BODY(S4 A8 B8 A8) if (N == 4) BODY(S4 A8 B8 A8 B8)
    if (N == 6) BODY(S4 A8 B8 A8 B8 A8 B8) if (N == 8) BODY(S4 A8 B8 A8 B8 A8 B8 A8 B8) if (N == 12) BODY(S4 A8 B8 A8 B8 A8 B8 A8 B8 A8 B8 A8 B8)
  } else {  // always misaligned: s_nop in front, then (V4 V4 + N x 8) repeated
#define BODY2(G) asm volatile(".p2align 6\n\ts_nop 0\n\t.rept 128\n\t" G ".endr" : "+v"(a), "+v"(b), "+v"(i), "+v"(j) : "v"(e), "v"(kk));
    if (N == 1) BODY2(V4 V4 A8) if (N == 2) BODY2(V4 V4 A8 B8) if (N == 3) BODY2(V4 V4 A8 B8 A8) if (N == 4) BODY2(V4 V4 A8 B8 A8 B8)
    if (N == 6) BODY2(V4 V4 A8 B8 A8 B8 A8 B8) if (N == 8) BODY2(V4 V4 A8 B8 A8 B8 A8 B8 A8 B8) if (N == 12) BODY2(V4 V4 A8 B8 A8 B8 A8 B8 A8 B8 A8 B8 A8 B8)
  }
  unsigned long long t1 = now();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  out[threadIdx.x] = a + b + i + j;
}
template <typename K>
static double run(K kern, double* out, unsigned long long* cyc) {
  unsigned long long h = 0, best = ~0ull;
  for (int r = 0; r < 5; r++) {
    kern<<<1, 64>>>(out, cyc);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    if (h < best) best = h;
  }
  return (double)best;
}
int main() {
  double* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 1024 * 8); (void)hipMalloc(&cyc, 64);
  printf("cycles per instruction; groups repeated 128 times\n");
#define ROW(N) printf("n = %2d:  [VALU4 + n x f64] %.2f   [SALU4 + n x f64] %.2f   [2 x VALU4 + n x f64, all f64 misaligned] %.2f\n", N, \
    run(k<N, 0>, out, cyc) / (128.0 * (N + 1)), run(k<N, 1>, out, cyc) / (128.0 * (N + 1)), run(k<N, 2>, out, cyc) / (128.0 * (N + 2)));
  ROW(1) ROW(2) ROW(3) ROW(4) ROW(6) ROW(8) ROW(12)
  return 0;
}
