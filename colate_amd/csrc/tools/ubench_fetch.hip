// tools/ubench_fetch.hip -- what does code placement cost a lone wave?  (The EM loops move by up to 5 % when shifted by 4 bytes.)
// (1) straight-line 8-byte VALU instructions starting 0 / 4 bytes behind a 32-byte boundary (at +4 every fourth instruction straddles one);
// (2) a taken branch whose target sits 0 .. 60 bytes behind a 64-byte boundary, followed by 8-byte instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define ADD8 "v_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\tv_add_f64 %2, %2, %4\n\tv_add_f64 %3, %3, %4\n\t"
#define REP16(x) x x x x x x x x x x x x x x x x
template <int PAD>
__global__ void straight(double* out, unsigned long long* cyc) {
  double a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e = 1e-9;
  unsigned long long t0 = now();
  if (PAD == 0) asm volatile(".p2align 6\n\t" REP16(REP16(ADD8)) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));
  if (PAD == 1) asm volatile(".p2align 6\n\ts_nop 0\n\t" REP16(REP16(ADD8)) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));
  unsigned long long t1 = now();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  out[threadIdx.x] = a + b + c + d;
}
// 256 times: jump over a 64-byte-aligned gap to a target OFF dwords behind the boundary, then eight 8-byte adds
#define HOP(OFF) "s_branch 1f\n\t.p2align 6\n\t.rept " #OFF "\n\ts_nop 0\n\t.endr\n\t1:\n\t" ADD8 ADD8
template <int OFF>
__global__ void hops(double* out, unsigned long long* cyc) {
  double a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e = 1e-9;
  unsigned long long t0 = now();
#define H(O) if (OFF == O) asm volatile(REP16(REP16(HOP(O))) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));
  H(0) H(1) H(2) H(3) H(5) H(7) H(9) H(11) H(13) H(15)
  unsigned long long t1 = now();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  out[threadIdx.x] = a + b + c + d;
}
template <typename K>
static double run(K k, double* out, unsigned long long* cyc) {
  unsigned long long h = 0, best = ~0ull;
  for (int r = 0; r < 5; r++) {
    k<<<1, 64>>>(out, cyc);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    if (h < best) best = h;
  }
  return (double)best;
}
int main() {
  double* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 1024 * 8); (void)hipMalloc(&cyc, 64);
  printf("1024 x v_add_f64 (8 bytes each), first one at a 64-byte boundary: %.2f cycles per instruction; 4 bytes behind it: %.2f\n",
         run(straight<0>, out, cyc) / 1024, run(straight<1>, out, cyc) / 1024);
  printf("256 x (s_branch to a target n dwords behind a 64-byte boundary + eight 8-byte adds): cycles per hop, minus 8 adds at 4.2\n");
  double t;
#define P(O) t = run(hops<O>, out, cyc) / 256; printf("  target at +%2d dwords: %.1f cycles per hop, %.1f beyond the adds\n", O, t, t - 8 * 4.2);
  P(0) P(1) P(2) P(3) P(5) P(7) P(9) P(11) P(13) P(15)
  return 0;
}
