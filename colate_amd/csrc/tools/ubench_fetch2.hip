// tools/ubench_fetch2.hip -- follow-up to ubench_fetch.hip: what do 4-byte instructions and their VOP3 re-encodings cost a
// lone wave, alone and mixed with 8-byte instructions at either alignment?  (cycles per instruction, 1024 in a row)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define REP16(x) x x x x x x x x x x x x x x x x
#define REP256(x) REP16(REP16(x))
#define KERNEL(NAME, BODY)                                                                      \
  __global__ void NAME(double* out, unsigned long long* cyc) {                                   \
    double a = threadIdx.x, b = a + 1, e = 1e-9;                                                 \
    int i = threadIdx.x, j = i + 1, k = 3;                                                       \
    unsigned long long t0 = now();                                                               \
    asm volatile(".p2align 6\n\t" BODY : "+v"(a), "+v"(b), "+v"(i), "+v"(j) : "v"(e), "v"(k));  \
    unsigned long long t1 = now();                                                               \
    if (threadIdx.x == 0) cyc[0] = t1 - t0;                                                      \
    out[threadIdx.x] = a + b + i + j;                                                            \
  }
// %0,%1 f64 pairs; %2,%3 ints; %4 f64; %5 int
KERNEL(k_add32_e32, REP256("v_add_u32_e32 %2, %5, %2\n\tv_add_u32_e32 %3, %5, %3\n\tv_add_u32_e32 %2, %5, %2\n\tv_add_u32_e32 %3, %5, %3\n\t"))
KERNEL(k_add32_e64, REP256("v_add_u32_e64 %2, %5, %2\n\tv_add_u32_e64 %3, %5, %3\n\tv_add_u32_e64 %2, %5, %2\n\tv_add_u32_e64 %3, %5, %3\n\t"))
KERNEL(k_mov_e32, REP256("v_mov_b32_e32 %2, %5\n\tv_mov_b32_e32 %3, %5\n\tv_mov_b32_e32 %2, %5\n\tv_mov_b32_e32 %3, %5\n\t"))
KERNEL(k_mov_e64, REP256("v_mov_b32_e64 %2, %5\n\tv_mov_b32_e64 %3, %5\n\tv_mov_b32_e64 %2, %5\n\tv_mov_b32_e64 %3, %5\n\t"))
KERNEL(k_salu4, REP256("s_mov_b32 s20, s21\n\ts_mov_b32 s22, s23\n\ts_mov_b32 s20, s21\n\ts_mov_b32 s22, s23\n\t"))
KERNEL(k_nop, REP256("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\t"))
// mixed: 4-byte VALU + 8-byte f64, alternating: half of the 8-byte instructions misaligned
KERNEL(k_mix_4_8, REP256("v_add_u32_e32 %2, %5, %2\n\tv_add_f64 %0, %0, %4\n\tv_add_u32_e32 %3, %5, %3\n\tv_add_f64 %1, %1, %4\n\t"))
// the same with the 4-byte instructions re-encoded: all aligned
KERNEL(k_mix_8_8, REP256("v_add_u32_e64 %2, %5, %2\n\tv_add_f64 %0, %0, %4\n\tv_add_u32_e64 %3, %5, %3\n\tv_add_f64 %1, %1, %4\n\t"))
// pairs of 4-byte instructions between 8-byte ones: all aligned without re-encoding
KERNEL(k_mix_44_8, REP256("v_add_u32_e32 %2, %5, %2\n\tv_add_u32_e32 %3, %5, %3\n\tv_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\t"))
// 4-byte SALU + 8-byte f64 alternating (half misaligned), and with an s_nop making pairs (all aligned, one more instruction)
KERNEL(k_mix_s4_8, REP256("s_mov_b32 s20, s21\n\tv_add_f64 %0, %0, %4\n\ts_mov_b32 s22, s23\n\tv_add_f64 %1, %1, %4\n\t"))
KERNEL(k_mix_s4n_8, REP256("s_mov_b32 s20, s21\n\ts_nop 0\n\tv_add_f64 %0, %0, %4\n\ts_mov_b32 s22, s23\n\ts_nop 0\n\tv_add_f64 %1, %1, %4\n\t"))
// DPP moves (8 bytes) aligned / misaligned
KERNEL(k_dpp_al, REP256("v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\t"))
KERNEL(k_dpp_mis, "s_nop 0\n\t" REP256("v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\t"))
template <typename K>
static double run(K k, double* out, unsigned long long* cyc, int n) {
  unsigned long long h = 0, best = ~0ull;
  for (int r = 0; r < 5; r++) {
    k<<<1, 64>>>(out, cyc);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    if (h < best) best = h;
  }
  return (double)best / n;
}
int main() {
  double* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 1024 * 8); (void)hipMalloc(&cyc, 64);
#define P(K, N, WHAT) printf("%-14s %.2f cycles per instruction   (%s)\n", #K, run(K, out, cyc, N), WHAT);
  P(k_add32_e32, 1024, "v_add_u32_e32, 4 bytes")
  P(k_add32_e64, 1024, "v_add_u32_e64, 8 bytes, aligned")
  P(k_mov_e32, 1024, "v_mov_b32_e32")
  P(k_mov_e64, 1024, "v_mov_b32_e64")
  P(k_salu4, 1024, "s_mov_b32, 4 bytes")
  P(k_nop, 1024, "s_nop 0")
  P(k_mix_4_8, 1024, "4-byte VALU, 8-byte f64 alternating: every other f64 misaligned")
  P(k_mix_8_8, 1024, "the same with the 4-byte ones re-encoded as VOP3: all aligned")
  P(k_mix_44_8, 1024, "two 4-byte VALU, two 8-byte f64: aligned without re-encoding")
  P(k_mix_s4_8, 1024, "4-byte SALU, 8-byte f64 alternating")
  P(k_mix_s4n_8, 1536, "SALU + s_nop + f64: aligned, one instruction more (per instruction incl. the nops)")
  P(k_dpp_al, 1024, "2 DPP movs + 2 f64 adds, aligned")
  P(k_dpp_mis, 1024, "the same 4 bytes behind")
  return 0;
}
