// tools/ubench_clock.hip -- what does the shader clock do under the EM kernel's kind of load (few waves, each issuing
// alone on its SIMD)?  s_nop 15 takes exactly 16 shader cycles; s_memrealtime ticks at a constant 100 MHz; s_memtime is
// what the probes of this repo count "cycles" in.  Prints the shader clock and the rate of s_memtime for grids of
// 1 .. 1024 one-wave workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* out, int reps) {
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  for (int i = 0; i < reps; i++) {
    asm volatile(".rept 64\n\ts_nop 15\n\t.endr" ::: "memory");
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = t1 - t0;
    out[2 * blockIdx.x + 1] = r1 - r0;
  }
}
int main() {
  unsigned long long* out;
  if (hipMalloc(&out, 2048 * 16) != hipSuccess) return 1;
  const int reps = 20000;  // 20000 x 64 x 16 = 20.5 M cycles, ~10 ms
  for (int pass = 0; pass < 3; pass++)
    for (int grid : {1, 100, 256, 1024}) {
      for (int threads : {64, 256}) {
        k<<<grid, threads>>>(out, reps);
        if (hipDeviceSynchronize() != hipSuccess) return 1;
        unsigned long long h[2];
        if (hipMemcpy(h, out, 16, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        const double cycles = (double)reps * 64 * 16, sec = h[1] / 100e6;
        printf("pass %d grid %4d x %3d threads: shader clock >= %.0f MHz (s_nop cycles / s_memrealtime), s_memtime rate %.1f MHz\n", pass, grid, threads,
               cycles / sec / 1e6, h[0] / sec / 1e6);
      }
    }
  return 0;
}
