// tools/ubench_branch.hip -- what does a TAKEN branch cost a lone wave (instruction fetch re-steer)?  The EM loop has
// about 25 of them per iteration.  Times 256 repetitions of: nothing | a not-taken conditional branch | a taken
// conditional branch over 1, 4, 16 and 64 dwords | an unconditional branch over 4 dwords.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define REP 256
__global__ void k(unsigned long long* cyc, double* out) {
  unsigned long long t0;
  double x = threadIdx.x * 1e-3 + 1.0;
  // baseline: one dependent fma per repetition
  t0 = now();
  asm volatile(".rept 256\n\tv_fma_f64 %0, %0, %0, %0\n\t.endr" : "+v"(x));
  cyc[0] = now() - t0;
  // not-taken conditional branch (scc = 0)
  t0 = now();
  asm volatile("s_cmp_eq_u32 0, 1\n\t.rept 256\n\tv_fma_f64 %0, %0, %0, %0\n\ts_cbranch_scc1 1\n\ts_nop 0\n\t.endr" : "+v"(x)::"scc");
  cyc[1] = now() - t0;
  // taken conditional branch over 1 dword
  t0 = now();
  asm volatile("s_cmp_eq_u32 0, 0\n\t.rept 256\n\tv_fma_f64 %0, %0, %0, %0\n\ts_cbranch_scc1 1\n\ts_nop 0\n\t.endr" : "+v"(x)::"scc");
  cyc[2] = now() - t0;
  // taken over 4 dwords
  t0 = now();
  asm volatile("s_cmp_eq_u32 0, 0\n\t.rept 256\n\tv_fma_f64 %0, %0, %0, %0\n\ts_cbranch_scc1 4\n\t.rept 4\n\ts_nop 0\n\t.endr\n\t.endr" : "+v"(x)::"scc");
  cyc[3] = now() - t0;
  // taken over 16 dwords
  t0 = now();
  asm volatile("s_cmp_eq_u32 0, 0\n\t.rept 256\n\tv_fma_f64 %0, %0, %0, %0\n\ts_cbranch_scc1 16\n\t.rept 16\n\ts_nop 0\n\t.endr\n\t.endr" : "+v"(x)::"scc");
  cyc[4] = now() - t0;
  // taken over 64 dwords
  t0 = now();
  asm volatile("s_cmp_eq_u32 0, 0\n\t.rept 256\n\tv_fma_f64 %0, %0, %0, %0\n\ts_cbranch_scc1 64\n\t.rept 64\n\ts_nop 0\n\t.endr\n\t.endr" : "+v"(x)::"scc");
  cyc[5] = now() - t0;
  // unconditional over 4 dwords
  t0 = now();
  asm volatile(".rept 256\n\tv_fma_f64 %0, %0, %0, %0\n\ts_branch 4\n\t.rept 4\n\ts_nop 0\n\t.endr\n\t.endr" : "+v"(x));
  cyc[6] = now() - t0;
  // saveexec + execz branch not taken (the usual shape of a divergent if)
  t0 = now();
  asm volatile(".rept 256\n\tv_fma_f64 %0, %0, %0, %0\n\ts_and_saveexec_b64 s[20:21], exec\n\ts_cbranch_execz 1\n\ts_nop 0\n\ts_or_b64 exec, exec, s[20:21]\n\t.endr" : "+v"(x)::"s20", "s21");
  cyc[7] = now() - t0;
  out[threadIdx.x] = x;
}
int main() {
  unsigned long long* d;
  double* o;
  (void)hipMalloc(&d, 64);
  (void)hipMalloc(&o, 64 * 8 * 16);
  const char* names[] = {"fma only", "+ cond. branch not taken", "+ cond. branch taken over 1 dword", "+ taken over 4 dwords",
                         "+ taken over 16 dwords", "+ taken over 64 dwords", "+ s_branch over 4 dwords", "+ saveexec/execz not taken/restore"};
  for (int threads : {64, 256}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, d, o);
    unsigned long long c[8];
    (void)hipMemcpy(c, d, 64, hipMemcpyDeviceToHost);
    printf("threads=%d (cycles per repetition, s_memtime)\n", threads);
    for (int i = 0; i < 8; i++) printf("  %-40s %6.1f\n", names[i], c[i] / 256.0);
  }
  return 0;
}
