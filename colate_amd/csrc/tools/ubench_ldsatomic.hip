// tools/ubench_ldsatomic.hip -- what does a per-epoch sum by LDS floating-point atomics cost against the in-register
// row-segmented reduce (4 DPP steps x 3 arrays = 36 instructions) of the EM kernel's bin phase?  Times, on one wave per
// SIMD, rounds of { 3 x ds_add_f64 ; s_waitcnt lgkmcnt(0) } for several lane -> address patterns, and checks that the
// sum a pattern leaves behind is the same in every round (the order in which the LDS serialises conflicting lanes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define ROUNDS 256
__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
__device__ __forceinline__ void lds_add(double* p, double v) {
  const unsigned addr = (unsigned)(size_t)p;  // LDS offset
  asm volatile("ds_add_f64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// kmap: [npat][64] epoch of each lane (-1: lane off); out: [npat][4 waves][ROUNDS][64] sums of array 0
__global__ void k(const int* kmap, int npat, unsigned long long* cyc, double* sums, unsigned long long* same) {
  __shared__ double acc[4][3][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int pat = 0; pat < npat; pat++) {
    const int kb = kmap[pat * 64 + lane];
    // values with full mantissas so that the order of the additions shows in the last bits
    const double v0 = 1.0 / (3.0 + lane), v1 = 1e-3 / (7.0 + lane), v2 = 1e3 / (1.0 + lane);
    double first = 0.0;
    unsigned long long all_same = 1;
    unsigned long long t = 0;
    for (int r = 0; r < ROUNDS; r++) {
      for (int a = 0; a < 3; a++) acc[wave][a][lane] = 0.0;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned long long t0 = now();
      if (kb >= 0) {
        lds_add(&acc[wave][0][kb], v0);
        lds_add(&acc[wave][1][kb], v1);
        lds_add(&acc[wave][2][kb], v2);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      t += now() - t0;
      const double s = acc[wave][0][lane] + acc[wave][2][lane];
      if (r == 0) first = s;
      if (__double_as_longlong(s) != __double_as_longlong(first)) all_same = 0;
    }
    if (lane == 0) cyc[pat * 4 + wave] = t / ROUNDS;
    sums[(pat * 4 + wave) * 64 + lane] = first;
    const unsigned long long bad = __ballot(!all_same);
    if (lane == 0) same[pat * 4 + wave] = bad;
    __syncthreads();
  }
}
int main() {
  std::vector<int> km;
  std::vector<const char*> names;
  auto add = [&](const char* n, auto f) {
    names.push_back(n);
    for (int l = 0; l < 64; l++) km.push_back(f(l));
  };
  add("no conflict (lane -> own address)", [](int l) { return l; });
  add("runs of 2", [](int l) { return l / 2; });
  add("runs of 4", [](int l) { return l / 4; });
  add("runs of 8", [](int l) { return l / 8; });
  add("runs of 16", [](int l) { return l / 16; });
  add("all 64 lanes -> one address", [](int) { return 0; });
  // --bins 3,7,0.2 on the 185-point age grid, first bin group: 19 bins in epoch 0, then runs of ~5
  add("E=23 group 0 (19, then runs of 5)", [](int l) { return l < 19 ? 0 : 1 + (l - 19) / 5; });
  add("E=23 group 1 (runs of 5, 47 live)", [](int l) { return l < 47 ? 10 + l / 5 : -1; });
  add("E=122 (runs of 1-2)", [](int l) { return (l * 2) / 3; });
  const int npat = names.size();
  int* d_km;
  unsigned long long *d_cyc, *d_same;
  double* d_sums;
  (void)hipMalloc(&d_km, km.size() * 4);
  (void)hipMalloc(&d_cyc, npat * 4 * 8);
  (void)hipMalloc(&d_same, npat * 4 * 8);
  (void)hipMalloc(&d_sums, npat * 4 * 64 * 8);
  (void)hipMemcpy(d_km, km.data(), km.size() * 4, hipMemcpyHostToDevice);
  for (int threads : {64, 256}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, d_km, npat, d_cyc, d_sums, d_same);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<unsigned long long> cyc(npat * 4), same(npat * 4);
    std::vector<double> sums(npat * 4 * 64);
    (void)hipMemcpy(cyc.data(), d_cyc, npat * 4 * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(same.data(), d_same, npat * 4 * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(sums.data(), d_sums, npat * 4 * 64 * 8, hipMemcpyDeviceToHost);
    printf("threads=%d: cycles per round of 3 x ds_add_f64 + wait (s_memtime)\n", threads);
    for (int p = 0; p < npat; p++) {
      // ascending-lane reference sum of array 0 + array 2 for the first address of the pattern
      bool waves_equal = true;
      for (int w = 1; w < threads / 64; w++) waves_equal &= !memcmp(&sums[(p * 4 + w) * 64], &sums[p * 4 * 64], 64 * 8);
      printf("  %-40s cycles %4llu   same sum in all %d rounds: %s   waves agree: %s\n", names[p], cyc[p * 4], ROUNDS,
             same[p * 4] == 0 ? "yes" : "NO", waves_equal ? "yes" : "NO");
    }
  }
  return 0;
}
