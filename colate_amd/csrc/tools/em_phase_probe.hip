// colate_amd/csrc/tools/em_phase_probe.hip -- DIAGNOSTIC build of the EM kernel with cycle stamps
// (s_memtime) around the phases of an iteration.  Not part of the product library: it compiles
// em_kernels.hip with -DCOLATE_EM_STAMPS into a stand-alone program and prints where the cycles of
// an iteration go.  Only the SHARES are meaningful (the stamps serialise the phases).
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -I../../include -I.. tools/em_phase_probe.hip -o em_phase_probe
#ifndef COLATE_NO_STAMPS
#define COLATE_EM_STAMPS 1
#endif
#include "../em_kernels.hip"
#include "../em_kernels_ilp.hip"  // (same template again: the probe builds one translation unit)

#include <cmath>
#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 100, E = argc > 2 ? atoi(argv[2]) : 23, A = 185;
  std::vector<double> grid(A), ep(E), init(E, 1.0 / 20000), sh((size_t)B * A, 0.0), ns((size_t)B * A, 0.0);
  grid[0] = 0;
  for (int b = 1; b < A; b++) grid[b] = std::exp((b - 1) / 10.0) / 10.0;
  ep[0] = 0;  // log-spaced epochs like --bins 3,7,step
  ep[1] = 0;
  for (int e = 2; e < E - 1; e++) ep[e] = std::pow(10.0, 3.0 + 4.0 * (e - 1) / (E - 3.0)) / 28.0;
  ep[E - 1] = 1e8 / 28.0;
  unsigned s = 12345;
  for (int r = 0; r < B; r++)
    for (int b = 40; b <= 150; b++) {
      s = s * 1664525u + 1013904223u;
      double tot = (50 + 450.0 * (s >> 8) / 16777216.0) * 11, pr = 1 - std::exp(-grid[b] / 12000);
      sh[(size_t)r * A + b] = 0.8 * pr * tot;
      ns[(size_t)r * A + b] = tot - sh[(size_t)r * A + b];
    }
  double *d_grid, *d_sh, *d_ns, *d_ep, *d_init, *d_rates, *d_ll, *d_dbg;
  int *d_it, *d_fl;
  hipMalloc(&d_grid, A * 8); hipMalloc(&d_sh, sh.size() * 8); hipMalloc(&d_ns, ns.size() * 8);
  hipMalloc(&d_ep, E * 8); hipMalloc(&d_init, E * 8); hipMalloc(&d_rates, (size_t)B * E * 8);
  hipMalloc(&d_ll, B * 8); hipMalloc(&d_it, B * 4); hipMalloc(&d_fl, B * 4); hipMalloc(&d_dbg, (size_t)B * 4 * 16 * 8);
  hipMemcpy(d_grid, grid.data(), A * 8, hipMemcpyHostToDevice);
  hipMemcpy(d_sh, sh.data(), sh.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(d_ns, ns.data(), ns.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(d_ep, ep.data(), E * 8, hipMemcpyHostToDevice);
  hipMemcpy(d_init, init.data(), E * 8, hipMemcpyHostToDevice);
  hipMemset(d_dbg, 0, (size_t)B * 4 * 16 * 8);
  ColateEmArgs a{};
  a.B = B, a.E = E, a.A = A, a.mode = 0;
  a.age_grid = d_grid, a.cnt_sh = d_sh, a.cnt_ns = d_ns, a.epochs = d_ep, a.rates_in = d_init;
  a.max_iter = 100000, a.min_iter = 1000, a.rel_tol = 1e-7, a.rate_floor = 5e-9;
#ifdef COLATE_ABL
  a.max_iter = 1002, a.min_iter = 1000000;  // ablation builds: fixed number of iterations (results are meaningless)
#endif
  a.out_rates = d_rates, a.out_iters = d_it, a.out_ll = d_ll, a.out_flags = d_fl, a.out_num = d_dbg;
  double* d_pro;
  hipMalloc(&d_pro, (size_t)B * 10 * 8);
  hipMemset(d_pro, 0, (size_t)B * 10 * 8);
  a.out_den = d_pro;
  if (argc > 3) a.max_iter = atoi(argv[3]), a.min_iter = 1000000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  colate_em_launch(a, nullptr);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipError_t err = colate_em_launch(a, nullptr);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> dbg((size_t)B * 4 * 16);
  std::vector<int> it(B);
  hipMemcpy(dbg.data(), d_dbg, dbg.size() * 8, hipMemcpyDeviceToHost);
  hipMemcpy(it.data(), d_it, B * 4, hipMemcpyDeviceToHost);
  printf("launch: %s, %.3f ms, iters[0]=%d\n", hipGetErrorString(err), ms, it[0]);
  {
    std::vector<unsigned long long> pro((size_t)B * 10);
    hipMemcpy(pro.data(), d_pro, pro.size() * 8, hipMemcpyDeviceToHost);
    const char* pn[8] = {"LDS clear + epoch starts", "bins: counts, ages, epoch of each bin", "epoch statics, starting rates", "bin statics", "total counts", "per-epoch bin ranges + later counts (+ barrier)", "verdict constants, masks", "all iterations"};
    printf("prologue of replicate 0, wave 0 (cycles at 2.4 GHz; us):\n");
    for (int i = 0; i < 8; i++) printf("   %-48s %8llu  %6.2f us\n", pn[i], pro[i + 1] - pro[i], (pro[i + 1] - pro[i]) / 2400.0);
    unsigned long long first = ~0ull, last = 0;
    for (int r = 0; r < B; r++) { if (pro[(size_t)r * 10] < first) first = pro[(size_t)r * 10]; if (pro[(size_t)r * 10 + 8] > last) last = pro[(size_t)r * 10 + 8]; }
    printf("   first workgroup start to last workgroup at its epilogue: %.2f us; workgroup starts spread over %.2f us\n", (last - first) / 2400.0,
           ([&] { unsigned long long mx = 0; for (int r = 0; r < B; r++) if (pro[(size_t)r * 10] > mx) mx = pro[(size_t)r * 10]; return (mx - first) / 2400.0; })());
  }
#ifndef COLATE_EM_STAMPS
  return 0;
#endif
  const char* names[16] = {"P2 start (gathers)", "P2 bin math", "P2 seg-reduce+store", "barrier 2", "P3 N,D + store", "P4 divide+masks", "P4 wait at barrier 3", "loop top/stop",
                           "P1 cs scan", "P1 exp/div/write", "P3 tail loads", "P3 suffix scan (A)", "P3 affine scan (B)", "P4 N,D LDS loads+adds", "-", "-"};
  for (int w = 0; w < 4; w++) {
    unsigned long long tot = 0;
    for (int i = 0; i < 16; i++) tot += dbg[(size_t)w * 16 + i];
    if (!tot) continue;
    printf("replicate 0 wave %d (role %d, bin group %d): total %llu cycles, %.0f per iteration\n", w, w & 1, w >> 1, tot, (double)tot / (it[0] + 1));
    for (int i = 0; i < 16; i++)
      if (dbg[(size_t)w * 16 + i])
        printf("   %-20s %8.0f cyc/iter  %5.1f%%\n", names[i], (double)dbg[(size_t)w * 16 + i] / (it[0] + 1), 100.0 * dbg[(size_t)w * 16 + i] / tot);
  }
  return 0;
}
