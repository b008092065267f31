// colate_amd/csrc/tools/residency_probe.hip -- DIAGNOSTIC: how many workgroups of the EM kernel are resident
// on one CU at a time when B exceeds the CU count.  Compiles em_kernels.hip with -DCOLATE_EM_TRACE (each
// workgroup records HW_ID, XCC_ID and its start/end s_memtime) and reconstructs per-CU concurrency.
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -I../../include -I.. tools/residency_probe.hip -o residency_probe
//   residency_probe [replicates] [age_bins]
#define COLATE_EM_TRACE 1
#include "../em_kernels.hip"
#include "../em_kernels_ilp.hip"  // (same template again: the probe builds one translation unit)

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <map>
#include <vector>

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 1024, A = argc > 2 ? atoi(argv[2]) : 185, E = 23, lo = (185 - A) / 2;
  std::vector<double> full(185), grid(A), ep(E), init(E, 1.0 / 20000), sh((size_t)B * A, 0.0), ns((size_t)B * A, 0.0);
  full[0] = 0;
  for (int b = 1; b < 185; b++) full[b] = std::exp((b - 1) / 10.0) / 10.0;
  for (int b = 0; b < A; b++) grid[b] = full[lo + b];
  ep[0] = 0;
  ep[1] = 0;
  for (int e = 2; e < E - 1; e++) ep[e] = std::pow(10.0, 3.0 + 4.0 * (e - 1) / (E - 3.0)) / 28.0;
  ep[E - 1] = 1e8 / 28.0;
  if (lo > 0) ep[1] = 0;  // (epochs[0] <= age_grid[0] holds: 0)
  unsigned s = 12345;
  for (int r = 0; r < B; r++)
    for (int b = 40; b <= 150; b++) {
      s = s * 1664525u + 1013904223u;
      if (b - lo < 0 || b - lo >= A) continue;
      double tot = (50 + 450.0 * (s >> 8) / 16777216.0) * 11, pr = 1 - std::exp(-full[b] / 12000);
      sh[(size_t)r * A + b - lo] = 0.8 * pr * tot;
      ns[(size_t)r * A + b - lo] = tot - sh[(size_t)r * A + b - lo];
    }
  double *d_grid, *d_sh, *d_ns, *d_ep, *d_init, *d_rates, *d_ll, *d_dbg;
  int *d_it, *d_fl;
  hipMalloc(&d_grid, A * 8); hipMalloc(&d_sh, sh.size() * 8); hipMalloc(&d_ns, ns.size() * 8);
  hipMalloc(&d_ep, E * 8); hipMalloc(&d_init, E * 8); hipMalloc(&d_rates, (size_t)B * E * 8);
  hipMalloc(&d_ll, B * 8); hipMalloc(&d_it, B * 4); hipMalloc(&d_fl, B * 4); hipMalloc(&d_dbg, (size_t)B * 4 * 8);
  hipMemcpy(d_grid, grid.data(), A * 8, hipMemcpyHostToDevice);
  hipMemcpy(d_sh, sh.data(), sh.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(d_ns, ns.data(), ns.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(d_ep, ep.data(), E * 8, hipMemcpyHostToDevice);
  hipMemcpy(d_init, init.data(), E * 8, hipMemcpyHostToDevice);
  ColateEmArgs a{};
  a.B = B, a.E = E, a.A = A, a.mode = 0;
  a.age_grid = d_grid, a.cnt_sh = d_sh, a.cnt_ns = d_ns, a.epochs = d_ep, a.rates_in = d_init;
  a.max_iter = 100000, a.min_iter = 1000, a.rel_tol = 1e-7, a.rate_floor = 5e-9;
  a.out_rates = d_rates, a.out_iters = d_it, a.out_ll = d_ll, a.out_flags = d_fl, a.out_num = d_dbg;
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
  std::vector<unsigned long long> dbg((size_t)B * 4);
  hipMemcpy(dbg.data(), d_dbg, dbg.size() * 8, hipMemcpyDeviceToHost);
  printf("launch: %s, B=%d A=%d: %.3f ms\n", hipGetErrorString(err), B, A, ms);
  // per CU: sweep the start/end events
  struct Ev { unsigned long long t; int d; };
  std::map<unsigned, std::vector<Ev>> per_cu;
  double life = 0;
  for (int r = 0; r < B; r++) {
    const unsigned hw = (unsigned)dbg[(size_t)r * 4], xcc = (unsigned)dbg[(size_t)r * 4 + 1] & 0xf;
    const unsigned cu = (hw >> 8) & 0xf, sh_id = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    const unsigned key = (xcc << 12) | (se << 8) | (sh_id << 4) | cu;
    per_cu[key].push_back({dbg[(size_t)r * 4 + 2], +1});
    per_cu[key].push_back({dbg[(size_t)r * 4 + 3], -1});
    life += (double)(dbg[(size_t)r * 4 + 3] - dbg[(size_t)r * 4 + 2]);
  }
  std::map<int, int> hist_max, hist_n;
  for (auto& kv : per_cu) {
    auto& v = kv.second;
    std::sort(v.begin(), v.end(), [](const Ev& x, const Ev& y) { return x.t < y.t || (x.t == y.t && x.d < y.d); });
    int cur = 0, mx = 0;
    for (auto& e : v) {
      cur += e.d;
      mx = std::max(mx, cur);
    }
    hist_max[mx]++;
    hist_n[(int)v.size() / 2]++;
  }
  printf("distinct (xcc,se,sh,cu) = %zu; mean workgroup lifetime %.0f ticks of s_memtime\n", per_cu.size(), life / B);
  for (auto& kv : hist_n) printf("  CUs that ran %d workgroups: %d\n", kv.first, kv.second);
  for (auto& kv : hist_max) printf("  CUs whose peak concurrency was %d workgroups: %d\n", kv.first, kv.second);
  return 0;
}
