// colate_amd/csrc/tools/coal_EM_shim_check.cpp -- compiles include/colate_coal_EM.hpp the way the reference's call site
// uses `coal_EM` (include/coal/coal.cpp:3698-3721: construct, then EM_shared / EM_notshared per age bin with
// count > 0, accumulate) and prints num/denom/logl as hex floats for tests/test_gpu_coal_em_shim.py to compare with
// the oracle.  stdin: E, epochs[E], rates[E], n, ages[n].
#include <cstdio>
#include <iostream>
#include <vector>

#include "colate_coal_EM.hpp"
using coal_EM = colate::coal_EM;  // (instead of the reference's #include "coal_EM.hpp")

int main() {
  int E, n;
  if (!(std::cin >> E)) return 2;
  std::vector<double> epochs(E), coal_rates(E);
  for (double& x : epochs) std::cin >> x;
  for (double& x : coal_rates) std::cin >> x;
  std::cin >> n;
  std::vector<double> ages(n);
  for (double& a : ages) std::cin >> a;
  try {
    coal_EM EM(epochs, coal_rates);  // coal.cpp:3698
    std::vector<double> num(E), denom(E);
    for (int bin = 0; bin < n; bin++) {
      for (int kind = 0; kind < 2; kind++) {
        const double logl = kind == 0 ? EM.EM_shared(ages[bin], ages[bin], num, denom)      // coal.cpp:3708
                                      : EM.EM_notshared(ages[bin], ages[bin], num, denom);  // coal.cpp:3721
        std::printf("%a", logl);
        for (int e = 0; e < E; e++) std::printf(" %a %a", num[e], denom[e]);
        std::printf("\n");
      }
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  return 0;
}
