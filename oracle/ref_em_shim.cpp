// oracle/ref_em_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Our own extern "C" window onto the *reference's* class coal_EM
// (reference include/coal/coal_EM.hpp:14-63), so that Python (ctypes) can call
// the reference E-step directly.  This file contains no reference code: it
// only #includes the reference's public header from /root/reference (via -I)
// and is linked against the reference's own object files, which oracle/Makefile
// compiles from where they lie into oracle/_ref/ (git-ignored).
//
// Used by tests/golden/make_golden.py to generate the committed golden vectors
// and by tests to pin oracle/colate_oracle.c against the real reference when
// oracle/_ref/libref_em.so is present.
#include "coal_EM.hpp"

#include <vector>

extern "C" {

// One call of coal_EM::EM_shared (kind=0) or coal_EM::EM_notshared (kind=1) on a
// freshly constructed coal_EM(epochs, rates), exactly as coal.cpp:3698-3721 uses it.
// num/denom are caller-owned [E]; they are zero-filled first, as the caller's
// vectors are at coal.cpp:3703.
double ref_em_call(int kind, int E, const double* epochs, const double* rates,
                   double age_begin, double age_end, double* num, double* denom) {
  std::vector<double> ep(epochs, epochs + E), cr(rates, rates + E);
  std::vector<double> n(E, 0.0), d(E, 0.0);
  coal_EM EM(ep, cr);
  double logl = (kind == 0) ? EM.EM_shared(age_begin, age_end, n, d)
                            : EM.EM_notshared(age_begin, age_end, n, d);
  for (int e = 0; e < E; e++) {
    num[e] = n[e];
    denom[e] = d[e];
  }
  return logl;
}

// One whole E-step over an age grid, accumulating with the loop structure of
// coal.cpp:3704-3733 (bins ascending, shared before not-shared, count > 0 only,
// num/denom vectors reused between calls).  Returns the log-likelihood.
double ref_em_estep(int E, int A, const double* epochs, const double* rates, const double* age_grid,
                    const double* cnt_shared, const double* cnt_notshared, double* num_acc,
                    double* den_acc) {
  std::vector<double> ep(epochs, epochs + E), cr(rates, rates + E);
  std::vector<double> n(E, 0.0), d(E, 0.0);
  coal_EM EM(ep, cr);
  double ll = 0.0;
  for (int e = 0; e < E; e++) num_acc[e] = den_acc[e] = 0.0;
  for (int b = 0; b < A; b++) {
    if (cnt_shared[b] > 0) {
      double c = cnt_shared[b];
      double logl = EM.EM_shared(age_grid[b], age_grid[b], n, d);
      ll += c * logl;
      for (int e = 0; e < E; e++) {
        num_acc[e] += c * n[e];
        den_acc[e] += c * d[e];
      }
    }
    if (cnt_notshared[b] > 0) {
      double c = cnt_notshared[b];
      double logl = EM.EM_notshared(age_grid[b], age_grid[b], n, d);
      ll += c * logl;
      for (int e = 0; e < E; e++) {
        num_acc[e] += c * n[e];
        den_acc[e] += c * d[e];
      }
    }
  }
  return ll;
}

}  // extern "C"
