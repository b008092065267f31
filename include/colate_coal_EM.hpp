/* include/colate_coal_EM.hpp -- a C++ class with the public face of the reference's `class coal_EM`
 * (include/coal/coal_EM.hpp:14-63: constructor (epochs, coal), UpdateCoal, EM_shared / EM_notshared(age_begin,
 * age_end, num, denom) -> log-normaliser) on top of libcolate_amd.so, so that the reference's call site
 * (include/coal/coal.cpp:3698-3721) and its Catch2 test (include/test/test_aDNA.cpp:68-212) compile against it unchanged:
 *
 *     #include "colate_coal_EM.hpp"
 *     using coal_EM = colate::coal_EM;        // instead of #include "coal_EM.hpp"
 *
 * Header-only, C ABI underneath (colate_em_estep: one E-step over a one-bin age grid with count 1 yields exactly
 * the reference's per-bin num / denom / logl).  One GPU launch per call: for parity work and experiments, not for
 * speed -- the fast path is colate_em_batch.  Only age_begin == age_end is implemented, the only
 * way mut() calls it (coal.cpp:3708, 3721); anything else throws std::invalid_argument.  A failing call throws
 * std::runtime_error with colate_last_error() (the reference would abort on its asserts). */
#ifndef COLATE_COAL_EM_HPP
#define COLATE_COAL_EM_HPP

#include <stdexcept>
#include <string>
#include <vector>

#include "colate_amd.h"

namespace colate {

class coal_EM {
 public:
  coal_EM(std::vector<double>& epochs, std::vector<double>& coal) : epochs_(epochs), coal_rates_(coal) {}
  void UpdateCoal(std::vector<double>& coal) { coal_rates_ = coal; }

  double EM_shared(double age_begin, double age_end, std::vector<double>& num, std::vector<double>& denom) {
    return one(age_begin, age_end, num, denom, true);
  }
  double EM_notshared(double age_begin, double age_end, std::vector<double>& num, std::vector<double>& denom) {
    return one(age_begin, age_end, num, denom, false);
  }

 private:
  double one(double age_begin, double age_end, std::vector<double>& num, std::vector<double>& denom, bool shared) {
    if (age_begin != age_end)
      throw std::invalid_argument("colate::coal_EM implements age_begin == age_end only (the path mut() uses)");
    const int E = (int)epochs_.size();
    num.assign(E, 0.0);    // coal_EM.cpp:157-158
    denom.assign(E, 0.0);
    const double one_count = 1.0, zero = 0.0;
    double loglik = 0.0;
    int flags = 0;
    const int rc = colate_em_estep(1, E, 1, &age_begin, shared ? &one_count : &zero, shared ? &zero : &one_count,
                                   epochs_.data(), coal_rates_.data(), num.data(), denom.data(), &loglik, &flags);
    if (rc != COLATE_OK) throw std::runtime_error(std::string("colate_em_estep: ") + colate_last_error());
    if (flags & (COLATE_FLAG_NAN | COLATE_FLAG_NEG))  // the reference aborts here (coal.cpp:3711-3714, coal_EM.cpp:128-129, 351)
      throw std::runtime_error("colate::coal_EM: NaN or negative sufficient statistics (the reference asserts on these)");
    return loglik;
  }
  std::vector<double> epochs_, coal_rates_;
};

}  // namespace colate
#endif
