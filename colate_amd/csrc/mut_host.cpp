// colate_amd/csrc/mut_host.cpp -- host-side pieces of `mut()` around the EM hot
// path (include/colate_amd.h, second half): age grid, epoch builders, block
// bootstrap with the F redistribution, .coal writer.  Plain C++ on the CPU; the
// reference does these once per run on the host too (they are O(nb*A) at most).
//
// Random numbers: the reference draws from one std::mt19937 through
// std::uniform_int_distribution<int> (coal.cpp:3330, 3355).  The same standard
// library facilities are used here, so a run with the same --seed on the same
// toolchain consumes the identical stream.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "colate_amd.h"
#include "colate_internal.h"

using colate::fail;

extern "C" {

// coal.cpp:3126-3137: age_bin[0] = 0, age_bin[k] = exp((k-1)/C)/10, C = 10,
// num_age_bins = (int)(log(1e8)*C) + 1 = 185
int colate_age_grid(double* age_grid, int cap) {
  const double C = 10;
  const int A = ((int)(std::log(1e8) * C)) + 1;
  if (!age_grid || cap < A) return fail(COLATE_EINVAL, "age grid needs room for %d values", A);
  age_grid[0] = 0.0;
  for (int bin = 0; bin < A - 1; bin++) age_grid[bin + 1] = std::exp(bin / C) / 10.0;
  return A;
}

// coal.cpp:3551-3632
int colate_epochs_from_bins(const char* bins, double age, double years_per_gen, double* epochs,
                            int cap, int* ep_null_out) {
  if (!bins || !epochs) return fail(COLATE_EINVAL, "NULL argument");
  const std::string s(bins);
  double field[3];
  size_t i = 0;
  for (int f = 0; f < 3; f++) {
    if (f > 0 && i >= s.size())
      return fail(COLATE_EINVAL, "Error: epochs format is wrong. Specify x,y,stepsize.");
    std::string tmp;
    while (i < s.size() && s[i] != ',') tmp += s[i++];
    i++;
    try {
      field[f] = std::stof(tmp);  // float precision, widened (coal.cpp:3566, 3578, 3590)
    } catch (...) {
      return fail(COLATE_EINVAL, "Error: epochs format is wrong. Specify x,y,stepsize.");
    }
  }
  const double epoch_lower = field[0], epoch_upper = field[1], epoch_step = field[2];
  if (!(epoch_step > 0)) return fail(COLATE_EINVAL, "--bins step must be positive");
  const double log_10 = std::log(10);
  double log_age = std::log(age * years_per_gen) / log_10;
  std::vector<double> ep;
  int ep_null = 0;
  ep.push_back(0.0);
  if (log_age < epoch_lower && age != 0.0) {  // coal.cpp:3597-3601
    ep.push_back(age);
    log_age = -1;
  }
  double epoch_boundary = epoch_lower;
  while (epoch_boundary < epoch_upper) {  // coal.cpp:3603-3627
    if (epoch_boundary > log_age && log_age != -1) {
      ep.push_back(age);
      if (epoch_boundary - log_age < 0.25 * epoch_step) epoch_boundary += epoch_step;
      log_age = -1;
    } else {
      if (log_age != -1) ep_null++;
      ep.push_back(std::exp(log_10 * epoch_boundary) / years_per_gen);
    }
    epoch_boundary += epoch_step;
    if ((int)ep.size() > cap) return fail(COLATE_ELIMIT, "more than %d epochs", cap);
  }
  ep.push_back(std::exp(log_10 * epoch_upper) / years_per_gen);
  ep.push_back(std::max(1e8, 10 * ep[ep.size() - 1]) / years_per_gen);
  if ((int)ep.size() > cap) return fail(COLATE_ELIMIT, "more than %d epochs", cap);
  std::copy(ep.begin(), ep.end(), epochs);
  if (ep_null_out) *ep_null_out = ep_null;
  return (int)ep.size();
}

// coal.cpp:3508-3549 (epochs: tokens of line 2 through std::stof, `age` inserted as
// epoch 1 for ancient samples) and coal.cpp:3638-3646 (rates: skip two numbers of
// line 3, then read num_epochs of them)
int colate_epochs_from_coal(const char* path, double age, double* epochs, double* init_rates,
                            int cap) {
  if (!path || !epochs || !init_rates) return fail(COLATE_EINVAL, "NULL argument");
  std::ifstream is(path);
  if (!is) return fail(COLATE_EIO, "cannot open %s", path);
  std::string line;
  std::getline(is, line);
  std::getline(is, line);
  std::vector<double> ep;
  std::string tmp;
  int n = 0;
  auto token = [&](const std::string& t) {
    const double v = std::stof(t);
    if (n == 1 && age < v && age != 0.0) {
      ep.push_back(age);
      n++;
    }
    if (n != 1 || age == 0.0) {
      ep.push_back(v);
      n++;
    }
  };
  try {
    for (char c : line) {
      if (c == ' ' || c == '\t') {
        token(tmp);  // the reference calls stof on empty tokens too (and throws)
        tmp.clear();
      } else {
        tmp += c;
      }
    }
    if (!tmp.empty()) token(tmp);
  } catch (...) {
    return fail(COLATE_EINVAL, "%s: malformed epoch line", path);
  }
  const int E = (int)ep.size();
  if (E < 2 || (age != 0.0 && n <= 2)) return fail(COLATE_EINVAL, "%s: too few epochs", path);
  if (E > cap) return fail(COLATE_ELIMIT, "more than %d epochs", cap);
  if (ep[0] != 0) return fail(COLATE_EINVAL, "%s: first epoch must be 0", path);
  for (int e = 1; e < E; e++)
    if (!(ep[e] > ep[e - 1])) return fail(COLATE_EINVAL, "%s: epochs must increase", path);
  std::copy(ep.begin(), ep.end(), epochs);
  for (int e = 0; e < E; e++) init_rates[e] = COLATE_DEFAULT_INIT_RATE;
  double dummy;
  is >> dummy >> dummy;
  for (int e = 0; e < E; e++) is >> init_rates[e];  // failed extractions leave the default/0 as iostreams do
  return E;
}

void* colate_rng_create(unsigned int seed) {
  std::mt19937* rng = new std::mt19937();
  rng->seed(seed);
  return rng;
}

void colate_rng_destroy(void* rng_state) { delete static_cast<std::mt19937*>(rng_state); }

// coal.cpp:3350-3357
int colate_bootstrap_weights(void* rng_state, int num_bootstrap, int nb, double* weights) {
  if (!rng_state || !weights || num_bootstrap < 1 || nb < 1) return fail(COLATE_EINVAL, "bad argument");
  std::mt19937& rng = *static_cast<std::mt19937*>(rng_state);
  std::uniform_int_distribution<int> dist_blocks(0, nb - 1);
  for (int i = 0; i < num_bootstrap; i++) {
    double* w = weights + (size_t)i * nb;
    if (num_bootstrap == 1) {
      std::fill(w, w + nb, 1.0);
    } else {
      std::fill(w, w + nb, 0.0);
      for (int j = 0; j < nb; j++) w[dist_blocks(rng)] += 1.0;
    }
  }
  return COLATE_OK;
}

// coal.cpp:3358-3451 for all replicates, from given block weights w[num_bootstrap][nb] (the emp tables are reduced to the
// row the reference reads: bin1 == 0, coal.cpp:3397).  The host twin of bootstrap_kernel (bootstrap_kernel.hip).
int colate_bootstrap_counts_from_weights(int num_bootstrap, int nb, int A, const double* age_grid, double age,
                                         const double* weights, const double* sh_block, const double* ns_block,
                                         const double* sh_emp_block, const double* ns_emp_block, double* cnt_shared,
                                         double* cnt_notshared) {
  if (!weights || !age_grid || !sh_block || !ns_block || !sh_emp_block || !ns_emp_block || !cnt_shared || !cnt_notshared)
    return fail(COLATE_EINVAL, "NULL argument");
  if (num_bootstrap < 1 || nb < 1 || A < 2) return fail(COLATE_EINVAL, "bad sizes");
  std::vector<double> sh_emp(A), ns_emp(A), F(A);
  for (int i = 0; i < num_bootstrap; i++) {
    const double* blocks = weights + (size_t)i * nb;
    double* csh = cnt_shared + (size_t)i * A;
    double* cns = cnt_notshared + (size_t)i * A;
    std::fill(csh, csh + A, 0.0);
    std::fill(cns, cns + A, 0.0);
    std::fill(sh_emp.begin(), sh_emp.end(), 0.0);
    std::fill(ns_emp.begin(), ns_emp.end(), 0.0);
    for (int j = 0; j < nb; j++) {
      if (blocks[j] > 0.0) {
        const double w = blocks[j];
        const double* s1 = sh_block + (size_t)j * A;
        const double* s2 = ns_block + (size_t)j * A;
        const double* s3 = sh_emp_block + (size_t)j * A;
        const double* s4 = ns_emp_block + (size_t)j * A;
        for (int b = 0; b < A; b++) csh[b] += w * s1[b];
        for (int b = 0; b < A; b++) cns[b] += w * s2[b];
        for (int b = 0; b < A; b++) sh_emp[b] += w * s3[b];
        for (int b = 0; b < A; b++) ns_emp[b] += w * s4[b];
      }
    }
    // F redistribution, coal.cpp:3392-3441
    int bin = 0;
    while (bin < A && age_grid[bin] <= age) bin++;
    if (bin >= A || bin < 1) return fail(COLATE_EINVAL, "sample age outside the age grid");
    const int bin_start = bin;
    double lower_age = age_grid[bin_start - 1];
    std::fill(F.begin(), F.end(), 0.0);
    double fcount = 0.0;
    for (bin = bin_start; bin < A; bin++) {
      fcount += sh_emp[bin];
      if (sh_emp[bin] > 0) F[bin] = sh_emp[bin] / (sh_emp[bin] + ns_emp[bin]);
    }
    for (bin = bin_start; bin < A; bin++) {
      F[bin - 1] *= (age_grid[bin] - lower_age);
      lower_age = age_grid[bin];
    }
    double normf = 0.0;
    for (bin = 0; bin < A; bin++) normf += F[bin];
    for (bin = 0; bin < A; bin++) {
      F[bin] /= normf;
      F[bin] *= fcount;
      csh[bin] += std::max(0.0, F[bin]);  // max(0.0, NaN) == 0.0 when nothing was redistributed
    }
  }
  return COLATE_OK;
}

// coal.cpp:3344-3451 for all replicates: the weights of replicate i are drawn right before its sums, as the reference
// does -- the generator is shared with nothing else at this point, so drawing them all first is the same stream
int colate_bootstrap_counts(void* rng_state, int num_bootstrap, int nb, int A,
                            const double* age_grid, double age, const double* sh_block,
                            const double* ns_block, const double* sh_emp_block,
                            const double* ns_emp_block, double* cnt_shared,
                            double* cnt_notshared) {
  if (!rng_state) return fail(COLATE_EINVAL, "NULL argument");
  if (num_bootstrap < 1 || nb < 1 || A < 2) return fail(COLATE_EINVAL, "bad sizes");
  std::vector<double> weights((size_t)num_bootstrap * nb);
  if (int rc = colate_bootstrap_weights(rng_state, num_bootstrap, nb, weights.data())) return rc;
  return colate_bootstrap_counts_from_weights(num_bootstrap, nb, A, age_grid, age, weights.data(), sh_block, ns_block,
                                              sh_emp_block, ns_emp_block, cnt_shared, cnt_notshared);
}

// coal.cpp:3660-3672 and 3830-3847.  operator<<(double) with default flags is "%g".
int colate_write_coal(const char* path, int B, int E, const double* epochs, const double* rates,
                      int is_ancient, int ep_null) {
  if (!path || !epochs || !rates || E < 1 || B < 0) return fail(COLATE_EINVAL, "bad argument");
  FILE* f = std::fopen(path, "w");
  if (!f) return fail(COLATE_EIO, "cannot write %s", path);
  std::fprintf(f, "0\n");
  if (is_ancient) {
    std::fprintf(f, "0 ");
    for (int e = ep_null + 1; e < E; e++) std::fprintf(f, "%g ", epochs[e]);
  } else {
    for (int e = 0; e < E; e++) std::fprintf(f, "%g ", epochs[e]);
  }
  std::fprintf(f, "\n");
  for (int i = 0; i < B; i++) {
    const double* r = rates + (size_t)i * E;
    std::fprintf(f, "0 %d ", i);
    if (is_ancient) {
      // coal_rates[0..ep_null] are zeroed, then printed from ep_null on
      for (int e = ep_null; e < E; e++) std::fprintf(f, "%g ", e <= ep_null ? 0.0 : r[e]);
    } else {
      for (int e = 0; e < E; e++) std::fprintf(f, "%g ", r[e]);
    }
    std::fprintf(f, "\n");
  }
  if (std::fclose(f) != 0) return fail(COLATE_EIO, "error closing %s", path);
  return COLATE_OK;
}

}  // extern "C"
