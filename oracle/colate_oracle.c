/* oracle/colate_oracle.c -- TEST INFRASTRUCTURE ONLY (see colate_oracle.h).
 *
 * CPU restatement, in plain C, of the reference's EM path.  Every function
 * cites the reference lines it follows (paths relative to /root/reference/).
 * The operation ORDER of the reference is kept on purpose (several of its
 * formulas cancel catastrophically; an algebraically nicer form would give
 * different doubles), and the file is compiled with -ffp-contract=off so no
 * multiply-add is fused, as in the reference's x86-64 build.
 */
#include "colate_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define LOG0 (-INFINITY) /* the reference's log(0.0), coal_EM.hpp:21 */

static int absent(double x) { return isinf(x) || isnan(x); }

/* ---- libm-noise hook (checker-side experiment, OFF by default) ----------------------------------------
 * With a non-zero seed every exp / log / log1p result inside the EM path below is moved to a neighbouring
 * double at random (1/4 up, 1/4 down, 1/2 unchanged): the result the same source would give when linked
 * against another libm whose functions are, like glibc's, accurate to < 1 ulp but not correctly rounded
 * (the reference's own shipped binaries are macOS builds).  tests/oracle_lib.stable_mask uses it to find the
 * epochs whose value is pinned by the reference's SOURCE rather than by the last bit of one libm.  With the
 * seed at 0 (the default) the three wrappers are the plain libm calls and the oracle is bit-equal to the
 * reference build in oracle/_ref (tests/test_oracle_golden.py). */
static unsigned long long g_libm_noise = 0;
void oracle_set_libm_noise(unsigned long long seed) { g_libm_noise = seed; }
static double jitter(double y) {
  if (!g_libm_noise || !isfinite(y) || y == 0.0) return y;
  g_libm_noise ^= g_libm_noise << 13;
  g_libm_noise ^= g_libm_noise >> 7;
  g_libm_noise ^= g_libm_noise << 17;
  if (!g_libm_noise) g_libm_noise = 0x9e3779b97f4a7c15ull;
  switch ((g_libm_noise >> 33) & 3) {
    case 0: return nextafter(y, INFINITY);
    case 1: return nextafter(y, -INFINITY);
    default: return y;
  }
}
static double o_exp(double x) { return jitter(exp(x)); }
static double o_log(double x) { return jitter(log(x)); }
static double o_log1p(double x) { return jitter(log1p(x)); }

/* include/coal/coal_EM.cpp:5-31 */
double oracle_logsumexp(double loga, double logb) {
  if (absent(loga)) return absent(logb) ? LOG0 : logb;
  if (absent(logb)) return loga;
  if (loga > logb) return loga + o_log1p(o_exp(logb - loga));
  return logb + o_log1p(o_exp(loga - logb));
}

/* include/coal/coal_EM.cpp:33-58 */
double oracle_logminusexp(double loga, double logb) {
  if (absent(loga)) return LOG0;
  if (absent(logb)) return loga;
  if (loga < logb) return LOG0;
  return loga + o_log1p(-o_exp(logb - loga));
}

/* include/coal/coal_EM.cpp:97-151 with t_int = epochs, ep_index[i] = i
 * (constructor, include/coal/coal_EM.hpp:38-50). */
void oracle_get_AB(int E, const double* epochs, const double* rates, double* A_ep, double* B_ep) {
  double* cs = (double*)malloc(sizeof(double) * (size_t)E);
  cs[0] = 0.0;
  for (int i = 1; i < E; i++) cs[i] = cs[i - 1] + rates[i - 1] * (epochs[i] - epochs[i - 1]);
  for (int i = 0; i < E - 1; i++) {
    double t_begin = epochs[i], t_end = epochs[i + 1];
    double rate = rates[i];
    double inv = 1.0 / rates[i];
    if (rate > 0 && t_end != 0 && t_end - t_begin > 0) {
      A_ep[i] = oracle_logminusexp(-cs[i], -cs[i + 1]);
      double b = (t_begin + inv) - (t_end + inv) * o_exp(-cs[i + 1] + cs[i]);
      B_ep[i] = o_log(b) - cs[i];
    } else {
      A_ep[i] = LOG0;
      B_ep[i] = LOG0;
    }
  }
  {
    int i = E - 1;
    double rate = rates[i];
    if (rate > 0) {
      A_ep[i] = -cs[i];
      B_ep[i] = o_log(epochs[i] + 1.0 / rate) - cs[i];
    } else {
      A_ep[i] = LOG0;
      B_ep[i] = LOG0;
    }
  }
  free(cs);
}

/* The merged time grid of include/coal/coal_EM.cpp:60-95 for age_begin == age_end == age,
 * reduced to what the identical-ages branches read.  With k = the largest e such
 * that epochs[e] <= age (the test at :66 is a strict `age < epochs[e]`), the
 * grid is  t_0..t_k, age, age, t_{k+1}..t_{E-1}  and the cumulative sums of
 * coal_EM.cpp:178-181 / 313-316 at the grid positions that are read are
 *   c[k]   = cs_k                                   (same adds as get_AB)
 *   c[k+1] = c[k]   + rates[k] * (age - t_k)
 *   c[k+2] = c[k+1] + rates[k] * (age - age)
 *   c[k+3] = c[k+2] + rates[k] * (t_{k+1} - age)    (only if k < E-1)
 */
typedef struct {
  int k;
  double ck, ck1, ck2, ck3;
} tint_t;

static tint_t merged_grid(int E, const double* epochs, const double* rates, double age) {
  tint_t g;
  int k = E - 1;
  for (int e = 0; e < E; e++) {
    if (age < epochs[e]) {
      k = e - 1;
      break;
    }
  }
  g.k = k;
  double c = 0.0;
  for (int i = 1; i <= k; i++) c = c + rates[i - 1] * (epochs[i] - epochs[i - 1]);
  g.ck = c;
  g.ck1 = g.ck + rates[k] * (age - epochs[k]);
  g.ck2 = g.ck1 + rates[k] * (age - age);
  g.ck3 = (k < E - 1) ? g.ck2 + rates[k] * (epochs[k + 1] - age) : 0.0;
  return g;
}

/* include/coal/coal_EM.cpp:153-295, times_identical branch (live lines 186-210, 244-261,
 * 263-293). */
double oracle_em_shared(int E, const double* epochs, const double* rates, const double* A_ep,
                        const double* B_ep, double age, double* num, double* denom) {
  for (int e = 0; e < E; e++) num[e] = denom[e] = 0.0; /* :157-158 */
  tint_t g = merged_grid(E, epochs, rates, age);
  int k = g.k;
  double nc = 1.0; /* "unset" sentinel, :184 */
  for (int e = 0; e <= k; e++) {
    if (e < k) { /* :191-197 */
      num[e] = A_ep[e];
      denom[e] = B_ep[e];
    } else { /* e == k, :198-210 and :244-248 */
      double t_begin = epochs[k], t_end = age;
      double inv = 1.0 / rates[k];
      if (rates[k] > 0) {
        num[e] = oracle_logminusexp(-g.ck, -g.ck1);
        denom[e] = o_log((t_begin + inv) / inv - (t_end + inv) / inv * o_exp(-g.ck1 + g.ck)) +
                   o_log(inv) - g.ck;
      } else {
        num[e] = LOG0;
        denom[e] = LOG0;
      }
    }
    if (nc == 1.0) /* :254-258 */
      nc = num[e];
    else
      nc = oracle_logsumexp(nc, num[e]);
  }
  if (!isinf(nc) && !isnan(nc)) { /* :263-287 */
    double integ = 1.0;
    int lim = (E - 1 < k + 1) ? E - 1 : k + 1;
    int e;
    for (e = 0; e < lim; e++) {
      num[e] -= nc;
      denom[e] -= nc;
      num[e] = o_exp(num[e]);
      if (integ > 0.0)
        integ -= num[e];
      else
        integ = 0.0;
      denom[e] = o_exp(denom[e]);
      denom[e] += -epochs[e] * num[e] + (epochs[e + 1] - epochs[e]) * integ;
      if (denom[e] < 0.0) denom[e] = 0.0;
    }
    if (k == E - 1) {
      e = E - 1;
      num[e] -= nc;
      denom[e] -= nc;
      num[e] = o_exp(num[e]);
      denom[e] = o_exp(denom[e]);
      denom[e] -= epochs[e] * num[e];
      if (denom[e] < 0.0) denom[e] = 0.0;
    }
  } else { /* :288-292 */
    nc = 0.0;
    for (int e = 0; e < E; e++) num[e] = denom[e] = 0.0;
  }
  return nc;
}

/* include/coal/coal_EM.cpp:297-468, times_identical branch (live lines 323-357, 435-466). */
double oracle_em_notshared(int E, const double* epochs, const double* rates, const double* A_ep,
                           const double* B_ep, double age, double* num, double* denom) {
  tint_t g = merged_grid(E, epochs, rates, age);
  int k = g.k;
  double rate = rates[k];
  double inv = 1.0 / rates[k];
  double nc;
  if (k != E - 1) { /* :330-349 */
    double t_begin = age, t_end = epochs[k + 1];
    if (rate > 0) {
      num[k] = oracle_logminusexp(-g.ck2, -g.ck3);
      denom[k] = o_log((t_begin + inv) - (t_end + inv) * o_exp(-g.ck3 + g.ck2)) - g.ck2;
      nc = num[k];
    } else {
      num[k] = LOG0;
      denom[k] = LOG0;
      nc = LOG0;
    }
    for (int e = k + 1; e < E; e++) {
      num[e] = A_ep[e];
      denom[e] = B_ep[e];
      nc = oracle_logsumexp(nc, num[e]);
    }
  } else { /* :350-357 (the reference asserts rate > 0 here) */
    num[k] = -g.ck2;
    denom[k] = o_log(age + inv) - g.ck2;
    nc = num[k];
  }
  if (!isinf(nc) && !isnan(nc)) { /* :435-460 */
    double integ = 1.0;
    int e;
    for (e = 0; e < k; e++) {
      num[e] = 0.0;
      denom[e] = epochs[e + 1] - epochs[e];
    }
    for (; e < E - 1; e++) {
      num[e] -= nc;
      denom[e] -= nc;
      num[e] = o_exp(num[e]);
      if (integ > 0.0)
        integ -= num[e];
      else
        integ = 0.0;
      denom[e] = o_exp(denom[e]);
      denom[e] += -epochs[e] * num[e] + (epochs[e + 1] - epochs[e]) * integ;
      if (denom[e] < 0.0) denom[e] = 0.0;
    }
    e = E - 1;
    num[e] -= nc;
    denom[e] -= nc;
    num[e] = o_exp(num[e]);
    denom[e] = o_exp(denom[e]);
    denom[e] -= epochs[e] * num[e];
    if (denom[e] < 0.0) denom[e] = 0.0;
  } else { /* :461-465 */
    nc = 0.0;
    for (int e = 0; e < E; e++) num[e] = denom[e] = 0.0;
  }
  return nc;
}

/* include/coal/coal.cpp:3698-3733 */
double oracle_estep(int E, int A, const double* epochs, const double* rates, const double* age_grid,
                    const double* cnt_shared, const double* cnt_notshared, double* num_acc,
                    double* den_acc, int* flags) {
  double* buf = (double*)malloc(sizeof(double) * (size_t)E * 4);
  double *A_ep = buf, *B_ep = buf + E, *num = buf + 2 * E, *denom = buf + 3 * E;
  oracle_get_AB(E, epochs, rates, A_ep, B_ep); /* coal_EM ctor at coal.cpp:3698 */
  for (int e = 0; e < E; e++) num[e] = denom[e] = num_acc[e] = den_acc[e] = 0.0;
  double ll = 0.0;
  for (int bin = 0; bin < A; bin++) {
    for (int kind = 0; kind < 2; kind++) {
      double count = kind == 0 ? cnt_shared[bin] : cnt_notshared[bin];
      if (!(count > 0)) continue;
      double logl = kind == 0
                        ? oracle_em_shared(E, epochs, rates, A_ep, B_ep, age_grid[bin], num, denom)
                        : oracle_em_notshared(E, epochs, rates, A_ep, B_ep, age_grid[bin], num,
                                              denom);
      ll += count * logl;
      /* coal_EM.cpp:351: for a not-shared count inside the last epoch the reference asserts rate > 0 (it aborts);
       * reported like the other aborts (coal.cpp:3711) */
      if (kind == 1 && !(age_grid[bin] < epochs[E - 1]) && !(rates[E - 1] > 0)) *flags |= ORACLE_FLAG_NAN;
      for (int e = 0; e < E; e++) {
        if (isnan(num[e]) || isnan(denom[e])) *flags |= ORACLE_FLAG_NAN;
        if (num[e] < 0.0 || denom[e] < 0.0) *flags |= ORACLE_FLAG_NEG;
        num_acc[e] += count * num[e];
        den_acc[e] += count * denom[e];
      }
    }
  }
  free(buf);
  return ll;
}

/* include/coal/coal.cpp:3771-3815, is_EM branch (regularise == 2, coal.cpp:3091) */
void oracle_mstep(int E, const double* num_acc, const double* den_acc, double rate_floor,
                  double* rates) {
  for (int e = 0; e < E; e++) {
    if (num_acc[e] == 0) {
      rates[e] = (e > 0) ? rates[e - 1] : 0.0;
    } else if (den_acc[e] == 0) {
      /* keep */
    } else {
      rates[e] = num_acc[e] / den_acc[e];
      if (rates[e] < rate_floor) rates[e] = rate_floor;
    }
  }
}

/* include/coal/coal.cpp:3675-3827 */
void oracle_em_run(int E, int A, const double* age_grid, const double* epochs,
                   const double* init_rates, const double* cnt_shared, const double* cnt_notshared,
                   int max_iter, int min_iter, double rel_tol, double rate_floor, double* out_rates,
                   int* out_iters, double* out_loglik, int* out_flags) {
  double* buf = (double*)malloc(sizeof(double) * (size_t)E * 3);
  double *rates = buf, *num_acc = buf + E, *den_acc = buf + 2 * E;
  memcpy(rates, init_rates, sizeof(double) * (size_t)E);
  double ll = LOG0, prev_ll = LOG0;
  int flags = 0;
  int iter;
  for (iter = 0; iter < max_iter; iter++) {
    prev_ll = ll;
    ll = oracle_estep(E, A, epochs, rates, age_grid, cnt_shared, cnt_notshared, num_acc, den_acc,
                      &flags);
    oracle_mstep(E, num_acc, den_acc, rate_floor, rates);
    if ((ll / prev_ll > 1.0 - rel_tol) & (iter > min_iter)) break; /* coal.cpp:3822 */
  }
  if (iter == max_iter) flags |= ORACLE_FLAG_MAXITER;
  memcpy(out_rates, rates, sizeof(double) * (size_t)E);
  *out_iters = iter;
  *out_loglik = ll;
  *out_flags = flags;
  free(buf);
}

void oracle_em_batch(int B, int E, int A, const double* age_grid, const double* cnt_shared,
                     const double* cnt_notshared, const double* epochs, const double* init_rates,
                     int max_iter, int min_iter, double rel_tol, double rate_floor,
                     double* out_rates, int* out_iters, double* out_loglik, int* out_flags) {
  for (int b = 0; b < B; b++)
    oracle_em_run(E, A, age_grid, epochs, init_rates, cnt_shared + (size_t)b * A,
                  cnt_notshared + (size_t)b * A, max_iter, min_iter, rel_tol, rate_floor,
                  out_rates + (size_t)b * E, out_iters + b, out_loglik + b, out_flags + b);
}

/* include/coal/coal.cpp:3126-3137 */
int oracle_age_grid(double* age_grid, int cap) {
  double C = 10;
  int A = ((int)(log(1e8) * C)) + 1;
  if (A > cap) return -1;
  age_grid[0] = 0.0;
  for (int bin = 0; bin < A - 1; bin++) age_grid[bin + 1] = exp(bin / C) / 10.0;
  return A;
}

/* include/coal/coal.cpp:3551-3632.  std::stof(tmp) == (float)strtof(tmp), widened to double. */
int oracle_epochs_from_bins(const char* bins, double age, double years_per_gen, double* epochs,
                            int cap, int* ep_null_out) {
  double log_10 = log(10);
  double log_age = log(age * years_per_gen) / log_10;
  char tmp[64];
  double val[3];
  size_t n = strlen(bins), i = 0;
  for (int f = 0; f < 3; f++) {
    size_t l = 0;
    if (f > 0 && i >= n) return -1; /* "epochs format is wrong", :3568, :3580 */
    while (i < n && bins[i] != ',') {
      if (l + 1 < sizeof(tmp)) tmp[l++] = bins[i];
      i++;
    }
    tmp[l] = 0;
    val[f] = (double)strtof(tmp, NULL);
    i++;
  }
  double epoch_lower = val[0], epoch_upper = val[1], epoch_step = val[2];
  int E = 0, ep_null = 0;
#define PUSH(x)              \
  do {                       \
    if (E >= cap) return -2; \
    epochs[E++] = (x);       \
  } while (0)
  PUSH(0.0);
  if (log_age < epoch_lower && age != 0.0) { /* :3597-3601 */
    PUSH(age);
    log_age = -1;
  }
  double epoch_boundary = epoch_lower;
  while (epoch_boundary < epoch_upper) { /* :3603-3627 */
    if (epoch_boundary > log_age && log_age != -1) {
      PUSH(age);
      if (epoch_boundary - log_age < 0.25 * epoch_step) epoch_boundary += epoch_step;
      log_age = -1;
    } else {
      if (log_age != -1) ep_null++;
      PUSH(exp(log_10 * epoch_boundary) / years_per_gen);
    }
    epoch_boundary += epoch_step;
  }
  PUSH(exp(log_10 * epoch_upper) / years_per_gen);
  {
    double last = epochs[E - 1];
    double m = 10 * last;
    PUSH((1e8 > m ? 1e8 : m) / years_per_gen);
  }
#undef PUSH
  if (ep_null_out) *ep_null_out = ep_null;
  return E;
}

/* include/coal/coal.cpp:3508-3549 (epochs from line 2 of a .coal file: blank/tab separated tokens through std::stof,
 * i.e. FLOAT precision; for an ancient sample `age` takes the place of the second epoch when it is younger than it)
 * and :3638-3646 (starting rates: skip two numbers, then one `is >> double` per epoch -- extraction simply runs on into
 * the next line when `age` added an epoch; at end of file the default 1/20000 stays, coal.cpp:3636).
 * Returns E, or < 0 where the reference asserts (:3540-3546) / cannot read. */
#include <stdio.h>
int oracle_epochs_from_coal(const char* path, double age, double* epochs, double* init_rates, int cap) {
  FILE* f = fopen(path, "rb");
  if (!f) return -3;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  char* buf = (char*)malloc((size_t)n + 1);
  if (fread(buf, 1, (size_t)n, f) != (size_t)n) {
    fclose(f);
    free(buf);
    return -3;
  }
  fclose(f);
  buf[n] = 0;
  char* l2 = strchr(buf, '\n'); /* getline x 2: the second line holds the epochs */
  if (!l2) {
    free(buf);
    return -1;
  }
  l2++;
  char* l2end = strchr(l2, '\n');
  char* rest = l2end ? l2end + 1 : buf + n;
  if (l2end) *l2end = 0;
  int E = 0, ep = 0;
  char tmp[128];
  size_t l = 0;
  for (char* c = l2;; c++) { /* :3514-3539 */
    int sep = (*c == ' ' || *c == '\t');
    if (sep || *c == 0) {
      if (*c == 0 && l == 0) break; /* `if(tmp != "")` */
      tmp[l] = 0;
      double v = (double)strtof(tmp, NULL);
      if (ep == 1 && age < v && age != 0.0) {
        if (E >= cap) return -2;
        epochs[E++] = age;
        ep++;
      }
      if (ep != 1 || age == 0.0) {
        if (E >= cap) return -2;
        epochs[E++] = v;
        ep++;
      }
      l = 0;
      if (*c == 0) break;
    } else if (l + 1 < sizeof(tmp)) {
      tmp[l++] = *c;
    }
  }
  if ((age != 0.0 && ep <= 2) || (age == 0.0 && ep <= 1) || epochs[0] != 0) { /* :3540-3543 */
    free(buf);
    return -1;
  }
  for (int e = 1; e < E; e++)
    if (!(epochs[e] > epochs[e - 1])) { /* :3544-3546 */
      free(buf);
      return -1;
    }
  for (int e = 0; e < E; e++) init_rates[e] = 1.0 / 20000.0; /* :3636-3637 */
  char* p = rest;
  int good = 1;
  for (int k = -2; k < E && good; k++) { /* :3640-3644, formatted extraction */
    while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r') p++;
    if (*p == 0) break; /* end of file while skipping blanks: the value is left alone */
    char* q;
    double v = strtod(p, &q);
    if (q == p) { /* not a number: C++11 stores 0 and the stream fails for good */
      v = 0.0;
      good = 0;
    }
    p = q;
    if (k >= 0) init_rates[k] = v;
  }
  free(buf);
  return E;
}

/* include/coal/coal.cpp:3358-3441 for one replicate; emp tables reduced to their row 0
 * (bin1 loop runs once, coal.cpp:3397). */
void oracle_bootstrap_counts(int nb, int A, const double* age_grid, double age,
                             const double* weights, const double* sh_block,
                             const double* ns_block, const double* sh_emp_block,
                             const double* ns_emp_block, double* cnt_shared,
                             double* cnt_notshared) {
  double* buf = (double*)calloc((size_t)A * 3, sizeof(double));
  double *sh_emp = buf, *ns_emp = buf + A, *F = buf + 2 * A;
  for (int b = 0; b < A; b++) cnt_shared[b] = cnt_notshared[b] = 0.0;
  for (int j = 0; j < nb; j++) {
    if (weights[j] > 0.0) {
      for (int b = 0; b < A; b++) cnt_shared[b] += weights[j] * sh_block[(size_t)j * A + b];
      for (int b = 0; b < A; b++) cnt_notshared[b] += weights[j] * ns_block[(size_t)j * A + b];
      for (int b = 0; b < A; b++) sh_emp[b] += weights[j] * sh_emp_block[(size_t)j * A + b];
      for (int b = 0; b < A; b++) ns_emp[b] += weights[j] * ns_emp_block[(size_t)j * A + b];
    }
  }
  int bin = 0;
  while (age_grid[bin] <= age) bin++; /* :3395-3396 */
  int bin_start = bin;
  double lower_age = age_grid[bin_start - 1]; /* bin1 == 0, :3399-3400 */
  double fcount = 0.0;
  for (bin = bin_start; bin < A; bin++) { /* :3406-3417 */
    fcount += sh_emp[bin];
    if (sh_emp[bin] > 0) F[bin] = sh_emp[bin] / (sh_emp[bin] + ns_emp[bin]);
  }
  for (bin = bin_start; bin < A; bin++) { /* :3420-3425 */
    F[bin - 1] *= (age_grid[bin] - lower_age);
    lower_age = age_grid[bin];
  }
  double normf = 0.0;
  for (bin = 0; bin < A; bin++) normf += F[bin]; /* :3429-3432 */
  for (bin = 0; bin < A; bin++) {                /* :3435-3441 */
    F[bin] /= normf;
    F[bin] *= fcount;
    /* std::max(0.0, F): returns 0.0 unless 0.0 < F, so NaN -> 0.0 */
    cnt_shared[bin] += (0.0 < F[bin]) ? F[bin] : 0.0;
  }
  free(buf);
}

/* ---- std::mt19937 (Matsumoto & Nishimura 1998; ISO C++ [rand.eng.mers] parameters) ---- */
void oracle_mt_seed(oracle_mt19937* g, unsigned int seed) {
  g->mt[0] = seed;
  for (int i = 1; i < 624; i++)
    g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (unsigned int)i;
  g->idx = 624;
}

unsigned int oracle_mt_next(oracle_mt19937* g) {
  if (g->idx >= 624) {
    for (int i = 0; i < 624; i++) {
      unsigned int y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
      g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    g->idx = 0;
  }
  unsigned int y = g->mt[g->idx++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

/* libstdc++ 11 <bits/uniform_int_dist.h>: 32-bit generator range, Lemire's
 * nearly-divisionless method (used at coal.cpp:3330, 3355). */
int oracle_uniform_int(oracle_mt19937* g, int n) {
  unsigned int range = (unsigned int)n;
  unsigned long long product = (unsigned long long)oracle_mt_next(g) * range;
  unsigned int low = (unsigned int)product;
  if (low < range) {
    unsigned int threshold = (0u - range) % range;
    while (low < threshold) {
      product = (unsigned long long)oracle_mt_next(g) * range;
      low = (unsigned int)product;
    }
  }
  return (int)(product >> 32);
}

/* libstdc++ generate_canonical<double,53> over a 32-bit generator: two draws. */
double oracle_uniform_real01(oracle_mt19937* g) {
  double sum = 0.0, tmp = 1.0;
  for (int k = 0; k < 2; k++) {
    sum += (double)oracle_mt_next(g) * tmp;
    tmp *= 4294967296.0;
  }
  double r = sum / tmp;
  if (r >= 1.0) r = nextafter(1.0, 0.0);
  return r;
}

/* include/coal/coal.cpp:3350-3357 */
void oracle_block_weights(oracle_mt19937* g, int nb, int num_bootstrap, double* weights) {
  if (num_bootstrap == 1) {
    for (int j = 0; j < nb; j++) weights[j] = 1.0;
  } else {
    for (int j = 0; j < nb; j++) weights[j] = 0.0;
    for (int j = 0; j < nb; j++) weights[oracle_uniform_int(g, nb)] += 1.0;
  }
}
