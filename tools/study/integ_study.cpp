// tools/study/integ_study.cpp -- RESEARCH TOOL (CPU only, not product, not oracle): what is the rounding residue of the
// reference's `integ = 1 - num[0] - num[1] - ...` recurrence (coal_EM.cpp:270-274, 445-449) per epoch and iteration?
// The reference's E-step is restated here as a template over the floating type (operation order of oracle/colate_oracle.c),
// run in double along the reference's own EM trajectory, and at every iteration compared with the same E-step in long
// double at the same rates: R_e = D_e(double) - D_e(long double), N likewise.  Output: per tail epoch the mean / std / min /
// max of R_e / D_e over the iterations, the fraction of (bin, kind) chains that clamp to 0, and the final rates.
//   g++ -O2 -std=c++17 -ffp-contract=off tools/study/integ_study.cpp -o /tmp/integ_study && /tmp/integ_study table.bin
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

template <typename T> T xexp(T x);
template <> double xexp(double x) { return std::exp(x); }
template <> long double xexp(long double x) { return expl(x); }
template <typename T> T xlog(T x);
template <> double xlog(double x) { return std::log(x); }
template <> long double xlog(long double x) { return logl(x); }
template <typename T> T xlog1p(T x);
template <> double xlog1p(double x) { return std::log1p(x); }
template <> long double xlog1p(long double x) { return log1pl(x); }

template <typename T> bool absent(T x) { return std::isinf(x) || std::isnan(x); }
template <typename T> T lse(T a, T b) {
  const T L0 = -INFINITY;
  if (absent(a)) return absent(b) ? L0 : b;
  if (absent(b)) return a;
  if (a > b) return a + xlog1p<T>(xexp<T>(b - a));
  return b + xlog1p<T>(xexp<T>(a - b));
}
template <typename T> T lme(T a, T b) {
  const T L0 = -INFINITY;
  if (absent(a)) return L0;
  if (absent(b)) return a;
  if (a < b) return L0;
  return a + xlog1p<T>(-xexp<T>(b - a));
}

struct Trace {  // per (bin, kind) chain: integ after each epoch (reference order), for inspection
  std::vector<double> integ;
};

template <typename T>
struct EStep {
  int E, A;
  std::vector<T> ep, rates, Aep, Bep, cs;
  std::vector<T> num, den, N, D;
  T ll;
  // per-epoch sum over bins of c_b * dt_e * integ_e(b)  (the part of D that the residue lives in)
  std::vector<T> Dinteg;
  long clamp_hits = 0, chains = 0;
  void get_AB() {
    const T L0 = -INFINITY;
    cs.assign(E, 0);
    for (int i = 1; i < E; i++) cs[i] = cs[i - 1] + rates[i - 1] * (ep[i] - ep[i - 1]);
    for (int i = 0; i < E - 1; i++) {
      T tb = ep[i], te = ep[i + 1], rate = rates[i], inv = T(1) / rates[i];
      if (rate > 0 && te != 0 && te - tb > 0) {
        Aep[i] = lme<T>(-cs[i], -cs[i + 1]);
        T b = (tb + inv) - (te + inv) * xexp<T>(-cs[i + 1] + cs[i]);
        Bep[i] = xlog<T>(b) - cs[i];
      } else
        Aep[i] = Bep[i] = L0;
    }
    int i = E - 1;
    if (rates[i] > 0) {
      Aep[i] = -cs[i];
      Bep[i] = xlog<T>(ep[i] + T(1) / rates[i]) - cs[i];
    } else
      Aep[i] = Bep[i] = L0;
  }
  void run(const std::vector<double>& epochs, const std::vector<double>& r, const std::vector<double>& grid,
           const double* csh, const double* cns, std::vector<std::vector<double>>* integ_trace = nullptr) {
    E = (int)epochs.size(), A = (int)grid.size();
    ep.assign(epochs.begin(), epochs.end());
    rates.assign(r.begin(), r.end());
    Aep.assign(E, 0), Bep.assign(E, 0), num.assign(E, 0), den.assign(E, 0), N.assign(E, 0), D.assign(E, 0), Dinteg.assign(E, 0);
    ll = 0;
    get_AB();
    const T L0 = -INFINITY;
    for (int bin = 0; bin < A; bin++)
      for (int kind = 0; kind < 2; kind++) {
        double count = kind == 0 ? csh[bin] : cns[bin];
        if (!(count > 0)) continue;
        T age = grid[bin];
        int k = E - 1;
        for (int e = 0; e < E; e++)
          if (age < ep[e]) { k = e - 1; break; }
        T c = 0;
        for (int i = 1; i <= k; i++) c = c + rates[i - 1] * (ep[i] - ep[i - 1]);
        T ck = c, ck1 = ck + rates[k] * (age - ep[k]), ck2 = ck1 + rates[k] * (age - age);
        T ck3 = (k < E - 1) ? ck2 + rates[k] * (ep[k + 1] - age) : T(0);
        for (int e = 0; e < E; e++) num[e] = den[e] = 0;
        T nc;
        std::vector<double> tr(E, 0.0);
        chains++;
        if (kind == 0) {
          nc = 1.0;
          for (int e = 0; e <= k; e++) {
            if (e < k) { num[e] = Aep[e]; den[e] = Bep[e]; }
            else {
              T inv = T(1) / rates[k];
              if (rates[k] > 0) {
                num[e] = lme<T>(-ck, -ck1);
                den[e] = xlog<T>((ep[k] + inv) / inv - (age + inv) / inv * xexp<T>(-ck1 + ck)) + xlog<T>(inv) - ck;
              } else num[e] = den[e] = L0;
            }
            if (nc == T(1)) nc = num[e]; else nc = lse<T>(nc, num[e]);
          }
          if (!absent(nc)) {
            T integ = 1;
            int lim = (E - 1 < k + 1) ? E - 1 : k + 1;
            bool clamped = false;
            for (int e = 0; e < lim; e++) {
              num[e] = xexp<T>(num[e] - nc);
              if (integ > 0) integ -= num[e]; else { integ = 0; clamped = true; }
              tr[e] = (double)integ;
              den[e] = xexp<T>(den[e] - nc);
              den[e] += -ep[e] * num[e] + (ep[e + 1] - ep[e]) * integ;
              Dinteg[e] += T(count) * ((ep[e + 1] - ep[e]) * integ);
              if (den[e] < 0) den[e] = 0;
            }
            if (clamped) clamp_hits++;
            if (k == E - 1) {
              int e = E - 1;
              num[e] = xexp<T>(num[e] - nc);
              den[e] = xexp<T>(den[e] - nc);
              den[e] -= ep[e] * num[e];
              if (den[e] < 0) den[e] = 0;
            }
          } else { nc = 0; for (int e = 0; e < E; e++) num[e] = den[e] = 0; }
        } else {
          T rate = rates[k], inv = T(1) / rates[k];
          if (k != E - 1) {
            if (rate > 0) {
              num[k] = lme<T>(-ck2, -ck3);
              den[k] = xlog<T>((age + inv) - (ep[k + 1] + inv) * xexp<T>(-ck3 + ck2)) - ck2;
              nc = num[k];
            } else num[k] = den[k] = nc = L0;
            for (int e = k + 1; e < E; e++) { num[e] = Aep[e]; den[e] = Bep[e]; nc = lse<T>(nc, num[e]); }
          } else {
            num[k] = -ck2; den[k] = xlog<T>(age + inv) - ck2; nc = num[k];
          }
          if (!absent(nc)) {
            T integ = 1;
            int e;
            bool clamped = false;
            for (e = 0; e < k; e++) { num[e] = 0; den[e] = ep[e + 1] - ep[e]; }
            for (; e < E - 1; e++) {
              num[e] = xexp<T>(num[e] - nc);
              if (integ > 0) integ -= num[e]; else { integ = 0; clamped = true; }
              tr[e] = (double)integ;
              den[e] = xexp<T>(den[e] - nc);
              den[e] += -ep[e] * num[e] + (ep[e + 1] - ep[e]) * integ;
              Dinteg[e] += T(count) * ((ep[e + 1] - ep[e]) * integ);
              if (den[e] < 0) den[e] = 0;
            }
            if (clamped) clamp_hits++;
            e = E - 1;
            num[e] = xexp<T>(num[e] - nc);
            den[e] = xexp<T>(den[e] - nc);
            den[e] -= ep[e] * num[e];
            if (den[e] < 0) den[e] = 0;
          } else { nc = 0; for (int e = 0; e < E; e++) num[e] = den[e] = 0; }
        }
        ll += T(count) * nc;
        for (int e = 0; e < E; e++) { N[e] += T(count) * num[e]; D[e] += T(count) * den[e]; }
        if (integ_trace) integ_trace->push_back(tr);
      }
  }
};

int main(int argc, char** argv) {
  // table.bin: int32 E, A; double epochs[E], grid[A], csh[A], cns[A]
  FILE* f = fopen(argv[1], "rb");
  int E, A;
  if (!f || fread(&E, 4, 1, f) != 1 || fread(&A, 4, 1, f) != 1) return 1;
  std::vector<double> ep(E), grid(A), csh(A), cns(A);
  if (fread(ep.data(), 8, E, f) != (size_t)E || fread(grid.data(), 8, A, f) != (size_t)A || fread(csh.data(), 8, A, f) != (size_t)A ||
      fread(cns.data(), 8, A, f) != (size_t)A) return 1;
  fclose(f);
  const int e_lo = argc > 2 ? atoi(argv[2]) : E - 24;
  const int dump_iter = argc > 3 ? atoi(argv[3]) : -1;
  std::vector<double> rates(E, 1.0 / 20000.0);
  EStep<double> d;
  EStep<long double> q;
  std::vector<double> sR(E, 0), sR2(E, 0), mn(E, 1e300), mx(E, -1e300), sX(E, 0), sRN(E, 0), sRN2(E, 0);
  double ll = -INFINITY, prev;
  int iter, nacc = 0;
  for (iter = 0; iter < 100000; iter++) {
    prev = ll;
    std::vector<std::vector<double>> tr_d;
    d.run(ep, rates, grid, csh.data(), cns.data(), iter == dump_iter ? &tr_d : nullptr);
    if (iter >= 50) {  // (skip the first iterations: rates far from anywhere)
      q.run(ep, rates, grid, csh.data(), cns.data());
      for (int e = e_lo; e < E; e++) {
        const double R = (double)((long double)d.D[e] - q.D[e]);
        const double rel = R / (double)q.D[e];
        sR[e] += rel, sR2[e] += rel * rel;
        if (rel < mn[e]) mn[e] = rel;
        if (rel > mx[e]) mx[e] = rel;
        sX[e] += (double)(q.Dinteg[e] / q.D[e]);
        const double relN = (double)(((long double)d.N[e] - q.N[e]) / q.N[e]);
        sRN[e] += relN, sRN2[e] += relN * relN;
      }
      nacc++;
    }
    if (iter == dump_iter) {
      std::vector<std::vector<double>> tr_q;
      EStep<long double> q2;
      q2.run(ep, rates, grid, csh.data(), cns.data(), &tr_q);
      printf("# iteration %d: per chain (bin-kind order), integ(double) - integ(long double) at epochs %d..%d, and integ(long double)\n", iter, e_lo, E - 2);
      for (size_t c = 0; c < tr_d.size(); c++) {
        printf("chain %3zu:", c);
        for (int e = e_lo; e < E - 1; e += 2) printf(" %9.2e/%8.1e", tr_d[c][e] - tr_q[c][e], tr_q[c][e]);
        printf("\n");
      }
    }
    ll = (double)d.ll;
    // M-step
    for (int e = 0; e < E; e++) {
      if (d.N[e] == 0) rates[e] = e > 0 ? rates[e - 1] : 0.0;
      else if (d.D[e] == 0) {}
      else { rates[e] = d.N[e] / d.D[e]; if (rates[e] < 5e-9) rates[e] = 5e-9; }
    }
    if ((ll / prev > 1.0 - 1e-7) & (iter > 1000)) break;
  }
  printf("iterations %d, chains %ld, chains that clamp to 0 per E-step %.1f\n", iter, d.chains / (iter + 1), (double)d.clamp_hits / (iter + 1));
  printf("%4s %12s %12s %12s %12s %12s %12s %12s %12s\n", "e", "rate", "mean R/D", "std R/D", "min", "max", "Dinteg/D", "mean RN/N", "std RN/N");
  for (int e = e_lo; e < E; e++) {
    const double m = sR[e] / nacc, s = std::sqrt(std::fmax(sR2[e] / nacc - m * m, 0.0));
    const double mN = sRN[e] / nacc, sN = std::sqrt(std::fmax(sRN2[e] / nacc - mN * mN, 0.0));
    printf("%4d %12.6g %12.3e %12.3e %12.3e %12.3e %12.3e %12.3e %12.3e\n", e, rates[e], m, s, mn[e], mx[e], sX[e] / nacc, mN, sN);
  }
  return 0;
}
