// tools/study/residue_models.cpp -- RESEARCH TOOL (CPU only): candidate models of the rounding behaviour of the
// reference's `integ` recurrence, tried on the FACTORISED E-step (the algebra of the HIP kernel, tests/factorized_model.py)
// and compared with the reference-order E-step run in double (= the oracle's arithmetic) on the same table.
//   g++ -O2 -std=c++17 -ffp-contract=off tools/study/residue_models.cpp -o /tmp/residue_models
//   /tmp/residue_models table.bin <first epoch to print>
// Models (not-shared bins; shared bins keep exact sums):
//   0  exact sums, no residue
//   1  round 2's kernel: + dt_e * 4e-17 * (all counts)
//   2  per (bin, epoch): integ = s_b H((x_be - D_b)/s_b), H(z) = phi(z) + z Phi(z); D_b = mass of the terms the
//      reference's log-sum-exp fold absorbs (term / running sum < ulp(Z_b)/2), s_b^2 = chain + fold rounding variance
//   3  per (bin, epoch): integ = max(0, x_be - D_b)  (no noise part)
//   4  per epoch, aggregated noise scale (an early candidate)
//   5  round 3's kernel: max(x_be - D_b, 0.4 s_b) through cut sets, D_b / s_b from the cheap per-bin estimates, refreshed sparsely
//   6  round 4's kernel: model 2's form s_b H((x_be - D_b) / s_b) with model 5's per-bin D_b, s_b and refresh schedule
//   7  as 6 with H replaced by a quadratic blend
//   8  as 6, the held correction linearised in S_{e+1}: R_e = R0_e + dR/dS (S_{e+1} - S0_e) between refreshes
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

static double Hfun(double z) {  // E[max(0, z + g)], g ~ N(0,1)
  if (z > 8) return z;
  if (z < -8) return 0.0;
  return std::exp(-0.5 * z * z) / std::sqrt(2 * M_PI) + z * 0.5 * std::erfc(-z / std::sqrt(2.0));
}
static double dbg_cD, dbg_cth, dbg_cs, dbg_csc;
static double KD = 0.52, KS = 2.6, S0 = 0; static int HSMOOTH = 0;
static double half_ulp(double z) {  // half the spacing of doubles at |z|
  z = std::fabs(z);
  if (z == 0) return 0;
  int ex;
  std::frexp(z, &ex);  // z = m 2^ex, m in [0.5, 1)
  return std::ldexp(1.0, ex - 1 - 53);
}

struct M5Bin { int b, k; double c, cs_a, xk; };
static int REFRESH = 32, EARLY = 0, POW2 = 0, SHRES = 0, SKIP12 = 0;
struct Fact {
  int E, A, model, iter_no = 0;
  std::vector<M5Bin> m5_bins;
  std::vector<double> R5, R5_1, R5_S0;
  std::vector<double> t, grid, csh, cns;
  std::vector<int> kb;
  std::vector<double> N, D;
  double ll;
  void estep(const std::vector<double>& lam) {
    std::vector<double> dt(E, 0), x(E, 0), cs(E + 1, 0), inv(E), q(E), S(E), p(E), beta(E), W(E), V(E), VW(E), PW(E + 1, 0), G(E + 1, 0), Xa(E);
    for (int e = 0; e < E - 1; e++) dt[e] = t[e + 1] - t[e];
    for (int e = 0; e < E; e++) x[e] = lam[e] * dt[e];
    for (int e = 0; e < E - 1; e++) cs[e + 1] = cs[e] + x[e];
    for (int e = 0; e < E; e++) {
      inv[e] = 1.0 / lam[e];
      q[e] = std::exp(-cs[e + 1] + cs[e]);
      S[e] = std::exp(-cs[e]);
      bool valid = e < E - 1 ? (lam[e] > 0 && t[e + 1] != 0 && dt[e] > 0) : lam[e] > 0;
      double tn = e < E - 1 ? t[e + 1] : 0.0;
      p[e] = valid ? 1.0 - q[e] : 0.0;
      beta[e] = valid ? (t[e] + inv[e]) - (tn + inv[e]) * q[e] : 0.0;
      if (e == E - 1) { p[e] = valid ? 1.0 : 0.0; beta[e] = valid ? t[e] + inv[e] : 0.0; q[e] = 0.0; }
      W[e] = S[e] * p[e]; V[e] = S[e] * beta[e]; VW[e] = V[e] - t[e] * W[e];
      Xa[e] = (t[e] + inv[e]) / inv[e];
    }
    for (int e = 0; e < E; e++) PW[e + 1] = PW[e] + W[e];
    G[E - 1] = p[E - 1];
    for (int e = E - 2; e >= 0; e--) G[e] = p[e] + q[e] * G[e + 1];
    std::vector<double> g(E, 0), gc(E, 0), gW(E, 0), gV(E, 0), h(E, 0), hc(E, 0), hN(E, 0), hD(E, 0);
    std::vector<double> Icorr(E, 0), ThetaAt(E, 0);  // ThetaAt[k]: sum over the not-shared bins of epoch k of c_b * half_ulp(Z_b)  // per-epoch correction to the not-shared integ sums (models 2, 3), in counts x mass
    ll = 0;
    double call = 0;
    for (int b = 0; b < A; b++) call += (csh[b] > 0 ? csh[b] : 0) + (cns[b] > 0 ? cns[b] : 0);
    for (int b = 0; b < A; b++) {
      int k = kb[b];
      double a = grid[b], lk = lam[k], ck = cs[k], ck1 = ck + lk * (a - t[k]);
      if (csh[b] > 0) {
        double c = csh[b], Wp = 0, Vp = 0;
        if (lk > 0) {
          double qd = std::exp(-ck1 + ck);
          Wp = S[k] * (1.0 - qd);
          double X = Xa[k] - (a + inv[k]) / inv[k] * qd;
          Vp = X * inv[k] * S[k];
        }
        double Sig = PW[k] + Wp;
        if (Sig > 0 && std::isfinite(Sig)) {
          double r = 1.0 / Sig;
          ll += c * std::log(Sig);
          double nk = Wp * r, dk = Vp * r - t[k] * nk;
          g[k] += c * r; gc[k] += c; gW[k] += c * nk; gV[k] += c * std::fmax(dk, 0.0);
        }
      }
      if (cns[b] > 0) {
        double c = cns[b], ck2 = ck1 + lk * (a - a);
        if (k < E - 1) {
          double u = 1, pn = 0, bn = 0;
          if (lk > 0) {
            double ck3 = ck2 + lk * (t[k + 1] - a);
            u = std::exp(-ck3 + ck2); pn = 1.0 - u; bn = (a + inv[k]) - (t[k + 1] + inv[k]) * u;
          }
          double Sig = pn + u * G[k + 1];
          if (Sig > 0 && std::isfinite(Sig)) {
            double rr = 1.0 / Sig;
            ll += c * (-ck2 + std::log(Sig));
            double nk = pn * rr, dk = bn * rr - t[k] * nk + dt[k] * (1.0 - nk);
            h[k] += c * (u * rr); hc[k] += c; hN[k] += c * nk; hD[k] += c * std::fmax(dk, 0.0);
            if (model == 4) ThetaAt[k] += c * half_ulp(-ck2 + std::log(Sig));
            if (model >= 5) m5_bins.push_back({b, k, c, ck2, lk * dt[k]});
            if (model == 2 || model == 3) {
              // the reference's fold for this bin: terms n_k = pn, n_e = xprev * p_e (e > k), running sum -> 1; a term is
              // absorbed when term / running sum < ulp(nc)/2 with nc = log(running sum) - cs(a)
              const double Zfin = -ck2 + std::log(Sig);
              double run = nk, xprev = u * rr, Db = 0;
              int nfold = 0, nchain = 0;
              std::vector<double> xe(E, 0.0);
              xe[k] = xprev;
              for (int e = k + 1; e < E; e++) {
                double term = xprev * p[e];
                double ncur = Zfin + std::log(run > 0 ? run : 1e-300);  // the fold's running value
                if (run <= 0 || term / run < half_ulp(ncur)) Db += term; else { run += term; nfold++; }
                xprev = xprev * q[e];
                xe[e] = xprev;
              }
              // chain noise: each `integ -= n_e` rounds at the spacing of integ's binade
              double sc2 = 0;
              {
                double xp = u * rr;
                sc2 += std::pow(half_ulp(xp), 2) / 3.0;
                for (int e = k + 1; e < E - 1; e++) {
                  xp *= q[e];
                  sc2 += std::pow(half_ulp(xp > 1e-300 ? xp : 1e-300), 2) / 3.0;
                  nchain++;
                  if (xp < 1e-30) break;
                }
              }
              const double th = half_ulp(Zfin);
              const double sb = std::sqrt(sc2 + th * th * nfold / 3.0);
              dbg_cD += c * Db, dbg_cth += c * th, dbg_cs += c * sb, dbg_csc += c * std::sqrt(sc2);
              for (int e = k + 1; e < E - 1; e++) {  // epochs after the bin's own (the own-epoch integ is in hD)
                double want;
                if (model == 2) want = sb > 0 ? sb * Hfun((xe[e] - Db) / sb) : std::fmax(xe[e] - Db, 0.0);
                else want = std::fmax(xe[e] - Db, 0.0);
                Icorr[e] += c * (want - xe[e]);
              }
            }
          }
        } else {
          ll += c * (-ck2); hc[k] += c; hN[k] += c; hD[k] += c * std::fmax((a + inv[k]) - t[k], 0.0);
        }
      }
    }
    // ---- model 5 = the algorithm of the HIP kernel (refresh every REFRESH iterations, R5 held in between)
    if (model >= 5 && !(SKIP12 && (iter_no == 1 || iter_no == 2)) && (POW2 ? (iter_no < EARLY || ((iter_no & (iter_no - 1)) == 0) || (iter_no % REFRESH) == 0) : ((iter_no % REFRESH) == 0 || iter_no < EARLY))) {
      int k_old = -1;
      for (auto& bn : m5_bins) if (bn.k > k_old) k_old = bn.k;
      // epochs whose fold term can be absorbed by some bin / whose survival can be below some bin's threshold
      const int nb = (int)m5_bins.size();
      std::vector<double> tau(nb), cq(nb), cm(nb), Dbv(nb), sbv(nb), mv(nb);
      double PDtot = 0;
      for (int i = 0; i < nb; i++) {
        const auto& bn = m5_bins[i];
        const double th = half_ulp(-bn.cs_a);
        double m = std::exp(bn.cs_a);
        if (!(m < 1e100)) m = 1e100;
        double Db = 0;
        int ndrop = 0;
        for (int e = bn.k + 1; e < E; e++) {
          const double w = W[e] * m;
          if (w < th) Db += w, ndrop++;
        }
        const int nfold = E - 1 - bn.k - ndrop;
        const double nhalf = (bn.xk > 0 ? std::fmin(0.69 / bn.xk, (double)(E - bn.k)) : (double)(E - bn.k)) + 1.5;
        const double sc2 = std::ldexp(1.0, -108) / 3.0 * nhalf;
        const double rho = 0.4 * std::sqrt(sc2 + th * th * nfold / 3.0);
        tau[i] = (Db + rho) / m;
        Dbv[i] = Db, sbv[i] = rho / 0.4, mv[i] = m;
        cq[i] = bn.c * (rho + Db), cm[i] = bn.c * m;
        PDtot += bn.c * Db;
      }
      for (int i = nb - 2; i >= 0; i--) tau[i] = std::fmax(tau[i], tau[i + 1]);  // suffix max: the cut set is a prefix
      std::vector<double> PQ(nb + 1, 0), PM(nb + 1, 0);
      for (int i = 0; i < nb; i++) PQ[i + 1] = PQ[i] + cq[i], PM[i + 1] = PM[i] + cm[i];
      R5.assign(E, 0.0), R5_1.assign(E, 0.0), R5_S0.assign(E, 0.0);
      for (int e = 0; e < E - 1; e++) {
        int lo_e = 0;  // bins of earlier epochs
        while (lo_e < nb && m5_bins[lo_e].k < e) lo_e++;
        int bs = 0;
        while (bs < nb && tau[bs] > S[e + 1]) bs++;
        if (bs > lo_e) bs = lo_e;
        R5[e] = PQ[bs] - S[e + 1] * PM[bs] - (e > k_old ? PDtot : 0.0);
        if (model >= 6) {  // per (bin, epoch): c_b (s_b H((x_be - D_b) / s_b) - x_be), D_b and s_b as the kernel computes them
          double r = 0, r1 = 0;
          for (int i = 0; i < lo_e; i++) {
            const double x = S[e + 1] * mv[i], z = (x - Dbv[i]) / sbv[i];
            const double Hq = z >= 2 ? z : (z <= -2 ? 0.0 : (z + 2) * (z + 2) / 8);
            r += m5_bins[i].c * (sbv[i] * (model != 7 ? Hfun(z) : Hq) - x);
            // d/dS of c s (H(z) - x / s) = c m (Phi(z) - 1) = -c m Q(z)
            r1 -= m5_bins[i].c * mv[i] * 0.5 * std::erfc(z / std::sqrt(2.0));
          }
          R5[e] = r;
          R5_1[e] = r1, R5_S0[e] = S[e + 1];
        }
      }
    }
    m5_bins.clear();
    std::vector<double> RS(E + 1, 0), CS(E + 1, 0), CN(E + 1, 0), T(E + 1, 0);
    for (int e = E - 1; e >= 0; e--) { RS[e] = RS[e + 1] + g[e]; CS[e] = CS[e + 1] + gc[e]; CN[e] = CN[e + 1] + hc[e]; }
    for (int e = 0; e < E - 1; e++) T[e + 1] = q[e] * T[e] + h[e];
    N.assign(E, 0); D.assign(E, 0);
    for (int e = 0; e < E; e++) {
      double rs = RS[e + 1], cs_ = CS[e + 1], cn = CN[e + 1];
      N[e] = W[e] * rs + gW[e] + p[e] * T[e] + hN[e];
      if (e < E - 1) {
        double integ_ns = G[e + 1] * (q[e] * T[e]);
        if (model == 2 || model == 3) integ_ns = std::fmax(integ_ns + Icorr[e], 0.0);
        if (model >= 5) integ_ns = std::fmax(integ_ns + R5[e] + (model == 8 ? R5_1[e] * (S[e + 1] - R5_S0[e]) : 0.0), 0.0);
        if (model == 4) {
          double Th = 0, Cp = 0;
          for (int j = 0; j < e; j++) Th += ThetaAt[j], Cp += hc[j];  // bins of earlier epochs
          if (Cp > 0) {
            // chain noise of `integ -= num[e]`: ~E/6 steps rounded at 2^-54 each (integ in [0.5, 1)), plus the fold's roundings
            const double s0 = S0 > 0 ? S0 : std::ldexp(1.0, -54) * std::sqrt(E / 6.0);
            const double sg = s0 * Cp + KS * Th, z = (integ_ns - KD * Th) / sg;
            const double Hq = z >= 2 ? z : (z <= -2 ? 0.0 : (z + 2) * (z + 2) / 8);
            integ_ns = sg * (HSMOOTH ? Hfun(z) : Hq);
          }
        }
        D[e] = (std::fmax(VW[e] * rs + dt[e] * std::fmax(cs_ - PW[e + 1] * rs, 0.0), 0.0) + gV[e]) +
               (std::fmax(dt[e] * cn + ((beta[e] - t[e] * p[e]) * T[e] + dt[e] * integ_ns), 0.0) + hD[e]);
        if (model == 1) D[e] += dt[e] * (4.0e-17 * call);
        // the shared kind's mean residue as the kernel adds it: SHRES=1 round 3 (every epoch, all shared counts), SHRES=2 round 4
        // (only the shared bins whose own chain reaches epoch e: k_b >= e -- coal_EM.cpp:266-278 stops at the bin's epoch)
        if (model >= 5 && SHRES == 1) D[e] += dt[e] * (4.0e-17 * CS[0]);
        if (model >= 5 && SHRES == 2) D[e] += dt[e] * (4.0e-17 * CS[e]);
      } else
        D[e] = gV[e] + (beta[e] - t[e] * p[e]) * T[e] + hD[e];
    }
  }
  int em(std::vector<double>& rates) {
    double l = -INFINITY, prev;
    int iter;
    R5.assign(E, 0.0), R5_1.assign(E, 0.0), R5_S0.assign(E, 0.0);
    for (iter = 0; iter < 100000; iter++) {
      prev = l;
      iter_no = iter;
      estep(rates);
      l = ll;
      for (int e = 0; e < E; e++) {
        if (N[e] == 0) rates[e] = e > 0 ? rates[e - 1] : 0.0;
        else if (D[e] == 0) {}
        else { rates[e] = N[e] / D[e]; if (rates[e] < 5e-9) rates[e] = 5e-9; }
      }
      if ((l / prev > 1.0 - 1e-7) & (iter > 1000)) break;
    }
    return iter;
  }
};

int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb");
  int E, A;
  if (!f || fread(&E, 4, 1, f) != 1 || fread(&A, 4, 1, f) != 1) return 1;
  Fact F;
  F.E = E, F.A = A;
  F.t.resize(E), F.grid.resize(A), F.csh.resize(A), F.cns.resize(A);
  if (fread(F.t.data(), 8, E, f) != (size_t)E || fread(F.grid.data(), 8, A, f) != (size_t)A || fread(F.csh.data(), 8, A, f) != (size_t)A ||
      fread(F.cns.data(), 8, A, f) != (size_t)A) return 1;
  fclose(f);
  F.kb.resize(A);
  for (int b = 0; b < A; b++) {
    int k = E - 1;
    for (int e = 0; e < E; e++) if (F.grid[b] < F.t[e]) { k = e - 1; break; }
    F.kb[b] = k < 0 ? 0 : k;
  }
  const int e_lo = argc > 2 ? atoi(argv[2]) : E - 24;
  std::vector<std::vector<double>> res;
  std::vector<int> its;
  if (argc > 3) KD = atof(argv[3]);
  if (argc > 4) KS = atof(argv[4]);
  if (argc > 5) HSMOOTH = atoi(argv[5]);
  if (argc > 6) S0 = atof(argv[6]);
  if (getenv("REFRESH")) REFRESH = atoi(getenv("REFRESH"));
  if (getenv("EARLY")) EARLY = atoi(getenv("EARLY"));
  if (getenv("POW2")) POW2 = atoi(getenv("POW2"));
  if (getenv("SHRES")) SHRES = atoi(getenv("SHRES"));
  if (getenv("SKIP12")) SKIP12 = atoi(getenv("SKIP12"));
  for (int m = 0; m <= 8; m++) {
    F.model = m;
    std::vector<double> rates(E, 1.0 / 20000.0);
    its.push_back(F.em(rates));
    if ((m == 2 || m == 5) && getenv("DBG5")) { F.estep(rates); unsetenv("DBG5"); if (m == 2) setenv("DBG5", "1", 1); }
    if (m == 2) { dbg_cD = dbg_cth = dbg_cs = dbg_csc = 0; F.estep(rates); printf("model 2 at the fixed point: sum c D_b / sum c theta_b = %.3f, sum c s_b / sum c theta_b = %.3f, sum c s_chain / sum c = %.3e, sum c theta / sum c = %.3e\n", dbg_cD / dbg_cth, dbg_cs / dbg_cth, dbg_csc, dbg_cth); }
    res.push_back(rates);
  }
  printf("iterations: %d %d %d %d %d %d %d %d %d\n", its[0], its[1], its[2], its[3], its[4], its[5], its[6], its[7], its[8]);
  for (int e = e_lo; e < E; e++) printf("%d %.10g %.10g %.10g %.10g %.10g %.10g %.10g %.10g %.10g\n", e, res[0][e], res[1][e], res[2][e], res[3][e], res[4][e], res[5][e], res[6][e], res[7][e], res[8][e]);
  return 0;
}
