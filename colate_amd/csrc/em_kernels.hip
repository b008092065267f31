// colate_amd/csrc/em_kernels.hip -- the EM hot path of `Colate --mode mut` on gfx950.
//
// Replaces, for B bootstrap replicates at once, the reference's
//   bootstrap EM driver            include/coal/coal.cpp:3675-3827
//   coal_EM ctor / get_AB          include/coal/coal_EM.hpp:38-50, coal_EM.cpp:97-151
//   coal_EM::EM_shared/_notshared  include/coal/coal_EM.cpp:153-295, 297-468 (age_begin == age_end)
//   E-step accumulation            include/coal/coal.cpp:3704-3733
//   M-step, floor, stop rule       include/coal/coal.cpp:3771-3815, 3822-3825
//
// One workgroup (256 threads, one wave per SIMD of a CU) owns one replicate and
// runs all of its EM iterations inside one launch: counts, age grid, epochs and
// the current rates never leave the CU (registers + LDS), so HBM sees each
// replicate's 2*A counts once on the way in and E rates on the way out.
//
// The reference evaluates exp(log-term - Z) for every (age bin, epoch) pair:
// O(A*E) transcendentals per iteration.  Here every such term is factored into
// a per-epoch piece times a per-bin piece, so that the sufficient
// statistics N_e = sum_b c_b num_e(b), D_e = sum_b c_b denom_e(b) and
// ll = sum_b c_b Z_b need O(A + E) transcendentals.  All formulas that the
// reference evaluates with catastrophic cancellation are kept operand for
// operand (no fused multiply-add: this file is built with -ffp-contract=off and
// uses fma only inside em_math.hpp and in recurrences that have no counterpart
// in the reference).
//
// Phases of one iteration (threads change role between barriers):
//   P1  epoch e : cs_e (sequential sum, as coal_EM.cpp:100-103), q_e, S_e, p_e, beta_e, W_e, V_e
//   P2  epoch e : PW_e = sum_{j<e} W_j,  G_e = p_e + q_e G_{e+1}
//   P3  bin b   : shared / not-shared bin terms  -> 8 per-bin values, ll partial
//   P4  (e,arr) : per-epoch sums of the 8 per-bin values over the bins inside epoch e
//   P5  epoch e : suffix sums over later epochs, forward recurrence T, N_e, D_e, M-step candidate
//   P6  epoch e : resolve "num == 0 -> copy previous rate" chain, stop rule
#include <hip/hip_runtime.h>

#include "em_kernels.h"
#include "em_math.hpp"

namespace {

constexpr int kThreads = COLATE_EM_THREADS;
constexpr int kNumBinArrays = 8;
enum { O_G = 0, O_GC, O_GW, O_GV, O_H, O_HC, O_HN, O_HD };

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ bool finite_pos(double x) { return x > 0.0 && x < __builtin_inf(); }

template <int MODE>  // 0: EM to convergence, 1: one E-step (num/den/ll out)
__global__ __launch_bounds__(kThreads) void em_kernel(ColateEmArgs p) {
  extern __shared__ double lds[];
  const int E = p.E, A = p.A;
  const int EP = E + 1;
  const int AP = (A + 63) & ~63;
  // ---- LDS carve-up (doubles) ----
  double* s_t = lds;             // [EP] epoch starts
  double* s_x = s_t + EP;        // [EP] lambda_e * dt_e
  double* s_cs = s_x + EP;       // [EP] cumulative hazard at epoch start
  double* s_lam = s_cs + EP;     // [EP]
  double* s_inv = s_lam + EP;    // [EP] 1/lambda
  double* s_q = s_inv + EP;      // [EP] exp(-x_e)
  double* s_p = s_q + EP;        // [EP] 1-q (0 if epoch invalid); last epoch: 1 or 0
  double* s_S = s_p + EP;        // [EP] exp(-cs_e)
  double* s_W = s_S + EP;        // [EP] exp(A_ep) = S p
  double* s_Xa = s_W + EP;       // [EP] (t_e + inv)/inv
  double* s_PW = s_Xa + EP;      // [EP] prefix sums of W
  double* s_G = s_PW + EP;       // [EP] relative suffix mass, s_G[E] = 0
  double* s_cand = s_G + EP;     // [EP] M-step candidate rate
  double* s_gs = s_cand + EP;    // [8][EP] per-epoch sums of the per-bin values
  double* s_out = s_gs + kNumBinArrays * EP;  // [8][AP] per-bin values
  double* s_ll = s_out + kNumBinArrays * AP;  // [4] per-wave log-likelihood partials
  int* s_kb = reinterpret_cast<int*>(s_ll + 4);  // [AP] epoch of each bin
  int* s_lo = s_kb + AP;                         // [EP] first bin of epoch e (clipped)
  int* s_hi = s_lo + EP;                         // [EP] one past last bin of epoch e (clipped)
  int* s_copy = s_hi + EP;                       // [EP] M-step: num == 0 -> copy previous
  int* s_misc = s_copy + EP;                     // [4] nzlo, nzhi, flags

  const int tid = threadIdx.x;
  const int rep = blockIdx.x;
  const bool is_ep = tid < E;
  const bool is_bin = tid < A;

  // ------------------------------------------------------------------ prologue
  const double* epochs = p.epochs + (size_t)rep * p.epochs_stride;
  if (is_ep) s_t[tid] = epochs[tid];
  if (tid == 0) {
    s_t[E] = 0.0;
    s_misc[0] = A;
    s_misc[1] = 0;
    s_misc[2] = 0;
  }
  if (tid < 4) s_ll[tid] = 0.0;
  for (int i = tid; i < kNumBinArrays * AP; i += kThreads) s_out[i] = 0.0;
  __syncthreads();

  // epoch-role statics
  double t_e = 0, tn_e = 0, dt_e = 0, lam_e = 0;
  bool valid_static = false;
  if (is_ep) {
    t_e = s_t[tid];
    if (tid < E - 1) {
      tn_e = s_t[tid + 1];
      dt_e = tn_e - t_e;
      valid_static = (tn_e != 0) && (dt_e > 0);  // coal_EM.cpp:117
    } else {
      valid_static = true;
    }
    lam_e = p.rates_in[(size_t)rep * p.rates_stride + tid];
  }
  // bin-role statics
  double a_b = 0, csh = 0, cns = 0, tk = 0, tkn = 0, dtk = 0, da = 0, db = 0;
  int kb = 0;
  if (is_bin) {
    a_b = p.age_grid[tid];
    double c1 = p.cnt_sh[(size_t)rep * A + tid];
    double c2 = p.cnt_ns[(size_t)rep * A + tid];
    csh = (c1 > 0) ? c1 : 0.0;  // coal.cpp:3706, 3719: only counts > 0 are visited
    cns = (c2 > 0) ? c2 : 0.0;
    kb = E - 1;  // coal_EM.cpp:60-95: largest e with epochs[e] <= age (strict `age < epochs[e]`)
    for (int e = 0; e < E; e++) {
      if (a_b < s_t[e]) {
        kb = e - 1;
        break;
      }
    }
    if (kb < 0) kb = 0;  // host validates age >= epochs[0]; never taken
    tk = s_t[kb];
    if (kb < E - 1) {
      tkn = s_t[kb + 1];
      dtk = tkn - tk;
    }
    da = a_b - tk;
    db = tkn - a_b;
    s_kb[tid] = kb;
    if (csh > 0 || cns > 0) {
      atomicMin(&s_misc[0], tid);
      atomicMax(&s_misc[1], tid + 1);
    }
  }
  __syncthreads();
  if (is_ep) {
    int lo = A, hi = 0;
    for (int b = 0; b < A; b++) {
      if (s_kb[b] == tid) {
        if (b < lo) lo = b;
        hi = b + 1;
      }
    }
    const int nzlo = s_misc[0], nzhi = s_misc[1];
    if (lo < nzlo) lo = nzlo;
    if (hi > nzhi) hi = nzhi;
    s_lo[tid] = lo;
    s_hi[tid] = hi;
  }
  const bool bin_live = is_bin && (csh > 0 || cns > 0);
  int my_flags = 0;

  const double thr = 1.0 - p.rel_tol;
  double ll = -__builtin_inf(), prev_ll = -__builtin_inf();  // coal.cpp:3685
  int iter = 0;
  const int max_iter = (MODE == 1) ? 1 : p.max_iter;

  for (iter = 0; iter < max_iter; iter++) {
    const bool need_ll = (MODE == 1) || (iter >= p.min_iter) || (iter == max_iter - 1);
    // ---------------------------------------------------------------- P1
    if (is_ep) {
      s_x[tid] = lam_e * dt_e;
      s_lam[tid] = lam_e;
    }
    __syncthreads();
    double q_e = 0, p_e = 0, beta_e = 0, W_e = 0, VW_e = 0;
    if (is_ep) {
      double cs = 0.0;  // cs_e = ((x_0 + x_1) + ...) + x_{e-1}, in this order
      for (int j = 0; j < tid; j++) cs = cs + s_x[j];
      const double csn = cs + lam_e * dt_e;
      const double inv = 1.0 / lam_e;
      const double S = em::em_exp(-cs);
      const bool valid = valid_static && (lam_e > 0);
      if (tid < E - 1) {
        q_e = em::em_exp(-csn + cs);  // exp(-cumsum[i+1] + cumsum[i]), coal_EM.cpp:120
        if (valid) {
          p_e = 1.0 - q_e;                              // exp(A_ep + cs), coal_EM.cpp:119
          beta_e = (t_e + inv) - (tn_e + inv) * q_e;    // exp(B_ep + cs), coal_EM.cpp:120
        }
      } else if (valid) {  // last epoch, coal_EM.cpp:136-141
        p_e = 1.0;
        beta_e = t_e + inv;
      }
      W_e = S * p_e;
      VW_e = S * beta_e - t_e * W_e;
      s_cs[tid] = cs;
      s_inv[tid] = inv;
      s_q[tid] = q_e;
      s_p[tid] = p_e;
      s_S[tid] = S;
      s_W[tid] = W_e;
      s_Xa[tid] = (t_e + inv) / inv;  // coal_EM.cpp:204
    }
    __syncthreads();
    // ---------------------------------------------------------------- P2
    double PWn_e = 0, Gn_e = 0;
    if (is_ep) {
      double pw = 0.0;
      for (int j = 0; j < tid; j++) pw = pw + s_W[j];
      double g = 0.0;
      for (int j = E - 1; j > tid; j--) g = em::fma_(s_q[j], g, s_p[j]);
      Gn_e = g;                          // G_{e+1}
      s_PW[tid] = pw;                    // PW_e
      PWn_e = pw + W_e;                  // PW_{e+1}
      s_G[tid] = em::fma_(q_e, g, p_e);  // G_e
      if (tid == 0) s_G[E] = 0.0;
    }
    __syncthreads();
    // ---------------------------------------------------------------- P3
    {
      double llp = 0.0;
      if (bin_live) {
        double o_g = 0, o_gc = 0, o_gW = 0, o_gV = 0, o_h = 0, o_hc = 0, o_hN = 0, o_hD = 0;
        const double lk = s_lam[kb], ik = s_inv[kb], ck = s_cs[kb];
        const bool lpos = lk > 0;
        const double ck1 = ck + lk * da;  // coal_EM.cpp:178-181 at the merged grid
        if (csh > 0) {                    // ---- EM_shared, coal_EM.cpp:198-210, 263-287
          const double Sk = s_S[kb];
          const double qd = em::em_exp(-ck1 + ck);
          double Wp = 0.0, Vp = 0.0;
          if (lpos) {
            Wp = Sk * (1.0 - qd);
            const double X = s_Xa[kb] - (a_b + ik) / ik * qd;
            Vp = X * ik * Sk;
          }
          const double Sig = s_PW[kb] + Wp;
          if (finite_pos(Sig)) {
            const double r = 1.0 / Sig;
            const double nk = Wp * r;
            double dk = Vp * r + (-tk * nk);
            if (dk < 0.0) dk = 0.0;
            o_g = csh * r;
            o_gc = csh;
            o_gW = csh * nk;
            o_gV = csh * dk;
            if (need_ll) llp += csh * em::em_log(Sig);
          }
        }
        if (cns > 0) {  // ---- EM_notshared, coal_EM.cpp:330-357, 435-460
          const double ck2 = ck1 + lk * (a_b - a_b);
          if (kb < E - 1) {
            const double ck3 = ck2 + lk * db;
            const double u = em::em_exp(-ck3 + ck2);
            double pn = 0.0, bn = 0.0;
            if (lpos) {
              pn = 1.0 - u;
              bn = (a_b + ik) - (tkn + ik) * u;
            }
            const double Sig = pn + u * s_G[kb + 1];
            if (finite_pos(Sig)) {
              const double rr = 1.0 / Sig;
              const double nk = pn * rr;
              double dk = bn * rr + (-tk * nk + dtk * (1.0 - nk));
              if (dk < 0.0) dk = 0.0;
              o_h = cns * (u * rr);
              o_hc = cns;
              o_hN = cns * nk;
              o_hD = cns * dk;
              if (need_ll) llp += cns * (-ck2 + em::em_log(Sig));
            }
          } else {  // bin beyond the start of the last epoch, coal_EM.cpp:350-357
            if (!lpos) my_flags |= COLATE_FLAG_NAN;  // reference: assert(coal_rate_e > 0)
            double dk = (a_b + ik) - tk;
            if (dk < 0.0) dk = 0.0;
            o_hc = cns;
            o_hN = cns;
            o_hD = cns * dk;
            if (need_ll) llp += cns * (-ck2);
          }
        }
        s_out[O_G * AP + tid] = o_g;
        s_out[O_GC * AP + tid] = o_gc;
        s_out[O_GW * AP + tid] = o_gW;
        s_out[O_GV * AP + tid] = o_gV;
        s_out[O_H * AP + tid] = o_h;
        s_out[O_HC * AP + tid] = o_hc;
        s_out[O_HN * AP + tid] = o_hN;
        s_out[O_HD * AP + tid] = o_hD;
      }
      if (need_ll) {
        llp = wave_sum(llp);
        if ((tid & 63) == 0) s_ll[tid >> 6] = llp;
      }
    }
    __syncthreads();
    // ---------------------------------------------------------------- P4
    for (int idx = tid; idx < kNumBinArrays * E; idx += kThreads) {
      const int e = idx >> 3, arr = idx & 7;
      const int lo = s_lo[e], hi = s_hi[e];
      const double* src = s_out + arr * AP;
      double acc = 0.0;
      for (int b = lo; b < hi; b++) acc += src[b];
      s_gs[arr * EP + e] = acc;
    }
    __syncthreads();
    // ---------------------------------------------------------------- P5
    double N_e = 0, D_e = 0;
    if (is_ep) {
      double RS = 0, CS = 0, CN = 0;  // over bins in LATER epochs
      for (int j = E - 1; j > tid; j--) {
        RS += s_gs[O_G * EP + j];
        CS += s_gs[O_GC * EP + j];
        CN += s_gs[O_HC * EP + j];
      }
      double T = 0;  // T_e = sum_{b: k_b < e} c_b u_b/Sig_b * S_e/S_{k_b+1}
      for (int j = 0; j < tid; j++) T = em::fma_(s_q[j], T, s_gs[O_H * EP + j]);
      const double gW = s_gs[O_GW * EP + tid], gV = s_gs[O_GV * EP + tid];
      const double hN = s_gs[O_HN * EP + tid], hD = s_gs[O_HD * EP + tid];
      N_e = W_e * RS + gW + p_e * T + hN;
      if (tid < E - 1) {
        D_e = VW_e * RS + dt_e * (CS - PWn_e * RS) + gV + dt_e * CN + (beta_e - t_e * p_e) * T +
              dt_e * Gn_e * (q_e * T) + hD;
      } else {
        D_e = gV + (beta_e - t_e * p_e) * T + hD;
      }
      if (N_e != N_e || D_e != D_e) my_flags |= COLATE_FLAG_NAN;  // coal.cpp:3711-3712
      if (N_e < 0.0 || D_e < 0.0) my_flags |= COLATE_FLAG_NEG;    // coal.cpp:3713-3714
    }
    if (need_ll) ll = ((s_ll[0] + s_ll[1]) + s_ll[2]) + s_ll[3];
    if (MODE == 1) {
      if (is_ep) {
        p.out_num[(size_t)rep * E + tid] = N_e;
        p.out_den[(size_t)rep * E + tid] = D_e;
      }
      break;
    }
    // M-step candidate, coal.cpp:3777-3804
    if (is_ep) {
      const bool copy = (N_e == 0);
      double cand = lam_e;
      if (!copy && D_e != 0) {
        cand = N_e / D_e;
        if (cand < p.rate_floor) cand = p.rate_floor;
      }
      s_cand[tid] = cand;
      s_copy[tid] = copy ? 1 : 0;
    }
    __syncthreads();
    // ---------------------------------------------------------------- P6
    if (is_ep) {
      int j = tid;
      while (j >= 0 && s_copy[j]) j--;  // coal.cpp:3779-3786 (already-updated previous rate)
      lam_e = (j >= 0) ? s_cand[j] : 0.0;
    }
    // stop rule, coal.cpp:3822 (evaluated after the update); uniform across the workgroup
    const bool stop = (ll / prev_ll > thr) & (iter > p.min_iter);
    prev_ll = ll;
    if (stop) break;
  }

  // ------------------------------------------------------------------ epilogue
  if (MODE == 0 && is_ep) {
    if (lam_e != lam_e) my_flags |= COLATE_FLAG_NAN;
    p.out_rates[(size_t)rep * E + tid] = lam_e;
  }
  if (my_flags) atomicOr(&s_misc[2], my_flags);
  __syncthreads();
  if (tid == 0) {
    int fl = s_misc[2];
    if (MODE == 0) {
      if (iter >= p.max_iter) fl |= COLATE_FLAG_MAXITER;
      p.out_iters[rep] = iter < p.max_iter ? iter : p.max_iter;
    }
    p.out_ll[rep] = ll;
    p.out_flags[rep] = fl;
  }
}

}  // namespace

size_t colate_em_lds_bytes(int E, int A) {
  const size_t EP = (size_t)E + 1;
  const size_t AP = ((size_t)A + 63) & ~(size_t)63;
  size_t doubles = 13 * EP + kNumBinArrays * EP + kNumBinArrays * AP + 4;
  size_t ints = AP + 3 * EP + 4;
  return doubles * sizeof(double) + ints * sizeof(int);
}

hipError_t colate_em_launch(const ColateEmArgs& args, hipStream_t stream) {
  const size_t lds = colate_em_lds_bytes(args.E, args.A);
  dim3 grid(args.B), block(kThreads);
  if (args.mode == 1)
    hipLaunchKernelGGL(em_kernel<1>, grid, block, lds, stream, args);
  else
    hipLaunchKernelGGL(em_kernel<0>, grid, block, lds, stream, args);
  return hipGetLastError();
}
