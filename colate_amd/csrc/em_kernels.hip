// colate_amd/csrc/em_kernels.hip -- the EM hot path of `Colate --mode mut` on gfx950.
//
// Replaces, for B bootstrap replicates at once, the reference's
//   bootstrap EM driver            include/coal/coal.cpp:3675-3827
//   coal_EM ctor / get_AB          include/coal/coal_EM.hpp:38-50, coal_EM.cpp:97-151
//   coal_EM::EM_shared/_notshared  include/coal/coal_EM.cpp:153-295, 297-468 (age_begin == age_end)
//   E-step accumulation            include/coal/coal.cpp:3704-3733
//   M-step, floor, stop rule       include/coal/coal.cpp:3771-3815, 3822-3825
//
// One workgroup owns one replicate and runs all of its EM iterations inside one
// launch: counts, age grid, epochs and the current rates never leave the CU
// (registers + LDS), so HBM sees each replicate's 2*A counts once on the way in
// and E rates on the way out.  Each iteration is a short dependent chain, so
// the kernel is built for LATENCY (BASELINE configs put <= 256 replicates on a
// 256-CU GPU: one workgroup per CU):
//
//  * every wave of the workgroup carries the whole epoch state redundantly
//    (epoch e in lane e & 63, chunk e >> 6), so the epoch-level recurrences are
//    wave-local DPP scans and no barrier separates them from the bin phase;
//  * age bins are one per thread; their per-epoch sums are reduced in registers
//    (row-segmented DPP) and handed to the epoch lanes through a double-buffered
//    LDS tile: ONE workgroup barrier per EM iteration;
//  * sums the model makes telescope are not summed: sum_{j<e} exp(A_j) = 1 - exp(-cs_e),
//    and the not-shared normaliser is exp(-cs(age)) whenever the last epoch absorbs.
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
#include <hip/hip_runtime.h>

#include "em_kernels.h"
#include "em_math.hpp"

namespace {

constexpr int kWave = 64;
enum { O_G = 0, O_H, O_N, O_D, kNumBinArrays };                        // per-bin values -> epochs
enum { G_LAM = 0, G_INV, G_CS, G_S, G_XA, G_PW, kNumGather };           // per-epoch values -> bins

// ----------------------------------------------------------------- lane plumbing
__device__ __forceinline__ double readlane_d(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// DPP move of a double.  Lanes whose source is out of range, or whose row is
// not in ROW_MASK, keep `old` (BOUND == false) or read 0 (BOUND == true).
template <int CTRL, int ROW_MASK = 0xf, bool BOUND = false>
__device__ __forceinline__ double dpp_d(double old, double v) {
  int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, 0xf, BOUND);
  int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, 0xf, BOUND);
  return __hiloint2double(hi, lo);
}
constexpr int ROW_SHR1 = 0x111, ROW_SHR2 = 0x112, ROW_SHR4 = 0x114, ROW_SHR8 = 0x118;
constexpr int ROW_SHL1 = 0x101, ROW_SHL2 = 0x102, ROW_SHL4 = 0x104, ROW_SHL8 = 0x108;
constexpr int ROW_BCAST15 = 0x142, ROW_BCAST31 = 0x143, WAVE_SHR1 = 0x138, WAVE_SHL1 = 0x130;

// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ double wave_prefix_sum(double v) {
  v += dpp_d<ROW_SHR1>(0.0, v);
  v += dpp_d<ROW_SHR2>(0.0, v);
  v += dpp_d<ROW_SHR4>(0.0, v);
  v += dpp_d<ROW_SHR8>(0.0, v);
  v += dpp_d<ROW_BCAST15, 0xa>(0.0, v);
  v += dpp_d<ROW_BCAST31, 0xc>(0.0, v);
  return v;
}
// inclusive suffix sum over the 64 lanes (lane l: sum of lanes l..63)
__device__ __forceinline__ double wave_suffix_sum(double v, int lane) {
  v += dpp_d<ROW_SHL1>(0.0, v);
  v += dpp_d<ROW_SHL2>(0.0, v);
  v += dpp_d<ROW_SHL4>(0.0, v);
  v += dpp_d<ROW_SHL8>(0.0, v);
  const double r1 = readlane_d(v, 16), r2 = readlane_d(v, 32), r3 = readlane_d(v, 48);
  const double s23 = r2 + r3, s123 = r1 + s23;
  const int row = lane >> 4;
  const double add = row == 0 ? s123 : (row == 1 ? s23 : (row == 2 ? r3 : 0.0));
  return v + add;
}
// inclusive prefix composition of the affine maps x -> a*x + b (lane order = application order):
// afterwards (a, b) of lane l is f_l o ... o f_0
__device__ __forceinline__ void wave_affine_scan(double& a, double& b) {
#define COLATE_AFF_STEP(CTRL, RM)                  \
  {                                                \
    const double as = dpp_d<CTRL, RM>(1.0, a);     \
    const double bs = dpp_d<CTRL, RM>(0.0, b);     \
    b = em::fma_(a, bs, b);                        \
    a = a * as;                                    \
  }
  COLATE_AFF_STEP(ROW_SHR1, 0xf)
  COLATE_AFF_STEP(ROW_SHR2, 0xf)
  COLATE_AFF_STEP(ROW_SHR4, 0xf)
  COLATE_AFF_STEP(ROW_SHR8, 0xf)
  COLATE_AFF_STEP(ROW_BCAST15, 0xa)
  COLATE_AFF_STEP(ROW_BCAST31, 0xc)
#undef COLATE_AFF_STEP
}

#ifdef COLATE_EM_STAMPS
// diagnostic build only (tools/em_phase_probe.hip): cycle stamps around the phases of an iteration
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define COLATE_STAMP(i)                     \
  {                                         \
    const unsigned long long now_ = stamp(); \
    st_acc[i] += now_ - st_prev;            \
    st_prev = now_;                         \
  }
#else
#define COLATE_STAMP(i)
#endif

// Marks a rarely-taken branch body: a volatile asm cannot be executed speculatively, so the compiler
// keeps the branch instead of if-converting it (it otherwise evaluates whole exp()/log() calls of
// cold paths unconditionally and selects the result).
#define COLATE_COLD() asm volatile("; cold path")

__device__ __forceinline__ bool finite_pos(double x) { return x > 0.0 && x < __builtin_inf(); }

// wave-local LDS hand-off: earlier ds_writes of this wave are visible to its later ds_reads
// (the LDS queue is in order per wave); this only stops the compiler from moving them.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// MODE 0: EM to convergence, 1: one E-step (num/den/ll out).  NCH = epoch chunks of 64 per lane.
template <int MODE, int NCH>
__global__ __launch_bounds__(COLATE_EM_MAX_A) void em_kernel(ColateEmArgs p) {
  extern __shared__ double lds[];
  const int E = p.E, A = p.A;
  constexpr int EPAD = NCH * kWave;
  const int AP = blockDim.x;  // A rounded up to a multiple of 64
  const int APZ = AP + 16;    // stride of the per-bin tiles; entries [AP, APZ) stay zero
  const int nwaves = AP >> 6;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rep = blockIdx.x;

  // ---- LDS carve-up ----
  double* s_t = lds;                                     // [EPAD + 1] epoch starts
  double* s_ep = s_t + EPAD + 1;                         // [nwaves][kNumGather][EPAD] per-wave epoch values
  double* s_out = s_ep + nwaves * kNumGather * EPAD;     // [2][kNumBinArrays][APZ] per-bin tails
  double* s_cfail = s_out + 2 * kNumBinArrays * APZ;     // [2][2][APZ] counts of bins whose normaliser failed
  double* s_cnt = s_cfail + 4 * APZ;                     // [2][APZ] counts (prologue only)
  double* s_ll = s_cnt + 2 * APZ;                        // [2][4] per-wave log-likelihood partials
  int* s_kb = reinterpret_cast<int*>(s_ll + 8);          // [AP + 1] epoch of each bin
  int* s_fail = s_kb + AP + 1;                           // [2][4] per-wave "a bin failed" flags
  int* s_misc = s_fail + 8;                              // [4] nzlo, nzhi, flags
  double* my_ep = s_ep + wave * kNumGather * EPAD;

  // ------------------------------------------------------------------ prologue
  const double* epochs = p.epochs + (size_t)rep * p.epochs_stride;
  for (int i = tid; i < EPAD + 1; i += AP) s_t[i] = (i < E) ? epochs[i] : 0.0;
  if (tid == 0) {
    s_misc[0] = A;
    s_misc[1] = 0;
    s_misc[2] = 0;
  }
  if (tid < 8) {
    s_ll[tid] = 0.0;
    s_fail[tid] = 0;
  }
  for (int i = tid; i < 2 * kNumBinArrays * APZ + 6 * APZ; i += AP) s_out[i] = 0.0;  // s_out, s_cfail, s_cnt
  __syncthreads();

  // epoch-role statics (identical in every wave)
  double t_e[NCH], tn_e[NCH], dt_e[NCH], lam_e[NCH];
  bool vstat[NCH], ep_on[NCH];
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const int e = c * kWave + lane;
    ep_on[c] = e < E;
    t_e[c] = s_t[e];
    tn_e[c] = 0.0;
    dt_e[c] = 0.0;
    vstat[c] = false;
    lam_e[c] = 0.0;
    if (ep_on[c]) {
      if (e < E - 1) {
        tn_e[c] = s_t[e + 1];
        dt_e[c] = tn_e[c] - t_e[c];
        vstat[c] = (tn_e[c] != 0) && (dt_e[c] > 0);  // coal_EM.cpp:117
      } else {
        vstat[c] = true;
      }
      lam_e[c] = p.rates_in[(size_t)rep * p.rates_stride + e];
    }
  }
  // bin-role statics
  const bool is_bin = tid < A;
  double a_b = 0, csh = 0, cns = 0, tk = 0, tkn = 0, dtk = 0, da = 0, db = 0;
  int kb = E;  // padding lanes: beyond every epoch
  if (is_bin) {
    a_b = p.age_grid[tid];
    const double c1 = p.cnt_sh[(size_t)rep * A + tid];
    const double c2 = p.cnt_ns[(size_t)rep * A + tid];
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
    s_cnt[tid] = csh;
    s_cnt[APZ + tid] = cns;
    if (csh > 0 || cns > 0) {
      atomicMin(&s_misc[0], tid);
      atomicMax(&s_misc[1], tid + 1);
    }
  }
  s_kb[tid] = kb;
  if (tid == 0) s_kb[AP] = E + 1;
  __syncthreads();
  const bool bin_live = is_bin && (csh > 0 || cns > 0);
  const bool last_bin = (kb == E - 1);
  // row-segmented reduction statics: f_d = 1 if the lane d to the left (same 16-lane row) is in
  // the same epoch; a lane is the "tail" of its (row, epoch) run if its right neighbour is not
  double f1 = 0, f2 = 0, f4 = 0, f8 = 0;
  {
    const int r = lane & 15;
    if (r >= 1 && s_kb[tid - 1] == kb) f1 = 1.0;
    if (r >= 2 && s_kb[tid - 2] == kb) f2 = 1.0;
    if (r >= 4 && s_kb[tid - 4] == kb) f4 = 1.0;
    if (r >= 8 && s_kb[tid - 8] == kb) f8 = 1.0;
  }
  const bool is_tail = is_bin && ((lane & 15) == 15 || s_kb[tid + 1] != kb);
  // epoch-role: where the tails of this epoch sit in the bin tile (clipped to the bins with data),
  // and the counts of the bins in LATER epochs (CS = shared, CN = not shared)
  int slot0[NCH], slot1[NCH], slot2[NCH], row_x[NCH], row_hi[NCH], seg_hi[NCH];
  double CS0[NCH], CN0[NCH];
  {
    const int nzlo = s_misc[0], nzhi = s_misc[1];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      const int e = c * kWave + lane;
      int lo = A, hi = 0;
      double cs_later = 0.0, cn_later = 0.0;
      if (ep_on[c]) {
        for (int b = 0; b < A; b++) {
          const int k = s_kb[b];
          if (k == e) {
            if (b < lo) lo = b;
            hi = b + 1;
          }
          if (k > e) {
            cs_later += s_cnt[b];
            cn_later += s_cnt[APZ + b];
          }
        }
      }
      CS0[c] = cs_later;
      CN0[c] = cn_later;
      seg_hi[c] = hi;
      const int clo = lo > nzlo ? lo : nzlo, chi = hi < nzhi ? hi : nzhi;
      slot0[c] = slot1[c] = slot2[c] = AP;  // a zero entry
      row_x[c] = 1;
      row_hi[c] = 0;
      if (clo < chi) {
        const int r0 = clo >> 4, r1 = (chi - 1) >> 4;
        slot0[c] = (r0 * 16 + 15 < hi - 1) ? r0 * 16 + 15 : hi - 1;
        if (r1 > r0) slot1[c] = ((r0 + 1) * 16 + 15 < hi - 1) ? (r0 + 1) * 16 + 15 : hi - 1;
        if (r1 > r0 + 1) slot2[c] = ((r0 + 2) * 16 + 15 < hi - 1) ? (r0 + 2) * 16 + 15 : hi - 1;
        row_x[c] = r0 + 3;  // rows beyond the first three (rare: an epoch spanning > 48 bins with data)
        row_hi[c] = r1;
      }
    }
  }
  int my_flags = 0;
  bool wrote_fail0 = false, wrote_fail1 = false;  // this lane published a failed count into buffer 0 / 1

  const double thr = 1.0 - p.rel_tol;
  double ll = -__builtin_inf(), prev_ll = -__builtin_inf();  // coal.cpp:3685
  int iter = 0;
  const int max_iter = (MODE == 1) ? 1 : p.max_iter;

#ifdef COLATE_EM_STAMPS
  unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_prev = stamp();
#endif
  for (iter = 0; iter < max_iter; iter++) {
    COLATE_STAMP(7)
    const bool need_ll = (MODE == 1) || (iter >= p.min_iter) || (iter == max_iter - 1);
    const int par = iter & 1;
    // ============================================================ epoch phase (every wave)
    double q_e[NCH], p_e[NCH], beta_e[NCH], W_e[NCH], VW_e[NCH], PWn_e[NCH], cs_e[NCH], csn_e[NCH];
    {
      // cs_e = sum_{j<e} lambda_j dt_j (coal_EM.cpp:100-103), as a wave scan
      double carry = 0.0;
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        const double x = lam_e[c] * dt_e[c];
        const double incl = wave_prefix_sum(x);
        cs_e[c] = carry + dpp_d<WAVE_SHR1, 0xf, true>(0.0, incl);
        csn_e[c] = cs_e[c] + x;
        carry = carry + readlane_d(incl, 63);
      }
      COLATE_STAMP(8)
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        const int e = c * kWave + lane;
        const double inv = 1.0 / lam_e[c];
        double omS;  // 1 - S_e = sum_{j<e} exp(A_ep[j])   (the shared normaliser's epoch part)
        const double S = em::em_exp_om(-cs_e[c], &omS);
        const bool valid = vstat[c] && (lam_e[c] > 0);
        q_e[c] = 0.0;
        p_e[c] = 0.0;
        beta_e[c] = 0.0;
        if (e < E - 1) {
          q_e[c] = em::em_exp(-csn_e[c] + cs_e[c]);  // exp(-cumsum[i+1] + cumsum[i]), coal_EM.cpp:120
          if (valid) {
            p_e[c] = 1.0 - q_e[c];                                  // exp(A_ep + cs), coal_EM.cpp:119
            beta_e[c] = (t_e[c] + inv) - (tn_e[c] + inv) * q_e[c];  // exp(B_ep + cs), coal_EM.cpp:120
          }
        } else if (e == E - 1 && valid) {  // last epoch, coal_EM.cpp:136-141
          p_e[c] = 1.0;
          beta_e[c] = t_e[c] + inv;
        }
        W_e[c] = ep_on[c] ? S * p_e[c] : 0.0;
        VW_e[c] = ep_on[c] ? S * beta_e[c] - t_e[c] * W_e[c] : 0.0;
        PWn_e[c] = omS + W_e[c];  // sum_{j<=e} exp(A_ep[j])
        COLATE_STAMP(9)
        if (ep_on[c]) {
          my_ep[G_LAM * EPAD + e] = lam_e[c];
          my_ep[G_INV * EPAD + e] = inv;
          my_ep[G_CS * EPAD + e] = cs_e[c];
          my_ep[G_S * EPAD + e] = S;
          my_ep[G_XA * EPAD + e] = (t_e[c] + inv) / inv;  // coal_EM.cpp:204
          my_ep[G_PW * EPAD + e] = omS;
        }
      }
    }
    // G_e = sum_{j>=e} W_j / S_e (mass still to coalesce, relative to survival at t_e) obeys
    // G_e = p_e + q_e G_{e+1} with p_e = 1 - q_e, i.e. 1 - G_e = (1 - G_{E-1}) prod q_j: it is 1
    // whenever the last epoch can absorb (lambda_{E-1} > 0).  The reference asserts that
    // (coal_EM.cpp:351) only for bins inside the last epoch; otherwise fall back to the product.
    double lam_last = 0.0, cs_last = 0.0;  // values of epoch E-1 (uniform)
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      if (c == ((E - 1) >> 6)) {
        lam_last = readlane_d(lam_e[c], (E - 1) & 63);
        cs_last = readlane_d(cs_e[c], (E - 1) & 63);
      }
    }
    const bool absorbing = lam_last > 0;
    wave_lds_fence();
    COLATE_STAMP(0)
    // ============================================================ bin phase (own bins)
    {
      double o_g = 0, o_h = 0, o_N = 0, o_D = 0, llp = 0.0;
      bool failS = false, failN = false;
      if (bin_live) {
        const double lk = my_ep[G_LAM * EPAD + kb], ik = my_ep[G_INV * EPAD + kb];
        const double ck = my_ep[G_CS * EPAD + kb], Sk = my_ep[G_S * EPAD + kb];
        const double Xak = my_ep[G_XA * EPAD + kb], PWk = my_ep[G_PW * EPAD + kb];
        const bool lpos = lk > 0;
        const double ck1 = ck + lk * da;            // coal_EM.cpp:178-181 at the merged grid
        const double ck2 = ck1 + lk * (a_b - a_b);  // second copy of `age` in the merged grid
        const double ck3 = ck2 + lk * db;
        const double qd = em::em_exp(-ck1 + ck);  // EM_shared, coal_EM.cpp:198-210, 263-287
        const double u = em::em_exp(-ck3 + ck2);  // EM_notshared, coal_EM.cpp:330-357, 435-460
        const double Y = (a_b + ik) / ik;
        const double Wp = lpos ? Sk * (1.0 - qd) : 0.0;
        const double X = Xak - Y * qd;
        const double Vp = lpos ? X * ik * Sk : 0.0;
        const double pn = lpos ? 1.0 - u : 0.0;
        const double bn = lpos ? (a_b + ik) - (tkn + ik) * u : 0.0;
        // ---- shared
        const double SigS = PWk + Wp;
        const bool okS = finite_pos(SigS);
        failS = (csh > 0) && !okS;
        if (csh > 0 && okS) {
          const double r = em::em_rcp(SigS);
          const double nk = Wp * r;
          double dk = Vp * r + (-tk * nk);
          if (dk < 0.0) dk = 0.0;
          o_g = csh * r;
          o_N = csh * nk;
          o_D = csh * dk;
          if (need_ll) {
            COLATE_COLD();
            llp = csh * em::em_log(SigS);
          }
        }
        // ---- not shared
        if (cns > 0) {
          if (last_bin) {  // bin beyond the start of the last epoch, coal_EM.cpp:350-357
            if (!lpos) my_flags |= COLATE_FLAG_NAN;  // reference: assert(coal_rate_e > 0)
            double dk = (a_b + ik) - tk;
            if (dk < 0.0) dk = 0.0;
            o_N += cns;
            o_D += cns * dk;
            llp += cns * (-ck2);
          } else if (absorbing) {  // normaliser = exp(-cs(age)) * ((1 - u) + u) = exp(-cs(age))
            double dk = bn + (-tk * pn + dtk * (1.0 - pn));
            if (dk < 0.0) dk = 0.0;
            o_h = cns * u;
            o_N += cns * pn;
            o_D += cns * dk;
            llp += cns * (-ck2);
          } else {  // last rate is 0: the mass beyond t_{k+1} is 1 - S_{E-1}/S_{k+1}
            COLATE_COLD();
            const double Gk1 = 1.0 - em::em_exp(-cs_last + my_ep[G_CS * EPAD + kb + 1]);
            const double SigN = pn + u * Gk1;
            if (finite_pos(SigN)) {
              const double rr = 1.0 / SigN;
              const double nk = pn * rr;
              double dk = bn * rr + (-tk * nk + dtk * (1.0 - nk));
              if (dk < 0.0) dk = 0.0;
              o_h = cns * (u * rr);
              o_N += cns * nk;
              o_D += cns * dk;
              if (need_ll) llp += cns * (-ck2 + em::em_log(SigN));
            } else {
              failN = true;
            }
          }
        }
      }
      COLATE_STAMP(1)
      // bins whose normaliser failed (coal_EM.cpp:288-292, 461-465) drop out of the static counts
      {
        const bool any_fail = __any(failS || failN);
        const bool wrote = par ? wrote_fail1 : wrote_fail0;
        if (failS || failN || wrote) {  // publish, or clear what this lane published two iterations ago
          COLATE_COLD();
          s_cfail[(par * 2 + 0) * APZ + tid] = failS ? csh : 0.0;
          s_cfail[(par * 2 + 1) * APZ + tid] = failN ? cns : 0.0;
        }
        if (par)
          wrote_fail1 = failS || failN;
        else
          wrote_fail0 = failS || failN;
        if (lane == 0) s_fail[par * 4 + wave] = any_fail ? 1 : 0;
      }
      // sums over the run of equal-epoch bins inside each 16-lane row, left to right
#define COLATE_SEG_STEP(CTRL, F)                              \
  o_g = em::fma_(dpp_d<CTRL, 0xf, true>(0.0, o_g), F, o_g);   \
  o_h = em::fma_(dpp_d<CTRL, 0xf, true>(0.0, o_h), F, o_h);   \
  o_N = em::fma_(dpp_d<CTRL, 0xf, true>(0.0, o_N), F, o_N);   \
  o_D = em::fma_(dpp_d<CTRL, 0xf, true>(0.0, o_D), F, o_D);
      COLATE_SEG_STEP(ROW_SHR1, f1)
      COLATE_SEG_STEP(ROW_SHR2, f2)
      COLATE_SEG_STEP(ROW_SHR4, f4)
      COLATE_SEG_STEP(ROW_SHR8, f8)
#undef COLATE_SEG_STEP
      if (is_tail) {
        double* dst = s_out + par * kNumBinArrays * APZ + tid;
        dst[O_G * APZ] = o_g;
        dst[O_H * APZ] = o_h;
        dst[O_N * APZ] = o_N;
        dst[O_D * APZ] = o_D;
      }
      if (need_ll) {
        COLATE_COLD();
        const double tot = readlane_d(wave_prefix_sum(llp), 63);
        if (lane == 0) s_ll[par * 4 + wave] = tot;
      }
    }
    COLATE_STAMP(2)
    __syncthreads();  // the one barrier of the iteration (the LDS tiles are double-buffered)
    COLATE_STAMP(3)
    // ============================================================ epoch accumulation (every wave)
    double N_e[NCH], D_e[NCH];
    {
      const double* src = s_out + par * kNumBinArrays * APZ;
      double g[NCH], h[NCH], oN[NCH], oD[NCH];
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        const double g0 = src[O_G * APZ + slot0[c]], g1 = src[O_G * APZ + slot1[c]], g2 = src[O_G * APZ + slot2[c]];
        const double h0 = src[O_H * APZ + slot0[c]], h1 = src[O_H * APZ + slot1[c]], h2 = src[O_H * APZ + slot2[c]];
        const double n0 = src[O_N * APZ + slot0[c]], n1 = src[O_N * APZ + slot1[c]], n2 = src[O_N * APZ + slot2[c]];
        const double d0 = src[O_D * APZ + slot0[c]], d1 = src[O_D * APZ + slot1[c]], d2 = src[O_D * APZ + slot2[c]];
        g[c] = (g0 + g1) + g2;
        h[c] = (h0 + h1) + h2;
        oN[c] = (n0 + n1) + n2;
        oD[c] = (d0 + d1) + d2;
        for (int r = row_x[c]; r <= row_hi[c]; r++) {
          COLATE_COLD();
          int slot = r * 16 + 15;
          if (slot > seg_hi[c] - 1) slot = seg_hi[c] - 1;
          g[c] += src[O_G * APZ + slot];
          h[c] += src[O_H * APZ + slot];
          oN[c] += src[O_N * APZ + slot];
          oD[c] += src[O_D * APZ + slot];
        }
      }
      COLATE_STAMP(10)
      // RS = sum c r over the shared bins of LATER epochs
      double RSn[NCH];
      {
        double cR = 0.0;
#pragma unroll
        for (int c = NCH - 1; c >= 0; c--) {
          const double sR = wave_suffix_sum(g[c], lane);
          RSn[c] = cR + dpp_d<WAVE_SHL1, 0xf, true>(0.0, sR);
          cR = cR + readlane_d(sR, 0);
        }
      }
      COLATE_STAMP(11)
      // forward recurrence T_{e+1} = q_e T_e + h_e, T_0 = 0
      // (T_e = sum over not-shared bins b in EARLIER epochs of c_b u_b/Sig_b * S_e/S_{k_b+1})
      double T[NCH];
      {
        double Tc = 0.0;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
          double a = ep_on[c] ? q_e[c] : 1.0, b = ep_on[c] ? h[c] : 0.0;
          wave_affine_scan(a, b);
          const double Tn = em::fma_(a, Tc, b);         // T_{e+1}
          T[c] = dpp_d<WAVE_SHR1, 0xf, false>(Tc, Tn);  // T_e (lane 0: carry-in)
          Tc = readlane_d(Tn, 63);
        }
      }
      COLATE_STAMP(12)
      // counts of the bins in LATER epochs, minus those whose normaliser failed this iteration
      double CSn[NCH], CNn[NCH];
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        CSn[c] = CS0[c];
        CNn[c] = CN0[c];
      }
      if (s_fail[par * 4 + 0] | s_fail[par * 4 + 1] | s_fail[par * 4 + 2] | s_fail[par * 4 + 3]) {
        COLATE_COLD();
#pragma unroll
        for (int c = 0; c < NCH; c++) {
          double fs = 0.0, fn = 0.0;
          for (int b = seg_hi[c]; ep_on[c] && b < A; b++) {
            if (s_kb[b] > c * kWave + lane) {
              fs += s_cfail[(par * 2 + 0) * APZ + b];
              fn += s_cfail[(par * 2 + 1) * APZ + b];
            }
          }
          CSn[c] -= fs;
          CNn[c] -= fn;
        }
      }
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        const int e = c * kWave + lane;
        N_e[c] = W_e[c] * RSn[c] + oN[c] + p_e[c] * T[c];
        if (e < E - 1) {
          double Gn = 1.0;  // G_{e+1}
          if (!absorbing) {
            COLATE_COLD();
            Gn = 1.0 - em::em_exp(-cs_last + csn_e[c]);
          }
          // shared bins of later epochs: sum_b c_b (exp(B_e - Z_b) - t_e num_e(b) + dt_e integ_e(b)); the
          // reference clamps every bin's term at 0 (coal_EM.cpp:277), here the (non-negative) sums are
          double integ = CSn[c] - PWn_e[c] * RSn[c];  // sum_b c_b (1 - r_b PW_{e+1})
          if (integ < 0.0) integ = 0.0;
          double dsh = VW_e[c] * RSn[c] + dt_e[c] * integ;
          if (dsh < 0.0) dsh = 0.0;
          // not-shared bins: later epochs contribute dt_e each, earlier ones their tail mass
          double dns = dt_e[c] * CNn[c] + ((beta_e[c] - t_e[c] * p_e[c]) * T[c] + dt_e[c] * Gn * (q_e[c] * T[c]));
          if (dns < 0.0) dns = 0.0;
          D_e[c] = dsh + oD[c] + dns;
        } else {
          double dns = (beta_e[c] - t_e[c] * p_e[c]) * T[c];
          if (dns < 0.0) dns = 0.0;
          D_e[c] = oD[c] + dns;
        }
        if (MODE == 1 && ep_on[c]) {  // (EM mode: a NaN sticks to the rate and is flagged at the end)
          if (N_e[c] != N_e[c] || D_e[c] != D_e[c]) my_flags |= COLATE_FLAG_NAN;  // coal.cpp:3711-3712
          if (N_e[c] < 0.0 || D_e[c] < 0.0) my_flags |= COLATE_FLAG_NEG;          // coal.cpp:3713-3714
        }
      }
    }
    COLATE_STAMP(4)
    if (need_ll) {
      COLATE_COLD();
      ll = ((s_ll[par * 4 + 0] + s_ll[par * 4 + 1]) + s_ll[par * 4 + 2]) + s_ll[par * 4 + 3];
    }
    if (MODE == 1) {
      if (wave == 0) {
#pragma unroll
        for (int c = 0; c < NCH; c++) {
          if (ep_on[c]) {
            p.out_num[(size_t)rep * E + c * kWave + lane] = N_e[c];
            p.out_den[(size_t)rep * E + c * kWave + lane] = D_e[c];
          }
        }
      }
      break;
    }
    // ============================================================ M-step, coal.cpp:3777-3804
    {
      double cand[NCH];
      unsigned long long keep[NCH];  // epochs that do NOT copy their predecessor
      bool simple = true;            // the copying epochs form a prefix 0..m-1: they all become 0
      bool lower_keep = false;
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        const bool copy = (N_e[c] == 0);
        cand[c] = lam_e[c];
        if (!copy && D_e[c] != 0) {
          cand[c] = N_e[c] / D_e[c];
          if (cand[c] < p.rate_floor) cand[c] = p.rate_floor;
        }
        keep[c] = __ballot(ep_on[c] && !copy);
        const unsigned long long cp = __ballot(ep_on[c] && copy);
        if (cp && (lower_keep || (cp & (cp + 1ull)))) simple = false;
        if (keep[c]) lower_keep = true;
      }
      if (simple) {
#pragma unroll
        for (int c = 0; c < NCH; c++) lam_e[c] = ((keep[c] >> lane) & 1ull) ? cand[c] : 0.0;
      } else {
        COLATE_COLD();
        // num == 0: take the (already updated) rate of the previous epoch, 0 if there is none
#pragma unroll
        for (int c = 0; c < NCH; c++) {
          const unsigned long long below = keep[c] & ((1ull << lane) - 1ull);
          const int src = below ? 63 - __builtin_clzll(below) : 0;
          const double from_chunk = __shfl(cand[c], src, 64);
          double from_lower = 0.0;  // nearest keeper in an earlier chunk (uniform)
          bool have_lower = false;
#pragma unroll
          for (int cc = NCH - 1; cc >= 0; cc--) {
            if (cc < c && !have_lower && keep[cc]) {
              from_lower = readlane_d(cand[cc], 63 - __builtin_clzll(keep[cc]));
              have_lower = true;
            }
          }
          const bool self = (keep[c] >> lane) & 1ull;
          lam_e[c] = ep_on[c] ? (self ? cand[c] : (below ? from_chunk : from_lower)) : 0.0;
        }
      }
    }
    COLATE_STAMP(5)
    // stop rule, coal.cpp:3822 (evaluated after the update); uniform across the workgroup
    bool stop = false;
    if (iter > p.min_iter) {
      COLATE_COLD();
      stop = (ll / prev_ll > thr);
    }
    prev_ll = ll;
    if (stop) break;
  }

#ifdef COLATE_EM_STAMPS
  if (p.out_num && lane == 0 && MODE == 0) {  // diagnostic build: per-wave phase cycles in place of out_num
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.out_num) + ((size_t)rep * 4 + wave) * 16;
    for (int i = 0; i < 16; i++) dbg[i] = st_acc[i];
  }
#endif
  // ------------------------------------------------------------------ epilogue
  if (MODE == 0 && wave == 0) {
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      if (ep_on[c]) {
        if (lam_e[c] != lam_e[c]) my_flags |= COLATE_FLAG_NAN;
        p.out_rates[(size_t)rep * E + c * kWave + lane] = lam_e[c];
      }
    }
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

template <int MODE, int NCH>
hipError_t launch_one(const ColateEmArgs& args, hipStream_t stream, size_t lds, int threads) {
  auto kern = em_kernel<MODE, NCH>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(args.B), dim3(threads), lds, stream, args);
  return hipGetLastError();
}

}  // namespace

static int em_threads(int A) { return (A + 63) & ~63; }
static int em_chunks(int E) { return E <= 64 ? 1 : (E <= 128 ? 2 : 4); }

size_t colate_em_lds_bytes(int E, int A) {
  const size_t EPAD = (size_t)em_chunks(E) * kWave;
  const size_t AP = (size_t)em_threads(A);
  const size_t APZ = AP + 16;
  const size_t nwaves = AP / kWave;
  const size_t doubles = (EPAD + 1) + nwaves * kNumGather * EPAD + 2 * kNumBinArrays * APZ + 6 * APZ + 8;
  const size_t ints = (AP + 1) + 8 + 4;
  return doubles * sizeof(double) + ints * sizeof(int);
}

hipError_t colate_em_launch(const ColateEmArgs& args, hipStream_t stream) {
  const size_t lds = colate_em_lds_bytes(args.E, args.A);
  const int threads = em_threads(args.A);
  const int nch = em_chunks(args.E);
  if (args.mode == 1) {
    if (nch == 1) return launch_one<1, 1>(args, stream, lds, threads);
    if (nch == 2) return launch_one<1, 2>(args, stream, lds, threads);
    return launch_one<1, 4>(args, stream, lds, threads);
  }
  if (nch == 1) return launch_one<0, 1>(args, stream, lds, threads);
  if (nch == 2) return launch_one<0, 2>(args, stream, lds, threads);
  return launch_one<0, 4>(args, stream, lds, threads);
}
