// colate_amd/csrc/em_kernel_impl.hpp -- the EM hot path of `Colate --mode mut` on gfx950.
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
// and E rates on the way out.  Each iteration is a short chain of dependent
// double-precision instructions, and BASELINE configs put <= 256 replicates on a
// 256-CU GPU (one workgroup per CU), so the kernel is built for LATENCY: what
// costs is the number of instructions ONE wave has to issue per iteration
// (a lone wave issues one instruction per 4.3 cycles whatever it is -- FP64, DPP, scalar,
// s_nop -- and in order: tools/ubench.hip, tools/ubench_fetch2.hip).
// The work of an iteration is therefore split by ROLE over waves that run
// concurrently on different SIMDs:
//
//   role A ("shared"):     cs scan, S_e = exp(-cs_e), shared-bin terms, suffix scan RS, N/D shared parts
//   role B ("not shared"): cs scan, q_e, p_e, beta_e, 1/lambda, not-shared-bin terms, affine scan T,
//                          N/D not-shared parts
//
// with age bins one per lane (compacted to the bins that carry data: NB groups of
// 64, so 2*NB waves; waves without bins retire before the loop, which keeps the
// s_barrier cheap), epoch e in lane e / NCH (slot e % NCH; NCH = 1, 2, 4 for up to 64, 128, 256 epochs) of the role leaders,
// and three workgroup barriers per iteration (epoch values -> bins -> per-epoch
// sums -> rates) -- two in the build for batches that leave every workgroup a CU to
// itself at up to 64 epochs, where each wave of role A computes the epoch values it
// needs itself (`kFree` in em_kernel).  Per-epoch sums of the per-bin terms are reduced in registers
// (row-segmented DPP) and handed over through an LDS tile at static "tail" slots.
// For batches far beyond the CU count the same code is instantiated as a THROUGHPUT
// variant (template flag TPUT: two waves per replicate that loop over the bin groups;
// see em_kernel below), with bit-identical results.
//
// The reference evaluates exp(log-term - Z) for every (age bin, epoch) pair:
// O(A*E) transcendentals per iteration.  Here every such term is factored into
// a per-epoch piece times a per-bin piece, so that the sufficient
// statistics N_e = sum_b c_b num_e(b), D_e = sum_b c_b denom_e(b) and
// ll = sum_b c_b Z_b need O(A + E) transcendentals; sums that telescope in the
// model are not summed (sum_{j<e} exp(A_j) = 1 - exp(-cs_e); the not-shared
// normaliser is exp(-cs(age)) whenever the last epoch absorbs).  The formulas
// the reference evaluates with catastrophic cancellation are kept operand for
// operand (no fused multiply-add: this file is built with -ffp-contract=off and
// uses fma only inside em_math.hpp and in recurrences that have no counterpart
// in the reference).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "em_kernels.h"
#include "em_math.hpp"

namespace {

constexpr int kWave = 64;
enum { O_W = 0, O_N, O_D, kNumBinArrays };  // per-bin values -> epochs: weight (c r | c u), own-epoch num, denom
enum { G_LAM = 0, G_INV, G_XA, G_P, G_BETA, G_CS, G_S, G_PW, kNumGather };  // per-epoch values in LDS
constexpr int G_Q = kNumGather;  // one more row, q_e: the split leaders of em_kernel hand it over; free mode: wave 3 to role B's leader
constexpr int num_gather_rows(int nch, bool tput) { return (nch >= 2 || !tput) ? kNumGather + 1 : kNumGather; }
// with the epochs split over two waves of a role the tail model's refresh cannot borrow the tile (the other owner may still be
// loading its tails), and each owner needs arrays of its own (they run the refresh side by side, unsynchronised: sharing one set
// would let the slower one's first pass overwrite what the faster one is searching)
// (the throughput variant has one wave per role: one set)
constexpr int tail_scratch_arrays(int nch, bool tput) { return nch >= 2 ? (tput ? 3 : 6) : 0; }

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

// The epoch-level scans only need to span the lanes that hold epochs: `rows` = number of 16-lane
// rows in use (uniform), so E <= 16 / <= 32 skip the cross-row steps.

// inclusive prefix sum over the lanes of the first `rows` rows
__device__ __forceinline__ double wave_prefix_sum(double v, int rows = 4) {
  v += dpp_d<ROW_SHR1, 0xf, true>(0.0, v);
  v += dpp_d<ROW_SHR2, 0xf, true>(0.0, v);
  v += dpp_d<ROW_SHR4, 0xf, true>(0.0, v);
  v += dpp_d<ROW_SHR8, 0xf, true>(0.0, v);
  // (row_bcast15 leaves the lanes of row 0 unwritten even with bound_ctrl -- tried with row_mask 0xf to save the two v_mov that
  // set up `old`: row 0 then keeps whatever the register held, and test_l2_golden failed; the masked form it is)
  if (rows > 1) v += dpp_d<ROW_BCAST15, 0xa>(0.0, v);
  if (rows > 2) v += dpp_d<ROW_BCAST31, 0xc>(0.0, v);
  return v;
}
// inclusive suffix sum (lane l: sum of lanes l..), lanes beyond the rows in use must hold 0
__device__ __forceinline__ double wave_suffix_sum(double v, int lane, int rows = 4) {
  v += dpp_d<ROW_SHL1, 0xf, true>(0.0, v);
  v += dpp_d<ROW_SHL2, 0xf, true>(0.0, v);
  v += dpp_d<ROW_SHL4, 0xf, true>(0.0, v);
  v += dpp_d<ROW_SHL8, 0xf, true>(0.0, v);
  if (rows > 1) {
    const double r1 = readlane_d(v, 16);
    double add = 0.0;
    if (rows > 2) {
      const double r2 = readlane_d(v, 32), r3 = readlane_d(v, 48);
      const double s23 = r2 + r3, s123 = r1 + s23;
      const int row = lane >> 4;
      add = row == 0 ? s123 : (row == 1 ? s23 : (row == 2 ? r3 : 0.0));
    } else {
      add = (lane >> 4) == 0 ? r1 : 0.0;
    }
    v = v + add;
  }
  return v;
}
// inclusive prefix composition of the affine maps x -> a*x + b (lane order = application order):
// afterwards (a, b) of lane l is f_l o ... o f_0
// (`nosrc`: null, or four per-lane constants -- 1.0 where the lane has no source 1 / 2 / 4 / 8 lanes to its left in its row, else 0.0.
// A lane without a source must see the identity map: its multiplier reads 1.  As `old` of the DPP move that is two v_mov per
// step -- the compiler sets the register pair up again every time --; as zero fill plus the constant it is one addition.)
__device__ __forceinline__ void wave_affine_scan(double& a, double& b, int rows = 4, const double* nosrc = nullptr) {
#define COLATE_AFF_STEP(CTRL, RM, BND, K)                                                                   \
  {                                                                                                         \
    const double as = nosrc ? dpp_d<CTRL, RM, true>(0.0, a) + nosrc[K] : dpp_d<CTRL, RM>(1.0, a);           \
    const double bs = dpp_d<CTRL, RM, BND>(0.0, b);                                                         \
    b = em::fma_(a, bs, b);                                                                                 \
    a = a * as;                                                                                             \
  }
  COLATE_AFF_STEP(ROW_SHR1, 0xf, true, 0)
  COLATE_AFF_STEP(ROW_SHR2, 0xf, true, 1)
  COLATE_AFF_STEP(ROW_SHR4, 0xf, true, 2)
  COLATE_AFF_STEP(ROW_SHR8, 0xf, true, 3)
#undef COLATE_AFF_STEP
  if (rows == 2) {  // (the last step: only b is used afterwards)
    const double bs = dpp_d<ROW_BCAST15, 0xa>(0.0, b);
    b = em::fma_(a, bs, b);
  }
#define COLATE_AFF_STEP2(CTRL, RM)                 \
  {                                                \
    const double as = dpp_d<CTRL, RM>(1.0, a);     \
    const double bs = dpp_d<CTRL, RM>(0.0, b);     \
    b = em::fma_(a, bs, b);                        \
    a = a * as;                                    \
  }
  if (rows > 2) COLATE_AFF_STEP2(ROW_BCAST15, 0xa)
  if (rows > 2) COLATE_AFF_STEP2(ROW_BCAST31, 0xc)
#undef COLATE_AFF_STEP2
}

// inclusive suffix maximum of non-negative values (lane l: max of lanes l..63)
__device__ __forceinline__ double wave_suffix_max(double v, int lane) {
  v = __builtin_fmax(v, dpp_d<ROW_SHL1, 0xf, true>(0.0, v));
  v = __builtin_fmax(v, dpp_d<ROW_SHL2, 0xf, true>(0.0, v));
  v = __builtin_fmax(v, dpp_d<ROW_SHL4, 0xf, true>(0.0, v));
  v = __builtin_fmax(v, dpp_d<ROW_SHL8, 0xf, true>(0.0, v));
  const double r1 = readlane_d(v, 16), r2 = readlane_d(v, 32), r3 = readlane_d(v, 48);
  const double m23 = __builtin_fmax(r2, r3), m123 = __builtin_fmax(r1, m23);
  const int row = lane >> 4;
  return __builtin_fmax(v, row == 0 ? m123 : (row == 1 ? m23 : (row == 2 ? r3 : 0.0)));
}
// half the spacing of the doubles around z > 0 (0 for z == 0 and below 2^-969): what an addend must reach to change z
__device__ __forceinline__ double half_ulp_pos(double z) {
  const int ex = (__double2hiint(z) >> 20) & 0x7ff;  // z = 1.f x 2^(ex - 1023)
  return ex > 53 ? __hiloint2double((ex - 53) << 20, 0) : 0.0;
}

// E[max(0, g - z)] for g ~ N(0, 1):  phi(z) - z Q(z)  (H(-z) of tools/study/residue_models.cpp; H(z) = z + H(-z)), and its
// derivative -Q(z) through `q`.  The upper tail Q by Abramowitz & Stegun 26.2.17 (absolute error < 7.5e-8: plenty for a model
// of rounding noise).
// (Evaluated with the single-precision units -- v_exp_f32, v_rcp_f32, five v_fma_f32: 1e-7 relative, deterministic on the
// device -- in a third of the instructions of the double-precision exp: the refresh evaluates it per (bin, epoch in transition).)
__device__ __forceinline__ double tail_hneg(double z, double& q) {
  const float a = __builtin_fminf(__builtin_fabsf((float)z), 30.0f);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.2316419f, a, 1.0f));
  float poly = __builtin_fmaf(t, 1.330274429f, -1.821255978f);
  poly = __builtin_fmaf(t, poly, 1.781477937f);
  poly = __builtin_fmaf(t, poly, -0.356563782f);
  poly = __builtin_fmaf(t, poly, 0.319381530f);
  poly = poly * t;
  const float phi = 0.3989422804f * __builtin_amdgcn_exp2f(-0.72134752f * (a * a));  // exp(-a^2 / 2)
  const float qa = phi * poly;                                                         // Q(|z|)
  const float base = __builtin_fmaxf(__builtin_fmaf(-a, qa, phi), 0.0f);
  q = z < 0.0 ? (double)(1.0f - qa) : (double)qa;
  return (z < 0.0 ? -z : 0.0) + (double)base;
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
// ... and absolute stamps along the prologue and around the loops (wave 0 of every replicate; written to out_den at the end)
#define COLATE_PSTAMP(i) pro_t[i] = stamp();
#else
#define COLATE_STAMP(i)
#define COLATE_PSTAMP(i)
#endif

// Timing-only ablations for tools/em_phase_probe.hip (-DCOLATE_ABL=<bit mask>): each bit removes one
// piece of the iteration so that its true cost shows up in the kernel time (results are garbage).
#ifdef COLATE_ABL
#define COLATE_ABL_HAS(b) (((COLATE_ABL) >> (b)) & 1)
#else
#define COLATE_ABL_HAS(b) 0
#endif

// Marks a rarely-taken branch body: a volatile asm cannot be executed speculatively, so the compiler
// keeps the branch instead of if-converting it (it otherwise evaluates whole exp()/log() calls of
// cold paths unconditionally and selects the result).
#define COLATE_COLD() asm volatile("; cold path")
#define COLATE_STR2(x) #x
#define COLATE_STR(x) COLATE_STR2(x)

__device__ __forceinline__ bool finite_pos(double x) { return x > 0.0 && x < __builtin_inf(); }

// the lanes' predicate as a scalar mask, straight from the compare (HIP's __ballot / __any go through a v_cndmask + v_cmp pair)
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// wave-local LDS hand-off: earlier ds_writes of this wave are visible to its later ds_reads
// (the LDS queue is in order per wave); this only stops the compiler from moving them.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// what the bin phase (P2) needs to know about one age bin
struct BinStat {
  double a_b, cnt, tk, tkn, dtk, da, db;  // age, count (this role's kind), epoch start / end / length, age - start, end - age
  double f1, f2, f4, f8;                  // 1.0 if the lane 1/2/4/8 to the left (same 16-lane row) is in the same epoch
  int kb, pos;                            // epoch of the bin; position in the compacted tile
  bool live, last_bin, is_tail;           // carries data; lies in the last epoch; last lane of its (row, epoch) run
};
// packed form of the static part, one int per compacted position (throughput variant)
enum { BF_F1 = 1, BF_F2 = 2, BF_F4 = 4, BF_F8 = 8, BF_TAIL = 16, BF_INRANGE = 32, BF_KB_SHIFT = 8 };

// dwords of padding between the 64-byte boundary and the EM loop, per instantiation and build (see the loop head);
// -DCOLATE_LOOP_PAD=n overrides all of them (tools/pad_sweep.sh)
constexpr int em_loop_pad(int mode, int nch, int erows, bool tput, int wpe) {
#ifdef COLATE_LOOP_PAD
  return (COLATE_LOOP_PAD) & 7;
#else
  (void)erows, (void)wpe;
  if (mode != 0) return 0;
  // measured on MI355X, kernel ms for pads 0..7, loop code of round 3 (gpurun_out/r03j/padsweep.txt -> profiles/r03_placement.txt).
  // The pad shifts everything behind it, the loops compiled per kind of wave included.
#ifdef COLATE_EM_ILP_BUILD
  (void)tput;
  // latency variant, max-ilp build, kernel ms for pads 0..7 (round 4, every loop anchored by itself: gpurun_out/r04_padsweep.txt ->
  // profiles/r04_placement.txt; round 3's library on the same box: 0.883 / 1.217 / 1.140):
  // E=23 B=100 (the build without the register cap, two barriers) 0.895 0.897 0.903 0.906 0.910 0.887 0.901 0.916;
  // E=122 B=100 1.308 1.309 1.308 1.314 1.316 1.317 1.335 1.338 (with the tail model's per-iteration load still in the loop);
  // E=23 B=400 (with the cap) 1.144 1.146 1.149 1.138 1.145 1.154 1.134 1.148
  // ... and again at the end of round 4 (refresh block and prologue trimmed; gpurun_out/t2/padsweep.txt -> profiles/r04_placement_raw.txt):
  // E=23 B=100 0.896 0.899 0.910 0.914 0.916 0.890 0.906 0.922 (the build before: 0.893, round 3's: 0.881);
  // E=122 B=100 1.256 1.258 1.249 1.254 1.252 1.263 1.267 1.275 (the build before: 1.310, round 3's: 1.224);
  // E=23 B=400 1.149 1.156 1.155 1.142 1.147 1.156 1.139 1.150 (1.146, 1.140)
  return nch == 1 ? (wpe == 2 ? 5 : 6) : 2;
#else
  (void)nch;
  if (!tput) return 6;  // latency variant, default build (not picked by colate_em_variant any more; COLATE_EM_VARIANT=latency)
  return 4;             // throughput variant: E=23 B=4096 6.347 6.291 6.272 6.320 6.193 6.286 6.279 6.313 (round 4, gpurun_out/r04_padsweep.txt; round 3's library: 6.206)
#endif
#endif
}

// A second placement point behind barrier 2 (a 32-byte boundary + this many dwords), or -1 for none
constexpr int em_loop_pad2(int mode, int nch, bool tput) {
#if defined(COLATE_EM_ILP_BUILD) && !defined(COLATE_LOOP_PAD)
  (void)mode, (void)nch, (void)tput;
  return -1;  // (was worth 1 % before the table-driven exp; not re-tuned for the final code)
#else
  (void)mode, (void)nch, (void)tput;
  return -1;
#endif
}

// MODE 0: EM to convergence, 1: one E-step (num/den/ll out).  NCH = epochs per lane (1, 2, 4: up to 64, 128, 256 epochs);
// EROWS = 16-lane rows that hold epochs (1, 2 or 4; 4 whenever NCH > 1).
// TPUT = false: the latency variant described at the top (a wave per role and bin group, one workgroup per CU
// in mind).  TPUT = true: the THROUGHPUT variant for batches far beyond the number of CUs: the same phases and
// the same arithmetic (results are bit-identical), but one replicate is two waves (one per role) that walk
// through the bin groups one after the other, their per-bin statics re-read from LDS: a third of the wave
// slots and fewer registers per replicate, so three times as many replicates are resident per CU and fill
// the issue slots that a lone workgroup leaves empty at its barriers.
// (second launch bound = waves per SIMD the register allocation must leave room for: three at up to 64 epochs -- two 6-wave
// workgroups of the latency variant, six 2-wave ones of the throughput variant per CU: B = 400 ran 2.01 instead of 1.21 ms
// when the tail model's refresh block took the kernel to 182 registers -- two at up to 128, one beyond.  WPE != 0 overrides
// it: a batch that leaves every workgroup a CU to itself runs the build without the cap, 0.5 % faster at B = 100.)
template <int MODE, int NCH, int EROWS, bool TPUT, int WPE = 0>
__global__ __launch_bounds__(TPUT ? 2 * kWave : 2 * COLATE_EM_MAX_A, WPE ? WPE : ((NCH == 1) ? 3 : (NCH == 2 ? 2 : 1))) void em_kernel(ColateEmArgs p) {
  extern __shared__ double lds[];
  const int E = p.E, A = p.A;
  constexpr int EPAD = NCH * kWave;
  const int NBMAX = (A + kWave - 1) / kWave;  // bin groups of 64 per role
  const int AP = NBMAX * kWave;       // >= A
  const int APZ = AP + 2;             // stride of the per-bin tiles; entry AP stays zero (the "no tail" slot)
  const int tid = threadIdx.x, lane = tid & 63;
  // Epoch e lives in lane e / NCH, slot e % NCH of the epoch-level waves: the NCH epochs of a lane are CONSECUTIVE, so a
  // scan over the epochs is one wave scan of the lanes' totals plus NCH - 1 local steps (round 2 kept chunks of 64
  // consecutive epochs per slot -- one wave scan per chunk -- and 65..128 epochs cost two of every scan)
  constexpr int kSlotShift = (NCH == 1) ? 0 : (NCH == 2 ? 1 : (NCH == 4 ? 2 : (NCH == 8 ? 3 : 4)));
  static_assert(NCH == 1 || NCH == 2 || NCH == 4 || NCH == 8 || NCH == 16, "NCH");
  auto ep_of = [&](int c) { return lane * NCH + c; };
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (a scalar: branches on role / group / wave are s_cbranch_scc)
  const int role = wave & 1;          // 0: shared (A), 1: not shared (B)
  const int grp = wave >> 1;          // waves 2g, 2g+1 own bin group g: the live waves are 0..2*NB-1, one per SIMD
  const int rep = blockIdx.x;

  // ---- LDS carve-up ----
  double* s_t = lds;                                 // [EPAD + 1] epoch starts
  double* s_ep = s_t + EPAD + 1;                     // [kNumGather][EPAD] epoch values (A writes CS,S,PW; B the rest)
  constexpr int kRows = num_gather_rows(NCH, TPUT);
  double* s_out = s_ep + kRows * EPAD;               // [2 roles][kNumBinArrays][APZ] per-bin tails
  double* s_nd = s_out + 2 * kNumBinArrays * APZ;    // [2 roles][2][EPAD] partial N, D per role
  double* s_cfail = s_nd + 4 * EPAD;                 // [2 roles][APZ] counts of bins whose normaliser failed
  double* s_cnt = s_cfail + 2 * APZ;                 // [2 roles][APZ] counts
  double* s_ll = s_cnt + 2 * APZ;                    // [8] per-wave log-likelihood partials + [2] total counts per kind + oldest data epoch
  double* s_exptab = s_ll + 12;                      // [64] 2^(j/32) as hi, lo pairs for em::em_exp_t (em_math.hpp)
  double* s_age = s_exptab + em::kExpTableDoubles;   // [AP] age grid (throughput variant; the tail model's refresh)
  double* s_tscr = s_age + AP;                       // [2 owners][3][APZ] scratch of the tail model's refresh (two or more epochs per lane only)
  int* s_kb = reinterpret_cast<int*>(s_tscr + tail_scratch_arrays(NCH, TPUT) * APZ);  // [AP + 1] epoch of each bin
  int* s_fail = s_kb + AP + 1;                       // [8] per-wave "a bin failed" flags
  int* s_misc = s_fail + 8;                          // [4] nzlo, nzhi, flags
  int* s_bflags = s_misc + 4;                        // [AP] packed per-position statics (throughput variant)

#ifdef COLATE_EM_TRACE  // diagnostic build (tools/residency_probe.hip): where and when this workgroup ran
  unsigned long long trace_t0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(trace_t0)::"memory");
#endif
  // ------------------------------------------------------------------ prologue
#ifdef COLATE_EM_STAMPS
  unsigned long long pro_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  COLATE_PSTAMP(0)
  const double* epochs = p.epochs + (size_t)rep * p.epochs_stride;
  for (int i = tid; i < EPAD + 1; i += blockDim.x) s_t[i] = (i < E) ? epochs[i] : 0.0;
  for (int i = tid; i < kRows * EPAD + 2 * kNumBinArrays * APZ + 4 * EPAD + 4 * APZ; i += blockDim.x) s_ep[i] = 0.0;
  if (tid == 0) {
    s_misc[0] = A;
    s_misc[1] = 0;
    s_misc[2] = 0;
    s_misc[3] = 0;
  }
  if (tid < 8) {
    s_ll[tid] = 0.0;
    s_fail[tid] = 0;
  }
  if (tid < em::kExpTableDoubles) s_exptab[tid] = em::kExpTableDevice[tid];
  __syncthreads();
  COLATE_PSTAMP(1)
  for (int t = tid; t < AP; t += blockDim.x) {
    int kb = E;  // padding: beyond every epoch
    bool has_sh = false, has_ns = false;
    if (t < A) {
      const double a = p.age_grid[t];
      const double c1 = p.cnt_sh[(size_t)rep * A + t];
      const double c2 = p.cnt_ns[(size_t)rep * A + t];
      const double csh = (c1 > 0) ? c1 : 0.0;  // coal.cpp:3706, 3719: only counts > 0 are visited
      const double cns = (c2 > 0) ? c2 : 0.0;
      // coal_EM.cpp:60-95: largest e with epochs[e] <= age (strict `age < epochs[e]`): the epoch starts are non-decreasing
      // (include/colate_amd.h), so the first e with age < epochs[e] comes from a bisection -- the linear scan of the reference
      // was 8 (12) of the 34 (59) us this prologue took at 23 (122) epochs, and the prologue 5 % (8 %) of the whole kernel
      int first_gt = 0;  // number of epochs with start <= age
      for (int len = E; len > 0;) {
        const int half = len >> 1;
        if (!(a < s_t[first_gt + half])) {
          first_gt += half + 1;
          len -= half + 1;
        } else {
          len = half;
        }
      }
      kb = first_gt - 1;
      if (kb < 0) kb = 0;  // host validates age >= epochs[0]; never taken
      s_cnt[t] = csh;
      s_cnt[APZ + t] = cns;
      s_age[t] = a;  // (throughput variant: per-bin statics; both: the tail model's refresh)
      has_sh = csh > 0;
      has_ns = cns > 0;
    }
    {  // first / last bin with data and the number of (bin, kind) pairs the reference evaluates (for the epilogue): one atomic per
       // wave from the lane masks (AP is a multiple of 64: a wave's lanes run this loop together) instead of three per lane
      const unsigned long long m_sh = ballot64(has_sh), m_ns = ballot64(has_ns), m_any = m_sh | m_ns;
      if (m_any != 0 && lane == 0) {
        atomicMin(&s_misc[0], t + (int)__builtin_ctzll(m_any));
        atomicMax(&s_misc[1], t + 64 - (int)__builtin_clzll(m_any));
        atomicAdd(&s_misc[3], (int)__builtin_popcountll(m_sh) + (int)__builtin_popcountll(m_ns));
      }
    }
    s_kb[t] = kb;
  }
  if (tid == 0) s_kb[AP] = E + 1;
  __syncthreads();
  COLATE_PSTAMP(2)
  int nzlo = s_misc[0], nzhi = s_misc[1];
  if (nzlo >= nzhi) {  // no data at all: keep one (empty) group so that the run mirrors the reference
    nzlo = 0;
    nzhi = 0;
  }
  const int NB = (nzhi - nzlo + 63) / 64 > 0 ? (nzhi - nzlo + 63) / 64 : 1;

  // epoch statics (every wave; only the role leaders and the M-step use them)
  double t_e[NCH], tn_e[NCH], dt_e[NCH], lam_e[NCH];
  bool vstat[NCH], ep_on[NCH];
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const int e = ep_of(c);
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
      // the starting rates wait in the output row for the epilogue's verdict (LDS is what limits the resident
      // workgroups of the throughput variant: 26.0 KB x 6 just fits a CU's 160 KB)
      if (MODE == 0 && wave == 0) p.out_rates[(size_t)rep * E + e] = lam_e[c];
    }
  }
  COLATE_PSTAMP(3)
  if (tid == 0) s_ll[10] = (nzhi > nzlo) ? (double)s_kb[nzhi - 1] : -1.0;  // epoch of the oldest bin that carries data
  // bin statics: this lane's bin (compacted to the bins that carry data) and role
  const int pos = grp * kWave + lane;  // position in the compacted tile
  const int bin = nzlo + pos;
  const bool in_range = (grp < NB) && (bin < nzhi);
  double a_b = 0, cnt = 0, tk = 0, tkn = 0, dtk = 0, da = 0, db = 0;
  int kb = E;
  if (in_range) {
    a_b = p.age_grid[bin];
    cnt = s_cnt[role * APZ + bin];
    kb = s_kb[bin];
    tk = s_t[kb];
    if (kb < E - 1) {
      tkn = s_t[kb + 1];
      dtk = tkn - tk;
    }
    da = a_b - tk;
    db = (kb < E - 1) ? tkn - a_b : 0.0;  // (bins in the last epoch: no end of epoch)
  }
  const bool live = in_range && cnt > 0;
  const bool last_bin = (kb == E - 1);
  // row-segmented reduction statics: f_d = 1 if the lane d to the left (same 16-lane row) is in
  // the same epoch; a lane is the "tail" of its (row, epoch) run if its right neighbour is not
  double f1 = 0, f2 = 0, f4 = 0, f8 = 0;
  bool is_tail = false;
  if (in_range) {
    const int r = lane & 15;
    if (r >= 1 && s_kb[bin - 1] == kb) f1 = 1.0;
    if (r >= 2 && s_kb[bin - 2] == kb) f2 = 1.0;
    if (r >= 4 && s_kb[bin - 4] == kb) f4 = 1.0;
    if (r >= 8 && s_kb[bin - 8] == kb) f8 = 1.0;
    is_tail = (r == 15) || (bin + 1 >= nzhi) || (s_kb[bin + 1] != kb);
  }
  const BinStat bs0{a_b, cnt, tk, tkn, dtk, da, db, f1, f2, f4, f8, kb, pos, live, last_bin, is_tail};
  if (TPUT) {  // the same statics for every compacted position, packed (the bin phase rebuilds a BinStat per group)
    for (int t = tid; t < AP; t += blockDim.x) {
      const int b = nzlo + t;
      int fl = 0;
      if (t < NB * kWave && b < nzhi) {
        const int k = s_kb[b], r = t & 15;
        fl = BF_INRANGE | (k << BF_KB_SHIFT);
        if (r >= 1 && s_kb[b - 1] == k) fl |= BF_F1;
        if (r >= 2 && s_kb[b - 2] == k) fl |= BF_F2;
        if (r >= 4 && s_kb[b - 4] == k) fl |= BF_F4;
        if (r >= 8 && s_kb[b - 8] == k) fl |= BF_F8;
        if (r == 15 || b + 1 >= nzhi || s_kb[b + 1] != k) fl |= BF_TAIL;
      }
      s_bflags[t] = fl;
    }
  }
  // epoch-role statics: where the tails of this epoch sit in the compacted tile, and the counts of
  // the bins in LATER epochs (this role's kind)
  int slot0[NCH], slot1[NCH], slot2[NCH], row_x[NCH], row_hi[NCH], seg_hi[NCH];
  int nlt[NCH];  // bins (compacted positions) in EARLIER epochs: where this epoch's bins start in the compacted order
  double C0[NCH];
  // The reference's denominators contain dt_e * integ with integ = 1 - num[0] - num[1] - ... (coal_EM.cpp:270-274,
  // 445-449): where the mass still to coalesce is below the resolution of that subtraction (survival < ~1e-16:
  // epochs behind a very high rate, or far older than all data) what remains is its rounding residue, which is
  // >= 0 after the reference's clamps and averages kIntegResidue per unit count (measured on the reference:
  // 2.7e-17 .. 5.7e-17, i.e. ~0.36 * 2^-53).  That residue is what drives the reference's rate to its floor in such
  // epochs.  The factored sums below are exact there (mass 0), so the residue is put in explicitly: without it
  // those epochs would get the ratio of two vanishing numbers instead of the reference's floor.
  constexpr double kIntegResidue = 4.0e-17;
  // epilogue: a denominator below this many residues is not reproducible to 1e-8.  Measured on the reference (its rates
  // under 1-ulp libm noise, tools/parity_sweep.py): relative spread of a rate ~ 30 .. 100 / (denominator / (dt_e * residue)) --
  // integ accumulates the rounding of each of the ~100 terms it subtracts -- so 1e-8 needs a ratio of ~1e10.  (Round 2 had a
  // x3 margin on top, which flagged one epoch more than the checker finds unstable; with the tail model below the flagged
  // epochs themselves stay within the reference's noise envelope, and the margin went.)
  constexpr double kResolvedRatio = 1.0e10;
  COLATE_PSTAMP(4)
  double c_all = 0.0;
  // (ascending; the counts outside [nzlo, nzhi) are 0, x + 0.0 == x, and the row is zero up to AP: eight loads in flight per
  // round, added in order -- one load per addition waited ~75 cycles for each)
  for (int b = nzlo; b < nzhi; b += 8) {
    double v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = s_cnt[role * APZ + b + j];
#pragma unroll
    for (int j = 0; j < 8; j++) c_all += (b + j < nzhi) ? v[j] : 0.0;  // (see c_later below: never past the data)
  }
  COLATE_PSTAMP(5)
  // role A: dt_e * residue of the shared bins that reach the epoch (set with the bin ranges below).  Role B: the tail
  // model's correction R_e to the not-shared integ mass of the epoch (see `tail model` in P3), refreshed there.
  // (role B, more than 64 epochs -- where deep tails are the rule --: R_e is held as a linear function of the exact mass X_e = q_e T_e
  // of the epoch (proportional to S_{e+1}), integ = eta_s X_e + eta_e with eta_s = 1 + dR_e / dX_e, between two refreshes that are up to 128 iterations apart; up to 64 epochs: as a constant, refreshed every 32nd
  // iteration while some epoch is in transition -- with --bins 3,7,0.2 none ever is --, no instruction in the iteration)
#ifdef COLATE_TAIL_CONST_HOLD  // (A/B switch: the correction held as a constant at every epoch count)
  constexpr bool kLinearHold = false;
#else
  constexpr bool kLinearHold = (NCH >= 2);
#endif
  double eta_e[NCH], eta_s[NCH];
#pragma unroll
  for (int c = 0; c < NCH; c++) eta_e[c] = 0.0, eta_s[c] = 1.0;
  bool tail_trivial_prev = true;  // (uniform) the last full refresh of the tail model found no epoch in its transition zone
  if (tid < 2 * kWave && lane == 0) s_ll[8 + role] = c_all;  // both kinds' totals, for the epilogue (no register carries them)
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const int e = ep_of(c);
    // The bins of epoch e are [lo, hi) -- the ages ascend, so the bins' epochs do --: two bisections of s_kb instead of a walk over
    // all bins per lane, which was 22 (42) us per launch at 23 (122) epochs.  An epoch without bins: lo == hi, no tail slots.
    int lo = A, hi = 0, n_before = 0;
    double c_later = 0.0;
    if (ep_on[c]) {
      lo = hi = 0;
      for (int len = A; len > 0;) {  // first b with s_kb[b] >= e
        const int half = len >> 1;
        if (s_kb[lo + half] < e) {
          lo += half + 1;
          len -= half + 1;
        } else {
          len = half;
        }
      }
      for (int len = A; len > 0;) {  // first b with s_kb[b] > e
        const int half = len >> 1;
        if (s_kb[hi + half] <= e) {
          hi += half + 1;
          len -= half + 1;
        } else {
          len = half;
        }
      }
      const int lo_c = lo < nzlo ? nzlo : (lo > nzhi ? nzhi : lo);
      n_before = lo_c - nzlo;  // compacted positions in earlier epochs
    }
    {  // counts of the later bins, b >= hi, summed in ascending order as before (a bin outside the data has a count of 0, and
       // x + 0.0 == x): one walk over the bins with data for the whole wave, every lane adding from its own hi on
      int b0 = __builtin_amdgcn_readfirstlane(hi);  // (lane 0 holds the slot's earliest epoch: the smallest hi)
      if (b0 < nzlo) b0 = nzlo;
      for (int b = b0; b < nzhi; b += 8) {  // (eight loads in flight; the row is zero from nzhi up to AP)
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = s_cnt[role * APZ + b + j];
#pragma unroll
        for (int j = 0; j < 8; j++) {
          // (b + j < nzhi: the last round of eight may reach past the row's zero padding -- up to AP + 1 -- into what lies behind
          // it in LDS, e.g. another launch's log-likelihood partials: round 4's fuzz, 46 cases, when the walk started elsewhere)
          c_later += (ep_on[c] && b + j >= hi && b + j < nzhi) ? v[j] : 0.0;
        }
      }
    }
    C0[c] = c_later;
    nlt[c] = n_before;
    const int clo = (lo > nzlo ? lo : nzlo) - nzlo, chi = (hi < nzhi ? hi : nzhi) - nzlo;  // compacted, clipped
    seg_hi[c] = chi;
    slot0[c] = slot1[c] = slot2[c] = AP;  // a zero entry
    row_x[c] = 1;
    row_hi[c] = 0;
    if (clo < chi) {
      const int r0 = clo >> 4, r1 = (chi - 1) >> 4;
      slot0[c] = (r0 * 16 + 15 < chi - 1) ? r0 * 16 + 15 : chi - 1;
      if (r1 > r0) slot1[c] = ((r0 + 1) * 16 + 15 < chi - 1) ? (r0 + 1) * 16 + 15 : chi - 1;
      if (r1 > r0 + 1) slot2[c] = ((r0 + 2) * 16 + 15 < chi - 1) ? (r0 + 2) * 16 + 15 : chi - 1;
      row_x[c] = r0 + 3;  // rows beyond the first three (rare: an epoch spanning > 48 bins with data)
      row_hi[c] = r1;
    }
  }
  // role A: dt_e * mean residue of the shared bins WHOSE CHAIN REACHES EPOCH e (0 in the last epoch, which has no dt_e * integ
  // term).  A shared bin's integ recurrence stops at the bin's own epoch (coal_EM.cpp:266-278: e = 0 .. min(E - 2, k)); behind
  // it the bin adds nothing to any denominator -- round 3 charged every epoch with the residue of ALL shared counts, which in
  // the flat epochs behind all data pulled the rate down ten times faster than any real build of the reference does
  // (profiles/parity/ref_self_reproducibility_e122.json: epoch 109 of --bins 2,7.95,0.05).  The counts of the epoch's own bins
  // and the later ones, b >= lo(e), are the previous epoch's c_later -- lo(e) = hi(e - 1): the same terms in the same order --,
  // one lane (or slot) over; epoch 0: all counts.
  if (role == 0) {
    const double from_below = dpp_d<WAVE_SHR1, 0xf, true>(0.0, C0[NCH - 1]);
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      const double c_ge = (c == 0) ? (lane == 0 ? c_all : from_below) : C0[c - 1];
      eta_e[c] = dt_e[c] * (kIntegResidue * c_ge);
    }
  }
  bool more_rows[NCH];  // (wave-uniform) some epoch of the slot spans more than three 16-lane rows
  bool third_row = false;  // ... more than two (the steady-state loops read two tail slots per epoch, not three)
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    more_rows[c] = (ballot64(row_x[c] <= row_hi[c]) != 0);
    third_row |= (ballot64(slot2[c] != AP) != 0);
  }
  __syncthreads();
  COLATE_PSTAMP(6)
  if (!TPUT && grp >= NB) return;  // waves without bins retire; later barriers count the remaining waves only
  const bool leader = (grp == 0);
  // The wave that keeps the verdict's history masks in the M-step and writes the verdict and the rates at the end: a
  // wave of the second bin group when there is one (it sits out P1 and P3, so this is off the critical chain; on the
  // leaders it cost 3.5 % at E=23 and 7 % at E=122, profiles/r02_placement.txt), else wave 0.
  // (with two epochs per lane and two bin groups all four waves own epoch work -- the split below -- and wave 0 is the one with
  // slack: its slot's scans need no hand-over from the other owner)
  // (the free mode of the loop -- see kFree below -- loads role A's waves up to barrier 2 and leaves role B's leader, whose
  // epoch values wave 3 computes, the most room: it keeps the history there)
  constexpr bool kFreeBuild = !TPUT && NCH == 1 && WPE == 2;
  const int verdict_wave = (!TPUT && NB >= 2 && NCH != 2) ? (kFreeBuild ? 1 : 2) : 0;
  const bool tracker = (wave == verdict_wave);
  constexpr int erows = EROWS;  // 16-lane rows that hold epochs: a compile-time constant (skipping the cross-row
                                // scan steps behind run-time uniform branches measured slower)
  // One epoch per lane, latency variant: NO BARRIER between the epoch values and the bin terms.  The iteration is a chain
  // of dependent latencies (LDS round trips, DPP steps), not of issue slots, and the hand-over "leader stores S_e -> wait ->
  // barrier -> every bin gathers S_k" was ~150 cycles of it.  Instead every wave of role A runs the cs scan and S_e =
  // exp(-cs_e) itself, in its epoch lanes (each wave has all rates: each runs the M-step), and its bins fetch S_k, 1 - S_k out
  // of the wave's own registers (ds_bpermute: one trip through the LDS crossbar, no store, no barrier), like the rate; the
  // bins' exp(-lambda_k (age - t_k)) fills the scan's stalls.  1 / lambda_k and (t_k + 1/lambda_k) lambda_k come from the
  // bin's own division instead of the gathered per-epoch values (same operands, same bits).  A role-B bin needs nothing but
  // its rate, so the waves of role B run straight from the M-step into their bin terms; role B's epoch values q_e, p_e,
  // beta_e, needed in P3 only, are computed by the wave of its second bin group (wave 3) when there is one, which otherwise
  // idles from its bin terms to the next M-step, and reach both leaders through LDS behind barrier 2.  The rows of S_e etc.
  // are still written (by wave 0) for the readers behind barrier 2 -- and for the two cases in which a role-B bin does need
  // them: the log-likelihood (cs_k) and a last rate of zero; then, and only then, barrier 1 is executed (both conditions are
  // the same in every wave).  The loops compiled per kind of wave run while the last rate is positive and leave at the first
  // iteration that ends with it at zero (the general loop, which tests it at run time, takes over).
  // Measured on one box (gpurun_out/r03z -> profiles/r03_free_mode.txt): 0.990 against 1.000 ms at B = 100 -- and 1.279 against
  // 1.238 ms at B = 400, where two workgroups share a CU and the instructions of the redundant scans (+8 % in total) are no
  // longer free: the mode belongs to the build for batches that leave every workgroup a CU to itself (WPE == 2).
  constexpr bool kFree = kFreeBuild;
  // (this build has registers to spare: the affine scan's "no source lane" constants, see wave_affine_scan)
  double aff_nosrc[4] = {0.0, 0.0, 0.0, 0.0};
  if (kFree) {
    aff_nosrc[0] = (lane & 15) < 1 ? 1.0 : 0.0;
    aff_nosrc[1] = (lane & 15) < 2 ? 1.0 : 0.0;
    aff_nosrc[2] = (lane & 15) < 4 ? 1.0 : 0.0;
    aff_nosrc[3] = (lane & 15) < 8 ? 1.0 : 0.0;
  }
  const int p1b_wave = (kFree && NB >= 2) ? 3 : 1;
  const int nwave_live = 2 * NB;
  (void)nwave_live;

  int my_flags = 0;
  // bit l: the numerator of epoch l of the chunk was below kTinyNum (0 included) / was not 0 in some iteration
  constexpr double kTinyNum = 1e-280;
  // (kept per lane in vector registers -- the smallest and the largest high word the numerator has had, one instruction
  // each per iteration -- and turned into the lane masks in the epilogue: as scalar masks updated by ballots the history
  // cost the wave that keeps it 460 cycles per iteration at two epochs per lane, most of it spill traffic of the masks)
  unsigned tr_min[NCH], tr_max[NCH];
  int tr_noisy = 0;
#pragma unroll
  for (int c = 0; c < NCH; c++) tr_min[c] = 0xffffffffu, tr_max[c] = 0u;
  // any epoch (up to the oldest data) whose rate was, in some iteration, a quotient with a denominator below kNoisyRatio residues (and neither a
  // copy nor clamped to the floor): the reference's own trajectory is then rounding noise of >= 1e-4 per iteration there
  // (its integ residue, see kResolvedRatio) -- harmless for epochs that converge to a fixed point, decisive for the
  // flat epochs behind all data, whose final value records the history (epilogue)
  constexpr double kNoisyRatio = 1.0e6;
  double noisy_thr[NCH];
#pragma unroll
  for (int c = 0; c < NCH; c++)  // (only epochs up to the oldest bin with data: the flat ones behind it do not move at all)
    noisy_thr[c] = (double)(ep_of(c)) <= s_ll[10] ? kNoisyRatio * (dt_e[c] * (kIntegResidue * (s_ll[8] + s_ll[9]))) : 0.0;
  unsigned long long prev_fail = 0;
  unsigned long long ep_mask[NCH];  // the lanes that hold an epoch, per slot
#pragma unroll
  for (int c = 0; c < NCH; c++) ep_mask[c] = ballot64(ep_on[c]);
  const double thr = 1.0 - p.rel_tol;
  double ll = -__builtin_inf(), prev_ll = -__builtin_inf();  // coal.cpp:3685
  int iter = 0;
  const int max_iter = (MODE == 1) ? 1 : p.max_iter;
  double* out_mine = s_out + role * kNumBinArrays * APZ;

#ifdef COLATE_EM_STAMPS
  unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  COLATE_PSTAMP(7)
  unsigned long long st_prev = stamp();
#endif
  // Code placement: the time of an iteration moves by up to 5 % with the position of the loop's code relative to 32-byte
  // instruction-fetch lines (measured: the same loop shifted in 4-byte steps has a period of 8 dwords, best to worst
  // 1.362 .. 1.430 ms at B = 100; profiles/r02_placement.txt) -- and every edit of the prologue used to shift it.  The loop
  // is therefore pinned to a 64-byte boundary plus em_loop_pad() dwords, tuned per instantiation on the GPU.
  // Round 4: EVERY loop compiled per kind of wave is pinned by itself (COLATE_LOOP_ANCHOR in front of each loop below) -- with
  // one anchor in front of them all, an edit of any loop, or of the refresh iteration that sits between them, moved every
  // loop behind it: the same iteration code measured 1.151, 1.161, 1.214 and 1.240 us at E = 122 in four builds of this round
  // that differed in the refresh block only (profiles/r04_placement.txt).  An anchor is executed once per entry of a loop.
  constexpr int kPad = em_loop_pad(MODE, NCH, EROWS, TPUT, WPE);
#define COLATE_PAD_CASE(n) \
  if constexpr (kPad == n) asm volatile(".p2align 6\n\t.rept " #n "\n\ts_nop 0\n\t.endr");
#define COLATE_LOOP_ANCHOR()                                            \
  {                                                                     \
    COLATE_PAD_CASE(0) COLATE_PAD_CASE(1) COLATE_PAD_CASE(2) COLATE_PAD_CASE(3) \
    COLATE_PAD_CASE(4) COLATE_PAD_CASE(5) COLATE_PAD_CASE(6) COLATE_PAD_CASE(7) \
  }
  // The tail model (P3, role B leader) is refreshed in iterations 0, 1, 2, 4, ..., 128 and then every 128th, and held in between
  // as a linear function of S_{e+1} (tools/study/residue_models.cpp, model 8: with a period of up to 256 the final rates of the
  // flat epochs stay inside the range of the reference's real builds; held as a constant, model 6, only up to 32; round 3's
  // powers of two held a constant for up to 512 iterations).
  // free mode: the last rate is positive (after the latest M-step; the same in every wave) -- what the loops compiled per kind of
  // wave assume; always true otherwise (those loops test it themselves where it matters)
  bool last_pos = true;
  // ... and how the loops learn of it: `lim` is the iteration bound of whichever per-kind loop is running, and the M-step sets it to 0
  // when the last rate is not positive -- one s_cselect in the iteration and a plain counted loop around it (as a second loop
  // condition it cost every wave twelve scalar instructions and two branches per iteration)
  int lim = 0;
  // (the throughput variant makes the same assumption in its per-kind loops -- it spares its role-B wave the branches around the
  // cold paths of a zero last rate: 6.18 against 6.33 ms at B = 4096 --; for the other latency builds it measured no gain)
  constexpr bool kAssumeAbsorbing = kFree || TPUT;
  // (the last epoch's lane, per slot: 0 in the slots that do not hold it)
  unsigned long long last_bit[NCH];
#pragma unroll
  for (int c = 0; c < NCH; c++) last_bit[c] = (((E - 1) & (NCH - 1)) == c) ? 1ull << (((E - 1) >> kSlotShift) & 63) : 0ull;
  auto last_rate_positive = [&]() {
    unsigned long long pos = 0;
#pragma unroll
    for (int c = 0; c < NCH; c++) pos |= ballot64(lam_e[c] > 0.0) & last_bit[c];
    return pos != 0ull;
  };
  const unsigned long long last_bit1 = 1ull << ((E - 1) & 63);  // (one epoch per lane: the last epoch's lane)
  if (kFree) last_pos = (ballot64(lam_e[0] > 0.0) & last_bit1) != 0ull;
  if (TPUT) last_pos = last_rate_positive();
  // (the tail model's refresh schedule -- see `tail model` in P3.  More than 64 epochs: the powers of two up to 128 and then
  // every 128th iteration -- fifteen refreshes in 1001 iterations.  Tried at the end of round 4: every 256th from 256 on (twelve)
  // saves 16 us per launch and puts epoch 111 of --bins 2,7.95,0.05 at 8.5e-7, outside the range of the reference's real builds
  // (<= 1.6e-7; tests/test_gpu_parity.py::test_tail_at_122_epochs_...): the study tool's "up to 256" did not survive the GPU.  Up to 64: the powers of two, and every 32nd iteration while the last refresh found an epoch in
  // transition; the flag only changes inside a refresh, so due / next-due stay consistent along a run.)
  auto tail_due = [&](int it) {
#ifdef COLATE_TAIL_LATE256  // (A/B switch: every 256th from 512 on)
    if (NCH >= 2) return it < 128 ? (it & (it - 1)) == 0 : (it < 512 ? (it & 127) == 0 : (it & 255) == 0);
#else
    if (NCH >= 2) return it < 128 ? (it & (it - 1)) == 0 : (it & 127) == 0;
#endif
    return (it & (it - 1)) == 0 || (!tail_trivial_prev && (it & 31) == 0);
  };
  auto tail_next_due = [&](int it) {  // first due iteration >= it
    if (it <= 1) return it;
    const int p2 = 1 << (32 - __builtin_clz((unsigned)(it - 1)));
#ifdef COLATE_TAIL_LATE256
    if (NCH >= 2) return it <= 128 ? p2 : (it <= 512 ? ((it + 127) & ~127) : ((it + 255) & ~255));
#else
    if (NCH >= 2) return it <= 128 ? p2 : ((it + 127) & ~127);
#endif
    const int m32 = (it + 31) & ~31;
    return (!tail_trivial_prev && m32 < p2) ? m32 : p2;
  };
  auto iteration = [&](auto role_c, auto leader_c, auto ll_c, auto track_c, auto refresh_c, auto p1_c) __attribute__((always_inline)) -> bool {
    COLATE_STAMP(7)
    // (compile-time role / leadership / "no log-likelihood needed" in the steady-state loops below; -1 = run-time value)
    constexpr int kRole = decltype(role_c)::value, kLeader = decltype(leader_c)::value, kNeedLL = decltype(ll_c)::value;
    constexpr int kTrack = decltype(track_c)::value;  // this wave keeps the verdict's history masks (1), does not (0)
    constexpr int kRefresh = decltype(refresh_c)::value;  // the tail model is refreshed in this iteration (1), is not (0), -1: if due
    constexpr bool kSteady = (kNeedLL >= 0);          // one of the loops compiled per kind of wave: no wave with `more_rows` / `third_row` runs it
    const int ROLE = kRole < 0 ? role : kRole;
    // kLeader: 0 no epoch work; 1 (or the run-time `leader`): all epochs; 2 / 3: the 65..128-epoch SPLIT -- this wave owns
    // slot 0 / slot 1 (the even / the odd epochs) of its role's epoch work and the wave of the other bin group owns the other
    // one (both run the M-step and the scans for all epochs: same operations, same bits)
    const bool LEADER = kLeader < 0 ? leader : (kLeader != 0);
    constexpr bool kSplit = (kLeader >= 2);
    constexpr int kOwn = (kLeader == 3) ? 1 : 0;
    static_assert(!kSplit || NCH == 2, "the split is for two epochs per lane");
    auto own = [&](int c) { return !kSplit || c == kOwn; };
    // which epoch values this wave computes in P1 -- bit 0: role A's (cs, S, 1 - S), bit 1: role B's (q, p, beta, ...); -1:
    // decided at run time
    constexpr int kP1 = decltype(p1_c)::value;
    const bool P1A = kP1 < 0 ? (ROLE == 0 && (kFree || LEADER)) : ((kP1 & 1) != 0);
    const bool P1B = kP1 < 0 ? (kFree ? wave == p1b_wave : (LEADER && ROLE == 1)) : ((kP1 & 2) != 0);
    constexpr bool CROSS = kFree;  // the bins divide for themselves and fetch S_k from the wave's own registers
    const bool need_ll = kNeedLL < 0 ? ((MODE == 1) || (iter >= p.min_iter) || (iter == max_iter - 1) || p.ll_trace != nullptr) : (kNeedLL != 0);
    // ============================================================ P1: epoch values (ROLE leaders)
    double q_e[NCH], p_e[NCH], beta_e[NCH], S_e[NCH], omS_e[NCH], cs_e[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) q_e[c] = p_e[c] = beta_e[c] = S_e[c] = omS_e[c] = cs_e[c] = 0.0;
    if ((P1A || P1B) && !COLATE_ABL_HAS(15)) {
      double x_e[NCH];
#pragma unroll
      for (int c = 0; c < NCH; c++) x_e[c] = lam_e[c] * dt_e[c];
      if (P1A) {
        // cs_e = sum_{j<e} lambda_j dt_j (coal_EM.cpp:100-103): a lane's epochs are consecutive, so one wave scan of the
        // lanes' totals plus the local prefix (split: every owner runs it in full -- the rates of all epochs are in every
        // wave -- and keeps its own slot)
        double loc[NCH];
        loc[0] = 0.0;
#pragma unroll
        for (int c = 1; c < NCH; c++) loc[c] = loc[c - 1] + x_e[c - 1];
        const double tot = (NCH == 1) ? x_e[0] : loc[NCH - 1] + x_e[NCH - 1];  // (no `0.0 + x` on the chain)
#if COLATE_ABL_HAS(11)
        const double incl = tot * 7.0;
#else
        const double incl = wave_prefix_sum(tot, erows);
#endif
        const double excl = dpp_d<WAVE_SHR1, 0xf, true>(0.0, incl);
#pragma unroll
        for (int c = 0; c < NCH; c++) cs_e[c] = (c == 0) ? excl : excl + loc[c];
      }
      COLATE_STAMP(8)
      if (P1A) {
#pragma unroll
        for (int c = 0; c < NCH; c++) {
          if (!own(c)) continue;
          const int e = ep_of(c);
#if COLATE_ABL_HAS(3)
          S_e[c] = 1.0 - cs_e[c] * 1e-3;
          omS_e[c] = cs_e[c] * 1e-3;
#else
          S_e[c] = em::em_exp_om_t<true>(cs_e[c], &omS_e[c], s_exptab);  // exp(-cs_e); omS = 1 - S_e = sum_{j<e} exp(A_ep[j])
#endif
          // (no `if (ep_on)`: the rows are EPAD wide, entries beyond E are written with whatever the idle lanes hold and
          // never read for an epoch; a not-taken skip branch costs a lone wave 8 cycles, its exec bookkeeping 10 more)
          if (!kFree || LEADER) {  // (free: every wave of role A has them in registers; wave 0 writes the rows)
            s_ep[G_CS * EPAD + e] = cs_e[c];
            s_ep[G_S * EPAD + e] = S_e[c];
            s_ep[G_PW * EPAD + e] = omS_e[c];
          }
        }
      }
      if (P1B) {
#pragma unroll
        for (int c = 0; c < NCH; c++) {
          if (!own(c)) continue;
          const int e = ep_of(c);
#if COLATE_ABL_HAS(7)
          const double inv = 2.0e4 - lam_e[c];
#else
          const double inv = em::em_rcp_ieee(lam_e[c]);
#endif
          const bool valid = vstat[c] && (lam_e[c] > 0);
          // exp(-cumsum[i+1] + cumsum[i]) of coal_EM.cpp:120, taken as exp(-lambda_e dt_e): the two arguments
          // differ by the rounding of cumsum (<= ulp(cs)/2), and role B then needs no scan at all.  No branch on the last
          // epoch (coal_EM.cpp:136-141: p = 1, beta = t + 1/lambda): with q = 0 there the same two lines give exactly that
          // (tn_e = 0); lanes beyond E compute something that is never stored.
#if COLATE_ABL_HAS(4)
          const double qx = 1.0 - x_e[c];
#else
          const double qx = em::em_exp_t(-x_e[c], s_exptab);
#endif
          q_e[c] = (e < E - 1) ? qx : 0.0;
          p_e[c] = valid ? 1.0 - q_e[c] : 0.0;                                  // exp(A_ep + cs), coal_EM.cpp:119
          beta_e[c] = valid ? (t_e[c] + inv) - (tn_e[c] + inv) * q_e[c] : 0.0;  // exp(B_ep + cs), coal_EM.cpp:120
          s_ep[G_LAM * EPAD + e] = lam_e[c];
          if (!CROSS) {  // (crossed: the bins divide for themselves)
            s_ep[G_INV * EPAD + e] = inv;
#if COLATE_ABL_HAS(7)
            s_ep[G_XA * EPAD + e] = (t_e[c] + inv) * lam_e[c];
#else
            s_ep[G_XA * EPAD + e] = em::em_div_known_rcp(t_e[c] + inv, inv, lam_e[c]);  // (t + 1/lambda)/(1/lambda), coal_EM.cpp:204
#endif
          }
          s_ep[G_P * EPAD + e] = p_e[c];
          s_ep[G_BETA * EPAD + e] = beta_e[c];
          if (NCH >= 2 || kFree) s_ep[G_Q * EPAD + e] = q_e[c];  // (the other slot's owner needs it for the affine scan; the tail model; free: role B's leader)
        }
      }
    }
    // The rate of this lane's own bin's epoch, lambda_k, is known to every wave before barrier 1 (each wave runs the
    // M-step itself): fetching it from the wave's own registers (ds_bpermute, issued here so that its latency falls into
    // the wait the barrier needs anyway) lets the bin's exp(-lambda_k (age - t_k)) start right behind the barrier instead
    // of behind the LDS gather of the other per-epoch values.  Same value, so nothing changes but the time.
    double lk_pre = 0.0;
    if (!TPUT) {
      const int kq = bs0.kb < E ? bs0.kb : 0;
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        const int lo = __builtin_amdgcn_ds_bpermute(((kq >> kSlotShift) & 63) << 2, __double2loint(lam_e[c]));
        const int hi = __builtin_amdgcn_ds_bpermute(((kq >> kSlotShift) & 63) << 2, __double2hiint(lam_e[c]));
        if (NCH == 1 || (kq & (NCH - 1)) == c) lk_pre = __hiloint2double(hi, lo);
      }
      asm volatile("" : "+v"(lk_pre));  // (keeps the fetch on this side of the barrier)
    }
    // free: the bin's S_k, 1 - S_k out of this wave's own epoch lanes, and its exp(-lambda_k (age - t_k)) -- outside the
    // `live` branch of the bin terms, so that it shares a basic block with the scan and fills its stalls
    double Sk_pre = 0.0, PWk_pre = 0.0, qd_pre = 0.0;
    if (kFree && ROLE == 0) {
      const int kq = bs0.kb < E ? bs0.kb : 0;
      const int s_lo = __builtin_amdgcn_ds_bpermute((kq & 63) << 2, __double2loint(S_e[0]));
      const int s_hi = __builtin_amdgcn_ds_bpermute((kq & 63) << 2, __double2hiint(S_e[0]));
      const int o_lo = __builtin_amdgcn_ds_bpermute((kq & 63) << 2, __double2loint(omS_e[0]));
      const int o_hi = __builtin_amdgcn_ds_bpermute((kq & 63) << 2, __double2hiint(omS_e[0]));
      Sk_pre = __hiloint2double(s_hi, s_lo);
      PWk_pre = __hiloint2double(o_hi, o_lo);
      qd_pre = em::em_exp_t(-(lk_pre * bs0.da), s_exptab);
    }
    COLATE_STAMP(9)
    // the last epoch absorbs (lambda_{E-1} > 0) in every valid run; the reference asserts it only
    // for bins inside the last epoch (coal_EM.cpp:351)
    // (from this wave's own copy of the rates -- every wave runs the M-step --, not from LDS: the read and its wait were the
    // first thing behind barrier 1 in role B's waves)
    // (the free-mode loops compiled per kind of wave run only while it does: the loop conditions below)
    bool absorbing = kAssumeAbsorbing && kSteady;
    if (!(kAssumeAbsorbing && kSteady)) {
      const int cl = (E - 1) & (NCH - 1), ll_ = (E - 1) >> kSlotShift;
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        const unsigned long long pos = ballot64(lam_e[c] > 0.0);
        if (NCH == 1 || c == cl) absorbing = (pos >> ll_) & 1ull;
      }
    }
    // ---- barrier 1: epoch values visible (free: only where a role-B bin reads them -- see above; uniform over the workgroup)
    if (!kFree || need_ll || !absorbing) __syncthreads();
    COLATE_STAMP(0)
    // ============================================================ P2: bin terms (own bins, own ROLE)
    // (cross: free mode -- Sk_in, PWk_in, qd_in are this bin's S_k, 1 - S_k, exp(-lambda_k (age - t_k)) from above)
    auto bin_terms = [&](const BinStat& bs, const double lk_own, const bool have_lk, const bool cross, const double Sk_in, const double PWk_in, const double qd_in) {
      const double a_b = bs.a_b, cnt = bs.cnt, tk = bs.tk, tkn = bs.tkn, dtk = bs.dtk, da = bs.da, db = bs.db;
      const double f1 = bs.f1, f2 = bs.f2, f4 = bs.f4, f8 = bs.f8;
      const int kb = bs.kb, pos = bs.pos;
      const bool live = bs.live, last_bin = bs.last_bin, is_tail = bs.is_tail;
      const int wslot = 2 * (pos >> 6) + ROLE;  // entry of this (bin group, ROLE) in s_fail / s_ll: the latency variant's wave
      double o_w = 0, o_N = 0, o_D = 0, llp = 0.0;
      bool fail = false;
      // the lanes whose normaliser failed, as scalar mask arithmetic where the branch-free paths know it (through the bool it took a
      // v_cndmask and a v_cmp to get the mask back); `fail` itself is for the paths behind branches
      unsigned long long fail_direct = 0;
      bool have_fail_direct = false;
      if (ROLE == 0 && !COLATE_ABL_HAS(12)) {
        // ---- EM_shared, coal_EM.cpp:198-210, 263-287, WITHOUT exec-masked branches on `live` and `finite_pos(Sig)`: every lane
        // computes, the three results are selected at the end.  The two branches cost the wave 14 scalar / branch instructions per
        // iteration -- each an issue slot like an FP64 instruction -- against 6 v_cndmask here; a lane without data computes on
        // zeros (its gathers read epoch E's column, inside the rows) and is dropped.
        const double lk = have_lk ? lk_own : s_ep[G_LAM * EPAD + kb];
        const bool lpos = lk > 0;
        // 1 / lambda_k: the per-epoch value of P1, or the bin's own division (the same operation on the same operand).
        // A rate of 0 (coal_EM.cpp:198-210 skips such an epoch's terms): with the bin's own division the reciprocal is taken as 0,
        // and Wp = S_k (1 - exp(-0)) = 0, Vp = X * 0 * S_k = 0 come out by themselves -- one select instead of two.
        const double ik = cross ? (lpos ? em::em_rcp_ieee(lk) : 0.0) : s_ep[G_INV * EPAD + kb];
        const double Sk = cross ? Sk_in : s_ep[G_S * EPAD + kb], PWk = cross ? PWk_in : s_ep[G_PW * EPAD + kb];
        const double Xak = cross ? em::em_div_known_rcp(tk + ik, ik, lk) : s_ep[G_XA * EPAD + kb];  // (t_k + 1/lambda)/(1/lambda), coal_EM.cpp:204
#if COLATE_ABL_HAS(5)
        const double qd = 1.0 - lk * da;
#else
        const double qd = cross ? qd_in : em::em_exp_t(-(lk * da), s_exptab);  // exp(-cumsum(age) + cumsum(t_k)): same up to the rounding of cumsum
#endif
#if COLATE_ABL_HAS(8)
        const double Y = (a_b + ik) * lk;
#else
        const double Y = em::em_div_known_rcp(a_b + ik, ik, lk);  // (age + 1/lambda)/(1/lambda), coal_EM.cpp:204
#endif
        const double Wp = (cross || lpos) ? Sk * (1.0 - qd) : 0.0;
        const double X = Xak - Y * qd;
        const double Vp = (cross || lpos) ? X * ik * Sk : 0.0;  // (not cross: the gathered (t_k + 1/lambda)/(1/lambda) of a zero rate is a NaN)
        const double Sig = PWk + Wp;
        const bool fin = finite_pos(Sig), ok = live && fin;
        // (the reciprocal is the one thing that is not finite where the bin drops out -- Sig = 0 behind rates of zero --: with
        // 0 in its place all three results are the exact zeros the branch used to leave, for one select instead of three)
        const double r = ok ? em::em_rcp(Sig) : 0.0;
        const double nk = Wp * r;
        double dk = Vp * r + (-tk * nk);
        dk = __builtin_fmax(dk, 0.0);
        o_w = cnt * r;
        o_N = cnt * nk;
        o_D = cnt * dk;
        fail_direct = ballot64(live) & ~ballot64(fin);
        have_fail_direct = true;
        if (need_ll) {
          COLATE_COLD();
          llp = ok ? cnt * em::em_log(Sig) : 0.0;
        }
      } else if (ROLE == 1 && !COLATE_ABL_HAS(12)) {
        // ---- EM_notshared, coal_EM.cpp:330-357, 435-460, likewise without a branch on `live`: a lane without data has a count of
        // (+)0 and finite factors -- every static of a lane beyond the data is 0, exp never returns a NaN, and the selects on
        // `lambda_k > 0` keep 1 / lambda_k = inf out -- so its three products are the +0.0 the branch used to leave there.
        const double lk = have_lk ? lk_own : s_ep[G_LAM * EPAD + kb];
        // 1 / lambda_k: the per-epoch value of P1, or the bin's own division (the same operation on the same operand)
        const double ik = cross ? em::em_rcp_ieee(lk) : s_ep[G_INV * EPAD + kb];
        const bool lpos = lk > 0;
        // -cumsum(age) at the merged grid (coal_EM.cpp:178-181): only the log-likelihood needs it
        auto neg_cs_age = [&]() {
          const double ck = s_ep[G_CS * EPAD + kb];
          const double ck1 = ck + lk * da;
          return -(ck1 + lk * (a_b - a_b));  // (second copy of `age` in the merged grid)
        };
        // every lane takes the general path (db = 0 in the last epoch, so it is harmless there); the few bins
        // beyond the start of the last epoch are put right afterwards, behind a wave-uniform test
#if COLATE_ABL_HAS(6)
        const double u = 1.0 - lk * db;
#else
        const double u = em::em_exp_t(-(lk * db), s_exptab);  // exp(-cumsum(t_{k+1}) + cumsum(age)), likewise
#endif
        const double pn = 1.0 - u;  // (a rate of 0: u = exp(-0) = 1 exactly, so this is the 0 a select would put here)
        const double bn = lpos ? (a_b + ik) - (tkn + ik) * u : 0.0;
        {  // the last epoch absorbs: normaliser = exp(-cs(age)) * ((1 - u) + u) = exp(-cs(age))
          double dk = bn + (-tk * pn + dtk * (1.0 - pn));
          dk = __builtin_fmax(dk, 0.0);
          o_w = cnt * u;
          o_N = cnt * pn;
          o_D = cnt * dk;
          if (need_ll) {
            COLATE_COLD();
            llp = live ? cnt * neg_cs_age() : 0.0;
          }
        }
        if (__builtin_expect(!absorbing, 0)) {  // last rate is 0: the mass beyond t_{k+1} is 1 - S_{E-1}/S_{k+1}
          COLATE_COLD();
          o_w = o_N = o_D = llp = 0.0;
          if (live) {
            const double Gk1 = 1.0 - em::em_exp_t(-s_ep[G_CS * EPAD + E - 1] + s_ep[G_CS * EPAD + kb + 1], s_exptab);
            const double SigN = pn + u * Gk1;
            if (finite_pos(SigN)) {
              const double rr = 1.0 / SigN;
              const double nk = pn * rr;
              double dk = bn * rr + (-tk * nk + dtk * (1.0 - nk));
              dk = __builtin_fmax(dk, 0.0);
              o_w = cnt * (u * rr);
              o_N = cnt * nk;
              o_D = cnt * dk;
              llp = cnt * (neg_cs_age() + em::em_log(SigN));
            } else {
              fail = true;
            }
          }
        }
        if (__builtin_expect(ballot64(last_bin && live) != 0, 0)) {
          COLATE_COLD();
          if (last_bin && live) {  // bin beyond the start of the last epoch, coal_EM.cpp:350-357
            if (!lpos) my_flags |= COLATE_FLAG_NAN;  // reference: assert(coal_rate_e > 0)
            double dk = (a_b + ik) - tk;
            dk = __builtin_fmax(dk, 0.0);
            o_w = 0.0;
            o_N = cnt;
            o_D = cnt * dk;
            fail = false;
            llp = need_ll ? cnt * neg_cs_age() : 0.0;
          }
        }
      }
      COLATE_STAMP(1)
      // bins whose normaliser failed (coal_EM.cpp:288-292, 461-465) drop out of the static counts
      // (s_fail: one byte per wave, the four bin groups of a role in one 32-bit word for the leader's single read)
      unsigned char* s_failb = reinterpret_cast<unsigned char*>(s_fail);
      const unsigned long long fail_mask = have_fail_direct ? fail_direct : ballot64(fail);
      if (have_fail_direct) fail = (fail_mask >> lane) & 1ull;  // (only read behind the rare branches below)
      if (TPUT) {  // (a wave serves several groups: publish every time)
        s_cfail[ROLE * APZ + pos] = fail ? cnt : 0.0;
        if (lane == 0) s_failb[ROLE * 4 + (pos >> 6)] = fail_mask ? 1 : 0;
      } else {
        // prev_fail: the lanes this wave published a failed bin for in the previous iteration
        const unsigned long long touch = fail_mask | prev_fail;
        if (__builtin_expect(touch != 0, 0)) {  // (uniform) something to publish, or to clear from last time
          COLATE_COLD();
          if ((touch >> lane) & 1ull) s_cfail[ROLE * APZ + pos] = fail ? cnt : 0.0;
          if ((fail_mask != 0) != (prev_fail != 0)) {  // the per-wave flag only when it changes
            if (lane == 0) s_failb[ROLE * 4 + (pos >> 6)] = fail_mask ? 1 : 0;
          }
          prev_fail = fail_mask;
        }
      }
      // sums over the run of equal-epoch bins inside each 16-lane row, left to right
#define COLATE_SEG_STEP(CTRL, F)                              \
  o_w = em::fma_(dpp_d<CTRL, 0xf, true>(0.0, o_w), F, o_w);   \
  o_N = em::fma_(dpp_d<CTRL, 0xf, true>(0.0, o_N), F, o_N);   \
  o_D = em::fma_(dpp_d<CTRL, 0xf, true>(0.0, o_D), F, o_D);
#if !COLATE_ABL_HAS(2)  // ablation 2: no segmented reduce
      COLATE_SEG_STEP(ROW_SHR1, f1)
      COLATE_SEG_STEP(ROW_SHR2, f2)
      COLATE_SEG_STEP(ROW_SHR4, f4)
      COLATE_SEG_STEP(ROW_SHR8, f8)
#endif
#undef COLATE_SEG_STEP
      {  // (no branch: lanes that are not the tail of a run write to the spare entry AP + 1, which nobody reads)
        const int tpos = is_tail ? pos : AP + 1;
        out_mine[O_W * APZ + tpos] = o_w;
        out_mine[O_N * APZ + tpos] = o_N;
        out_mine[O_D * APZ + tpos] = o_D;
      }
      if (need_ll) {
        COLATE_COLD();
        const double tot = readlane_d(wave_prefix_sum(llp), 63);
        if (lane == 0) s_ll[wslot] = tot;
      }
    };
    if (TPUT) {
      for (int g = 0; g < NB; g++) {  // this ROLE's bin groups, one after the other
        const int gpos = g * kWave + lane, fl = s_bflags[gpos], gbin = nzlo + gpos;
        const bool inr = fl & BF_INRANGE;
        BinStat b;
        b.kb = inr ? (fl >> BF_KB_SHIFT) : 0;
        b.pos = gpos;
        b.a_b = inr ? s_age[gbin] : 0.0;
        b.cnt = inr ? s_cnt[ROLE * APZ + gbin] : 0.0;
        b.tk = s_t[b.kb];
        b.tkn = (b.kb < E - 1) ? s_t[b.kb + 1] : 0.0;
        b.dtk = (b.kb < E - 1) ? b.tkn - b.tk : 0.0;
        b.da = b.a_b - b.tk;
        b.db = (b.kb < E - 1) ? b.tkn - b.a_b : 0.0;
        b.f1 = (fl & BF_F1) ? 1.0 : 0.0;
        b.f2 = (fl & BF_F2) ? 1.0 : 0.0;
        b.f4 = (fl & BF_F4) ? 1.0 : 0.0;
        b.f8 = (fl & BF_F8) ? 1.0 : 0.0;
        b.live = inr && b.cnt > 0;
        b.last_bin = (b.kb == E - 1);
        b.is_tail = fl & BF_TAIL;
        bin_terms(b, 0.0, false, false, 0.0, 0.0, 0.0);
      }
    } else {
      bin_terms(bs0, lk_pre, true, CROSS, Sk_pre, PWk_pre, qd_pre);
    }
    COLATE_STAMP(2)
    __syncthreads();  // ---- barrier 2: per-bin tails visible
    if constexpr (em_loop_pad2(MODE, NCH, TPUT) >= 0) {
      // second placement point (these few s_nop are executed every iteration; they pay for themselves: 1.391 -> 1.378 ms)
      static_assert(em_loop_pad2(MODE, NCH, TPUT) < 0 || em_loop_pad2(MODE, NCH, TPUT) == 4, "add the asm for another pad");
      asm volatile(".p2align 5\n\t.rept 4\n\ts_nop 0\n\t.endr");
    }
    COLATE_STAMP(3)
    // ============================================================ P3: per-epoch sums (ROLE leaders)
    if (LEADER && !COLATE_ABL_HAS(13)) {
      // did a bin of this ROLE fail this iteration? (entries of retired waves stay 0; loaded with the
      // tails: one LDS wait; fixed count -- a runtime-bounded loop here compiles to a vectorised monster)
      const int anyf = s_fail[ROLE];  // (four bytes: this role's bin groups)
      double w[NCH], oN[NCH], oD[NCH];
#pragma unroll
      for (int c = 0; c < NCH; c++) w[c] = oN[c] = oD[c] = 0.0;
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        // split: besides its own slot's three sums a wave needs the weights (only) of the other slot's epochs: its scan
        // runs over all epochs
        const bool mine = own(c);
#if COLATE_ABL_HAS(1)  // ablation: no tail loads
        const double w0 = 1e-3 * lane, w1 = 0, w2 = 0, n0 = 1e-3, n1 = 0, n2 = 0, d0 = 1.0, d1 = 0, d2 = 0;
#else
        const double w0 = out_mine[O_W * APZ + slot0[c]], w1 = out_mine[O_W * APZ + slot1[c]];
        // (a run of equal-epoch bins reaching into a third row: never in a wave that runs a steady-state loop; adding
        // the zero entry there would give the same sums)
        const double w2 = kSteady ? 0.0 : out_mine[O_W * APZ + slot2[c]];
        double n0 = 0, n1 = 0, n2 = 0, d0 = 0, d1 = 0, d2 = 0;
        if (mine) {
          n0 = out_mine[O_N * APZ + slot0[c]], n1 = out_mine[O_N * APZ + slot1[c]];
          d0 = out_mine[O_D * APZ + slot0[c]], d1 = out_mine[O_D * APZ + slot1[c]];
          n2 = kSteady ? 0.0 : out_mine[O_N * APZ + slot2[c]];
          d2 = kSteady ? 0.0 : out_mine[O_D * APZ + slot2[c]];
        }
#endif
        if (ROLE == 0 && mine) {  // the shared leader also needs the not-shared leader's p_e, beta_e
          p_e[c] = s_ep[G_P * EPAD + ep_of(c)];
          beta_e[c] = s_ep[G_BETA * EPAD + ep_of(c)];
        }
        if (ROLE == 1 && mine && !P1B) {  // (free: role B's epoch values are wave 3's work)
          q_e[c] = s_ep[G_Q * EPAD + ep_of(c)];
          p_e[c] = s_ep[G_P * EPAD + ep_of(c)];
          beta_e[c] = s_ep[G_BETA * EPAD + ep_of(c)];
        }
        w[c] = kSteady ? (w0 + w1) : (w0 + w1) + w2;  // (x + 0.0 == x for the non-negative sums here: same bits)
        oN[c] = kSteady ? (n0 + n1) : (n0 + n1) + n2;
        oD[c] = (kSteady ? (d0 + d1) : (d0 + d1) + d2);
        if (ROLE == 0) oD[c] = oD[c] + eta_e[c];  // (the shared residue joins the own-epoch sum: off the scan's dependency chain)
        if (!kSteady && __builtin_expect(more_rows[c], 0)) for (int r = row_x[c]; r <= row_hi[c]; r++) {
          COLATE_COLD();
          int slot = r * 16 + 15;
          if (slot > seg_hi[c] - 1) slot = seg_hi[c] - 1;
          w[c] += out_mine[O_W * APZ + slot];
          oN[c] += out_mine[O_N * APZ + slot];
          oD[c] += out_mine[O_D * APZ + slot];
        }
      }
      COLATE_STAMP(10)
      // counts of this ROLE's bins in LATER epochs, minus those whose normaliser failed this iteration
      double Cn[NCH];
#pragma unroll
      for (int c = 0; c < NCH; c++) Cn[c] = C0[c];
      {
        if (__builtin_expect(anyf != 0, 0)) {
          COLATE_COLD();
#pragma unroll
          for (int c = 0; c < NCH; c++) {
            double fs = 0.0;
            for (int q = seg_hi[c] > 0 ? seg_hi[c] : 0; ep_on[c] && q < nzhi - nzlo; q++) {
              if (s_kb[nzlo + q] > ep_of(c)) fs += s_cfail[ROLE * APZ + q];
            }
            Cn[c] -= fs;
          }
        }
      }
      double Npart[NCH], Dpart[NCH];
      if (ROLE == 0) {
        // RS = sum c r over the shared bins of LATER epochs (suffix sums over epochs)
        double RSn[NCH], ls[NCH];
        ls[NCH - 1] = 0.0;
#pragma unroll
        for (int c = NCH - 2; c >= 0; c--) ls[c] = ls[c + 1] + w[c + 1];
        {
          const double tot = (NCH == 1) ? w[0] : ls[0] + w[0];
#if COLATE_ABL_HAS(9)
          const double sR = tot * 3.0;
#else
          const double sR = wave_suffix_sum(tot, lane, erows);
#endif
          const double excl = dpp_d<WAVE_SHL1, 0xf, true>(0.0, sR);
#pragma unroll
          for (int c = 0; c < NCH; c++) RSn[c] = (c == NCH - 1) ? excl : excl + ls[c];
        }
        COLATE_STAMP(11)
#pragma unroll
        for (int c = 0; c < NCH; c++) {
          Npart[c] = Dpart[c] = 0.0;
          if (!own(c)) continue;
          const int e = ep_of(c);
          const double W = S_e[c] * p_e[c];                    // exp(A_ep)
          const double VW = S_e[c] * beta_e[c] - t_e[c] * W;   // exp(B_ep) - t_e exp(A_ep)
          const double PWn = omS_e[c] + W;                     // sum_{j<=e} exp(A_ep[j])
          Npart[c] = W * RSn[c] + oN[c];
          // sum_b c_b (exp(B_e - Z_b) - t_e num_e(b) + dt_e integ_e(b)) over the shared bins of later epochs;
          // the reference clamps every bin's term at 0 (coal_EM.cpp:277), here the (non-negative) sums are.
          // (No branch on the last epoch: dt_e = 0 and RS = 0 there, the term vanishes by itself.)
          double integ = Cn[c] - PWn * RSn[c];  // sum_b c_b (1 - r_b PW_{e+1})
          integ = __builtin_fmax(integ, 0.0);
          double dsh = VW * RSn[c] + dt_e[c] * integ;
          dsh = __builtin_fmax(dsh, 0.0);
          Dpart[c] = dsh + oD[c];
          (void)e;
        }
      } else {
        // forward recurrence T_{e+1} = q_e T_e + h_e, T_0 = 0
        // (T_e = sum over not-shared bins b in EARLIER epochs of c_b u_b/Sig_b * S_e/S_{k_b+1})
        double T[NCH];
        {
          // a lane's map T -> T after its epochs, composed locally; one wave scan of the lanes' maps; the local steps again
          double qa[NCH], wb[NCH];
#pragma unroll
          for (int c = 0; c < NCH; c++) {
            if (kSplit && !own(c)) q_e[c] = s_ep[G_Q * EPAD + ep_of(c)];  // (the other owner's value, same bits)
            qa[c] = ep_on[c] ? q_e[c] : 1.0;
            wb[c] = ep_on[c] ? w[c] : 0.0;
          }
          double a = qa[0], b = wb[0];
#pragma unroll
          for (int c = 1; c < NCH; c++) {
            b = em::fma_(qa[c], b, wb[c]);
            a = a * qa[c];
          }
#if !COLATE_ABL_HAS(9)
          wave_affine_scan(a, b, erows, kFree ? aff_nosrc : nullptr);
#endif
          // (b of lane l: T after the epochs of lanes 0..l, starting from T = 0)
          T[0] = dpp_d<WAVE_SHR1, 0xf, true>(0.0, b);  // T at the lane's first epoch (lane 0: 0)
#pragma unroll
          for (int c = 1; c < NCH; c++) T[c] = em::fma_(qa[c - 1], T[c - 1], wb[c - 1]);
        }
        COLATE_STAMP(12)
        // ---- tail model: what the reference's `integ` recurrence makes of the mass beyond t_{e+1}.
        // For a not-shared bin b the reference normalises its terms n_j = exp(A_j - Z_b) with a log-sum-exp fold over the
        // epochs (coal_EM.cpp:345-349) and then runs integ = 1 - n_k - n_{k+1} - ... (coal_EM.cpp:445-449).  In IEEE double
        // (i) a fold step whose increment is below half an ulp of the running value Z_b (|Z_b| ~ cs(age)) leaves it unchanged:
        // those terms are missing from the normaliser but are still subtracted, so integ_ref = x_be - D_b with D_b the mass
        // of the absorbed terms (a NEGATIVE bias of ~ulp(cs(age))/2, which accumulates over the iterations in the flat epochs
        // behind all data); (ii) the roundings of the chain (2^-54 per step while integ is in [0.5, 1)) and of the fold
        // (ulp(Z_b)/2 per step) add an error eps_b of standard deviation s_b that is made while integ is large and stays
        // FROZEN from there on: integ(b, e) = x_be - D_b + eps_b for all later epochs; (iii) the clamp `integ > 0 ? integ - n
        // : 0` cuts it at zero for good.  eps_b itself depends on the last bits of every exp() and log(); its expectation
        // does not:
        //     integ_ref(b, e) ~ E max(0, x_be - D_b + eps_b) = s_b H((x_be - D_b) / s_b),   H(z) = phi(z) + z Phi(z),
        // x_be = S_{e+1} / S(age_b).  The correction to the exact mass q_e T_e = sum_b c_b x_be of the epoch is therefore
        //     R_e = sum_{b: k_b < e} c_b s_b (H(z_be) - x_be / s_b) = sum w_b (Hneg(z_be) - d_b),
        // w_b = c_b s_b, d_b = D_b / s_b, z_be = S_{e+1} a_b - d_b, a_b = 1 / (S(age_b) s_b), Hneg(z) = H(-z) = H(z) - z.
        // Three regimes per epoch: every bin far above its threshold (z > 8: R_e = -sum c_b D_b), every bin far below (x <<
        // s: R_e = sum w_b H(-d_b) - S_{e+1} sum c_b / S(age_b)), and the two to eight epochs in between, where the sum
        // over the bins is evaluated term by term.  Round 3 used max(x_be - D_b, 0.4 s_b) -- the same at both ends, but up to
        // 0.4 s_b too much wherever D_b exceeds s_b, i.e. for every bin with absorbed terms -- and held R_e for up to 512
        // iterations; against five real builds of the reference (tools/ref_self_reproducibility.py) that put epoch 109 of
        // --bins 2,7.95,0.05 at 6e-6 where the builds print 3e-5 .. 5e-5.  R_e depends on the tail epochs' own rates through
        // S_{e+1}, which drift for hundreds of iterations: it is refreshed in iterations 0, 1, 2, 4, ..., 128 and then every
        // 128th (tail_due above) together with its derivative -sum w_b a_b Q(z_be), and held in between as A + B S_{e+1}
        // (tools/study/residue_models.cpp, model 8: every refresh period up to 256 ends inside the builds' range; a
        // held constant, model 6, only up to 32).
        if constexpr (kRefresh != 0) {
          const bool due = kRefresh > 0 || tail_due(iter);
          if (__builtin_expect(due, kRefresh > 0)) {
            if (kRefresh < 0) COLATE_COLD();
            const int k_old = (int)s_ll[10];  // epoch of the oldest bin with data (-1: none)
            const int e_o1 = (k_old + 1 < E - 1) ? (k_old + 1 > 0 ? k_old + 1 : 0) : E - 1;
            const double S_old1 = s_ep[G_S * EPAD + e_o1];  // no bin's S(age) is below this
            double S1[NCH], We[NCH];
            bool alive_l[NCH], dead_l[NCH];
            unsigned long long between[NCH], small_w[NCH], any_between = 0;
#pragma unroll
            for (int c = 0; c < NCH; c++) {
              const int e = ep_of(c);
              const bool has = e < E - 1;  // (the last epoch has no dt_e * integ term)
              // (split: the other slot's p_e comes from the row its owner has written; the same bits as that owner's register)
              const double pe = (kSplit && !own(c)) ? s_ep[G_P * EPAD + e] : p_e[c];
              We[c] = ep_on[c] ? s_ep[G_S * EPAD + e] * pe : 0.0;  // the fold's term of epoch e per unit S(age)
              S1[c] = has ? s_ep[G_S * EPAD + e + 1] : 1.0;
              // Behind all data (the flat epochs, where a bias accumulates over the iterations): every bin is far above its
              // threshold while S_{e+1} >= 1e-11 (x_be >= S_{e+1}; D_b + 8 s_b < 1e-12: D_b is at most ~20 half-ulps of
              // cs <= 32, s_b a few of them).  In an epoch with data the correction only shifts the fixed point by its ratio to
              // the mass: nothing to do while that mass is >= 1e-8 per unit count of the earlier bins (and no bin is below
              // S_{e+1} = 1e-14).  Every bin is far below its threshold (x_be < 1e-3 s_b) once S_{e+1} / S(oldest age) < 4e-20
              // (s_b >= 0.7 2^-54) -- taken only behind all data, where every bin is in an earlier epoch.
              // (an epoch without bins in earlier epochs has no such mass at all)
              // (mass beyond t_{e+1} per unit count of the earlier bins: exact with one epoch per lane; with more, where a wave
              // may own one slot only, its lower bound S_{e+1} (every 1 / S(age) >= 1), so that all owners decide alike)
              const double Ie = (NCH == 1) ? q_e[c] * T[c] : S1[c] * (c_all - C0[c]);
              alive_l[c] = !has || nlt[c] == 0 ||
                           ((e > k_old) ? S1[c] >= 1e-11 : (S1[c] >= 1e-14 && Ie >= 1e-8 * (c_all - C0[c])));
              dead_l[c] = has && e > k_old && S1[c] < 4e-20 * S_old1;
              between[c] = ballot64(ep_on[c] && !(alive_l[c] || dead_l[c]));
              any_between |= between[c];
              // a bin absorbs epoch e's term iff W_e / S(age) < ulp(cs(age))/2 <= cs 2^-53, and cs e^-cs <= 1/e
              // (a term of exactly 0 -- an epoch without a valid rate, or underflow -- adds nothing either way)
              small_w[c] = ballot64(ep_on[c] && We[c] < 0x1p-53 * 0.37 && We[c] > 0.0);
            }
            // (nothing in transition now nor at the last refresh: every epoch keeps one of the two closed forms, whose inputs
            // -- D_b, s_b of the bins -- move slowly: they are brought up to date every 256 iterations)
            const bool trivial = (any_between == 0);
            if (!(trivial && tail_trivial_prev && (iter & 255) != 0)) {
              tail_trivial_prev = trivial;
              int e_sm = E;  // first epoch whose term some bin may absorb
#pragma unroll
              for (int c = 0; c < NCH; c++)
                if (small_w[c]) {
                  const int e1 = __builtin_ctzll(small_w[c]) * NCH + c;
                  e_sm = e1 < e_sm ? e1 : e_sm;
                }
              // scratch: with one epoch per lane this role's tile (its tails are in registers by now and the bin waves write it
              // again only behind barriers 3 and 1); with more, the owners of a role run this refresh side by side (same inputs,
              // same values, not synchronised) while the other may still be loading its tails: each has arrays of its own
              double* const scr = s_tscr + (kSplit ? kOwn * 3 * APZ : 0);
              double* const s_w = (NCH == 1) ? out_mine + O_W * APZ : scr;
              double* const s_a = (NCH == 1) ? out_mine + O_N * APZ : scr + APZ;
              double* const s_d = (NCH == 1) ? out_mine + O_D * APZ : scr + 2 * APZ;
              double PDtot = 0.0, PHtot = 0.0, PMtot = 0.0;
              for (int g = 0; g < NB; g++) {  // the not-shared bins, youngest group first
                const int gpos = g * kWave + lane, gbin = nzlo + gpos;
                const bool inr = gbin < nzhi;
                const int kbb = inr ? s_kb[inr ? gbin : 0] : E;
                const double cntb = inr ? s_cnt[APZ + (inr ? gbin : 0)] : 0.0;
                const bool liveb = inr && cntb > 0 && kbb < E - 1;
                const int kq = liveb ? kbb : 0;
                const double ab = liveb ? s_age[gbin] : 0.0, tkb = s_t[kq], tknb = s_t[kq + 1];
                const double lkb = s_ep[G_LAM * EPAD + kq], ckb = s_ep[G_CS * EPAD + kq];
                const double csa = ckb + lkb * (ab - tkb);            // cs(age) = -Z_b
                const double th = half_ulp_pos(csa);                  // what a fold increment must reach to change Z_b
                const double mb = em::em_exp_t(__builtin_fmin(csa, 230.0), s_exptab);  // 1 / S(age), capped at 1e100
                double Db = 0.0;
                int ndrop = 0;
                // (uniform; W_e from the epoch lanes of this wave, a lane's NCH epochs per round: the slot as a run-time index was
                // two branches per epoch -- 25 instructions and ~140 cycles each; an epoch in front of e_sm has no bin that
                // drops its term, and one beyond E a term of 0 that the bound keeps out)
                for (int eb = e_sm & ~(NCH - 1); eb < E; eb += NCH) {
#pragma unroll
                  for (int c = 0; c < NCH; c++) {
                    const int e = eb + c;
                    const double w = readlane_d(We[c], eb >> kSlotShift) * mb;
                    const bool dr = (e < E) && (e > kq) && (w < th);
                    Db += dr ? w : 0.0;
                    ndrop += dr ? 1 : 0;
                  }
                }
                const int nfold = E - 1 - kq - ndrop;
                const double xk = lkb * (tknb - tkb);
                double nh = (double)(E - kq);  // chain steps taken while integ is still in [0.5, 1): ~ln 2 / (lambda dt)
                if (xk > 0.0) nh = __builtin_fmin(0.69 * em::em_rcp(xk), nh);
                nh += 1.5;
                const double sb = __builtin_amdgcn_sqrt((0x1p-108 / 3.0) * nh + th * th * ((double)nfold * (1.0 / 3.0)));
                const double rsb = em::em_rcp(sb);
                const double wv = liveb ? cntb * sb : 0.0, av = liveb ? mb * rsb : 0.0, dv = liveb ? Db * rsb : 0.0;
                s_w[gpos] = wv;
                s_a[gpos] = av;
                s_d[gpos] = dv;
                double qd;
                const double ch = wv * tail_hneg(dv, qd);  // c_b s_b H(-d_b): what is left of a bin deep in the tail
                const double cm = liveb ? cntb * mb : 0.0, cd = liveb ? cntb * Db : 0.0;
                PHtot = PHtot + readlane_d(wave_prefix_sum(ch), 63);
                PMtot = PMtot + readlane_d(wave_prefix_sum(cm), 63);
                PDtot = PDtot + readlane_d(wave_prefix_sum(cd), 63);
              }
              wave_lds_fence();
#pragma unroll
              for (int c = 0; c < NCH; c++) {
                const int e = ep_of(c);
                // R_e as a linear function of S_{e+1} around its current value, R_e = A + B S_{e+1}: what is held until the next
                // refresh (the tail epochs' own rates drift for hundreds of iterations, and S_{e+1} with them).  The two closed
                // forms are linear as they stand (an epoch in transition is overwritten below)
                double Ra = alive_l[c] ? ((e > k_old) ? -PDtot : 0.0) : PHtot;
                double Rb = alive_l[c] ? 0.0 : -PMtot;
                // (the epochs of this slot in transition, one after the other -- uniform --; with the epochs split over two waves
                // each owner needs the correction of its own slot only)
                for (unsigned long long m = own(c) ? between[c] : 0ull; m != 0; m &= m - 1) {
                  const int l = __builtin_ctzll(m);
                  const double S1e = readlane_d(S1[c], l);
                  const int nlte = __builtin_amdgcn_readlane(nlt[c], l);  // only bins of earlier epochs have a term in this epoch's integ
                  double acc = 0.0, acc1 = 0.0;
                  for (int g = 0; g < NB; g++) {
                    const int gpos = g * kWave + lane;
                    const double dv = s_d[gpos], av = s_a[gpos], wv = s_w[gpos];
                    double qz;
                    const double term = wv * (tail_hneg(S1e * av - dv, qz) - dv);
                    acc += (gpos < nlte) ? term : 0.0;
                    acc1 += (gpos < nlte) ? (wv * av) * qz : 0.0;  // -d term / d S_{e+1}
                  }
                  const double r0 = readlane_d(wave_prefix_sum(acc), 63), r1 = -readlane_d(wave_prefix_sum(acc1), 63);
                  Ra = (lane == l) ? r0 - r1 * S1e : Ra;
                  Rb = (lane == l) ? r1 : Rb;
                }
                const bool on = ep_on[c] && e < E - 1;
                if (kLinearHold) {
                  // in terms of what the iteration has in registers anyway: the exact mass X_e = q_e T_e = S_{e+1} sum_b c_b / S(age_b),
                  // whose second factor is a matter of the data epochs' rates (as a_b is): R_e = Ra + Rb (S1_0 / X_0) X_e, and
                  // integ = X_e + R_e = (1 + Rb S1_0 / X_0) X_e + Ra -- the factor replaces the 1.0 the mass is multiplied with
                  const double X0 = q_e[c] * T[c];
                  const bool lin = on && own(c) && X0 > 0.0;
                  eta_e[c] = on ? (lin ? Ra : Ra + Rb * S1[c]) : 0.0;
                  eta_s[c] = lin ? 1.0 + Rb * (S1[c] * em::em_rcp(X0)) : 1.0;
                } else {
                  eta_e[c] = on ? Ra + Rb * S1[c] : 0.0;  // the value at the current S_{e+1}
                }
              }
            }
          }
        }
#pragma unroll
        for (int c = 0; c < NCH; c++) {
          Npart[c] = Dpart[c] = 0.0;
          if (!own(c)) continue;
          const int e = ep_of(c);
          Npart[c] = p_e[c] * T[c] + oN[c];
          // mass still to coalesce after t_{e+1}, relative to survival there (more than 64 epochs: times the slope of the tail
          // model's held correction, 1 + dR_e / dX_e, see the refresh)
          double Gn = kLinearHold ? eta_s[c] : 1.0;
          if (__builtin_expect(!absorbing, 0)) {
            COLATE_COLD();
            if (e < E - 1) Gn = (1.0 - em::em_exp_t(-s_ep[G_CS * EPAD + E - 1] + s_ep[G_CS * EPAD + e + 1], s_exptab)) + (Gn - 1.0);
          }
          // later not-shared bins contribute dt_e each, earlier ones their tail mass (last epoch: dt_e = 0 and q_e = 0
          // leave (beta - t p) T, coal_EM.cpp:136-141, without a branch)
          const double integ_ns = __builtin_fmax(Gn * (q_e[c] * T[c]) + eta_e[c], 0.0);  // (eta_e: the tail model's R_e, its intercept)
          double dns = dt_e[c] * Cn[c] + ((beta_e[c] - t_e[c] * p_e[c]) * T[c] + dt_e[c] * integ_ns);
          dns = __builtin_fmax(dns, 0.0);
          Dpart[c] = dns + oD[c];
        }
      }
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        if (!own(c)) continue;
        s_nd[(ROLE * 2 + 0) * EPAD + ep_of(c)] = Npart[c];
        s_nd[(ROLE * 2 + 1) * EPAD + ep_of(c)] = Dpart[c];
      }
    }
    COLATE_STAMP(4)
    __syncthreads();  // ---- barrier 3: partial N, D visible
    COLATE_STAMP(6)
    // ============================================================ P4: M-step (every wave) and stop rule
    double N_e[NCH], D_e[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      const int e = ep_of(c);
      N_e[c] = s_nd[0 * EPAD + e] + s_nd[2 * EPAD + e];
      D_e[c] = s_nd[1 * EPAD + e] + s_nd[3 * EPAD + e];
      if (MODE == 1 && ep_on[c]) {  // (EM mode: a NaN sticks to the rate and is flagged at the end)
        if (N_e[c] != N_e[c] || D_e[c] != D_e[c]) my_flags |= COLATE_FLAG_NAN;  // coal.cpp:3711-3712
        if (N_e[c] < 0.0 || D_e[c] < 0.0) my_flags |= COLATE_FLAG_NEG;          // coal.cpp:3713-3714
      }
    }
    COLATE_STAMP(13)
    if (need_ll) {
      COLATE_COLD();
      ll = ((s_ll[0] + s_ll[1]) + (s_ll[2] + s_ll[3])) + ((s_ll[4] + s_ll[5]) + (s_ll[6] + s_ll[7]));  // retired waves: 0
      if (kNeedLL < 0 && p.ll_trace && tid == 0 && iter < p.ll_trace_cap) p.ll_trace[(size_t)rep * p.ll_trace_cap + iter] = ll;
    }
    if (MODE == 1) {
      if (wave == 0) {
#pragma unroll
        for (int c = 0; c < NCH; c++) {
          if (ep_on[c]) {
            p.out_num[(size_t)rep * E + ep_of(c)] = N_e[c];
            p.out_den[(size_t)rep * E + ep_of(c)] = D_e[c];
          }
        }
      }
      return true;
    }
    // ---- M-step, coal.cpp:3777-3804
    if (!COLATE_ABL_HAS(14)) {
      double cand[NCH];
      bool self[NCH];  // this lane's epoch keeps its own quotient (does not copy)
      unsigned long long keep[NCH];  // epochs that do NOT copy their predecessor
      unsigned long long cpm[NCH];   // epochs that do
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        const bool copy = (N_e[c] == 0);
        {  // (as selects: the quotient is computed in every lane, 0/0 and x/0 included, and dropped where it does not apply)
#if COLATE_ABL_HAS(10)
          double qn = N_e[c] * 1e-4 + D_e[c] * 1e-9;
#else
          double qn = N_e[c] / D_e[c];
#endif
          qn = em::max_c(qn, p.rate_floor);  // (one v_max_f64; a NaN quotient -- 0 / 0 -- is a copying epoch and not used)
          cand[c] = (!copy && D_e[c] != 0) ? qn : lam_e[c];
        }
        // (the compare's own lane mask and the static mask of the live epoch lanes: scalar arithmetic from here on)
        const unsigned long long zero_n = __builtin_amdgcn_fcmp(N_e[c], 0.0, 1 /* FCMP_OEQ */);
        keep[c] = ep_mask[c] & ~zero_n;
        cpm[c] = ep_mask[c] & zero_n;
        self[c] = ep_on[c] && !copy;
      }
      // The copying epochs form a prefix 0..m-1 of the epoch order (they all become 0 then) unless a bit of `bad` is set: as
      // scalar mask arithmetic, so that one compare and one branch decide.  With NCH consecutive epochs per lane that is: every
      // slot's copying lanes are a prefix of the lanes, the prefixes do not grow with the slot, and slot 0's is at most one
      // lane longer than the last slot's.
      unsigned long long bad = (NCH == 1) ? 0ull : (cpm[0] & ~((cpm[NCH - 1] << 1) | 1ull));
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        bad |= cpm[c] & (cpm[c] + 1ull);
        if (c > 0) bad |= cpm[c] & ~cpm[c - 1];
      }
      if (kTrack < 0 ? tracker : (kTrack != 0)) {  // scalar masks for the epilogue's verdict, kept by a wave that has time for it
#pragma unroll
        for (int c = 0; c < NCH; c++) {
          const unsigned nh = (unsigned)__double2hiint(N_e[c]);  // (N >= 0: the high words order like the values)
          tr_min[c] = nh < tr_min[c] ? nh : tr_min[c];
          tr_max[c] = nh > tr_max[c] ? nh : tr_max[c];
          tr_noisy |= (ep_on[c] && N_e[c] != 0.0 && D_e[c] < noisy_thr[c] && cand[c] > p.rate_floor) ? 1 : 0;
        }
      }
      if (__builtin_expect(bad == 0, 1)) {
#pragma unroll
        for (int c = 0; c < NCH; c++) lam_e[c] = self[c] ? cand[c] : 0.0;
      } else {
        COLATE_COLD();
        // num == 0: take the (already updated) rate of the previous epoch, 0 if there is none: the quotient of the nearest
        // earlier epoch that keeps its own -- in this lane's earlier slots, else the last one of the nearest lane below
        unsigned long long any_keep = 0;
        double lastc = 0.0;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
          any_keep |= keep[c];
          lastc = self[c] ? cand[c] : lastc;
        }
        const unsigned long long below = any_keep & ((1ull << lane) - 1ull);
        const int src = below ? 63 - __builtin_clzll(below) : 0;
        const double from_lanes = __shfl(lastc, src, 64);
        double prev = below ? from_lanes : 0.0;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
          lam_e[c] = ep_on[c] ? (self[c] ? cand[c] : prev) : 0.0;
          prev = self[c] ? cand[c] : prev;
        }
      }
    }
    // (for the loop conditions, see `kFree`; as scalar arithmetic on the compare's lane mask -- written as a shift of the mask the
    // compiler made a per-lane value and an exec-masked loop of it: 14 instructions at the top of every iteration)
    if (kAssumeAbsorbing && kSteady) {
      // (the two-barrier build keeps its one-slot form: the same test, but this kernel's code is tuned to its placement)
      lim = (kFree ? ((ballot64(lam_e[0] > 0.0) & last_bit1) != 0ull) : last_rate_positive()) ? lim : 0;
      asm volatile("" : "+s"(lim));  // (opaque: the compiler otherwise turns the select back into a second loop condition)
    }
    COLATE_STAMP(5)
    // stop rule, coal.cpp:3822 (evaluated after the update); uniform across the workgroup
    bool stop = false;
    if (kNeedLL != 0 && iter > p.min_iter) {  // (never in the steady-state loops)
      COLATE_COLD();
      stop = (ll / prev_ll > thr);
    }
    prev_ll = ll;
    return stop;
  };
  // Steady state: the iterations before min_iter need neither the log-likelihood nor the stop test, and a wave's role and
  // leadership never change -- so each kind of wave runs them in a loop of its own, compiled for exactly that kind: no
  // wave-uniform branch on role / leader / need_ll is left in it.  A taken branch costs a lone wave ~20 cycles and a
  // not-taken one ~8 (csrc/tools/ubench_branch.hip, profiles/r02/ubench_branch.txt), against ~5 for an FP64 instruction.
  // The waves of a workgroup run different loops but the same sequence of barriers.  The general loop below is what is
  // left for MODE 1 and for the rare wave whose epochs need more than two tail slots.
  bool stopped = false;
  // (8 and 16 epochs per lane -- 257 .. 1024 epochs, em_kernels_big.hip -- run everything in the general loop: the reference has no
  // limit on the number of epochs, coal.cpp:3551-3632, and neither has the drop-in; speed is secondary there)
  if constexpr (NCH <= 4) if (MODE == 0 && p.ll_trace == nullptr) {  // (the per-iteration log-likelihood trace runs everything in the general loop)
    int n_steady = p.min_iter < max_iter - 1 ? p.min_iter : max_iter - 1;
    if (n_steady < 0) n_steady = 0;
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    using CR = std::integral_constant<int, -1>;
#define COLATE_STEADY(R, L, T, P)                             \
  {                                                           \
    lim = n_steady;                                           \
    COLATE_LOOP_ANCHOR()                                      \
    do {                                                      \
      iteration(R{}, L{}, C0{}, T{}, C0{}, P{});              \
    } while (__builtin_expect(++iter < lim, 1));              \
    if (kAssumeAbsorbing && lim == 0) last_pos = false;                  \
  }
    // (the role B leader: the iterations that refresh the tail model are peeled out of the hot loop, which then carries
    // nothing of it but the held correction; same schedule as tail_due())
#define COLATE_STEADY_B(R, L, T, P)                                                    \
  while (iter < n_steady && last_pos) {                                                \
    if (tail_due(iter)) {                                                              \
      lim = n_steady;                                                                  \
      iteration(R{}, L{}, C0{}, T{}, C1{}, P{});                                       \
      ++iter;                                                                          \
      if (kAssumeAbsorbing && lim == 0) last_pos = false;                                         \
      if (iter >= n_steady || !last_pos) break;                                        \
    }                                                                                  \
    lim = tail_next_due(iter);                                                         \
    if (lim > n_steady) lim = n_steady;                                                \
    COLATE_LOOP_ANCHOR()                                                               \
    while (__builtin_expect(iter < lim, 1)) {                                          \
      iteration(R{}, L{}, C0{}, T{}, C0{}, P{});                                       \
      ++iter;                                                                          \
    }                                                                                  \
    if (kAssumeAbsorbing && lim == 0) last_pos = false;                                           \
  }
    bool any_more_rows = false;
#pragma unroll
    for (int c = 0; c < NCH; c++) any_more_rows |= more_rows[c];
    // (a wave with an epoch spanning more than two rows of data bins stays in the general loop)
    // ... and the iterations from min_iter on (log-likelihood and stop test in every one) in a second set of loops
    // compiled the same way: runs on sparse tables go on for up to 1e5 iterations there (1.55 -> 1.31 us per iteration).
    // (Round 2 compiled them up to 64 epochs only: with two epochs per lane the extra loops cost the steady ones 2.5 % then,
    // 1.53 -> 1.57 ms at E = 122.  Round 3, with the placement re-tuned: 1.270 against 1.282 ms for the steady state and 1.65
    // against 2.07 us per iteration from min_iter on -- gpurun_out/r03z/ll2_e122.txt -> profiles/r03_placement.txt.)
#define COLATE_STEADY_LL(R, L, T, P)                          \
  if (last_pos) {                                             \
    lim = max_iter;                                           \
    COLATE_LOOP_ANCHOR()                                      \
    for (; iter < lim; iter++) {                              \
      if (iteration(R{}, L{}, C1{}, T{}, C0{}, P{})) {        \
        stopped = true;                                       \
        break;                                                \
      }                                                       \
    }                                                         \
    if (kAssumeAbsorbing && lim == 0) last_pos = false;                  \
  }
#define COLATE_STEADY_LL_B(R, L, T, P)                                                 \
  while (iter < max_iter && !stopped && last_pos) {                                    \
    if (tail_due(iter)) {                                                              \
      lim = max_iter;                                                                  \
      if (iteration(R{}, L{}, C1{}, T{}, C1{}, P{})) {                                 \
        stopped = true;                                                                \
        break;                                                                         \
      }                                                                                \
      ++iter;                                                                          \
      if (kAssumeAbsorbing && lim == 0) last_pos = false;                                         \
      if (iter >= max_iter || !last_pos) break;                                        \
    }                                                                                  \
    lim = tail_next_due(iter);                                                         \
    if (lim > max_iter) lim = max_iter;                                                \
    COLATE_LOOP_ANCHOR()                                                               \
    for (; iter < lim; iter++) {                                                       \
      if (iteration(R{}, L{}, C1{}, T{}, C0{}, P{})) {                                 \
        stopped = true;                                                                \
        break;                                                                         \
      }                                                                                \
    }                                                                                  \
    if (kAssumeAbsorbing && lim == 0) last_pos = false;                                           \
  }
#ifndef COLATE_LL_MAX_NCH
#define COLATE_LL_MAX_NCH 2  // (epochs per lane up to which the log-likelihood-phase loops are compiled, see below)
#endif
#ifdef COLATE_NO_LL_LOOPS  // (A/B switch: the iterations from min_iter on in the general loop)
#define COLATE_BOTH(R, L, T, P)                               \
  {                                                           \
    if (iter < n_steady && last_pos) COLATE_STEADY(R, L, T, P); \
  }
#define COLATE_BOTH_B(R, L, T, P)                             \
  {                                                           \
    COLATE_STEADY_B(R, L, T, P)                               \
  }
#else
#define COLATE_BOTH(R, L, T, P)                               \
  {                                                           \
    if (iter < n_steady && last_pos) COLATE_STEADY(R, L, T, P); \
    if constexpr (NCH <= COLATE_LL_MAX_NCH) COLATE_STEADY_LL(R, L, T, P) \
  }
#define COLATE_BOTH_B(R, L, T, P)                             \
  {                                                           \
    COLATE_STEADY_B(R, L, T, P)                               \
    if constexpr (NCH <= COLATE_LL_MAX_NCH) COLATE_STEADY_LL_B(R, L, T, P) \
  }
#endif
    using C2 = std::integral_constant<int, 2>;
    using C3 = std::integral_constant<int, 3>;
    // 65..128 epochs with at least two bin groups: the epoch work of a role is SPLIT over its first two waves -- the even
    // epochs (slot 0 of every lane) to the wave of bin group 0, the odd ones (slot 1) to that of bin group 1, which otherwise
    // sits out P1 and P3.  Each owner runs the (single) scans in full -- they need all epochs -- and the per-epoch work
    // (exp, the N and D terms, the stores) for its own slot: per wave about what 64 epochs cost.
    // any_more_rows / third_row are functions of the epochs' bin spans only, the same in every wave, so all four choose alike.
    const bool split = (NCH == 2) && !TPUT && NB >= 2;
    // (last argument: the epoch values the wave computes in P1 -- 1 role A's, 2 role B's)
    if (!(any_more_rows || third_row)) {
      if constexpr (kFree) {  // every wave of role A computes role A's epoch values, wave p1b_wave role B's
        if (role == 0) {
          if (leader) {
            if (tracker) {
              COLATE_BOTH(C0, C1, C1, C1)
            } else {
              COLATE_BOTH(C0, C1, C0, C1)
            }
          } else {
            COLATE_BOTH(C0, C0, C0, C1)
          }
        } else if (leader) {
          if (wave == p1b_wave) {
            COLATE_BOTH_B(C1, C1, C0, C2)
          } else {
            COLATE_BOTH_B(C1, C1, C1, C0)  // (two or more bin groups: keeps the verdict's history)
          }
        } else if (wave == p1b_wave) {
          COLATE_BOTH(C1, C0, C0, C2)
        } else {
          COLATE_BOTH(C1, C0, C0, C0)
        }
      } else if (role == 0) {  // (the wave that keeps the verdict's history is of role 0: wave 0 or wave 2)
        if (split && grp <= 1) {
          if constexpr (NCH == 2) {
            if (grp == 0) {
              COLATE_BOTH(C0, C2, C1, C1)  // (wave 0: keeps the verdict's history in the split)
            } else {
              COLATE_BOTH(C0, C3, C0, C1)
            }
          }
        } else if (leader) {
          if (tracker) {
            COLATE_BOTH(C0, C1, C1, C1)
          } else {
            COLATE_BOTH(C0, C1, C0, C1)
          }
        } else {
          if (tracker) {
            COLATE_BOTH(C0, C0, C1, C0)
          } else {
            COLATE_BOTH(C0, C0, C0, C0)
          }
        }
      } else {
        if (split && grp <= 1) {
          if constexpr (NCH == 2) {
            if (grp == 0) {
              COLATE_BOTH_B(C1, C2, C0, C2)
            } else {
              COLATE_BOTH_B(C1, C3, C0, C2)
            }
          }
        } else if (leader) {
          COLATE_BOTH_B(C1, C1, C0, C2)
        } else {
          COLATE_BOTH(C1, C0, C0, C0)
        }
      }
    }
#undef COLATE_BOTH_B
#undef COLATE_BOTH
#undef COLATE_STEADY_LL_B
#undef COLATE_STEADY_LL
#undef COLATE_STEADY_B
#undef COLATE_STEADY
    // ... and where the per-kind loops of that phase are not built (more than 64 epochs), one loop compiled for "log-likelihood
    // needed" only: no cost for the steady loops, 2.20 -> 2.08 us per iteration at E = 122
    if constexpr (NCH > COLATE_LL_MAX_NCH) if (!(any_more_rows || third_row) && last_pos) {
      lim = max_iter;
      COLATE_LOOP_ANCHOR()
      for (; iter < lim; iter++) {
        if (iteration(CR{}, CR{}, C1{}, CR{}, CR{}, CR{})) {
          stopped = true;
          break;
        }
      }
      if (kAssumeAbsorbing && lim == 0) last_pos = false;
    }
  }
  COLATE_LOOP_ANCHOR()
  for (; !stopped && iter < max_iter; iter++) {
    using CRt = std::integral_constant<int, -1>;
    if (iteration(CRt{}, CRt{}, CRt{}, CRt{}, CRt{}, CRt{})) break;
  }

#ifdef COLATE_EM_STAMPS
  COLATE_PSTAMP(8)
  if (p.out_den && tid == 0 && MODE == 0) {  // diagnostic build: wave 0's absolute stamps along the prologue in place of out_den
    unsigned long long* dbg2 = reinterpret_cast<unsigned long long*>(p.out_den) + (size_t)rep * 10;
    for (int i = 0; i < 9; i++) dbg2[i] = pro_t[i];
  }
  if (p.out_num && lane == 0 && MODE == 0 && wave < 4) {  // diagnostic build: per-wave phase cycles in place of out_num
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.out_num) + ((size_t)rep * 4 + wave) * 16;
    for (int i = 0; i < 16; i++) dbg[i] = st_acc[i];
  }
#endif
#ifdef COLATE_EM_TRACE
  if (p.out_num && tid == 0 && MODE == 0) {
    unsigned long long trace_t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(trace_t1)::"memory");
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.out_num) + (size_t)rep * 4;
    dbg[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
    dbg[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
    dbg[2] = trace_t0;
    dbg[3] = trace_t1;
  }
#endif
  // ------------------------------------------------------------------ epilogue
  if (MODE == 0 && tracker) {
    // Which of the printed rates are determined by the reference's SOURCE, and which only by the last bits of the libm
    // it happens to be linked with?  The denominator of epoch e contains dt_e * integ (coal_EM.cpp:270-274, 445-449),
    // and integ carries an absolute rounding error of ~kIntegResidue per unit count: once dt_e * residue is no longer
    // negligible against the whole denominator, the reference's own rate moves by more than 1e-8 (relative) when its
    // exp()/log() return a neighbouring double (tests/oracle_lib.stable_mask measures exactly that on the oracle), and
    // through the coupling of the EM every OLDER epoch moves with it.  Exception: where the denominator is nothing but
    // residue and the rate sits on the floor, the floor is what every build prints.  The count of such trailing epochs
    // goes out in the high bits of out_flags (COLATE_FLAG_UNRESOLVED, COLATE_UNRESOLVED_EPOCHS()).
    int first_bad = E;
    const unsigned long long ever_noisy = ballot64(tr_noisy != 0);
    const int n_evals = s_misc[3];  // (read before this wave stores the verdict there)
    constexpr int kMinEvals = 32;
#pragma unroll
    for (int c = NCH - 1; c >= 0; c--) {
      const int e = ep_of(c);
      const double D = s_nd[1 * EPAD + e] + s_nd[3 * EPAD + e];  // denominators of the last E-step
      const double Nfin = s_nd[0 * EPAD + e] + s_nd[2 * EPAD + e];
      const double eta = dt_e[c] * (kIntegResidue * (s_ll[8] + s_ll[9]));  // dt_e * residue of ALL bins (both kinds); 0 in the last epoch and beyond E
      // A numerator of exactly 0 makes the rate copy its neighbour's (coal.cpp:3779-3788); when it is 0 in some
      // iterations and not in others (underflow: survival below ~1e-308) the rate ends as a snapshot of the neighbour at
      // the moment the underflow ended -- or never copies at all, depending on where exactly an implementation's products
      // reach zero (the reference's exp(A_e - Z_b) per bin, the kernel's W_e RS_e and q_e T_e chains).  So an epoch whose
      // numerator was ever within a factor 1e-280 of that edge while not being a structural zero (0 in every iteration,
      // like epoch 0) is not reproducible.
      // (high words: below kTinyNum up to its last 32 bits; not 0 = at least 2^-1042)
      const bool snapshot = ep_on[c] && tr_min[c] < (unsigned)__double2hiint(kTinyNum) && tr_max[c] != 0u;
      // An epoch that starts after the oldest bin with data: every contribution to its statistics has num/denom equal to
      // the current rate (the likelihood does not depend on it), so the EM leaves it where it is -- normally at its starting
      // value, which every build prints alike.  If it has moved, rounding moved it (early iterations far from the optimum),
      // and the reference's own value is as arbitrary (tests/golden/l3_coal_modern: 42 % under 1-ulp libm noise).
      const double init = ep_on[c] ? p.out_rates[(size_t)rep * E + e] : 0.0;
      const bool drifted = (double)e > s_ll[10] && !(__builtin_fabs(lam_e[c] - init) <= 1e-9 * init);
      // ... and even unmoved it is a record of the path: had the numerator been 0 in one iteration it would have copied
      // its neighbour.  Whether that happens in the reference depends on how high the rates of the epochs before it went
      // on the way, and where one of those was ever a quotient of rounding residue (ever_noisy) the reference's path is its
      // own (measured: sparse tables, the last epoch with data overshoots x4 on the reference, its survival underflows and
      // the flat epoch behind it ends as a copy, while the exact sums never come near; profiles/parity/sweep2_sparse_*).
      const bool path_dependent = (double)e > s_ll[10] && ever_noisy != 0;  // (any lane, any slot)
      // "Denominator nothing but residue, rate on the floor" is what every build prints only if the reference's own residue
      // cannot be so much smaller than the modelled one that its quotient leaves the floor: the model is an average over
      // the (bin, kind) pairs it evaluates -- with a handful of them the actual residue may be 0 (tools/fuzz_parity.py
      // found tables of 5 mutations where the reference prints 2e-4 and the kernel the floor) -- and the numerator must
      // stay below floor x (exact part + a fifth of the modelled residue).
      const bool deep_floor = lam_e[c] <= p.rate_floor && D < 3.0 * eta && n_evals >= kMinEvals &&
                              Nfin <= p.rate_floor * (D - 0.8 * eta);
      const bool resolved = !ep_on[c] || (!snapshot && !drifted && !path_dependent &&
                            (D >= kResolvedRatio * eta || deep_floor));
      const unsigned long long bad = ballot64(!resolved);
      if (bad) {
        const int e1 = __builtin_ctzll(bad) * NCH + c;
        first_bad = e1 < first_bad ? e1 : first_bad;
      }
    }
    if (lane == 0) s_misc[3] = E - first_bad;
  }
  if (MODE == 0 && tracker) {  // (after the verdict: it reads the starting rates out of this row)
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      if (ep_on[c]) {
        if (lam_e[c] != lam_e[c]) my_flags |= COLATE_FLAG_NAN;
        p.out_rates[(size_t)rep * E + ep_of(c)] = lam_e[c];
      }
    }
  }
  if (my_flags) atomicOr(&s_misc[2], my_flags);
  __syncthreads();
  if (tid == 0) {
    int fl = s_misc[2];
    if (MODE == 0) {
      if (iter >= p.max_iter) fl |= COLATE_FLAG_MAXITER;
      if (s_misc[3] > 0) fl |= COLATE_FLAG_UNRESOLVED | (s_misc[3] << COLATE_UNRESOLVED_SHIFT);
      p.out_iters[rep] = iter < p.max_iter ? iter : p.max_iter;
    }
    p.out_ll[rep] = ll;
    p.out_flags[rep] = fl;
  }
}


template <int MODE, int NCH, int EROWS, bool TPUT, int WPE = 0>
hipError_t launch_one(const ColateEmArgs& args, hipStream_t stream, size_t lds, int threads) {
  auto kern = em_kernel<MODE, NCH, EROWS, TPUT, WPE>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(args.B), dim3(threads), lds, stream, args);
  return hipGetLastError();
}

inline int em_groups(int A) { return (A + 63) / 64; }           // bin groups of 64 per role
inline int em_threads(int A) { return 2 * 64 * em_groups(A); }  // two roles (latency variant)
inline int em_chunks(int E) { return E <= 64 ? 1 : (E <= 128 ? 2 : (E <= 256 ? 4 : (E <= 512 ? 8 : 16))); }
inline int em_rows(int E) { return E <= 16 ? 1 : (E <= 32 ? 2 : 4); }  // BASELINE's `--bins 3,7,0.2` gives E = 23

inline size_t em_lds_bytes(int E, int A, bool tput) {
  const size_t EPAD = (size_t)em_chunks(E) * kWave;
  const size_t AP = (size_t)em_groups(A) * kWave;
  const size_t APZ = AP + 2;
  const size_t doubles = (EPAD + 1) + num_gather_rows(em_chunks(E), tput) * EPAD + 2 * kNumBinArrays * APZ + 4 * EPAD + 4 * APZ + 12 +
                         em::kExpTableDoubles + AP + tail_scratch_arrays(em_chunks(E), tput) * APZ;
  const size_t ints = (AP + 1) + 8 + 4 + AP;
  return doubles * sizeof(double) + ints * sizeof(int);
}

// the latency variant (a wave per role and bin group) for 1 or 2 epochs per lane; each translation unit that
// includes this header gets its own copy, compiled with that unit's flags
// `alone`: every workgroup has a CU to itself (B <= #CUs): the register cap that keeps three waves per SIMD is not needed
inline hipError_t launch_latency(const ColateEmArgs& args, hipStream_t stream, bool alone = false) {
  const size_t lds = em_lds_bytes(args.E, args.A, false);
  const int threads = em_threads(args.A);
  if (em_chunks(args.E) == 2) return launch_one<0, 2, 4, false>(args, stream, lds, threads);
#ifdef COLATE_EM_ILP_BUILD
  if (alone) switch (em_rows(args.E)) {
    case 1: return launch_one<0, 1, 1, false, 2>(args, stream, lds, threads);
    case 2: return launch_one<0, 1, 2, false, 2>(args, stream, lds, threads);
    default: return launch_one<0, 1, 4, false, 2>(args, stream, lds, threads);
  }
#endif
  switch (em_rows(args.E)) {
    case 1: return launch_one<0, 1, 1, false>(args, stream, lds, threads);
    case 2: return launch_one<0, 1, 2, false>(args, stream, lds, threads);
    default: return launch_one<0, 1, 4, false>(args, stream, lds, threads);
  }
}

}  // namespace
