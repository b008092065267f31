// colate_amd/csrc/em_kernels.h -- internal launch interface of the EM kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

#define COLATE_EM_THREADS 256
#define COLATE_EM_MAX_E 1024  // up to 16 epochs per lane of a 64-lane wave (em_kernels_big.hip beyond 256)
#define COLATE_EM_MAX_A 256  // one age bin per thread

// per-replicate diagnostic flags (the reference aborts on the corresponding asserts)
#define COLATE_FLAG_NAN 1      // coal.cpp:3711-3712, coal_EM.cpp:128-129, 351
#define COLATE_FLAG_NEG 2      // coal.cpp:3713-3714
#define COLATE_FLAG_MAXITER 4  // iteration cap reached without meeting the stop rule
#define COLATE_FLAG_UNRESOLVED 8  // some trailing epochs are below the resolution of the reference's arithmetic
#define COLATE_UNRESOLVED_SHIFT 8 // out_flags >> 8 = number of such trailing epochs

struct ColateEmArgs {
  int B, E, A;
  int mode;                 // 0 = EM to convergence, 1 = one E-step
  const double* age_grid;   // [A]      device
  const double* cnt_sh;     // [B][A]   device
  const double* cnt_ns;     // [B][A]   device
  const double* epochs;     // [E] (epochs_stride 0) or [B][E] (epochs_stride E)
  long epochs_stride;
  const double* rates_in;   // initial rates (mode 0) / rates (mode 1); [E] or [B][E]
  long rates_stride;
  int max_iter, min_iter;
  double rel_tol, rate_floor;
  double* out_rates;  // [B][E] (mode 0)
  int* out_iters;     // [B]    (mode 0)
  double* out_ll;     // [B]
  int* out_flags;     // [B]
  double* out_num;    // [B][E] (mode 1)
  double* out_den;    // [B][E] (mode 1)
  double* ll_trace;   // [B][ll_trace_cap] or NULL: the log-likelihood of every iteration (diagnostic; general loop only)
  int ll_trace_cap;
};

size_t colate_em_lds_bytes(int E, int A);
hipError_t colate_em_launch(const ColateEmArgs& args, hipStream_t stream);
// which build of the kernel a launch of this shape picks on the current device:
// 0 = latency (max-ilp build), 1 = latency (default build), 2 = throughput (em_kernels.hip)
int colate_em_variant(int B, int E);
// force the build for E <= 128 (0, 1, 2 as above; anything else = automatic again)
void colate_em_set_forced_variant(int v);

// block bootstrap on the device (bootstrap_kernel.hip)
hipError_t colate_bootstrap_launch(int B, int nb, int A, const double* age_grid, double age,
                                   const double* weights, const double* sh_block, const double* ns_block,
                                   const double* sh_emp_block, const double* ns_emp_block, double* cnt_sh,
                                   double* cnt_ns, int* status, hipStream_t stream);
// the same for rows [row_lo, row_lo + rows) of G groups x B replicates with per-group tables / weights / ages
// (device arrays indexed from group `group_first`; bootstrap_groups_kernel)
hipError_t colate_bootstrap_groups_launch(int B, int row_lo, int rows, int group_first, int A, const double* age_grid,
                                          const int* group_nb, const long long* group_block_off,
                                          const long long* group_weight_off, const double* group_age,
                                          const double* weights, const double* sh_block, const double* ns_block,
                                          const double* sh_emp_block, const double* ns_emp_block, double* cnt_sh,
                                          double* cnt_ns, int* status, hipStream_t stream);
