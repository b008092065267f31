// colate_amd/csrc/bootstrap_kernel.hip -- the block bootstrap of `mut()` on the GPU.
//
// Replaces include/coal/coal.cpp:3358-3441 for all replicates at once: the weighted sums of the
// per-block age-bin tables (cnt[i][:] = sum_j w[i][j] * block[j][:], four tables; only row 0 of the
// reference's A*A "emp" tables is live) and the F redistribution of the age_begin <= 0 mutations
// into the shared counts.  The multinomial weights w[B][nb] still come from the host's std::mt19937
// (coal.cpp:3350-3357: a sequential generator shared with the age sampling).
//
// One workgroup per replicate, one thread per age bin; every sum runs in the reference's order
// (blocks ascending; bins ascending for fcount / normf, by thread 0 out of LDS) with separate
// multiply and add (-ffp-contract=off), so the tables are bit-identical to the host path
// (colate_bootstrap_counts) and to the reference.
#include <hip/hip_runtime.h>

#include "em_kernels.h"

namespace {

// one replicate: `w` = its nb block weights, the four tables [nb][A], `out_sh` / `out_ns` = its two rows of counts
__device__ __forceinline__ void bootstrap_replicate(
    int nb, int A, const double* __restrict__ age_grid, double age, const double* __restrict__ w,
    const double* __restrict__ sh_block, const double* __restrict__ ns_block,
    const double* __restrict__ sh_emp_block, const double* __restrict__ ns_emp_block,
    double* __restrict__ out_sh, double* __restrict__ out_ns, int* __restrict__ status) {
  __shared__ double s_she[COLATE_EM_MAX_A], s_F[COLATE_EM_MAX_A], s_grid[COLATE_EM_MAX_A];
  __shared__ double s_fcount, s_normf;
  __shared__ int s_bin_start;
  const int b = threadIdx.x;
  double sh = 0.0, ns = 0.0, she = 0.0, nse = 0.0;
  if (b < A) {
    for (int j = 0; j < nb; j++) {
      const double wj = w[j];
      if (wj > 0.0) {  // coal.cpp:3359
        sh += wj * sh_block[(size_t)j * A + b];
        ns += wj * ns_block[(size_t)j * A + b];
        she += wj * sh_emp_block[(size_t)j * A + b];
        nse += wj * ns_emp_block[(size_t)j * A + b];
      }
    }
    s_she[b] = she;
    s_grid[b] = age_grid[b];
  }
  __syncthreads();
  if (b == 0) {  // coal.cpp:3395-3396, 3406-3408: bin_start and fcount, bins ascending
    int bin = 0;
    while (bin < A && s_grid[bin] <= age) bin++;
    s_bin_start = bin;
    double fc = 0.0;
    for (int k = bin; k < A; k++) fc += s_she[k];
    s_fcount = fc;
    if (bin < 1 || bin >= A) atomicOr(status, 1);  // sample age outside the age grid
  }
  __syncthreads();
  const int bin_start = s_bin_start;
  if (bin_start >= 1 && bin_start < A) {
    // F[bin] = sh_emp/(sh_emp + ns_emp) for bin >= bin_start (coal.cpp:3409-3411), then F[bin-1] *=
    // (age_bin[bin] - lower_age) with lower_age trailing one bin behind (coal.cpp:3420-3425): i.e.
    // F[m] (m >= bin_start - 1, m <= A - 2) is scaled by grid[m+1] - grid[m'] with m' = max(m, bin_start-1);
    // F[A-1] stays unscaled
    double F = 0.0;
    if (b < A) {
      if (b >= bin_start && she > 0) F = she / (she + nse);
      if (b >= bin_start - 1 && b <= A - 2) {
        const double lower = (b == bin_start - 1) ? s_grid[bin_start - 1] : s_grid[b];
        F *= (s_grid[b + 1] - lower);
      }
      s_F[b] = F;
    }
    __syncthreads();
    if (b == 0) {  // coal.cpp:3429-3432, bins ascending
      double nf = 0.0;
      for (int k = 0; k < A; k++) nf += s_F[k];
      s_normf = nf;
    }
    __syncthreads();
    if (b < A) {
      F /= s_normf;  // coal.cpp:3437-3440
      F *= s_fcount;
      sh += (0.0 < F) ? F : 0.0;  // std::max(0.0, F): NaN -> 0.0
    }
  }
  if (b < A) {
    out_sh[b] = sh;
    out_ns[b] = ns;
  }
}

__global__ __launch_bounds__(COLATE_EM_MAX_A) void bootstrap_kernel(
    int nb, int A, const double* __restrict__ age_grid, double age, const double* __restrict__ weights,
    const double* __restrict__ sh_block, const double* __restrict__ ns_block,
    const double* __restrict__ sh_emp_block, const double* __restrict__ ns_emp_block,
    double* __restrict__ cnt_sh, double* __restrict__ cnt_ns, int* __restrict__ status) {
  const int rep = blockIdx.x;
  bootstrap_replicate(nb, A, age_grid, age, weights + (size_t)rep * nb, sh_block, ns_block, sh_emp_block, ns_emp_block,
                      cnt_sh + (size_t)rep * A, cnt_ns + (size_t)rep * A, status);
}

// Batched all-pairs: row r = group * B + replicate of G (target, reference) pairs, each with its
// own block tables (nb[g] blocks from block_off[g] on in the concatenated tables), weights (from weight_off[g] on:
// [B][nb[g]]) and sample age.  The launch covers rows [row_lo, row_lo + gridDim.x); the group arrays start at group
// `group_first`; row r writes counts row r - row_lo.
__global__ __launch_bounds__(COLATE_EM_MAX_A) void bootstrap_groups_kernel(
    int B, int row_lo, int group_first, int A, const double* __restrict__ age_grid, const int* __restrict__ group_nb,
    const long long* __restrict__ group_block_off, const long long* __restrict__ group_weight_off,
    const double* __restrict__ group_age, const double* __restrict__ weights, const double* __restrict__ sh_block,
    const double* __restrict__ ns_block, const double* __restrict__ sh_emp_block,
    const double* __restrict__ ns_emp_block, double* __restrict__ cnt_sh, double* __restrict__ cnt_ns,
    int* __restrict__ status) {
  const int r = row_lo + blockIdx.x;
  const int g = r / B - group_first, rep = r % B;
  const int nb = group_nb[g];
  const size_t t0 = (size_t)group_block_off[g] * A;
  bootstrap_replicate(nb, A, age_grid, group_age[g], weights + group_weight_off[g] + (size_t)rep * nb, sh_block + t0,
                      ns_block + t0, sh_emp_block + t0, ns_emp_block + t0, cnt_sh + (size_t)blockIdx.x * A,
                      cnt_ns + (size_t)blockIdx.x * A, status);
}

}  // namespace

hipError_t colate_bootstrap_groups_launch(int B, int row_lo, int rows, int group_first, int A, const double* age_grid,
                                          const int* group_nb, const long long* group_block_off,
                                          const long long* group_weight_off, const double* group_age,
                                          const double* weights, const double* sh_block, const double* ns_block,
                                          const double* sh_emp_block, const double* ns_emp_block, double* cnt_sh,
                                          double* cnt_ns, int* status, hipStream_t stream) {
  const int threads = (A + 63) & ~63;
  hipLaunchKernelGGL(bootstrap_groups_kernel, dim3(rows), dim3(threads), 0, stream, B, row_lo, group_first, A, age_grid,
                     group_nb, group_block_off, group_weight_off, group_age, weights, sh_block, ns_block, sh_emp_block,
                     ns_emp_block, cnt_sh, cnt_ns, status);
  return hipGetLastError();
}

hipError_t colate_bootstrap_launch(int B, int nb, int A, const double* age_grid, double age,
                                   const double* weights, const double* sh_block, const double* ns_block,
                                   const double* sh_emp_block, const double* ns_emp_block, double* cnt_sh,
                                   double* cnt_ns, int* status, hipStream_t stream) {
  const int threads = (A + 63) & ~63;
  hipLaunchKernelGGL(bootstrap_kernel, dim3(B), dim3(threads), 0, stream, nb, A, age_grid, age, weights, sh_block,
                     ns_block, sh_emp_block, ns_emp_block, cnt_sh, cnt_ns, status);
  return hipGetLastError();
}
