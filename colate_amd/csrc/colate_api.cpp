// colate_amd/csrc/colate_api.cpp -- extern "C" entry points of libcolate_amd.so
// for the EM hot path (see include/colate_amd.h).  Argument checking, device
// buffers for the host-pointer variants, kernel launch.  No CPU fallback.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "colate_amd.h"
#include "colate_internal.h"
#include "em_kernels.h"

static_assert(COLATE_FLAG_NAN == 1 && COLATE_FLAG_NEG == 2 && COLATE_FLAG_MAXITER == 4, "flags");

namespace colate {

static thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

static int hip_fail(hipError_t e, const char* what) {
  return fail(e == hipErrorNoDevice || e == hipErrorInvalidDevice ? COLATE_ENODEVICE : COLATE_EHIP,
              "%s: %s", what, hipGetErrorString(e));
}

#define HIP_TRY(expr)                                \
  do {                                               \
    hipError_t e_ = (expr);                          \
    if (e_ != hipSuccess) return hip_fail(e_, #expr); \
  } while (0)

static int check_sizes(int B, int E, int A) {
  if (B < 0 || E < 1 || A < 1) return fail(COLATE_EINVAL, "bad sizes B=%d E=%d A=%d", B, E, A);
  if (E > COLATE_EM_MAX_E || A > COLATE_EM_MAX_A)
    return fail(COLATE_ELIMIT, "E=%d / A=%d above compiled limits (%d / %d)", E, A,
                COLATE_EM_MAX_E, COLATE_EM_MAX_A);
  return COLATE_OK;
}

static int ensure_device() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(COLATE_ENODEVICE, "no usable HIP device (%s); libcolate_amd has no CPU fallback",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  return COLATE_OK;
}

// grids that the kernel's contiguous-segment logic relies on
static int check_grids(int E, int A, const double* age_grid, const double* epochs) {
  for (int b = 0; b < A; b++) {
    if (!(age_grid[b] >= 0.0) || (b > 0 && age_grid[b] < age_grid[b - 1]))
      return fail(COLATE_EINVAL, "age_grid must be non-negative and non-decreasing (index %d)", b);
  }
  for (int e = 1; e < E; e++) {
    if (!(epochs[e] >= epochs[e - 1]))
      return fail(COLATE_EINVAL, "epochs must be non-decreasing (index %d)", e);
  }
  if (!(epochs[0] <= age_grid[0]))
    return fail(COLATE_EINVAL, "epochs[0] must not exceed age_grid[0]");
  return COLATE_OK;
}

struct DevBuf {
  void* p = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) {
      if (p) (void)hipFree(p);
      p = o.p;
      o.p = nullptr;
    }
    return *this;
  }
  DevBuf(DevBuf&& o) noexcept : p(o.p) { o.p = nullptr; }
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
  template <typename T>
  T* as() {
    return static_cast<T*>(p);
  }
};

static int launch(const ColateEmArgs& a, hipStream_t s) {
  if (a.B == 0) return COLATE_OK;
  hipError_t e = colate_em_launch(a, s);
  if (e != hipSuccess) return hip_fail(e, "EM kernel launch");
  return COLATE_OK;
}

}  // namespace colate

using namespace colate;

extern "C" {

const char* colate_version(void) { return "colate_amd 0.1 (gfx950)"; }
const char* colate_last_error(void) { return g_last_error.c_str(); }

int colate_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(COLATE_ENODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  return n;
}

int colate_set_device(int ordinal) {
  HIP_TRY(hipSetDevice(ordinal));
  return COLATE_OK;
}

int colate_em_batch_device(int B, int E, int A, const double* age_grid, const double* cnt_shared,
                           const double* cnt_notshared, const double* epochs,
                           int epochs_per_replicate, const double* init_rates,
                           int rates_per_replicate, int max_iter, int min_iter, double rel_tol,
                           double rate_floor, double* out_rates, int* out_iters,
                           double* out_loglik, int* out_flags, void* hip_stream) {
  if (int rc = check_sizes(B, E, A)) return rc;
  if (!age_grid || !cnt_shared || !cnt_notshared || !epochs || !init_rates || !out_rates ||
      !out_iters || !out_loglik || !out_flags)
    return fail(COLATE_EINVAL, "NULL pointer argument");
  if (max_iter < 1) return fail(COLATE_EINVAL, "max_iter must be >= 1");
  ColateEmArgs a{};
  a.B = B, a.E = E, a.A = A, a.mode = 0;
  a.age_grid = age_grid, a.cnt_sh = cnt_shared, a.cnt_ns = cnt_notshared;
  a.epochs = epochs, a.epochs_stride = epochs_per_replicate ? E : 0;
  a.rates_in = init_rates, a.rates_stride = rates_per_replicate ? E : 0;
  a.max_iter = max_iter, a.min_iter = min_iter, a.rel_tol = rel_tol, a.rate_floor = rate_floor;
  a.out_rates = out_rates, a.out_iters = out_iters, a.out_ll = out_loglik, a.out_flags = out_flags;
  return launch(a, static_cast<hipStream_t>(hip_stream));
}

int colate_em_estep_device(int B, int E, int A, const double* age_grid, const double* cnt_shared,
                           const double* cnt_notshared, const double* epochs, const double* rates,
                           double* num_acc, double* den_acc, double* loglik, int* flags,
                           void* hip_stream) {
  if (int rc = check_sizes(B, E, A)) return rc;
  if (!age_grid || !cnt_shared || !cnt_notshared || !epochs || !rates || !num_acc || !den_acc ||
      !loglik || !flags)
    return fail(COLATE_EINVAL, "NULL pointer argument");
  ColateEmArgs a{};
  a.B = B, a.E = E, a.A = A, a.mode = 1;
  a.age_grid = age_grid, a.cnt_sh = cnt_shared, a.cnt_ns = cnt_notshared;
  a.epochs = epochs, a.epochs_stride = 0;
  a.rates_in = rates, a.rates_stride = E;
  a.max_iter = 1, a.min_iter = 0, a.rel_tol = 0, a.rate_floor = 0;
  a.out_ll = loglik, a.out_flags = flags, a.out_num = num_acc, a.out_den = den_acc;
  return launch(a, static_cast<hipStream_t>(hip_stream));
}

int colate_em_batch(int B, int E, int A, const double* age_grid, const double* cnt_shared,
                    const double* cnt_notshared, const double* epochs, const double* init_rates,
                    int max_iter, int min_iter, double rel_tol, double rate_floor,
                    double* out_rates, int* out_iters, double* out_loglik, int* out_flags) {
  if (int rc = check_sizes(B, E, A)) return rc;
  if (!age_grid || !cnt_shared || !cnt_notshared || !epochs || !init_rates || !out_rates ||
      !out_iters || !out_loglik || !out_flags)
    return fail(COLATE_EINVAL, "NULL pointer argument");
  if (int rc = check_grids(E, A, age_grid, epochs)) return rc;
  if (int rc = ensure_device()) return rc;
  if (B == 0) return COLATE_OK;
  const size_t nBA = (size_t)B * A, nBE = (size_t)B * E;
  DevBuf d_grid, d_sh, d_ns, d_ep, d_init, d_rates, d_iters, d_ll, d_flags;
  HIP_TRY(d_grid.alloc(A * sizeof(double)));
  HIP_TRY(d_sh.alloc(nBA * sizeof(double)));
  HIP_TRY(d_ns.alloc(nBA * sizeof(double)));
  HIP_TRY(d_ep.alloc(E * sizeof(double)));
  HIP_TRY(d_init.alloc(E * sizeof(double)));
  HIP_TRY(d_rates.alloc(nBE * sizeof(double)));
  HIP_TRY(d_iters.alloc(B * sizeof(int)));
  HIP_TRY(d_ll.alloc(B * sizeof(double)));
  HIP_TRY(d_flags.alloc(B * sizeof(int)));
  HIP_TRY(hipMemcpy(d_grid.p, age_grid, A * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_sh.p, cnt_shared, nBA * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_ns.p, cnt_notshared, nBA * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_ep.p, epochs, E * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_init.p, init_rates, E * sizeof(double), hipMemcpyHostToDevice));
  int rc = colate_em_batch_device(B, E, A, d_grid.as<double>(), d_sh.as<double>(),
                                  d_ns.as<double>(), d_ep.as<double>(), 0, d_init.as<double>(), 0,
                                  max_iter, min_iter, rel_tol, rate_floor, d_rates.as<double>(),
                                  d_iters.as<int>(), d_ll.as<double>(), d_flags.as<int>(), nullptr);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out_rates, d_rates.p, nBE * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_iters, d_iters.p, B * sizeof(int), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_loglik, d_ll.p, B * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_flags, d_flags.p, B * sizeof(int), hipMemcpyDeviceToHost));
  return COLATE_OK;
}

int colate_bootstrap_counts_device(int B, int nb, int A, const double* age_grid, double age,
                                   const double* weights, const double* sh_block, const double* ns_block,
                                   const double* sh_emp_block, const double* ns_emp_block,
                                   double* cnt_shared, double* cnt_notshared, int* status,
                                   void* hip_stream) {
  if (B < 0 || nb < 1 || A < 2 || A > COLATE_EM_MAX_A) return fail(COLATE_EINVAL, "bad sizes B=%d nb=%d A=%d", B, nb, A);
  if (!age_grid || !weights || !sh_block || !ns_block || !sh_emp_block || !ns_emp_block || !cnt_shared ||
      !cnt_notshared)
    return fail(COLATE_EINVAL, "NULL pointer argument");
  if (B == 0) return COLATE_OK;
  DevBuf scratch;
  int* st = status;
  if (!st) {  // the kernel always reports; give it somewhere to write
    HIP_TRY(scratch.alloc(sizeof(int)));
    HIP_TRY(hipMemsetAsync(scratch.p, 0, sizeof(int), static_cast<hipStream_t>(hip_stream)));
    st = scratch.as<int>();
  }
  hipError_t e = colate_bootstrap_launch(B, nb, A, age_grid, age, weights, sh_block, ns_block, sh_emp_block,
                                         ns_emp_block, cnt_shared, cnt_notshared, st,
                                         static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "bootstrap kernel launch");
  if (!status) HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(hip_stream)));  // scratch dies with this call
  return COLATE_OK;
}

int colate_bootstrap_em_batch(int B, int nb, int E, int A, const double* age_grid, double age,
                              const double* weights, const double* sh_block, const double* ns_block,
                              const double* sh_emp_block, const double* ns_emp_block,
                              const double* epochs, const double* init_rates, int max_iter,
                              int min_iter, double rel_tol, double rate_floor, double* out_rates,
                              int* out_iters, double* out_loglik, int* out_flags,
                              double* out_cnt_shared, double* out_cnt_notshared) {
  if (int rc = check_sizes(B, E, A)) return rc;
  if (nb < 1 || A < 2) return fail(COLATE_EINVAL, "bad sizes nb=%d A=%d", nb, A);
  if (!age_grid || !weights || !sh_block || !ns_block || !sh_emp_block || !ns_emp_block || !epochs ||
      !init_rates || !out_rates || !out_iters || !out_loglik || !out_flags)
    return fail(COLATE_EINVAL, "NULL pointer argument");
  if (int rc = check_grids(E, A, age_grid, epochs)) return rc;
  if (int rc = ensure_device()) return rc;
  if (B == 0) return COLATE_OK;
  const size_t nT = (size_t)nb * A, nBA = (size_t)B * A, nBE = (size_t)B * E;
  DevBuf d_grid, d_w, d_t[4], d_sh, d_ns, d_ep, d_init, d_rates, d_iters, d_ll, d_flags, d_status;
  HIP_TRY(d_grid.alloc(A * sizeof(double)));
  HIP_TRY(d_w.alloc((size_t)B * nb * sizeof(double)));
  const double* tabs[4] = {sh_block, ns_block, sh_emp_block, ns_emp_block};
  for (int k = 0; k < 4; k++) {
    HIP_TRY(d_t[k].alloc(nT * sizeof(double)));
    HIP_TRY(hipMemcpy(d_t[k].p, tabs[k], nT * sizeof(double), hipMemcpyHostToDevice));
  }
  HIP_TRY(d_sh.alloc(nBA * sizeof(double)));
  HIP_TRY(d_ns.alloc(nBA * sizeof(double)));
  HIP_TRY(d_ep.alloc(E * sizeof(double)));
  HIP_TRY(d_init.alloc(E * sizeof(double)));
  HIP_TRY(d_rates.alloc(nBE * sizeof(double)));
  HIP_TRY(d_iters.alloc(B * sizeof(int)));
  HIP_TRY(d_ll.alloc(B * sizeof(double)));
  HIP_TRY(d_flags.alloc(B * sizeof(int)));
  HIP_TRY(d_status.alloc(sizeof(int)));
  HIP_TRY(hipMemset(d_status.p, 0, sizeof(int)));
  HIP_TRY(hipMemcpy(d_grid.p, age_grid, A * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_w.p, weights, (size_t)B * nb * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_ep.p, epochs, E * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_init.p, init_rates, E * sizeof(double), hipMemcpyHostToDevice));
  int rc = colate_bootstrap_counts_device(B, nb, A, d_grid.as<double>(), age, d_w.as<double>(), d_t[0].as<double>(),
                                          d_t[1].as<double>(), d_t[2].as<double>(), d_t[3].as<double>(),
                                          d_sh.as<double>(), d_ns.as<double>(), d_status.as<int>(), nullptr);
  if (rc) return rc;
  rc = colate_em_batch_device(B, E, A, d_grid.as<double>(), d_sh.as<double>(), d_ns.as<double>(), d_ep.as<double>(), 0,
                              d_init.as<double>(), 0, max_iter, min_iter, rel_tol, rate_floor, d_rates.as<double>(),
                              d_iters.as<int>(), d_ll.as<double>(), d_flags.as<int>(), nullptr);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  int status = 0;
  HIP_TRY(hipMemcpy(&status, d_status.p, sizeof(int), hipMemcpyDeviceToHost));
  if (status) return fail(COLATE_EINVAL, "sample age outside the age grid");
  HIP_TRY(hipMemcpy(out_rates, d_rates.p, nBE * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_iters, d_iters.p, B * sizeof(int), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_loglik, d_ll.p, B * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_flags, d_flags.p, B * sizeof(int), hipMemcpyDeviceToHost));
  if (out_cnt_shared) HIP_TRY(hipMemcpy(out_cnt_shared, d_sh.p, nBA * sizeof(double), hipMemcpyDeviceToHost));
  if (out_cnt_notshared) HIP_TRY(hipMemcpy(out_cnt_notshared, d_ns.p, nBA * sizeof(double), hipMemcpyDeviceToHost));
  return COLATE_OK;
}

int colate_em_batch_rows(int B, int E, int A, const double* age_grid, const double* cnt_shared,
                         const double* cnt_notshared, const double* epochs, const double* init_rates,
                         int max_iter, int min_iter, double rel_tol, double rate_floor,
                         double* out_rates, int* out_iters, double* out_loglik, int* out_flags) {
  if (int rc = check_sizes(B, E, A)) return rc;
  if (!age_grid || !cnt_shared || !cnt_notshared || !epochs || !init_rates || !out_rates ||
      !out_iters || !out_loglik || !out_flags)
    return fail(COLATE_EINVAL, "NULL pointer argument");
  for (int b = 0; b < B; b++)
    if (int rc = check_grids(E, A, age_grid, epochs + (size_t)b * E)) return rc;
  if (int rc = ensure_device()) return rc;
  if (B == 0) return COLATE_OK;
  const size_t nBA = (size_t)B * A, nBE = (size_t)B * E;
  DevBuf d_grid, d_sh, d_ns, d_ep, d_init, d_rates, d_iters, d_ll, d_flags;
  HIP_TRY(d_grid.alloc(A * sizeof(double)));
  HIP_TRY(d_sh.alloc(nBA * sizeof(double)));
  HIP_TRY(d_ns.alloc(nBA * sizeof(double)));
  HIP_TRY(d_ep.alloc(nBE * sizeof(double)));
  HIP_TRY(d_init.alloc(nBE * sizeof(double)));
  HIP_TRY(d_rates.alloc(nBE * sizeof(double)));
  HIP_TRY(d_iters.alloc(B * sizeof(int)));
  HIP_TRY(d_ll.alloc(B * sizeof(double)));
  HIP_TRY(d_flags.alloc(B * sizeof(int)));
  HIP_TRY(hipMemcpy(d_grid.p, age_grid, A * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_sh.p, cnt_shared, nBA * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_ns.p, cnt_notshared, nBA * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_ep.p, epochs, nBE * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_init.p, init_rates, nBE * sizeof(double), hipMemcpyHostToDevice));
  int rc = colate_em_batch_device(B, E, A, d_grid.as<double>(), d_sh.as<double>(),
                                  d_ns.as<double>(), d_ep.as<double>(), 1, d_init.as<double>(), 1,
                                  max_iter, min_iter, rel_tol, rate_floor, d_rates.as<double>(),
                                  d_iters.as<int>(), d_ll.as<double>(), d_flags.as<int>(), nullptr);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out_rates, d_rates.p, nBE * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_iters, d_iters.p, B * sizeof(int), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_loglik, d_ll.p, B * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_flags, d_flags.p, B * sizeof(int), hipMemcpyDeviceToHost));
  return COLATE_OK;
}

// per_row: epochs and init_rates are [B][E] (one grid per replicate row) instead of [E]
static int em_batch_sharded_impl(bool per_row, int num_devices, const int* devices, int B, int E, int A,
                                 const double* age_grid, const double* cnt_shared,
                                 const double* cnt_notshared, const double* epochs,
                                 const double* init_rates, int max_iter, int min_iter, double rel_tol,
                                 double rate_floor, double* out_rates, int* out_iters, double* out_loglik,
                                 int* out_flags) {
  if (num_devices < 1 || !devices) return fail(COLATE_EINVAL, "need at least one device");
  if (int rc = check_sizes(B, E, A)) return rc;
  if (!age_grid || !cnt_shared || !cnt_notshared || !epochs || !init_rates || !out_rates ||
      !out_iters || !out_loglik || !out_flags)
    return fail(COLATE_EINVAL, "NULL pointer argument");
  for (int b = 0; b < (per_row ? B : 1); b++)
    if (int rc = check_grids(E, A, age_grid, epochs + (size_t)b * E)) return rc;
  if (int rc = ensure_device()) return rc;
  int ndev_avail = 0;
  HIP_TRY(hipGetDeviceCount(&ndev_avail));
  for (int d = 0; d < num_devices; d++)
    if (devices[d] < 0 || devices[d] >= ndev_avail)
      return fail(COLATE_EINVAL, "device ordinal %d out of range (%d devices)", devices[d], ndev_avail);
  int prev_dev = 0;
  HIP_TRY(hipGetDevice(&prev_dev));
  struct Shard {
    int lo = 0, n = 0;
    hipStream_t stream = nullptr;
    DevBuf grid, sh, ns, ep, init, rates, iters, ll, flags;
  };
  std::vector<Shard> shards(num_devices);
  int rc = COLATE_OK;
  const int base = B / num_devices, rem = B % num_devices;
  // enqueue everything (copies in, kernel, copies out) on one stream per shard, then wait for all
  for (int d = 0; d < num_devices && rc == COLATE_OK; d++) {
    Shard& s = shards[d];
    s.lo = d * base + (d < rem ? d : rem);
    s.n = base + (d < rem ? 1 : 0);
    if (s.n == 0) continue;
    const size_t nA = (size_t)s.n * A, nE = (size_t)s.n * E;
    const size_t nEp = per_row ? nE : (size_t)E, ep_off = per_row ? (size_t)s.lo * E : 0;  // this shard's epoch rows
    auto step = [&](hipError_t e, const char* what) {
      if (e != hipSuccess && rc == COLATE_OK) rc = hip_fail(e, what);
      return e == hipSuccess;
    };
    if (!step(hipSetDevice(devices[d]), "hipSetDevice")) break;
    if (!step(hipStreamCreate(&s.stream), "hipStreamCreate")) break;
    bool ok = step(s.grid.alloc(A * sizeof(double)), "hipMalloc") && step(s.sh.alloc(nA * sizeof(double)), "hipMalloc") &&
              step(s.ns.alloc(nA * sizeof(double)), "hipMalloc") && step(s.ep.alloc(nEp * sizeof(double)), "hipMalloc") &&
              step(s.init.alloc(nEp * sizeof(double)), "hipMalloc") && step(s.rates.alloc(nE * sizeof(double)), "hipMalloc") &&
              step(s.iters.alloc(s.n * sizeof(int)), "hipMalloc") && step(s.ll.alloc(s.n * sizeof(double)), "hipMalloc") &&
              step(s.flags.alloc(s.n * sizeof(int)), "hipMalloc");
    if (!ok) break;
    ok = step(hipMemcpyAsync(s.grid.p, age_grid, A * sizeof(double), hipMemcpyHostToDevice, s.stream), "copy") &&
         step(hipMemcpyAsync(s.sh.p, cnt_shared + (size_t)s.lo * A, nA * sizeof(double), hipMemcpyHostToDevice, s.stream), "copy") &&
         step(hipMemcpyAsync(s.ns.p, cnt_notshared + (size_t)s.lo * A, nA * sizeof(double), hipMemcpyHostToDevice, s.stream), "copy") &&
         step(hipMemcpyAsync(s.ep.p, epochs + ep_off, nEp * sizeof(double), hipMemcpyHostToDevice, s.stream), "copy") &&
         step(hipMemcpyAsync(s.init.p, init_rates + ep_off, nEp * sizeof(double), hipMemcpyHostToDevice, s.stream), "copy");
    if (!ok) break;
    int r2 = colate_em_batch_device(s.n, E, A, s.grid.as<double>(), s.sh.as<double>(), s.ns.as<double>(),
                                    s.ep.as<double>(), per_row ? 1 : 0, s.init.as<double>(), per_row ? 1 : 0, max_iter,
                                    min_iter, rel_tol, rate_floor, s.rates.as<double>(), s.iters.as<int>(),
                                    s.ll.as<double>(), s.flags.as<int>(), s.stream);
    if (r2) {
      rc = r2;
      break;
    }
    ok = step(hipMemcpyAsync(out_rates + (size_t)s.lo * E, s.rates.p, nE * sizeof(double), hipMemcpyDeviceToHost, s.stream), "copy") &&
         step(hipMemcpyAsync(out_iters + s.lo, s.iters.p, s.n * sizeof(int), hipMemcpyDeviceToHost, s.stream), "copy") &&
         step(hipMemcpyAsync(out_loglik + s.lo, s.ll.p, s.n * sizeof(double), hipMemcpyDeviceToHost, s.stream), "copy") &&
         step(hipMemcpyAsync(out_flags + s.lo, s.flags.p, s.n * sizeof(int), hipMemcpyDeviceToHost, s.stream), "copy");
    if (!ok) break;
  }
  for (int d = 0; d < num_devices; d++) {  // always drain and release, also after an error
    Shard& s = shards[d];
    if (!s.stream) continue;
    (void)hipSetDevice(devices[d]);
    hipError_t e = hipStreamSynchronize(s.stream);
    if (e != hipSuccess && rc == COLATE_OK) rc = hip_fail(e, "hipStreamSynchronize");
    (void)hipStreamDestroy(s.stream);
    s.grid = DevBuf();
  }
  for (int d = 0; d < num_devices; d++) {  // free on the owning device
    (void)hipSetDevice(devices[d]);
    shards[d] = Shard();
  }
  (void)hipSetDevice(prev_dev);
  return rc;
}

int colate_em_batch_sharded(int num_devices, const int* devices, int B, int E, int A,
                            const double* age_grid, const double* cnt_shared,
                            const double* cnt_notshared, const double* epochs,
                            const double* init_rates, int max_iter, int min_iter, double rel_tol,
                            double rate_floor, double* out_rates, int* out_iters, double* out_loglik,
                            int* out_flags) {
  return em_batch_sharded_impl(false, num_devices, devices, B, E, A, age_grid, cnt_shared, cnt_notshared, epochs,
                               init_rates, max_iter, min_iter, rel_tol, rate_floor, out_rates, out_iters, out_loglik,
                               out_flags);
}

int colate_em_batch_rows_sharded(int num_devices, const int* devices, int B, int E, int A,
                                 const double* age_grid, const double* cnt_shared,
                                 const double* cnt_notshared, const double* epochs,
                                 const double* init_rates, int max_iter, int min_iter, double rel_tol,
                                 double rate_floor, double* out_rates, int* out_iters, double* out_loglik,
                                 int* out_flags) {
  return em_batch_sharded_impl(true, num_devices, devices, B, E, A, age_grid, cnt_shared, cnt_notshared, epochs,
                               init_rates, max_iter, min_iter, rel_tol, rate_floor, out_rates, out_iters, out_loglik,
                               out_flags);
}

int colate_em_estep(int B, int E, int A, const double* age_grid, const double* cnt_shared,
                    const double* cnt_notshared, const double* epochs, const double* rates,
                    double* num_acc, double* den_acc, double* loglik, int* flags) {
  if (int rc = check_sizes(B, E, A)) return rc;
  if (!age_grid || !cnt_shared || !cnt_notshared || !epochs || !rates || !num_acc || !den_acc ||
      !loglik || !flags)
    return fail(COLATE_EINVAL, "NULL pointer argument");
  if (int rc = check_grids(E, A, age_grid, epochs)) return rc;
  if (int rc = ensure_device()) return rc;
  if (B == 0) return COLATE_OK;
  const size_t nBA = (size_t)B * A, nBE = (size_t)B * E;
  DevBuf d_grid, d_sh, d_ns, d_ep, d_rates, d_num, d_den, d_ll, d_flags;
  HIP_TRY(d_grid.alloc(A * sizeof(double)));
  HIP_TRY(d_sh.alloc(nBA * sizeof(double)));
  HIP_TRY(d_ns.alloc(nBA * sizeof(double)));
  HIP_TRY(d_ep.alloc(E * sizeof(double)));
  HIP_TRY(d_rates.alloc(nBE * sizeof(double)));
  HIP_TRY(d_num.alloc(nBE * sizeof(double)));
  HIP_TRY(d_den.alloc(nBE * sizeof(double)));
  HIP_TRY(d_ll.alloc(B * sizeof(double)));
  HIP_TRY(d_flags.alloc(B * sizeof(int)));
  HIP_TRY(hipMemcpy(d_grid.p, age_grid, A * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_sh.p, cnt_shared, nBA * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_ns.p, cnt_notshared, nBA * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_ep.p, epochs, E * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_rates.p, rates, nBE * sizeof(double), hipMemcpyHostToDevice));
  int rc = colate_em_estep_device(B, E, A, d_grid.as<double>(), d_sh.as<double>(),
                                  d_ns.as<double>(), d_ep.as<double>(), d_rates.as<double>(),
                                  d_num.as<double>(), d_den.as<double>(), d_ll.as<double>(),
                                  d_flags.as<int>(), nullptr);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(num_acc, d_num.p, nBE * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(den_acc, d_den.p, nBE * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(loglik, d_ll.p, B * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(flags, d_flags.p, B * sizeof(int), hipMemcpyDeviceToHost));
  return COLATE_OK;
}

}  // extern "C"
