// colate_amd/csrc/colate_api.cpp -- extern "C" entry points of libcolate_amd.so
// for the EM hot path (see include/colate_amd.h).  Argument checking, device
// buffers for the host-pointer variants, kernel launch.  No CPU fallback.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <unistd.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "colate_amd.h"
#include "colate_internal.h"
#include "em_kernels.h"

static_assert(COLATE_FLAG_NAN == 1 && COLATE_FLAG_NEG == 2 && COLATE_FLAG_MAXITER == 4 && COLATE_FLAG_UNRESOLVED == 8 &&
                  COLATE_UNRESOLVED_SHIFT == 8, "flags");

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

// ---- profiler ranges (roctx), bound on first use
namespace {
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
};
const Roctx& roctx() {
  static const Roctx r = [] {
    Roctx x;
    if (!std::getenv("COLATE_ROCTX")) return x;  // (opt-in: COLATE_ROCTX=1 under `rocprofv3 --marker-trace`; otherwise no profiler library is loaded)
    for (const char* name : {"librocprofiler-sdk-roctx.so.1", "libroctx64.so.4", "librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
      if (void* h = dlopen(name, RTLD_NOW | RTLD_LOCAL)) {
        x.push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        x.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (x.push && x.pop) break;
        x.push = nullptr, x.pop = nullptr;
      }
    }
    return x;
  }();
  return r;
}
}  // namespace
ProfRange::ProfRange(const char* name) {
  if (roctx().push) roctx().push(name);
}
ProfRange::~ProfRange() {
  if (roctx().pop) roctx().pop();
}

static std::atomic<int> g_device_touched{0};
void mark_device_touched() { g_device_touched.store(1, std::memory_order_relaxed); }

int ensure_device() {
  mark_device_touched();
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(COLATE_ENODEVICE, "no usable HIP device (%s); libcolate_amd has no CPU fallback",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  return COLATE_OK;
}

// grids that the kernel's contiguous-segment logic relies on
int check_grids(int E, int A, const double* age_grid, const double* epochs) {
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

// ---- per-thread workspace of the host-pointer entry points -------------------------------------------
// One device buffer, one pinned host staging buffer and one stream per calling thread, grown on demand and
// kept between calls (colate_release_workspace frees them): a call costs one staged host-to-device copy,
// the launch(es), one device-to-host copy and one stream synchronisation instead of nine hipMalloc/hipFree
// and nine synchronous copies.  A thread that ends while the process lives frees its workspace (thread_local
// destructor below); the main thread's and whatever is left at process exit are not touched (a destructor there
// could run after the HIP runtime has shut down; the driver reclaims everything anyway).
static std::atomic<int> g_process_exiting{0};
struct Workspace {
  int device = -1;
  char* d = nullptr;
  size_t dcap = 0;
  char* h = nullptr;
  size_t hcap = 0;
  hipStream_t stream = nullptr;

  void release() {
    if (device >= 0) {
      int cur = -1;
      (void)hipGetDevice(&cur);
      if (cur != device) (void)hipSetDevice(device);
      if (stream) (void)hipStreamDestroy(stream);
      if (d) (void)hipFree(d);
      if (h) (void)hipHostFree(h);
      if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
    }
    *this = Workspace();
  }
  int reserve(size_t dbytes, size_t hbytes) {
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur));
    if (cur != device) {  // the thread moved to another GPU: start over there
      release();
      device = cur;
    }
    if (!stream) {
      static std::atomic<int> registered{0};  // (behind HIP's own exit handlers in the list, so it runs before them)
      if (!registered.exchange(1)) std::atexit([] { g_process_exiting.store(1); });
      HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    }
    if (dbytes > dcap) {
      if (d) (void)hipFree(d);
      d = nullptr, dcap = 0;
      const size_t want = dbytes + dbytes / 4 + 4096;
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), want));
      dcap = want;
    }
    if (hbytes > hcap) {
      if (h) (void)hipHostFree(h);
      h = nullptr, hcap = 0;
      const size_t want = hbytes + hbytes / 4 + 4096;
      HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h), want, hipHostMallocDefault));
      hcap = want;
    }
    return COLATE_OK;
  }
};
struct WorkspaceOwner {
  Workspace ws;
  ~WorkspaceOwner() {  // a short-lived worker thread must not leak HBM, pinned memory and a stream per thread
    const bool main_thread = (::getpid() == (pid_t)::gettid());
    if (!main_thread && !g_process_exiting.load()) ws.release();
  }
};
static thread_local WorkspaceOwner g_ws_owner;
#define g_ws (g_ws_owner.ws)

// One staged call on the workspace: declare what goes in, what stays on the device and what comes out,
// commit() (one H2D copy), launch on stream(), finish() (one D2H copy, synchronise, scatter to the caller).
class Stage {
 public:
  template <typename T>
  int in(const T* host, size_t n) {
    return add(kIn, host, nullptr, n * sizeof(T));
  }
  template <typename T>
  int out(T* host, size_t n) {  // host may be NULL: the space exists on the device, nothing is returned
    return add(kOut, nullptr, host, n * sizeof(T));
  }
  template <typename T>
  int scratch(size_t n) {
    return add(kScratch, nullptr, nullptr, n * sizeof(T));
  }
  template <typename T>
  T* dev(int idx) const {
    return reinterpret_cast<T*>(ws_->d + seg_[idx].doff);
  }
  hipStream_t stream() const { return ws_->stream; }

  int commit() {
    size_t sizes[3] = {0, 0, 0};
    for (Seg& s : seg_) {
      s.koff = sizes[s.kind];
      sizes[s.kind] += round_up(s.bytes);
    }
    const size_t base[3] = {0, sizes[kIn], sizes[kIn] + sizes[kScratch]};  // device: in | scratch | out
    for (Seg& s : seg_) s.doff = base[s.kind] + s.koff;
    in_bytes_ = sizes[kIn], out_bytes_ = sizes[kOut], out_base_ = base[kOut];
    ws_ = &g_ws;
    if (int rc = ws_->reserve(base[kOut] + sizes[kOut], sizes[kIn] + sizes[kOut])) return rc;  // host: in | out
    for (const Seg& s : seg_)
      if (s.kind == kIn && s.bytes) std::memcpy(ws_->h + s.koff, s.src, s.bytes);
    if (in_bytes_) HIP_TRY(hipMemcpyAsync(ws_->d, ws_->h, in_bytes_, hipMemcpyHostToDevice, ws_->stream));
    return COLATE_OK;
  }
  int finish() {
    char* hout = ws_->h + in_bytes_;
    if (out_bytes_) HIP_TRY(hipMemcpyAsync(hout, ws_->d + out_base_, out_bytes_, hipMemcpyDeviceToHost, ws_->stream));
    HIP_TRY(hipStreamSynchronize(ws_->stream));
    for (const Seg& s : seg_)
      if (s.kind == kOut && s.dst && s.bytes) std::memcpy(s.dst, hout + s.koff, s.bytes);
    return COLATE_OK;
  }
  // staged host copy of an `out` segment after finish() (for values the caller inspects before returning them)
  template <typename T>
  const T* host_out(int idx) const {
    return reinterpret_cast<const T*>(ws_->h + in_bytes_ + seg_[idx].koff);
  }

 private:
  enum Kind { kIn = 0, kScratch = 1, kOut = 2 };
  struct Seg {
    Kind kind;
    const void* src;
    void* dst;
    size_t bytes, koff = 0, doff = 0;
  };
  static size_t round_up(size_t b) { return (b + 255) & ~size_t(255); }
  int add(Kind k, const void* src, void* dst, size_t bytes) {
    seg_.push_back(Seg{k, src, dst, bytes});
    return (int)seg_.size() - 1;
  }
  std::vector<Seg> seg_;
  Workspace* ws_ = nullptr;
  size_t in_bytes_ = 0, out_bytes_ = 0, out_base_ = 0;
};

static int launch(const ColateEmArgs& a, hipStream_t s) {
  mark_device_touched();
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

int colate_device_touched(void) { return g_device_touched.load(std::memory_order_relaxed); }

int colate_device_count(void) {
  mark_device_touched();
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(COLATE_ENODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  return n;
}

int colate_set_device(int ordinal) {
  mark_device_touched();
  HIP_TRY(hipSetDevice(ordinal));
  return COLATE_OK;
}

int colate_warm_up(int ordinal) {
  // creates the HIP context of the device (a few hundred ms in a fresh process): a host that still has input files to
  // parse calls this from a second thread first (mut_driver.cpp)
  mark_device_touched();
  HIP_TRY(hipSetDevice(ordinal));
  HIP_TRY(hipFree(nullptr));
  return COLATE_OK;
}

int colate_em_kernel_variant(int B, int E) {
  if (B < 0 || E < 1 || E > COLATE_EM_MAX_E) return fail(COLATE_EINVAL, "bad sizes B=%d E=%d", B, E);
  if (int rc = ensure_device()) return rc;
  return colate_em_variant(B, E);
}

int colate_em_force_variant(int variant) {
  colate_em_set_forced_variant(variant);
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

// host-pointer EM on the workspace; per_row: epochs / init_rates are [B][E] instead of [E]
static int em_batch_host(bool per_row, int B, int E, int A, const double* age_grid, const double* cnt_shared,
                         const double* cnt_notshared, const double* epochs, const double* init_rates,
                         int max_iter, int min_iter, double rel_tol, double rate_floor, double* out_rates,
                         int* out_iters, double* out_loglik, int* out_flags) {
  if (int rc = check_sizes(B, E, A)) return rc;
  if (!age_grid || !cnt_shared || !cnt_notshared || !epochs || !init_rates || !out_rates ||
      !out_iters || !out_loglik || !out_flags)
    return fail(COLATE_EINVAL, "NULL pointer argument");
  for (int b = 0; b < (per_row ? B : 1); b++)
    if (int rc = check_grids(E, A, age_grid, epochs + (size_t)b * E)) return rc;
  if (int rc = ensure_device()) return rc;
  if (B == 0) return COLATE_OK;
  ProfRange range("colate_em_batch: H2D + EM kernel + D2H");
  const size_t nBA = (size_t)B * A, nBE = (size_t)B * E, nEp = per_row ? nBE : (size_t)E;
  Stage st;
  const int i_grid = st.in(age_grid, A), i_sh = st.in(cnt_shared, nBA), i_ns = st.in(cnt_notshared, nBA);
  const int i_ep = st.in(epochs, nEp), i_init = st.in(init_rates, nEp);
  const int o_rates = st.out(out_rates, nBE), o_ll = st.out(out_loglik, B), o_iters = st.out(out_iters, B),
            o_flags = st.out(out_flags, B);
  // COLATE_LL_TRACE=<file>: the log-likelihood of every iteration (the reference's commented-out trace, coal.cpp:3659,
  // 3674, 3817, 3821), "replicate iteration loglik" per line.  Diagnostic: the run then takes the general loop with the
  // log-likelihood evaluated in every iteration (same rates, slower); at most the first kTraceCap iterations are kept.
  const char* trace_path = std::getenv("COLATE_LL_TRACE");
  constexpr int kTraceCap = 8192;
  const int cap = trace_path ? (max_iter < kTraceCap ? max_iter : kTraceCap) : 0;
  std::vector<double> trace;
  int o_trace = -1;
  if (cap > 0) {
    trace.resize((size_t)B * cap);
    o_trace = st.out(trace.data(), trace.size());
  }
  if (int rc = st.commit()) return rc;
  if (cap > 0) {
    HIP_TRY(hipMemsetAsync(st.dev<double>(o_trace), 0xff, trace.size() * sizeof(double), st.stream()));  // NaN = "not reached"
    if (int rc = check_sizes(B, E, A)) return rc;
    ColateEmArgs a{};
    a.B = B, a.E = E, a.A = A, a.mode = 0;
    a.age_grid = st.dev<double>(i_grid), a.cnt_sh = st.dev<double>(i_sh), a.cnt_ns = st.dev<double>(i_ns);
    a.epochs = st.dev<double>(i_ep), a.epochs_stride = per_row ? E : 0;
    a.rates_in = st.dev<double>(i_init), a.rates_stride = per_row ? E : 0;
    a.max_iter = max_iter, a.min_iter = min_iter, a.rel_tol = rel_tol, a.rate_floor = rate_floor;
    a.out_rates = st.dev<double>(o_rates), a.out_iters = st.dev<int>(o_iters), a.out_ll = st.dev<double>(o_ll);
    a.out_flags = st.dev<int>(o_flags);
    a.ll_trace = st.dev<double>(o_trace), a.ll_trace_cap = cap;
    if (max_iter < 1) return fail(COLATE_EINVAL, "max_iter must be >= 1");
    if (int rc = launch(a, st.stream())) return rc;
    if (int rc = st.finish()) return rc;
    FILE* f = std::fopen(trace_path, "w");
    if (!f) return fail(COLATE_EIO, "cannot write %s", trace_path);
    for (int b = 0; b < B; b++)
      for (int it = 0; it < cap; it++) {
        const double v = trace[(size_t)b * cap + it];
        if (v == v) std::fprintf(f, "%d %d %.17g\n", b, it, v);
      }
    std::fclose(f);
    return COLATE_OK;
  }
  if (int rc = colate_em_batch_device(B, E, A, st.dev<double>(i_grid), st.dev<double>(i_sh), st.dev<double>(i_ns),
                                      st.dev<double>(i_ep), per_row, st.dev<double>(i_init), per_row, max_iter,
                                      min_iter, rel_tol, rate_floor, st.dev<double>(o_rates), st.dev<int>(o_iters),
                                      st.dev<double>(o_ll), st.dev<int>(o_flags), st.stream()))
    return rc;
  return st.finish();
}

int colate_em_batch(int B, int E, int A, const double* age_grid, const double* cnt_shared,
                    const double* cnt_notshared, const double* epochs, const double* init_rates,
                    int max_iter, int min_iter, double rel_tol, double rate_floor,
                    double* out_rates, int* out_iters, double* out_loglik, int* out_flags) {
  return em_batch_host(false, B, E, A, age_grid, cnt_shared, cnt_notshared, epochs, init_rates, max_iter, min_iter,
                       rel_tol, rate_floor, out_rates, out_iters, out_loglik, out_flags);
}

int colate_release_workspace(void) {
  g_ws.release();
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
  ProfRange range("colate_bootstrap_em_batch: H2D + bootstrap kernel + EM kernel + D2H");
  const size_t nT = (size_t)nb * A, nBA = (size_t)B * A, nBE = (size_t)B * E;
  Stage st;
  const int i_grid = st.in(age_grid, A), i_w = st.in(weights, (size_t)B * nb);
  const int i_t0 = st.in(sh_block, nT), i_t1 = st.in(ns_block, nT), i_t2 = st.in(sh_emp_block, nT),
            i_t3 = st.in(ns_emp_block, nT);
  const int i_ep = st.in(epochs, E), i_init = st.in(init_rates, E);
  const int zero = 0;
  const int i_status = st.in(&zero, 1);  // device int the bootstrap kernel ORs into; read back with the outputs
  const int o_rates = st.out(out_rates, nBE), o_ll = st.out(out_loglik, B), o_iters = st.out(out_iters, B),
            o_flags = st.out(out_flags, B);
  // the count tables stay on the device between the two kernels; they travel back only if asked for
  const bool want_counts = out_cnt_shared || out_cnt_notshared;
  const int c_sh = want_counts ? st.out(out_cnt_shared, nBA) : st.scratch<double>(nBA);
  const int c_ns = want_counts ? st.out(out_cnt_notshared, nBA) : st.scratch<double>(nBA);
  int status_host = 0;
  const int o_status = st.out(&status_host, 1);
  if (int rc = st.commit()) return rc;
  if (int rc = colate_bootstrap_counts_device(B, nb, A, st.dev<double>(i_grid), age, st.dev<double>(i_w),
                                              st.dev<double>(i_t0), st.dev<double>(i_t1), st.dev<double>(i_t2),
                                              st.dev<double>(i_t3), st.dev<double>(c_sh), st.dev<double>(c_ns),
                                              st.dev<int>(i_status), st.stream()))
    return rc;
  if (int rc = colate_em_batch_device(B, E, A, st.dev<double>(i_grid), st.dev<double>(c_sh), st.dev<double>(c_ns),
                                      st.dev<double>(i_ep), 0, st.dev<double>(i_init), 0, max_iter, min_iter, rel_tol,
                                      rate_floor, st.dev<double>(o_rates), st.dev<int>(o_iters), st.dev<double>(o_ll),
                                      st.dev<int>(o_flags), st.stream()))
    return rc;
  HIP_TRY(hipMemcpyAsync(st.dev<int>(o_status), st.dev<int>(i_status), sizeof(int), hipMemcpyDeviceToDevice, st.stream()));
  if (int rc = st.finish()) return rc;
  if (status_host) return fail(COLATE_EINVAL, "sample age outside the age grid");
  return COLATE_OK;
}

int colate_bootstrap_counts_groups_device(int G, int B, int group_first, int row_lo, int row_hi, int A,
                                          const double* age_grid, const int* group_nb,
                                          const long long* group_block_off, const long long* group_weight_off,
                                          const double* group_age, const double* weights, const double* sh_block,
                                          const double* ns_block, const double* sh_emp_block,
                                          const double* ns_emp_block, double* cnt_shared, double* cnt_notshared,
                                          int* status, void* hip_stream) {
  if (G < 1 || B < 1 || A < 2 || A > COLATE_EM_MAX_A || group_first < 0 || row_lo < 0 || row_hi < row_lo)
    return fail(COLATE_EINVAL, "bad sizes G=%d B=%d A=%d rows [%d, %d)", G, B, A, row_lo, row_hi);
  if (row_hi > row_lo && (row_lo / B < group_first || (row_hi - 1) / B >= group_first + G))
    return fail(COLATE_EINVAL, "rows [%d, %d) lie outside groups [%d, %d)", row_lo, row_hi, group_first, group_first + G);
  if (!age_grid || !group_nb || !group_block_off || !group_weight_off || !group_age || !weights || !sh_block ||
      !ns_block || !sh_emp_block || !ns_emp_block || !cnt_shared || !cnt_notshared || !status)
    return fail(COLATE_EINVAL, "NULL pointer argument");
  if (row_hi == row_lo) return COLATE_OK;
  mark_device_touched();
  hipError_t e = colate_bootstrap_groups_launch(B, row_lo, row_hi - row_lo, group_first, A, age_grid, group_nb,
                                                group_block_off, group_weight_off, group_age, weights, sh_block,
                                                ns_block, sh_emp_block, ns_emp_block, cnt_shared, cnt_notshared,
                                                status, static_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "bootstrap kernel launch");
  return COLATE_OK;
}

int colate_bootstrap_em_batch_groups(int G, int B, int E, int A, const double* age_grid, const int* group_nb,
                                     const double* group_age, const double* weights, const double* sh_block,
                                     const double* ns_block, const double* sh_emp_block,
                                     const double* ns_emp_block, const double* epochs, const double* init_rates,
                                     int max_iter, int min_iter, double rel_tol, double rate_floor,
                                     double* out_rates, int* out_iters, double* out_loglik, int* out_flags,
                                     double* out_cnt_shared, double* out_cnt_notshared) {
  if (G < 0 || B < 1) return fail(COLATE_EINVAL, "bad sizes G=%d B=%d", G, B);
  if ((long long)G * B > 0x7fffffffLL) return fail(COLATE_ELIMIT, "G x B = %lld rows", (long long)G * B);
  const int R = G * B;
  if (int rc = check_sizes(R, E, A)) return rc;
  if (A < 2) return fail(COLATE_EINVAL, "bad sizes A=%d", A);
  if (!age_grid || !group_nb || !group_age || !weights || !sh_block || !ns_block || !sh_emp_block || !ns_emp_block ||
      !epochs || !init_rates || !out_rates || !out_iters || !out_loglik || !out_flags)
    return fail(COLATE_EINVAL, "NULL pointer argument");
  if (max_iter < 1) return fail(COLATE_EINVAL, "max_iter must be >= 1");
  std::vector<long long> block_off(G), weight_off(G);
  long long nblocks = 0, nweights = 0;
  for (int g = 0; g < G; g++) {
    if (group_nb[g] < 1) return fail(COLATE_EINVAL, "group %d has %d genome blocks", g, group_nb[g]);
    if (int rc = check_grids(E, A, age_grid, epochs + (size_t)g * E)) return rc;
    block_off[g] = nblocks, weight_off[g] = nweights;
    nblocks += group_nb[g], nweights += (long long)B * group_nb[g];
  }
  if (int rc = ensure_device()) return rc;
  if (G == 0) return COLATE_OK;
  ProfRange range("colate_bootstrap_em_batch_groups: H2D + bootstrap kernel + EM kernel + D2H");
  const size_t nT = (size_t)nblocks * A, nRA = (size_t)R * A, nRE = (size_t)R * E;
  // epochs and starting rates per row (the EM kernel's per-replicate layout)
  std::vector<double> row_ep(nRE), row_init(nRE);
  for (int r = 0; r < R; r++) {
    std::memcpy(row_ep.data() + (size_t)r * E, epochs + (size_t)(r / B) * E, (size_t)E * sizeof(double));
    std::memcpy(row_init.data() + (size_t)r * E, init_rates + (size_t)(r / B) * E, (size_t)E * sizeof(double));
  }
  Stage st;
  const int i_grid = st.in(age_grid, A), i_w = st.in(weights, (size_t)nweights);
  const int i_t0 = st.in(sh_block, nT), i_t1 = st.in(ns_block, nT), i_t2 = st.in(sh_emp_block, nT), i_t3 = st.in(ns_emp_block, nT);
  const int i_nb = st.in(group_nb, G), i_bo = st.in(block_off.data(), G), i_wo = st.in(weight_off.data(), G),
            i_age = st.in(group_age, G);
  const int i_ep = st.in(row_ep.data(), nRE), i_init = st.in(row_init.data(), nRE);
  const int zero = 0;
  const int i_status = st.in(&zero, 1);
  const int o_rates = st.out(out_rates, nRE), o_ll = st.out(out_loglik, R), o_iters = st.out(out_iters, R),
            o_flags = st.out(out_flags, R);
  const bool want_counts = out_cnt_shared || out_cnt_notshared;
  const int c_sh = want_counts ? st.out(out_cnt_shared, nRA) : st.scratch<double>(nRA);
  const int c_ns = want_counts ? st.out(out_cnt_notshared, nRA) : st.scratch<double>(nRA);
  int status_host = 0;
  const int o_status = st.out(&status_host, 1);
  if (int rc = st.commit()) return rc;
  if (int rc = colate_bootstrap_counts_groups_device(G, B, 0, 0, R, A, st.dev<double>(i_grid), st.dev<int>(i_nb),
                                                     st.dev<long long>(i_bo), st.dev<long long>(i_wo), st.dev<double>(i_age),
                                                     st.dev<double>(i_w), st.dev<double>(i_t0), st.dev<double>(i_t1),
                                                     st.dev<double>(i_t2), st.dev<double>(i_t3), st.dev<double>(c_sh),
                                                     st.dev<double>(c_ns), st.dev<int>(i_status), st.stream()))
    return rc;
  if (int rc = colate_em_batch_device(R, E, A, st.dev<double>(i_grid), st.dev<double>(c_sh), st.dev<double>(c_ns),
                                      st.dev<double>(i_ep), 1, st.dev<double>(i_init), 1, max_iter, min_iter, rel_tol,
                                      rate_floor, st.dev<double>(o_rates), st.dev<int>(o_iters), st.dev<double>(o_ll),
                                      st.dev<int>(o_flags), st.stream()))
    return rc;
  HIP_TRY(hipMemcpyAsync(st.dev<int>(o_status), st.dev<int>(i_status), sizeof(int), hipMemcpyDeviceToDevice, st.stream()));
  if (int rc = st.finish()) return rc;
  if (status_host) return fail(COLATE_EINVAL, "sample age outside the age grid");
  return COLATE_OK;
}

int colate_em_batch_rows(int B, int E, int A, const double* age_grid, const double* cnt_shared,
                         const double* cnt_notshared, const double* epochs, const double* init_rates,
                         int max_iter, int min_iter, double rel_tol, double rate_floor,
                         double* out_rates, int* out_iters, double* out_loglik, int* out_flags) {
  return em_batch_host(true, B, E, A, age_grid, cnt_shared, cnt_notshared, epochs, init_rates, max_iter, min_iter,
                       rel_tol, rate_floor, out_rates, out_iters, out_loglik, out_flags);
}

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
    bool launched = false;
    hipStream_t stream = nullptr;
    DevBuf grid, sh, ns, ep, init, rates, iters, ll, flags;
  };
  std::vector<Shard> shards(num_devices);
  int rc = COLATE_OK;
  const int base = B / num_devices, rem = B % num_devices;
  auto step = [&](hipError_t e, const char* what) {
    if (e != hipSuccess && rc == COLATE_OK) rc = hip_fail(e, what);
    return e == hipSuccess;
  };
  // Pass 1: per shard, allocate, copy in and launch.  The caller's buffers are pageable, so each copy-in blocks
  // the host until it is done -- but nothing here waits for a KERNEL, so every GPU has its launch queued before
  // the first result is asked for.
  for (int d = 0; d < num_devices && rc == COLATE_OK; d++) {
    Shard& s = shards[d];
    s.lo = d * base + (d < rem ? d : rem);
    s.n = base + (d < rem ? 1 : 0);
    if (s.n == 0) continue;
    const size_t nA = (size_t)s.n * A, nE = (size_t)s.n * E;
    const size_t nEp = per_row ? nE : (size_t)E, ep_off = per_row ? (size_t)s.lo * E : 0;  // this shard's epoch rows
    if (!step(hipSetDevice(devices[d]), "hipSetDevice")) break;
    if (!step(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking), "hipStreamCreate")) break;
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
    s.launched = true;
  }
  // Pass 2: collect.  The first copy-out of a shard waits for that shard's kernel only; the other GPUs keep running.
  for (int d = 0; d < num_devices && rc == COLATE_OK; d++) {
    Shard& s = shards[d];
    if (!s.launched) continue;
    const size_t nE = (size_t)s.n * E;
    if (!step(hipSetDevice(devices[d]), "hipSetDevice")) break;
    bool ok = step(hipMemcpyAsync(out_rates + (size_t)s.lo * E, s.rates.p, nE * sizeof(double), hipMemcpyDeviceToHost, s.stream), "copy") &&
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
    s.stream = nullptr;
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
  Stage st;
  const int i_grid = st.in(age_grid, A), i_sh = st.in(cnt_shared, nBA), i_ns = st.in(cnt_notshared, nBA);
  const int i_ep = st.in(epochs, E), i_rates = st.in(rates, nBE);
  const int o_num = st.out(num_acc, nBE), o_den = st.out(den_acc, nBE), o_ll = st.out(loglik, B), o_flags = st.out(flags, B);
  if (int rc = st.commit()) return rc;
  if (int rc = colate_em_estep_device(B, E, A, st.dev<double>(i_grid), st.dev<double>(i_sh), st.dev<double>(i_ns),
                                      st.dev<double>(i_ep), st.dev<double>(i_rates), st.dev<double>(o_num),
                                      st.dev<double>(o_den), st.dev<double>(o_ll), st.dev<int>(o_flags), st.stream()))
    return rc;
  return st.finish();
}

}  // extern "C"
