// colate_amd/csrc/colate_comm.cpp -- the one-process-per-GPU form of the replicate sharding, in C++:
// every rank runs its contiguous range of bootstrap replicates on its own GPU and ONE ncclAllGather (RCCL over
// xGMI) hands every rank all results.  There is no counterpart in the reference (its mut() loops over the
// replicates sequentially, include/coal/coal.cpp:3675-3846); SURVEY.md section 8(e) defines this row.
//
// RCCL is bound lazily with dlopen: a process that never creates a communicator never loads it, and inside a
// process that already carries an RCCL (PyTorch ships its own librccl.so.1) the same copy is used instead of a
// second one with the same SONAME.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "colate_amd.h"
#include "colate_internal.h"

static_assert(COLATE_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

namespace {

using colate::check_grids;
using colate::ensure_device;
using colate::fail;

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  const char* error = nullptr;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
      r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.handle) break;
    }
    if (!r.handle) {
      r.error = "librccl.so.1 could not be loaded";
      return;
    }
    auto sym = [&](const char* n) {
      void* p = dlsym(r.handle, n);
      if (!p) r.error = "RCCL symbol missing";
      return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  });
  return r;
}

int rccl_ready() {
  Rccl& r = rccl();
  if (r.error) return fail(COLATE_EHIP, "RCCL unavailable: %s", r.error);
  return COLATE_OK;
}

#define NCCL_TRY(expr)                                                                         \
  do {                                                                                         \
    ncclResult_t r_ = (expr);                                                                  \
    if (r_ != ncclSuccess) return fail(COLATE_EHIP, "%s: %s", #expr, rccl().GetErrorString(r_)); \
  } while (0)
#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) return fail(COLATE_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

struct Comm {
  ncclComm_t nccl = nullptr;
  int nranks = 1, rank = 0, device = 0;
  hipStream_t stream = nullptr;
  char* d_send = nullptr;
  char* d_recv = nullptr;
  size_t cap = 0;  // bytes per rank the two buffers are sized for
  char* h_recv = nullptr;
  size_t hcap = 0;
};

// bytes of one rank's packed results: rates[n_max][E] f64 | loglik[n_max] f64 | iters[n_max] i32 | flags[n_max] i32 |
// this rank's return code (i32, padded to 8): a rank whose local work failed still takes part in the collective --
// otherwise the others would wait for it forever -- and every rank learns of the failure from the gathered codes
size_t payload_bytes(int n_max, int E) { return (size_t)n_max * ((size_t)E * 8 + 8 + 4 + 4); }
size_t packed_bytes(int n_max, int E) { return ((payload_bytes(n_max, E) + 7) & ~size_t(7)) + 8; }

// the two device buffers of the collective; without them this rank cannot take part in it
int reserve_device(Comm* c, size_t per_rank) {
  if (per_rank > c->cap) {
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    c->d_send = c->d_recv = nullptr, c->cap = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_send), per_rank));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_recv), per_rank * c->nranks));
    c->cap = per_rank;
  }
  return COLATE_OK;
}
// the pinned landing buffer of the gathered results
int reserve_host(Comm* c, size_t per_rank) {
  if (per_rank * c->nranks > c->hcap) {
    if (c->h_recv) (void)hipHostFree(c->h_recv);
    c->h_recv = nullptr, c->hcap = 0;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_recv), per_rank * c->nranks, hipHostMallocDefault));
    c->hcap = per_rank * c->nranks;
  }
  return COLATE_OK;
}

// Run `local(n, lo, d_rates, d_ll, d_iters, d_flags, stream)` for this rank's replicates [lo, lo+n) with the four
// outputs placed directly in the send buffer, all-gather, and scatter every rank's part into the caller's arrays.
template <typename Local>
int run_and_gather(Comm* c, int B, int E, double* out_rates, int* out_iters, double* out_loglik, int* out_flags,
                   Local&& local) {
  // Everything that can fail on THIS rank before the collective is turned into `local_rc`, and the rank still joins the
  // all-gather with its code in the trailing slot: a rank that returned early would leave the others waiting in
  // ncclAllGather forever.  The one exception is a rank that cannot even get its two device buffers (or its device): it
  // has nothing to join with and returns -- `Colate --ranks` (run_ranked) then ends the remaining ranks after a grace
  // period; a host that drives the ranks itself needs the same watchdog.
  HIP_TRY(hipSetDevice(c->device));
  const int n_max = (B + c->nranks - 1) / c->nranks;
  const size_t per_rank = packed_bytes(n_max, E);
  if (int rc = reserve_device(c, per_rank)) return rc;
  int local_rc = reserve_host(c, per_rank);
  int lo = 0, hi = 0;
  colate_shard_bounds(B, c->nranks, c->rank, &lo, &hi);
  auto carve = [&](char* base, double*& rates, double*& ll, int*& iters, int*& flags) {
    rates = reinterpret_cast<double*>(base);
    ll = rates + (size_t)n_max * E;
    iters = reinterpret_cast<int*>(ll + n_max);
    flags = iters + n_max;
  };
  double *d_rates, *d_ll;
  int *d_iters, *d_flags;
  carve(c->d_send, d_rates, d_ll, d_iters, d_flags);
  if (!local_rc) {  // rows of a short shard beyond its n stay zero
    const hipError_t e = hipMemsetAsync(c->d_send, 0, per_rank, c->stream);
    if (e != hipSuccess) local_rc = fail(COLATE_EHIP, "hipMemsetAsync: %s", hipGetErrorString(e));
  }
#ifdef COLATE_TEST_HOOKS  // (only in lib/testhooks/libcolate_amd.so: failure injection for the tests of exactly this path)
  if (const char* inj = getenv("COLATE_TEST_FAIL_RANK")) {
    if (atoi(inj) == c->rank) local_rc = fail(COLATE_EHIP, "injected failure on rank %d (COLATE_TEST_FAIL_RANK)", c->rank);
  }
#endif
  if (!local_rc && hi > lo) {
    colate::ProfRange range("colate shard: bootstrap + EM on this rank's replicates");
    local_rc = local(hi - lo, lo, d_rates, d_ll, d_iters, d_flags, c->stream);
  }
  std::string local_msg = local_rc ? colate_last_error() : "";
  if (local_rc) (void)hipMemcpyAsync(c->d_send + per_rank - 8, &local_rc, sizeof(int), hipMemcpyHostToDevice, c->stream);
  // the ONE collective of the path: per_rank bytes from every rank to every rank
  colate::ProfRange gather_range("colate all-gather of the packed results (RCCL)");
  NCCL_TRY(rccl().AllGather(c->d_send, c->d_recv, per_rank, ncclChar, c->nccl, c->stream));
  if (c->h_recv && c->hcap >= per_rank * c->nranks)
    HIP_TRY(hipMemcpyAsync(c->h_recv, c->d_recv, per_rank * c->nranks, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (local_rc) return fail(local_rc, "%s", local_msg.c_str());  // this rank's own message
  for (int r = 0; r < c->nranks; r++) {
    int code = 0;
    std::memcpy(&code, c->h_recv + (size_t)(r + 1) * per_rank - 8, sizeof(int));
    if (code) return fail(code, "rank %d of %d failed (code %d); see its own error output", r, c->nranks, code);
  }
  for (int r = 0; r < c->nranks; r++) {
    int rlo = 0, rhi = 0;
    colate_shard_bounds(B, c->nranks, r, &rlo, &rhi);
    const size_t n = (size_t)(rhi - rlo);
    double *h_rates, *h_ll;
    int *h_iters, *h_flags;
    carve(c->h_recv + (size_t)r * per_rank, h_rates, h_ll, h_iters, h_flags);
    std::memcpy(out_rates + (size_t)rlo * E, h_rates, n * E * sizeof(double));
    std::memcpy(out_loglik + rlo, h_ll, n * sizeof(double));
    std::memcpy(out_iters + rlo, h_iters, n * sizeof(int));
    std::memcpy(out_flags + rlo, h_flags, n * sizeof(int));
  }
  return COLATE_OK;
}

// device copies of host arrays that live for one call
struct Upload {
  std::vector<void*> ptrs;
  ~Upload() {
    for (void* p : ptrs) (void)hipFree(p);
  }
  template <typename T>
  int put(const T* host, size_t n, T** dev, hipStream_t s) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, n * sizeof(T) + 8));
    ptrs.push_back(p);
    if (host) HIP_TRY(hipMemcpyAsync(p, host, n * sizeof(T), hipMemcpyHostToDevice, s));
    *dev = static_cast<T*>(p);
    return COLATE_OK;
  }
};

}  // namespace

extern "C" {

int colate_shard_bounds(int B, int nranks, int rank, int* lo, int* hi) {
  if (B < 0 || nranks < 1 || rank < 0 || rank >= nranks || !lo || !hi) return fail(COLATE_EINVAL, "bad shard arguments");
  const int base = B / nranks, rem = B % nranks;
  *lo = rank * base + (rank < rem ? rank : rem);
  *hi = *lo + base + (rank < rem ? 1 : 0);
  return COLATE_OK;
}

int colate_comm_unique_id(void* id) {
  if (!id) return fail(COLATE_EINVAL, "NULL pointer argument");
  colate::mark_device_touched();
  if (int rc = rccl_ready()) return rc;
  ncclUniqueId u;
  NCCL_TRY(rccl().GetUniqueId(&u));
  std::memcpy(id, &u, sizeof(u));
  return COLATE_OK;
}

int colate_comm_create(const void* id, int nranks, int rank, void** comm) {
  if (!id || !comm || nranks < 1 || rank < 0 || rank >= nranks) return fail(COLATE_EINVAL, "bad communicator arguments");
  if (int rc = rccl_ready()) return rc;
  colate::mark_device_touched();
  Comm* c = new Comm;
  c->nranks = nranks, c->rank = rank;
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  hipError_t e = hipGetDevice(&c->device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete c;
    return fail(COLATE_ENODEVICE, "colate_comm_create: %s", hipGetErrorString(e));
  }
  ncclResult_t r = rccl().CommInitRank(&c->nccl, nranks, u, rank);
  if (r != ncclSuccess) {
    (void)hipStreamDestroy(c->stream);
    delete c;
    return fail(COLATE_EHIP, "ncclCommInitRank: %s", rccl().GetErrorString(r));
  }
  *comm = c;
  return COLATE_OK;
}

int colate_comm_destroy(void* comm) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c) return COLATE_OK;
  (void)hipSetDevice(c->device);
  if (c->nccl) (void)rccl().CommDestroy(c->nccl);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->d_send) (void)hipFree(c->d_send);
  if (c->d_recv) (void)hipFree(c->d_recv);
  if (c->h_recv) (void)hipHostFree(c->h_recv);
  delete c;
  return COLATE_OK;
}

int colate_em_batch_allgather(void* comm, int B, int E, int A, const double* age_grid, const double* cnt_shared,
                              const double* cnt_notshared, const double* epochs, const double* init_rates,
                              int max_iter, int min_iter, double rel_tol, double rate_floor, double* out_rates,
                              int* out_iters, double* out_loglik, int* out_flags) {
  Comm* c = static_cast<Comm*>(comm);
  if (!age_grid || !cnt_shared || !cnt_notshared || !epochs || !init_rates || !out_rates || !out_iters ||
      !out_loglik || !out_flags)
    return fail(COLATE_EINVAL, "NULL pointer argument");
  if (B < 0 || E < 1 || A < 1 || E > COLATE_MAX_EPOCHS || A > COLATE_MAX_AGE_BINS)
    return fail(COLATE_EINVAL, "bad sizes B=%d E=%d A=%d", B, E, A);
  // (the same inputs on every rank: a bad grid is refused by all of them alike, none enters the collective)
  if (int rc = check_grids(E, A, age_grid, epochs)) return rc;
  if (!c) return fail(COLATE_EINVAL, "NULL communicator");
  if (int rc = ensure_device()) return rc;
  if (B == 0) return COLATE_OK;
  Upload up;
  return run_and_gather(c, B, E, out_rates, out_iters, out_loglik, out_flags,
                        [&](int n, int lo, double* d_rates, double* d_ll, int* d_iters, int* d_flags, hipStream_t s) {
                          double *d_grid, *d_sh, *d_ns, *d_ep, *d_init;
                          if (int rc = up.put(age_grid, A, &d_grid, s)) return rc;
                          if (int rc = up.put(cnt_shared + (size_t)lo * A, (size_t)n * A, &d_sh, s)) return rc;
                          if (int rc = up.put(cnt_notshared + (size_t)lo * A, (size_t)n * A, &d_ns, s)) return rc;
                          if (int rc = up.put(epochs, E, &d_ep, s)) return rc;
                          if (int rc = up.put(init_rates, E, &d_init, s)) return rc;
                          return colate_em_batch_device(n, E, A, d_grid, d_sh, d_ns, d_ep, 0, d_init, 0, max_iter, min_iter,
                                                        rel_tol, rate_floor, d_rates, d_iters, d_ll, d_flags, s);
                        });
}

int colate_bootstrap_em_batch_allgather(void* comm, int B, int nb, int E, int A, const double* age_grid, double age,
                                        const double* weights, const double* sh_block, const double* ns_block,
                                        const double* sh_emp_block, const double* ns_emp_block, const double* epochs,
                                        const double* init_rates, int max_iter, int min_iter, double rel_tol,
                                        double rate_floor, double* out_rates, int* out_iters, double* out_loglik,
                                        int* out_flags) {
  Comm* c = static_cast<Comm*>(comm);
  if (!age_grid || !weights || !sh_block || !ns_block || !sh_emp_block || !ns_emp_block || !epochs ||
      !init_rates || !out_rates || !out_iters || !out_loglik || !out_flags)
    return fail(COLATE_EINVAL, "NULL pointer argument");
  if (B < 0 || nb < 1 || E < 1 || A < 2 || E > COLATE_MAX_EPOCHS || A > COLATE_MAX_AGE_BINS)
    return fail(COLATE_EINVAL, "bad sizes B=%d nb=%d E=%d A=%d", B, nb, E, A);
  if (int rc = check_grids(E, A, age_grid, epochs)) return rc;
  if (!c) return fail(COLATE_EINVAL, "NULL communicator");
  if (int rc = ensure_device()) return rc;
  if (B == 0) return COLATE_OK;
  Upload up;
  int* d_status = nullptr;
  int rc = run_and_gather(
      c, B, E, out_rates, out_iters, out_loglik, out_flags,
      [&](int n, int lo, double* d_rates, double* d_ll, int* d_iters, int* d_flags, hipStream_t s) {
        const size_t nT = (size_t)nb * A;
        double *d_grid, *d_w, *d_t0, *d_t1, *d_t2, *d_t3, *d_ep, *d_init, *d_sh, *d_ns;
        const int zero = 0;
        if (int r = up.put(age_grid, A, &d_grid, s)) return r;
        if (int r = up.put(weights + (size_t)lo * nb, (size_t)n * nb, &d_w, s)) return r;  // this rank's rows of w[B][nb]
        if (int r = up.put(sh_block, nT, &d_t0, s)) return r;
        if (int r = up.put(ns_block, nT, &d_t1, s)) return r;
        if (int r = up.put(sh_emp_block, nT, &d_t2, s)) return r;
        if (int r = up.put(ns_emp_block, nT, &d_t3, s)) return r;
        if (int r = up.put(epochs, E, &d_ep, s)) return r;
        if (int r = up.put(init_rates, E, &d_init, s)) return r;
        if (int r = up.put<double>(nullptr, (size_t)n * A, &d_sh, s)) return r;
        if (int r = up.put<double>(nullptr, (size_t)n * A, &d_ns, s)) return r;
        if (int r = up.put(&zero, 1, &d_status, s)) return r;
        HIP_TRY(hipStreamSynchronize(s));  // (`zero` leaves scope)
        if (int r = colate_bootstrap_counts_device(n, nb, A, d_grid, age, d_w, d_t0, d_t1, d_t2, d_t3, d_sh, d_ns, d_status, s))
          return r;
        return colate_em_batch_device(n, E, A, d_grid, d_sh, d_ns, d_ep, 0, d_init, 0, max_iter, min_iter, rel_tol,
                                      rate_floor, d_rates, d_iters, d_ll, d_flags, s);
      });
  if (rc) return rc;
  if (d_status) {
    int status = 0;
    HIP_TRY(hipMemcpy(&status, d_status, sizeof(int), hipMemcpyDeviceToHost));
    if (status) return fail(COLATE_EINVAL, "sample age outside the age grid");
  }
  return COLATE_OK;
}

int colate_bootstrap_em_batch_groups_allgather(void* comm, int G, int B, int group_first, int group_count, int E, int A,
                                               const double* age_grid, const int* group_nb, const double* group_age,
                                               const double* weights, const double* sh_block, const double* ns_block,
                                               const double* sh_emp_block, const double* ns_emp_block,
                                               const double* epochs, const double* init_rates, int max_iter,
                                               int min_iter, double rel_tol, double rate_floor, double* out_rates,
                                               int* out_iters, double* out_loglik, int* out_flags) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c) return fail(COLATE_EINVAL, "NULL communicator");
  if (G < 0 || B < 1 || E < 1 || A < 2 || E > COLATE_MAX_EPOCHS || A > COLATE_MAX_AGE_BINS || (long long)G * B > 0x7fffffffLL)
    return fail(COLATE_EINVAL, "bad sizes G=%d B=%d E=%d A=%d", G, B, E, A);
  if (!age_grid || !out_rates || !out_iters || !out_loglik || !out_flags) return fail(COLATE_EINVAL, "NULL pointer argument");
  if (max_iter < 1) return fail(COLATE_EINVAL, "max_iter must be >= 1");
  const int R = G * B;
  int lo = 0, hi = 0;
  colate_shard_bounds(R, c->nranks, c->rank, &lo, &hi);
  // A rank with rows must bring exactly the groups they belong to.  A mismatch is this rank's own error and is found on
  // its inputs alone, but the other ranks may be fine: it joins the collective with its code (run_and_gather) instead of
  // leaving them waiting.
  int early = COLATE_OK;
  if (hi > lo) {
    if (group_first != lo / B || group_count != (hi - 1) / B - lo / B + 1)
      early = fail(COLATE_EINVAL, "rank %d computes rows [%d, %d) = groups [%d, %d], but was given groups [%d, %d)", c->rank, lo, hi,
                   lo / B, (hi - 1) / B, group_first, group_first + group_count);
    else if (!group_nb || !group_age || !weights || !sh_block || !ns_block || !sh_emp_block || !ns_emp_block || !epochs || !init_rates)
      early = fail(COLATE_EINVAL, "NULL pointer argument");
    for (int g = 0; g < group_count && !early; g++) {
      if (group_nb[g] < 1) early = fail(COLATE_EINVAL, "group %d has %d genome blocks", group_first + g, group_nb[g]);
      if (!early) early = check_grids(E, A, age_grid, epochs + (size_t)g * E);
    }
  }
  const std::string early_msg = early ? colate_last_error() : "";
  if (int rc = ensure_device()) return rc;
  if (R == 0) return COLATE_OK;
  Upload up;
  int* d_status = nullptr;
  int rc = run_and_gather(
      c, R, E, out_rates, out_iters, out_loglik, out_flags,
      [&](int n, int row_lo, double* d_rates, double* d_ll, int* d_iters, int* d_flags, hipStream_t s) {
        if (early) return fail(early, "%s", early_msg.c_str());
        std::vector<long long> block_off(group_count), weight_off(group_count);
        long long nblocks = 0, nweights = 0;
        for (int g = 0; g < group_count; g++) {
          block_off[g] = nblocks, weight_off[g] = nweights;
          nblocks += group_nb[g], nweights += (long long)B * group_nb[g];
        }
        const size_t nT = (size_t)nblocks * A;
        std::vector<double> row_ep((size_t)n * E), row_init((size_t)n * E);
        for (int r = 0; r < n; r++) {
          const size_t g = (size_t)((row_lo + r) / B - group_first);
          std::memcpy(row_ep.data() + (size_t)r * E, epochs + g * E, (size_t)E * sizeof(double));
          std::memcpy(row_init.data() + (size_t)r * E, init_rates + g * E, (size_t)E * sizeof(double));
        }
        double *d_grid, *d_w, *d_t0, *d_t1, *d_t2, *d_t3, *d_ep, *d_init, *d_sh, *d_ns, *d_age;
        int* d_nb;
        long long *d_bo, *d_wo;
        const int zero = 0;
        if (int r = up.put(age_grid, A, &d_grid, s)) return r;
        if (int r = up.put(weights, (size_t)nweights, &d_w, s)) return r;
        if (int r = up.put(sh_block, nT, &d_t0, s)) return r;
        if (int r = up.put(ns_block, nT, &d_t1, s)) return r;
        if (int r = up.put(sh_emp_block, nT, &d_t2, s)) return r;
        if (int r = up.put(ns_emp_block, nT, &d_t3, s)) return r;
        if (int r = up.put(group_nb, group_count, &d_nb, s)) return r;
        if (int r = up.put(block_off.data(), group_count, &d_bo, s)) return r;
        if (int r = up.put(weight_off.data(), group_count, &d_wo, s)) return r;
        if (int r = up.put(group_age, group_count, &d_age, s)) return r;
        if (int r = up.put(row_ep.data(), (size_t)n * E, &d_ep, s)) return r;
        if (int r = up.put(row_init.data(), (size_t)n * E, &d_init, s)) return r;
        if (int r = up.put<double>(nullptr, (size_t)n * A, &d_sh, s)) return r;
        if (int r = up.put<double>(nullptr, (size_t)n * A, &d_ns, s)) return r;
        if (int r = up.put(&zero, 1, &d_status, s)) return r;
        HIP_TRY(hipStreamSynchronize(s));  // (the host vectors of this scope leave it)
        if (int r = colate_bootstrap_counts_groups_device(group_count, B, group_first, row_lo, row_lo + n, A, d_grid, d_nb, d_bo, d_wo,
                                                          d_age, d_w, d_t0, d_t1, d_t2, d_t3, d_sh, d_ns, d_status, s))
          return r;
        return colate_em_batch_device(n, E, A, d_grid, d_sh, d_ns, d_ep, 1, d_init, 1, max_iter, min_iter, rel_tol, rate_floor,
                                      d_rates, d_iters, d_ll, d_flags, s);
      });
  if (rc) return rc;
  if (d_status) {
    int status = 0;
    HIP_TRY(hipMemcpy(&status, d_status, sizeof(int), hipMemcpyDeviceToHost));
    if (status) return fail(COLATE_EINVAL, "sample age outside the age grid");
  }
  return COLATE_OK;
}

}  // extern "C"
