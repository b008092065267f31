// colate_amd/csrc/fill_kernel.hip -- the age sampling of the table fill on the GPU: see fill_device.h.
// Reference: include/coal/coal.cpp:2260-2273 (age_begin <= sample age: the F path), 2279-2295 (the 100 draws of a SNP).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "fill_device.h"

namespace colate_drv {
namespace {

constexpr int kWave = 64;
constexpr int kSlots = 4;          // a lane's bins: lane, lane + 64, lane + 128, lane + 192 (A <= 256)
constexpr int kJobsPerBlock = 4;   // one wave per job
constexpr int kDraws = 100;        // coal.cpp:2073 num_samples

__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// orders this wave's LDS accesses (the histogram) -- and only those: a fence over all address spaces waits for the global loads too
// (s_waitcnt vmcnt(0)), i.e. for the next SNP's record and uniforms, which are fetched ahead precisely so that nobody waits for them
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// One wave per job = the used SNPs of one (pair, genome block) in file order.  The loop is uniform over the wave: a SNP's record
// is the same for every lane; lanes 0..63 take its uniforms 0..63, lanes 0..35 also 64..99.
__global__ __launch_bounds__(kJobsPerBlock* kWave) void fill_sample_kernel(const FillJob* __restrict__ jobs, int njobs,
                                                                            const FillRec* __restrict__ recs, const double* __restrict__ U,
                                                                            const double* __restrict__ g_lo, const double* __restrict__ g_hi,
                                                                            int A, double* __restrict__ tables, int* __restrict__ flags) {
  __shared__ double2 s_band[kSlots * kWave + 2];  // (lower, upper) edge of the guard band around step k = 1..A ([0] = -inf, [A + 1] = +inf)
  __shared__ unsigned s_hist[kJobsPerBlock][kSlots * kWave];
  for (int i = threadIdx.x; i < A + 2; i += blockDim.x) s_band[i] = make_double2(g_lo[i], g_hi[i]);
  __syncthreads();  // (the only workgroup barrier: the waves of a block are independent jobs from here on)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = blockIdx.x * kJobsPerBlock + wave;
  if (j >= njobs) return;
  const FillJob job = jobs[j];
  unsigned* const hist = s_hist[wave];  // (accessed through wavefront-scope atomics between LDS-only fences: a `volatile` pointer made every access wait for the global loads in flight)
  double* const T = tables + (size_t)job.table * 2 * (size_t)A;
  double sh[kSlots], ns[kSlots];
#pragma unroll
  for (int s = 0; s < kSlots; s++) {
    const int b = lane + kWave * s;
    __hip_atomic_store(&hist[b], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    sh[s] = b < A ? T[b] : 0.0;
    ns[s] = b < A ? T[A + b] : 0.0;
  }
  wave_lds_fence();
  int bad = 0;
  const FillRec* r = recs + job.rec_off;
  const double* u = U + job.u_off;
  // (a SNP's record and uniforms are fetched while the SNP before it is worked on: alone on its SIMD -- a hand-over has a few
  // hundred jobs for 1024 SIMDs -- a wave waited ~3 us per SNP for the two loads, five times what the arithmetic takes)
  const bool v1 = lane < kDraws - kWave;
  float n_fb = 0.f, n_fe = 0.f;
  double n_wsh = 0.0, n_wns = 0.0, n_u0 = 0.0, n_u1 = 0.0;
  if (job.nrec > 0) {
    n_fb = r[0].begin, n_fe = r[0].end, n_wsh = r[0].w_sh, n_wns = r[0].w_ns;
    n_u0 = u[lane];
    n_u1 = v1 ? u[kWave + lane] : 0.0;
  }
  for (uint32_t i = 0; i < job.nrec; i++, u += kDraws) {
    const float fb = n_fb, fe = n_fe;
    const double w_sh = n_wsh, w_ns = n_wns, u0 = n_u0, u1 = n_u1;
    if (i + 1 < job.nrec) {
      n_fb = r[i + 1].begin, n_fe = r[i + 1].end, n_wsh = r[i + 1].w_sh, n_wns = r[i + 1].w_ns;
      n_u0 = u[kDraws + lane];
      n_u1 = v1 ? u[kDraws + kWave + lane] : 0.0;
    }
    const double begin = (double)fb, end = (double)fe;
    const double span = end - begin;
    const bool emp = !(begin > 0.0);  // age_begin <= age, coal.cpp:2245
    // the samples (coal.cpp:2262, 2281: a separate multiply and add)
    const double x0 = __dadd_rn(__dmul_rn(u0, span), begin);
    const double x1 = v1 ? __dadd_rn(__dmul_rn(u1, span), begin) : begin;
    // bin(x) = #{k : upper edge of band k <= x} for every x outside all bands.  A candidate from a single-precision logarithm
    // (bin k is centred on exp((k - 1) / 10) / 10), settled by the four bands around it: inside a band, or further off than one
    // step -- the host decides.  (Counting the steps between the bins of `begin` and of the largest possible sample, one LDS
    // round trip each, was a third of the kernel's time; for the F path, whose ranges start at 0, seventy steps.)
    auto settle = [&](double x, bool& unsure) -> int {
      int c = (int)rintf(6.9314718f * __log2f((float)x * 10.0f)) + 1;  // (x = 0: -inf -> clamped)
      c = c < 0 ? 0 : (c > A ? A : c);
      const double2 bm = s_band[c > 0 ? c - 1 : 0], b0_ = s_band[c], bp = s_band[c + 1], bq = s_band[c + 2 <= A + 1 ? c + 2 : A + 1];
      int b;
      bool ok;
      if (x >= bp.y) {            // at or beyond the upper edge of step c + 1
        b = c + 1, ok = x < bq.x;
      } else if (x >= b0_.y) {    // ... of step c (c = 0: -inf)
        b = c, ok = x < bp.x;
      } else {                    // below step c
        b = c - 1, ok = x < b0_.x && x >= bm.y;
      }
      unsure = !ok;
      return b < 0 ? 0 : b;
    };
    bool un0, un1;
    const int b0 = settle(x0, un0), b1 = settle(x1, un1);
    // inside a band, or -- not the F path -- beyond the grid, where the reference draws again and the stream
    // no longer lines up: the host decides (the whole pair is filled again there)
    bool trouble = un0 || (v1 && un1);
    if (!emp) trouble = trouble || b0 >= A || (v1 && b1 >= A) || x0 < 0.0 || (v1 && x1 < 0.0);
    if (ballot64(trouble) != 0ull) bad = 1;
    // how many samples fell into each bin (integers: the order does not matter) ...
    if (b0 < A) __hip_atomic_fetch_add(&hist[b0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    if (v1 && b1 < A) __hip_atomic_fetch_add(&hist[b1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    wave_lds_fence();
    unsigned cnt[kSlots];
#pragma unroll
    for (int s = 0; s < kSlots; s++) {
      cnt[s] = __hip_atomic_load(&hist[lane + kWave * s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      __hip_atomic_store(&hist[lane + kWave * s], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    wave_lds_fence();
    // ... and that many additions of the SNP's weight to the bin, one after the other (what the sample-by-sample loop of the
    // reference does to it: additions to different bins commute, those to one bin are all of the same addend); the F path adds to
    // the not-shared table only -- w_sh = +0.0 leaves the shared sum as it is
#pragma unroll
    for (int s = 0; s < kSlots; s++) {
      unsigned c = cnt[s];
      double a = sh[s], n = ns[s];
      while (ballot64(c > 0u) != 0ull) {
        if (c > 0u) {
          a = __dadd_rn(a, w_sh);
          n = __dadd_rn(n, w_ns);
          c--;
        }
      }
      sh[s] = a, ns[s] = n;
    }
  }
#pragma unroll
  for (int s = 0; s < kSlots; s++) {
    const int b = lane + kWave * s;
    if (b < A) {
      T[b] = sh[s];
      T[A + b] = ns[s];
    }
  }
  if (bad && lane == 0) flags[job.table] = 1;
}

}  // namespace

bool DeviceFill::fail(const char* what, int code) {
  char buf[256];
  std::snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString((hipError_t)code));
  err_ = buf;
  return false;
}

#define FILL_TRY(call)                                     \
  do {                                                     \
    const hipError_t e_ = (call);                          \
    if (e_ != hipSuccess) return fail(#call, (int)e_);     \
  } while (0)

bool DeviceFill::available() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
    (void)hipGetLastError();
    return false;
  }
  return true;
}

DeviceFill* DeviceFill::create(int device, int A, const double* guard_lo, const double* guard_hi, size_t max_tables, size_t batch_recs,
                               std::string& why) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
    (void)hipGetLastError();
    why = "no HIP device";
    return nullptr;
  }
  if (A < 1 || A > kSlots * kWave) {
    why = "more than 256 age bins";
    return nullptr;
  }
  DeviceFill* d = new DeviceFill;
  d->device_ = device % n, d->A_ = A, d->max_uniforms_ = 0, d->max_tables_ = max_tables, d->batch_recs_ = batch_recs;
  d->max_jobs_ = 1u << 16;
  auto init = [&]() -> bool {
    DeviceFill& o = *d;
    if (hipSetDevice(o.device_) != hipSuccess) return o.fail("hipSetDevice", (int)hipGetLastError());
    const size_t tb = max_tables * 2 * (size_t)A * sizeof(double);
    if (hipMalloc(&o.d_lo_, (A + 2) * sizeof(double)) != hipSuccess || hipMalloc(&o.d_hi_, (A + 2) * sizeof(double)) != hipSuccess ||
        hipMalloc(&o.d_tables_, tb) != hipSuccess || hipMalloc(&o.d_flags_, max_tables * sizeof(int)) != hipSuccess)
      return o.fail("hipMalloc (tables)", (int)hipGetLastError());
    if (hipMemcpy(o.d_lo_, guard_lo, (A + 2) * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(o.d_hi_, guard_hi, (A + 2) * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(o.d_tables_, 0, tb) != hipSuccess || hipMemset(o.d_flags_, 0, max_tables * sizeof(int)) != hipSuccess)
      return o.fail("hipMemcpy (thresholds)", (int)hipGetLastError());
    for (int b = 0; b < 2; b++) {
      if (hipHostMalloc(reinterpret_cast<void**>(&o.h_jobs_[b]), o.max_jobs_ * sizeof(FillJob), hipHostMallocDefault) != hipSuccess)
        return o.fail("hipHostMalloc (jobs)", (int)hipGetLastError());
      hipEvent_t e;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return o.fail("hipEventCreate", (int)hipGetLastError());
      o.ev_h2d_[b] = e;
    }
    for (int b = 0; b < kDev; b++) {
      if (hipMalloc(&o.d_jobs_[b], o.max_jobs_ * sizeof(FillJob)) != hipSuccess) return o.fail("hipMalloc (jobs)", (int)hipGetLastError());
      hipStream_t s;
      if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return o.fail("hipStreamCreate", (int)hipGetLastError());
      o.stream_[b] = s;
      for (int k = 0; k < 2; k++) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return o.fail("hipEventCreate", (int)hipGetLastError());
        o.ev_[b][k] = e;
      }
    }
    hipStream_t cs;
    if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) return o.fail("hipStreamCreate", (int)hipGetLastError());
    o.copy_stream_ = cs;
    hipEvent_t ue;
    if (hipEventCreateWithFlags(&ue, hipEventDisableTiming) != hipSuccess) return o.fail("hipEventCreate", (int)hipGetLastError());
    o.upload_ev_ = ue;
    return true;
  };
  if (!init()) {
    why = d->err_;
    delete d;
    return nullptr;
  }
  return d;
}

DeviceFill::~DeviceFill() {
  (void)hipSetDevice(device_);
  for (void* p : pinned_) (void)hipHostUnregister(p);
  for (int b = 0; b < kDev; b++)
    if (stream_[b]) (void)hipStreamSynchronize((hipStream_t)stream_[b]);
  for (int b = 0; b < 2; b++) {
    if (stage_[b]) (void)hipHostFree(stage_[b]);
    if (h_jobs_[b]) (void)hipHostFree(h_jobs_[b]);
    if (ev_h2d_[b]) (void)hipEventDestroy((hipEvent_t)ev_h2d_[b]);
  }
  for (int b = 0; b < kDev; b++) {
    if (d_recs_[b]) (void)hipFree(d_recs_[b]);
    if (d_jobs_[b]) (void)hipFree(d_jobs_[b]);
    for (int k = 0; k < 2; k++)
      if (ev_[b][k]) (void)hipEventDestroy((hipEvent_t)ev_[b][k]);
    if (stream_[b]) (void)hipStreamDestroy((hipStream_t)stream_[b]);
  }
  if (copy_stream_) (void)hipStreamSynchronize((hipStream_t)copy_stream_);
  if (upload_ev_) (void)hipEventDestroy((hipEvent_t)upload_ev_);
  if (copy_stream_) (void)hipStreamDestroy((hipStream_t)copy_stream_);
  for (void* p : {(void*)d_u_, (void*)d_lo_, (void*)d_hi_, (void*)d_tables_, (void*)d_flags_})
    if (p) (void)hipFree(p);
}

bool DeviceFill::alloc_staging() {
  FILL_TRY(hipSetDevice(device_));
  for (int b = 0; b < kDev; b++) FILL_TRY(hipMalloc(&d_recs_[b], batch_recs_ * sizeof(FillRec)));
  for (int b = 0; b < 2; b++) FILL_TRY(hipHostMalloc(reinterpret_cast<void**>(&stage_[b]), batch_recs_ * sizeof(FillRec), hipHostMallocDefault));
  return true;
}

bool DeviceFill::alloc_uniforms(uint64_t max_uniforms) {
  FILL_TRY(hipSetDevice(device_));
  FILL_TRY(hipMalloc(&d_u_, (max_uniforms + kDraws) * sizeof(double)));
  max_uniforms_ = max_uniforms;
  return true;
}

void DeviceFill::pin(void* p, size_t bytes) {
  if (hipSetDevice(device_) != hipSuccess) return;
  if (hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess) pinned_.push_back(p);
  else (void)hipGetLastError();
}

bool DeviceFill::upload_uniforms(uint64_t off, const double* src, size_t n) {
  if (off + n > max_uniforms_ + kDraws) {
    err_ = "uniform stream longer than the device buffer";
    return false;
  }
  FILL_TRY(hipSetDevice(device_));
  FILL_TRY(hipMemcpyAsync(d_u_ + off, src, n * sizeof(double), hipMemcpyHostToDevice, (hipStream_t)copy_stream_));
  uploads_pending_ = true;
  return true;
}

bool DeviceFill::sync_uploads() {
  if (!uploads_pending_) return true;
  FILL_TRY(hipSetDevice(device_));
  FILL_TRY(hipStreamSynchronize((hipStream_t)copy_stream_));
  uploads_pending_ = false;
  return true;
}

bool DeviceFill::submit(const std::vector<FillJob>& jobs, size_t nrecs) {
  if (jobs.empty()) return true;
  if (jobs.size() > max_jobs_ || nrecs > batch_recs_) {
    err_ = "batch larger than its buffers";
    return false;
  }
  FILL_TRY(hipSetDevice(device_));
  const int h = cur_, d = dcur_;
  hipStream_t s = (hipStream_t)stream_[d];
  if (launched_[d]) {  // the device buffer's previous kernel (kDev launches ago)
    FILL_TRY(hipEventSynchronize((hipEvent_t)ev_[d][1]));
    float ms = 0;
    if (hipEventElapsedTime(&ms, (hipEvent_t)ev_[d][0], (hipEvent_t)ev_[d][1]) == hipSuccess) gpu_s_ += ms * 1e-3;
    launched_[d] = false;
  }
  if (uploads_pending_) {  // the launch reads uniforms whose copies may still be in flight on the copy stream
    FILL_TRY(hipEventRecord((hipEvent_t)upload_ev_, (hipStream_t)copy_stream_));
    FILL_TRY(hipStreamWaitEvent(s, (hipEvent_t)upload_ev_, 0));
  }
  std::memcpy(h_jobs_[h], jobs.data(), jobs.size() * sizeof(FillJob));
  FILL_TRY(hipMemcpyAsync(d_recs_[d], stage_[h], nrecs * sizeof(FillRec), hipMemcpyHostToDevice, s));
  FILL_TRY(hipMemcpyAsync(d_jobs_[d], h_jobs_[h], jobs.size() * sizeof(FillJob), hipMemcpyHostToDevice, s));
  FILL_TRY(hipEventRecord((hipEvent_t)ev_h2d_[h], s));
  copied_[h] = true;
  FILL_TRY(hipEventRecord((hipEvent_t)ev_[d][0], s));
  const int nj = (int)jobs.size();
  hipLaunchKernelGGL(fill_sample_kernel, dim3((nj + kJobsPerBlock - 1) / kJobsPerBlock), dim3(kJobsPerBlock * kWave), 0, s, d_jobs_[d], nj, d_recs_[d],
                     d_u_, d_lo_, d_hi_, A_, d_tables_, d_flags_);
  FILL_TRY(hipGetLastError());
  FILL_TRY(hipEventRecord((hipEvent_t)ev_[d][1], s));
  launched_[d] = true;
  // one (pair, block) table is touched by one job only: launches need no order among themselves.  The other staging buffer is what
  // the host fills next: the copy out of it (two submits ago) must have completed
  dcur_ = (dcur_ + 1) % kDev;
  cur_ ^= 1;
  if (copied_[cur_]) {
    FILL_TRY(hipEventSynchronize((hipEvent_t)ev_h2d_[cur_]));
    copied_[cur_] = false;
  }
  return true;
}

bool DeviceFill::finish(std::vector<double>& tables, std::vector<int>& flags) {
  FILL_TRY(hipSetDevice(device_));
  if (!sync_uploads()) return false;
  for (int b = 0; b < kDev; b++) {
    FILL_TRY(hipStreamSynchronize((hipStream_t)stream_[b]));
    if (launched_[b]) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, (hipEvent_t)ev_[b][0], (hipEvent_t)ev_[b][1]) == hipSuccess) gpu_s_ += ms * 1e-3;
      launched_[b] = false;
    }
  }
  tables.resize(max_tables_ * 2 * (size_t)A_);
  flags.resize(max_tables_);
  FILL_TRY(hipMemcpy(tables.data(), d_tables_, tables.size() * sizeof(double), hipMemcpyDeviceToHost));
  FILL_TRY(hipMemcpy(flags.data(), d_flags_, flags.size() * sizeof(int), hipMemcpyDeviceToHost));
  return true;
}

}  // namespace colate_drv
