// colate_amd/csrc/em_kernels.hip -- instantiation and dispatch of the EM kernel (template in em_kernel_impl.hpp).
//
//   shape                                   instantiation                     build
//   one E-step (colate_em_estep)            <1, NCH, 4, false>                this unit
//   B <= 2 x #CUs, E <= 128                 <0, NCH, EROWS, false> latency    em_kernels_ilp.hip (max-ilp scheduling; fits 3 waves/SIMD)
//   only by COLATE_EM_VARIANT=latency       <0, NCH, EROWS, false> latency    this unit (default scheduler; A/B runs)
//   B > 2 x #CUs, or E > 128                <0, NCH, EROWS, true> throughput  this unit
//   256 < E <= 1024                         <*, 8 | 16, 4, *> two-wave layout   em_kernels_big.hip (general loop only)
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "em_kernel_impl.hpp"

hipError_t colate_em_launch_latency_ilp(const ColateEmArgs& args, hipStream_t stream, bool alone);  // em_kernels_ilp.hip
hipError_t colate_em_launch_big(const ColateEmArgs& args, hipStream_t stream);                          // em_kernels_big.hip: 257 .. 1024 epochs

size_t colate_em_lds_bytes(int E, int A) { return em_lds_bytes(E, A, false); }  // (the larger of the two variants' needs)

// number of CUs of the current device (cached per ordinal; atomics: launches may come from several host threads)
static int device_cus() {
  static std::atomic<int> cus[64];  // 0 = not asked yet, -1 = the query failed
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  int n = cus[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = -1;
    cus[dev].store(n, std::memory_order_relaxed);  // (two threads asking at once store the same value)
  }
  return n > 0 ? n : 0;
}

// Which build a launch takes for E <= 128 can be forced (tests, A/B runs): at run time by colate_em_force_variant()
// (include/colate_amd.h), initially by COLATE_EM_VARIANT=latency|latency-ilp|throughput, which is read ONCE per process
// (not per launch).  -1 = automatic.
static std::atomic<int>& variant_forced() {
  static std::atomic<int> forced{[] {
    const char* v = getenv("COLATE_EM_VARIANT");
    if (!v) return -1;
    if (!strcmp(v, "throughput")) return 2;
    if (!strcmp(v, "latency")) return 1;
    if (!strcmp(v, "latency-ilp")) return 0;
    return -1;
  }()};
  return forced;
}
void colate_em_set_forced_variant(int v) { variant_forced().store((v >= 0 && v <= 2) ? v : -1, std::memory_order_relaxed); }
static int variant_override() { return variant_forced().load(std::memory_order_relaxed); }

// 0 = latency (max-ilp build), 1 = latency (default build), 2 = throughput.
int colate_em_variant(int B, int E) {
  if (em_chunks(E) > 2) return 2;  // 129..256 epochs: only the two-wave layout fits the register file without scratch
  if (variant_override() >= 0) return variant_override();
  // Measured (tools/variant_sweep.sh, profiles/r02/variants.txt): since the steady-state loops the max-ilp build fits
  // 3 waves per SIMD as well (157 VGPRs) and is the faster latency build at every batch size (1.29 against 1.33 ms at
  // B = 400); two 6-wave workgroups per CU beat the two-wave layout up to 2 x #CUs (B = 512: 1.30 against 2.09 ms),
  // beyond that the throughput variant wins (B = 640: 2.09 against 2.12 ms; B = 1536: 2.70 against 3.65 ms).  The
  // default-scheduler latency build stays selectable (COLATE_EM_VARIANT=latency) for A/B runs.
  const int cus = device_cus();
  if (cus <= 0 || B <= 2 * cus) return 0;
  return 2;
}

hipError_t colate_em_launch(const ColateEmArgs& args, hipStream_t stream) {
  const int nch = em_chunks(args.E);
  if (nch > 4) return colate_em_launch_big(args, stream);
  if (args.mode == 1) {
    const int threads = em_threads(args.A);
    const size_t lds = em_lds_bytes(args.E, args.A, false);
    if (nch == 1) return launch_one<1, 1, 4, false>(args, stream, lds, threads);
    if (nch == 2) return launch_one<1, 2, 4, false>(args, stream, lds, threads);
    return launch_one<1, 4, 4, false>(args, stream, lds, threads);
  }
  switch (colate_em_variant(args.B, args.E)) {
    case 0: {
      const int cus = device_cus();
      return colate_em_launch_latency_ilp(args, stream, cus > 0 && args.B <= cus);
    }
    case 1: return launch_latency(args, stream);
  }
  const size_t lds = em_lds_bytes(args.E, args.A, true);
  if (nch == 4) return launch_one<0, 4, 4, true>(args, stream, lds, 2 * kWave);
  if (nch == 2) return launch_one<0, 2, 4, true>(args, stream, lds, 2 * kWave);
  switch (em_rows(args.E)) {
    case 1: return launch_one<0, 1, 1, true>(args, stream, lds, 2 * kWave);
    case 2: return launch_one<0, 1, 2, true>(args, stream, lds, 2 * kWave);
    default: return launch_one<0, 1, 4, true>(args, stream, lds, 2 * kWave);
  }
}
