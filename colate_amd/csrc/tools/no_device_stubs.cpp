// colate_amd/csrc/tools/no_device_stubs.cpp -- ONLY for the host sanitizer binary (`make asan` -> bin/Colate_asan):
// the host side of the library (mut_host.cpp, mut_driver.cpp) built with g++ -fsanitize=address,undefined, without
// HIP.  Every compute entry point that needs a device is defined here to FAIL with COLATE_ENODEVICE -- the sanitizer
// runs cover the readers, the table fill, the bootstrap, make_tmp and the writers (--counts_only); nothing here
// computes anything, and libcolate_amd.so does not contain this file.
#include <cstdarg>
#include <cstdio>
#include <string>

#include "colate_amd.h"
#include "colate_internal.h"

namespace colate {
static thread_local std::string g_err;
int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
ProfRange::ProfRange(const char*) {}
ProfRange::~ProfRange() {}
}  // namespace colate

static int nodev() { return colate::fail(COLATE_ENODEVICE, "host sanitizer build: no device code linked"); }

extern "C" {
const char* colate_version(void) { return "colate_amd host sanitizer build (no device code)"; }
const char* colate_last_error(void) { return colate::g_err.c_str(); }
int colate_device_touched(void) { return 0; }
int colate_device_count(void) { return nodev(); }
int colate_set_device(int) { return nodev(); }
int colate_warm_up(int) { return nodev(); }
int colate_em_force_variant(int) { return nodev(); }
int colate_em_batch(int, int, int, const double*, const double*, const double*, const double*, const double*, int, int,
                    double, double, double*, int*, double*, int*) { return nodev(); }
int colate_em_batch_rows(int, int, int, const double*, const double*, const double*, const double*, const double*, int, int,
                         double, double, double*, int*, double*, int*) { return nodev(); }
int colate_em_batch_sharded(int, const int*, int, int, int, const double*, const double*, const double*, const double*,
                            const double*, int, int, double, double, double*, int*, double*, int*) { return nodev(); }
int colate_em_batch_rows_sharded(int, const int*, int, int, int, const double*, const double*, const double*, const double*,
                                 const double*, int, int, double, double, double*, int*, double*, int*) { return nodev(); }
int colate_bootstrap_em_batch(int, int, int, int, const double*, double, const double*, const double*, const double*,
                              const double*, const double*, const double*, const double*, int, int, double, double, double*,
                              int*, double*, int*, double*, double*) { return nodev(); }
int colate_shard_bounds(int B, int nranks, int rank, int* lo, int* hi) {
  const int base = B / nranks, rem = B % nranks;
  *lo = rank * base + (rank < rem ? rank : rem);
  *hi = *lo + base + (rank < rem ? 1 : 0);
  return COLATE_OK;
}
int colate_bootstrap_em_batch_groups(int, int, int, int, const double*, const int*, const double*, const double*, const double*,
                                     const double*, const double*, const double*, const double*, const double*, int, int, double,
                                     double, double*, int*, double*, int*, double*, double*) { return nodev(); }
int colate_bootstrap_em_batch_groups_allgather(void*, int, int, int, int, int, int, const double*, const int*, const double*,
                                               const double*, const double*, const double*, const double*, const double*,
                                               const double*, const double*, int, int, double, double, double*, int*, double*,
                                               int*) { return nodev(); }
int colate_comm_unique_id(void*) { return nodev(); }
int colate_comm_create(const void*, int, int, void**) { return nodev(); }
int colate_comm_destroy(void*) { return COLATE_OK; }
int colate_em_batch_allgather(void*, int, int, int, const double*, const double*, const double*, const double*, const double*,
                              int, int, double, double, double*, int*, double*, int*) { return nodev(); }
int colate_bootstrap_em_batch_allgather(void*, int, int, int, int, const double*, double, const double*, const double*,
                                        const double*, const double*, const double*, const double*, const double*, int, int,
                                        double, double, double*, int*, double*, int*) { return nodev(); }
}

// the age sampling on the device (fill_device.h): there is none in this build, the host code samples
#include "fill_device.h"
namespace colate_drv {
DeviceFill* DeviceFill::create(int, int, const double*, const double*, size_t, size_t, std::string& why) {
  why = "built without a device";
  return nullptr;
}
bool DeviceFill::available() { return false; }
DeviceFill::~DeviceFill() {}
bool DeviceFill::alloc_staging() { return false; }
bool DeviceFill::alloc_uniforms(uint64_t) { return false; }
void DeviceFill::pin(void*, size_t) {}
bool DeviceFill::upload_uniforms(uint64_t, const double*, size_t) { return false; }
bool DeviceFill::sync_uploads() { return false; }
bool DeviceFill::submit(const std::vector<FillJob>&, size_t) { return false; }
bool DeviceFill::finish(std::vector<double>&, std::vector<int>&) { return false; }
bool DeviceFill::fail(const char*, int) { return false; }
}  // namespace colate_drv
