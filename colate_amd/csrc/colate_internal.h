// colate_amd/csrc/colate_internal.h -- shared between the translation units of libcolate_amd.so
#pragma once
namespace colate {
// records the message for colate_last_error() and returns `code`
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// COLATE_OK, or COLATE_ENODEVICE with a message (there is no CPU fallback)
int ensure_device();
// the grids the kernel's contiguous-segment logic relies on: age_grid non-negative and non-decreasing, epochs
// non-decreasing, epochs[0] <= age_grid[0] (every host-pointer entry point runs this before anything is launched)
int check_grids(int E, int A, const double* age_grid, const double* epochs);
// Set (process-wide, never cleared) by every entry point that makes this process talk to the HIP runtime.  A process
// that has done so must not fork() children that use the GPU: `Colate --ranks N` (run_ranked, mut_driver.cpp) refuses
// when it is set (colate_device_touched, include/colate_amd.h).
void mark_device_touched();
// A named range for profilers (rocprofv3 --marker-trace): roctxRangePush / roctxRangePop, bound lazily with dlopen so that the
// library has no link-time dependency on a profiler; a no-op where librocprofiler-sdk-roctx / libroctx64 is not present.
struct ProfRange {
  explicit ProfRange(const char* name);
  ~ProfRange();
  ProfRange(const ProfRange&) = delete;
  ProfRange& operator=(const ProfRange&) = delete;
};
}  // namespace colate
