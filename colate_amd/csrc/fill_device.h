// colate_amd/csrc/fill_device.h -- the age sampling of the table fill on the GPU (internal to libcolate_amd.so).
//
// Reference: include/coal/coal.cpp:2260-2273, 2279-2295 -- for every SNP a pair uses, 100 ages are drawn uniformly between the
// mutation's lower and upper age and the SNP's weight is added to the age bin of each of them, in the (pair, genome block)'s tables
// of shared / not-shared counts.  csrc/mut_pairs.cpp does that on the host (Engine::sample); this is the same arithmetic on the
// device, bit for bit:
//   * the uniforms of a --seed are ONE stream, the same for every pair: uploaded once, addressed by offset;
//   * a sample's age is u * span + begin by a separate multiply and add (no contraction), its bin the number of grid steps at or below
//     it, counted against both edges of the host's guard bands around the located steps (FastBin): where the two counts differ the
//     sample lies within 64 ulps of a step, the library expression would have to decide, and the pair is handed back to the host;
//   * the additions to one bin are a chain -- the same addend, one addition after the other, SNP after SNP --, and the chains of
//     different bins and different (pair, block) tables are independent: a wave per (pair, block), its bins in its lanes (the tables
//     live in registers), the counts of a SNP's samples per bin from an LDS histogram (integers: no order), then as many additions
//     as the fullest bin got, every lane on its own bins.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace colate_drv {

struct FillRec {  // one used SNP (coal.cpp:2245-2297): the sampled range and what a sample adds
  float begin, end;    // age_begin (already clamped to the sample age 0), age_end; begin <= 0: the F path (not-shared weight only, no redraws)
  double w_sh, w_ns;   // f_DAF_target * DAF_ref / (N_ref * 100) and the same with f_AAF_target, as the host computes them
};
static_assert(sizeof(FillRec) == 24, "FillRec");

struct FillJob {        // the SNPs of one (pair, genome block), in file order
  uint64_t rec_off;     // first record, in the batch's staging buffer
  uint64_t u_off;       // stream offset of the first SNP's first uniform (SNP i: u_off + 100 i)
  uint32_t nrec;
  uint32_t table;       // which [2][A] table (shared | not shared) on the device
};

class DeviceFill {
 public:
  static bool available();  // a HIP device is there
  // null (and a reason in `why`) when there is no device to use
  static DeviceFill* create(int device, int A, const double* guard_lo, const double* guard_hi, size_t max_tables, size_t batch_recs, std::string& why);
  // the two page-locked record buffers and their device copies (the slow part of the set-up -- gigabytes to lock --: callable from
  // another thread while uniforms are uploaded; before the first staging() / submit())
  bool alloc_staging();
  // room for the uniform stream (known once the inputs have been read: a pair takes at most 100 per row)
  bool alloc_uniforms(uint64_t max_uniforms);
  ~DeviceFill();
  FillRec* staging() { return stage_[cur_]; }  // pinned: where the next batch's records go
  size_t staging_capacity() const { return batch_recs_; }
  // page-locks a host buffer the uniforms will be uploaded from (the ring of the stream's producer): pageable memory goes through
  // a staging copy at a few GB/s, 6 GB of uniforms are then seconds of wall time.  Failure is not an error (the upload works either way).
  void pin(void* p, size_t bytes);
  // uniforms [off, off + n) of the stream: the copy is enqueued (launches submitted later wait for it on the device); the source may
  // be reused after sync_uploads()
  bool upload_uniforms(uint64_t off, const double* src, size_t n);
  bool sync_uploads();
  // records staging()[0 .. nrecs) and the jobs that refer to them: copied and launched asynchronously; staging() is another buffer afterwards
  bool submit(const std::vector<FillJob>& jobs, size_t nrecs);
  // waits for everything; tables [max_tables][2][A]; flags per table: 1 = a sample needs the host (guard band, redraw, range): redo the pair
  bool finish(std::vector<double>& tables, std::vector<int>& flags);
  const std::string& error() const { return err_; }
  double gpu_seconds() const { return gpu_s_; }

 private:
  DeviceFill() = default;
  int device_ = 0, A_ = 0, cur_ = 0;
  uint64_t max_uniforms_ = 0;
  size_t max_tables_ = 0, batch_recs_ = 0, max_jobs_ = 0;
  double *d_u_ = nullptr, *d_lo_ = nullptr, *d_hi_ = nullptr, *d_tables_ = nullptr;
  int* d_flags_ = nullptr;
  // Two page-locked staging buffers on the host (one is filled while the other is copied), kDev record buffers on the device: a
  // staging buffer is free again when its copy has completed, a device buffer when its kernel has -- so up to kDev launches run at
  // the same time (a launch of ~1200 jobs is about one wave per SIMD, each waiting most of the time: they interleave).
  static constexpr int kDev = 4;
  FillRec* stage_[2] = {nullptr, nullptr};
  FillJob* h_jobs_[2] = {nullptr, nullptr};
  void* ev_h2d_[2] = {nullptr, nullptr};
  bool copied_[2] = {false, false};
  FillRec* d_recs_[kDev] = {nullptr, nullptr, nullptr, nullptr};
  FillJob* d_jobs_[kDev] = {nullptr, nullptr, nullptr, nullptr};
  void* stream_[kDev] = {nullptr, nullptr, nullptr, nullptr};
  void* ev_[kDev][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};  // per device buffer: start / end of its latest kernel
  bool launched_[kDev] = {false, false, false, false};
  int dcur_ = 0;
  void* copy_stream_ = nullptr;
  void* upload_ev_ = nullptr;
  bool uploads_pending_ = false;
  double gpu_s_ = 0;
  std::vector<void*> pinned_;
  std::string err_;
  bool fail(const char* what, int code);
};

}  // namespace colate_drv
