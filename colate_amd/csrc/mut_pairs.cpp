// colate_amd/csrc/mut_pairs.cpp -- `Colate --mode mut --pairs FILE`: the batched all-pairs front end
// (BASELINE configs[4]: 10 target x 10 reference .colate.in, 20 replicates each).
//
// The reference has no such mode: it is run once per (target, reference) pair and every run re-reads the .mut files,
// walks both .colate.in streams (include/coal/coal.cpp:2071-2321) and draws 100 ages per used SNP from the run's own
// std::mt19937 (coal.cpp:2260-2295) before mut() (coal.cpp:3071-3863) bootstraps and fits.  Every pair here is processed
// exactly as its own `--mode mut` run with the same --seed would be -- same tables bit for bit, same bootstrap weights, same
// .coal -- but the work that does not depend on the pair is done once:
//   * every .mut file is inflated and tokenised ONCE (all chromosomes in parallel), reduced to the rows that pass the
//     row-level filters of coal.cpp:2150-2176 (16 bytes each);
//   * every .colate.in file is read and decoded ONCE, whatever number of pairs it takes part in;
//   * the uniform stream of the seed is the same for every pair (each run seeds its generator alike; only HOW MANY draws a
//     pair takes differs): it is generated ONCE, by one producer thread, into a ring of chunks that all pairs read;
//   * the pairs advance through that stream window by window, so the ring stays bounded (COLATE_UNIFORM_WINDOW_MB) however
//     long the stream a pair needs; inside a window every pair's SNP walk is a task and every (pair, genome block) a
//     sampling job on one pool of COLATE_THREADS workers: the pairs fill in parallel, and so do the blocks of one pair;
//   * where there is a GPU the sampling jobs run THERE (fill_device.h: the uniform stream uploaded once, a wave per (pair, block),
//     same arithmetic, same tables bit for bit; COLATE_DEVICE_FILL=0 keeps them on the host): the workers then only walk;
//   * the block bootstrap of all pairs runs on the GPU in one launch (bootstrap_groups_kernel) in front of ONE EM launch per
//     distinct number of epochs (per-row epochs: an ancient sample inserts its age as an epoch, coal.cpp:3597-3624);
//   * `--ranks N`: the rows (pair, replicate) are sharded over N processes, one per GPU; a rank fills only the pairs its
//     rows belong to, and one RCCL all-gather per launch returns every rank all results.
#include <fcntl.h>
#include <immintrin.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <iostream>
#include <limits>
#include <memory>
#include <mutex>
#include <new>
#include <thread>

#include "colate_amd.h"
#include "colate_internal.h"
#include "fill_device.h"
#include "mut_feeder.h"

namespace colate_drv {
namespace {

double now_s() { return StageTimes::now(); }

// thread-seconds per kind of work (COLATE_TIMING=1 prints them)
struct WorkSeconds {
  std::atomic<double> parse_mut{0}, load_tmp{0}, index{0}, walk{0}, sample{0};
  static void add(std::atomic<double>& a, double dt) {
    double v = a.load();
    while (!a.compare_exchange_weak(v, v + dt)) {}
  }
};
WorkSeconds g_work;

// Workers of the pool: COLATE_THREADS, else the hardware threads -- but no more than the CPU quota of the control group where
// there is one (a container with cpu.max = 16 CPUs on a 256-thread host: 256 workers fight over 16 CPUs' worth of time slices,
// 21-26 s for BASELINE configs[4] against 17 s with 32, profiles/r04/bench/pairs100.txt; with the vectorised sampling the run is
// CPU-bound at the quota and 16 workers, 11.6 s, beat 32, 12.8 s: pairs100_final.txt).
int pairs_threads() {
  int n = (int)std::thread::hardware_concurrency();
  if (const char* e = std::getenv("COLATE_THREADS")) return std::max(1, std::min(std::atoi(e), 1024));
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    long long quota = 0, period = 0;
    if (std::fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)  // ("max 100000": no quota, fscanf fails)
      n = (int)std::min<long long>(n, std::max<long long>(2, (quota + period - 1) / period));
    std::fclose(f);
  }
  return std::max(1, std::min(n, 256));
}

// ------------------------------------------------------------------ big arrays on transparent huge pages
// The decoded inputs and the ring of uniforms are hundreds of megabytes that are written once, front to back: with 4 KB
// pages that is a page fault per 4 KB (15 us each inside a VM: more than the decoding itself).  Allocations of 2 MB and
// more are mapped directly, 2 MB-aligned, with MADV_HUGEPAGE (a no-op where the kernel has THP switched off).
template <typename T>
struct HugeAlloc {
  using value_type = T;
  HugeAlloc() = default;
  template <typename U>
  HugeAlloc(const HugeAlloc<U>&) {}
  static constexpr size_t kHuge = size_t(2) << 20;
  static size_t mapped_bytes(size_t n) { return (n * sizeof(T) + 2 * kHuge - 1) & ~(kHuge - 1); }  // room to align + the header
  T* allocate(size_t n) {
    if (n * sizeof(T) < kHuge) return static_cast<T*>(::operator new(n * sizeof(T)));
    const size_t len = mapped_bytes(n) + kHuge;
    char* raw = static_cast<char*>(::mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
    if (raw == MAP_FAILED) throw std::bad_alloc();
    char* aligned = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(raw) + kHuge - 1) & ~(uintptr_t)(kHuge - 1));
    if (aligned > raw) ::munmap(raw, (size_t)(aligned - raw));
    const size_t keep = mapped_bytes(n);
    ::munmap(aligned + keep, len - (size_t)(aligned - raw) - keep);
    ::madvise(aligned, keep, MADV_HUGEPAGE);
    return reinterpret_cast<T*>(aligned);
  }
  void deallocate(T* p, size_t n) {
    if (n * sizeof(T) < kHuge) ::operator delete(p);
    else ::munmap(p, mapped_bytes(n));
  }
  template <typename U>
  bool operator==(const HugeAlloc<U>&) const { return true; }
  template <typename U>
  bool operator!=(const HugeAlloc<U>&) const { return false; }
};
template <typename T>
using HugeVector = std::vector<T, HugeAlloc<T>>;

// ------------------------------------------------------------------ a pool of workers over one FIFO of tasks
class Pool {
 public:
  explicit Pool(int nthreads) {
    for (int i = 0; i < nthreads; i++) workers_.emplace_back([this] { run(); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_work_.notify_all();
    for (std::thread& t : workers_) t.join();
  }
  int size() const { return (int)workers_.size(); }
  void submit(std::function<void()> f) {
    {
      std::lock_guard<std::mutex> lk(m_);
      q_.push_back(std::move(f));
      open_++;
    }
    cv_work_.notify_one();
  }
  size_t queued() {
    std::lock_guard<std::mutex> lk(m_);
    return q_.size();
  }
  void wait_idle() {  // every task submitted so far (and every task those submitted) has run
    std::unique_lock<std::mutex> lk(m_);
    cv_idle_.wait(lk, [this] { return open_ == 0; });
  }

 private:
  void run() {
    for (;;) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_work_.wait(lk, [this] { return stop_ || !q_.empty(); });
        if (q_.empty()) return;
        f = std::move(q_.front());
        q_.pop_front();
      }
      f();
      std::lock_guard<std::mutex> lk(m_);
      if (--open_ == 0) cv_idle_.notify_all();
    }
  }
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_work_, cv_idle_;
  std::deque<std::function<void()>> q_;
  size_t open_ = 0;
  bool stop_ = false;
};

// ------------------------------------------------------------------ .mut rows, reduced to what a pair's walk needs
// A row that fails the row-level conditions of coal.cpp:2150 (flipped, one branch, age_begin < age_end) or whose alleles are
// not single bases (coal.cpp:2160-2176) touches neither stream nor generator in the reference's loop (without masks, which
// --pairs refuses): such rows are dropped when the file is parsed.
struct CompactRow {
  int pos;
  float age_begin, age_end;
  char anc, der;
};

bool compact_row(const MutRow& m, CompactRow& c) {
  if (!(m.flipped == 0 && m.num_branches == 1 && m.age_begin < m.age_end && m.age_end >= 0.0)) return false;
  const std::string& mt = m.mutation_type;  // "anc/der" (mutations.cpp:236-246 splits at the first '/')
  const size_t slash = mt.find('/');
  if (slash != 1 || mt.size() != 3) return false;  // both sides exactly one character (empty sides: `continue`; longer: use = false)
  const char a = mt[0], d = mt[2];
  if (!(a == 'A' || a == 'C' || a == 'G' || a == 'T' || a == '0')) return false;
  if (!(d == 'A' || d == 'C' || d == 'G' || d == 'T' || d == '1')) return false;
  c.pos = m.pos, c.age_begin = m.age_begin, c.age_end = m.age_end, c.anc = a, c.der = d;
  return true;
}

// ------------------------------------------------------------------ a .colate.in file, in memory once
// Record (little-endian, no header), coal.cpp:2505-2514 / 2126-2133:
//   int32 lchrom; char chrom[lchrom]; int32 bp; char anc; char der; int32 AAF; int32 DAF
// The file is mapped once (one copy in memory, the page cache's, whatever number of pairs walk it) and every pair's cursor
// decodes records straight out of the mapping.
// ... and decoded ONCE: the walk of every pair that takes the file then steps through 16-byte records instead of decoding the bytes
// (a length, a name to copy and compare, five fields) again -- with 10 x 10 pairs every file was decoded ten times, 4.4 G records, most
// of the 60 thread-seconds the walks of BASELINE configs[4] took.  The records are what the byte cursor below (the reference's fread
// calls) yields, call by call, so nothing about a short last record or a persisting name buffer changes.
struct DecRec {
  int32_t bp, AAF, DAF;
  uint16_t chrom;  // index into TmpFile::names (0: the empty name the buffer starts with)
  char anc, der;
};
static_assert(sizeof(DecRec) == 16, "DecRec");

struct TmpFile {
  std::string path;
  const char* data = nullptr;
  size_t size = 0;
  bool ok = false;
  HugeVector<DecRec> recs;
  std::vector<std::string> names{std::string()};
  bool decoded = false;  // false: more than 65535 distinct names -- the walks decode the bytes themselves
  // What a walk finds in this file, row by row of the .mut files, as far as it does not depend on the other sample of the pair
  // (build_walk_index below): as the reference sample of a pair / as its target.
  struct RefIdx {
    int32_t prev_pass;  // position of the latest earlier row of the chromosome that passes as reference (-1: none)
    uint16_t DAF, N;    // the record's DAF and DAF + AAF where the row passes the tests of coal.cpp:2181-2199, else 0, 0
  };
  struct TgtIdx {
    int32_t prev_bp;    // position of the record in front of the row's (-2: the row's record is the chromosome's first, or there is none)
    uint16_t DAF, AAF;  // the record's counts where position and alleles match (coal.cpp:2201-2219), else 0, 0
  };
  std::vector<HugeVector<RefIdx>> ref_idx;  // [chromosome][row]
  std::vector<HugeVector<TgtIdx>> tgt_idx;
  bool indexable = false, want_ref = false, want_tgt = false;
  TmpFile() = default;
  TmpFile(const TmpFile&) = delete;
  TmpFile& operator=(const TmpFile&) = delete;
  ~TmpFile() {
    if (data && size) ::munmap(const_cast<char*>(data), size);
  }
};

bool load_tmp_file(TmpFile& f) {
  const int fd = ::open(f.path.c_str(), O_RDONLY);
  if (fd < 0) return false;
  struct stat st;
  if (::fstat(fd, &st) != 0) {
    ::close(fd);
    return false;
  }
  f.size = (size_t)st.st_size;
  if (f.size) {
    void* m = ::mmap(nullptr, f.size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);  // (populate: read in now, on this pool thread)
    if (m == MAP_FAILED) {
      ::close(fd);
      f.size = 0;
      return false;
    }
    f.data = static_cast<const char*>(m);
    ::madvise(m, f.size, MADV_SEQUENTIAL);
  }
  ::close(fd);
  f.ok = true;
  return true;
}

// The reference's FILE* together with the variables its fread calls fill (coal.cpp:2085-2087, 2126-2133): a field the file
// ends in front of (or inside) keeps the bytes it had, exactly as with fread; the name buffer persists from record to record.
struct ByteCursor {
  const char *p = nullptr, *end = nullptr;
  char chrom[1025] = {0};
  const char* name = "";  // the chromosome the walk is at
  bool match = true;      // strcmp(chrom, name) == 0
  int bp = 0, AAF = 0, DAF = 0;  // AAF, DAF: the walk resets them between SNPs (coal.cpp:2182-2183)
  char anc = 0, der = 0;
  bool partial = false;
  void open(const TmpFile& f) { p = f.data, end = f.data + f.size; }
  void set_name(const char* n) {
    name = n;
    match = std::strcmp(chrom, name) == 0;
  }
  bool next() {
    if (end - p < 4) {
      p = end;
      return false;
    }
    int l;
    std::memcpy(&l, p, 4), p += 4;
    if (l < 0 || l > 1023) l = 0;
    if (end - p >= l + 14) {  // the whole record is there
      std::memcpy(chrom, p, (size_t)l), p += l;
      std::memcpy(&bp, p, 4);
      anc = p[4], der = p[5];
      std::memcpy(&AAF, p + 6, 4);
      std::memcpy(&DAF, p + 10, 4);
      p += 14;
    } else {
      partial = true;  // (the file ends inside this record)
      take(chrom, (size_t)l);
      take(&bp, 4), take(&anc, 1), take(&der, 1), take(&AAF, 4), take(&DAF, 4);
    }
    chrom[l] = 0;
    match = std::strcmp(chrom, name) == 0;
    return true;
  }

 private:
  void take(void* dst, size_t n) {
    const size_t k = std::min(n, (size_t)(end - p));
    std::memcpy(dst, p, k);
    p += k;
  }
};

// decode the whole file through the byte cursor (once per file, on a pool thread)
void decode_tmp_file(TmpFile& f) {
  ByteCursor c;
  c.open(f);
  c.set_name("");
  f.recs.reserve(f.size / 19 + 16);  // (a record with a one-character name is 19 bytes)
  uint16_t last = 0;
  f.decoded = true;
  while (c.next()) {
    if (c.partial) {  // a short last record keeps, field by field, what the walk's variables held: left to the byte cursor of each walk
      f.decoded = false;
      f.recs = HugeVector<DecRec>();
      return;
    }
    if (f.names[last] != c.chrom) {
      size_t k = 0;
      while (k < f.names.size() && f.names[k] != c.chrom) k++;
      if (k == f.names.size()) {
        if (f.names.size() >= 65535) {
          f.decoded = false;
          f.recs = HugeVector<DecRec>();
          return;
        }
        f.names.emplace_back(c.chrom);
      }
      last = (uint16_t)k;
    }
    f.recs.push_back(DecRec{c.bp, c.AAF, c.DAF, last, c.anc, c.der});
  }
}

// A pair's cursor into one file: the decoded records where there are any (the same sequence of states the byte cursor goes through),
// else the bytes.
struct Cursor {
  const DecRec *r = nullptr, *rend = nullptr;
  const TmpFile* file = nullptr;
  ByteCursor bytes;
  uint16_t chrom = 0;
  int name_id = -1;       // index of the walk's chromosome among the file's names (-1: no record has it)
  bool match = true;
  int bp = 0, AAF = 0, DAF = 0;
  char anc = 0, der = 0;
  void open(const TmpFile& f) {
    file = &f;
    if (f.decoded) r = f.recs.data(), rend = r + f.recs.size();
    else bytes.open(f);
  }
  void set_name(const char* n) {
    if (!file->decoded) {
      bytes.set_name(n);
      match = bytes.match;
      return;
    }
    name_id = -1;
    for (size_t k = 0; k < file->names.size(); k++)
      if (file->names[k] == n) name_id = (int)k;
    match = (int)chrom == name_id;
  }
  bool next() {
    if (!file->decoded) {
      bytes.AAF = AAF, bytes.DAF = DAF;  // (the walk resets these between SNPs; a short last record keeps what it does not reach)
      const bool ok = bytes.next();
      bp = bytes.bp, AAF = bytes.AAF, DAF = bytes.DAF, anc = bytes.anc, der = bytes.der, match = bytes.match;
      return ok;
    }
    if (r == rend) return false;
    bp = r->bp, AAF = r->AAF, DAF = r->DAF, anc = r->anc, der = r->der, chrom = r->chrom;
    match = (int)chrom == name_id;
    r++;
    return true;
  }
};

// ------------------------------------------------------------------ what a pair's walk finds in one file, computed once per file
// The walk of coal.cpp:2125-2243 steps two cursors through the two samples' records, row by row of the .mut file.  What it finds is
// almost a property of each file alone.  For files in which every chromosome of the list is one run of records, the runs in the list's
// order, positions not descending -- and .mut rows whose positions do not descend --:
//   * the REFERENCE cursor is advanced at every row, whatever the target: the row passes iff the cursor had to move in this row's
//     search (its DAF / AAF are reset in front of every search, coal.cpp:2182-2183, and only a record read now sets them again:
//     a record reached while an earlier row was searched, or the chromosome's first, which the skip loop reads, gives DAF = 0),
//     stops on a record of the row's position and alleles, and that record's DAF is not 0;
//   * the TARGET cursor is advanced only at rows that passed as reference.  Its record for a row is the first at or behind the
//     row's position; it counts iff position and alleles match and the cursor moved in this row's search, i.e. iff the record in
//     front of it lies at or behind the latest earlier row that passed as reference (which that search had started from) -- a
//     number of the target file (prev_bp) against a number of the reference file (prev_pass).
// So a pair's walk is one pass over two 8-byte arrays instead of two cursor merges over 16-byte records with a name to track:
// 100 pairs x 1 GB of streaming became 100 x 0.3 GB, and a few instructions per row.  Anything else (a chromosome missing in a file,
// runs out of order, a position below the one in front of it, a file that was not decoded, an empty chromosome name) keeps the cursors.
struct WalkRows {
  const std::vector<std::string>* names;
  const std::vector<HugeVector<CompactRow>>* rows;
  bool rows_ascend = false;
};

bool find_runs(const TmpFile& f, const std::vector<std::string>& names, std::vector<std::pair<size_t, size_t>>& runs) {
  if (!f.decoded) return false;
  const size_t C = names.size(), n = f.recs.size();
  runs.assign(C, {0, 0});
  std::vector<int> list_of(f.names.size(), -1);  // file's name index -> position in the list (-1: not listed)
  for (size_t c = 0; c < C; c++) {
    if (names[c].empty()) return false;
    for (size_t d = 0; d < c; d++)
      if (names[d] == names[c]) return false;
    for (size_t k = 0; k < f.names.size(); k++)
      if (f.names[k] == names[c]) list_of[k] = (int)c;
  }
  int last = -1;          // list position of the latest run of a listed chromosome
  bool in_listed = false;  // inside such a run
  for (size_t k = 0; k < n; k++) {
    const bool starts = k == 0 || f.recs[k].chrom != f.recs[k - 1].chrom;
    if (starts) {
      if (in_listed) runs[(size_t)last].second = k;
      const int li = list_of[f.recs[k].chrom];
      in_listed = li >= 0;
      if (in_listed) {
        if (li <= last) return false;  // a second run of a chromosome, or the runs not in the list's order
        last = li;
        runs[(size_t)li].first = k;
      }
    } else if (in_listed && f.recs[k].bp < f.recs[k - 1].bp) {
      return false;  // (equal positions are fine: a cursor stops at the first of them, and so do the indices)
    }
  }
  if (in_listed) runs[(size_t)last].second = n;
  for (size_t c = 0; c < C; c++)
    if (runs[c].second <= runs[c].first) return false;  // (a chromosome without records: the skip loop would run to the end of the file)
  return true;
}

void build_walk_index(TmpFile& f, const WalkRows& w) {
  f.indexable = false;
  std::vector<std::pair<size_t, size_t>> runs;
  if (!w.rows_ascend || !find_runs(f, *w.names, runs)) return;
  const size_t C = w.names->size();
  if (f.want_ref) f.ref_idx.assign(C, HugeVector<TmpFile::RefIdx>());
  if (f.want_tgt) f.tgt_idx.assign(C, HugeVector<TmpFile::TgtIdx>());
  for (size_t c = 0; c < C; c++) {
    const HugeVector<CompactRow>& rr = (*w.rows)[c];
    const DecRec* const R = f.recs.data();
    const size_t b = runs[c].first, e = runs[c].second;
    if (f.want_ref) {
      HugeVector<TmpFile::RefIdx>& out = f.ref_idx[c];
      out.resize(rr.size());
      size_t k = b;  // the record the cursor is on: the chromosome's first, read by the skip loop (or by the overrun of the chromosome before)
      int32_t prev_pass = -1;
      bool off_end = false;
      for (size_t i = 0; i < rr.size(); i++) {
        TmpFile::RefIdx x{prev_pass, 0, 0};
        if (!off_end) {
          size_t k2 = k;
          while (k2 < e && R[k2].bp < rr[i].pos) k2++;
          if (k2 == e) {
            off_end = true;  // the cursor has left the chromosome: no match for this row nor any later one
          } else {
            if (k2 > k && R[k2].bp == rr[i].pos && R[k2].anc == rr[i].anc && R[k2].der == rr[i].der && R[k2].DAF != 0) {
              const long long N = (long long)R[k2].DAF + R[k2].AAF;
              if (R[k2].DAF < 0 || R[k2].DAF > 65535 || N <= 0 || N > 65535) return;  // (counts beyond the index's fields: cursors)
              x.DAF = (uint16_t)R[k2].DAF, x.N = (uint16_t)N;
              prev_pass = rr[i].pos;
            }
            k = k2;
          }
        }
        out[i] = x;
      }
    }
    if (f.want_tgt) {
      HugeVector<TmpFile::TgtIdx>& out = f.tgt_idx[c];
      out.resize(rr.size());
      size_t k = b;
      for (size_t i = 0; i < rr.size(); i++) {
        while (k < e && R[k].bp < rr[i].pos) k++;
        TmpFile::TgtIdx x{-2, 0, 0};
        if (k < e && k > b) x.prev_bp = R[k - 1].bp;
        if (k < e && R[k].bp == rr[i].pos && R[k].anc == rr[i].anc && R[k].der == rr[i].der) {
          if (R[k].DAF < 0 || R[k].DAF > 65535 || R[k].AAF < 0 || R[k].AAF > 65535) return;
          x.DAF = (uint16_t)R[k].DAF, x.AAF = (uint16_t)R[k].AAF;
        }
        if (x.prev_bp < -2) return;  // (negative positions: cursors)
        out[i] = x;
      }
    }
  }
  f.indexable = true;
}

// ------------------------------------------------------------------ the uniform stream of the seed, generated once
// std::uniform_real_distribution<double>(0, 1) on std::mt19937 = generate_canonical<double, 53>: two 32-bit draws per value
// (mut_driver.cpp, canonical_fast).  One producer thread fills a ring of chunks; readers address the stream by offset.
inline double canonical_from_words(uint32_t r1, uint32_t r2) {
  double ret = ((double)r1 + (double)r2 * 4294967296.0) * 0x1p-64;
  if (ret >= 1.0) ret = std::nextafter(1.0, 0.0);
  return ret;
}

// does the bulk generator reproduce this machine's library on this seed?  (the sequence is part of the result)
bool bulk_stream_ok(unsigned seed) {
  std::mt19937 lib(seed);
  BulkMt19937 b;
  if (!b.load(lib)) return false;
  std::uniform_real_distribution<double> d(0, 1);
  uint32_t w[2 * 1300];
  b.generate(w, 2 * 1300);  // (across two regenerations of the state)
  for (int i = 0; i < 1300; i++)
    if (d(lib) != canonical_from_words(w[2 * i], w[2 * i + 1])) return false;
  std::mt19937 back;
  return b.store(back) && back == lib;
}

class SharedUniforms {
 public:
  static constexpr uint64_t kChunk = 1u << 18;  // doubles per chunk (2 MB)
  SharedUniforms(unsigned seed, size_t ring_chunks) : ring_(ring_chunks), mem_(ring_chunks * kChunk) {
    std::mt19937 g(seed);
    bulk_.load(g);
    for (size_t k = 0; k < ring_.size(); k++) ring_[k].u = mem_.data() + k * kChunk;  // (one allocation: consecutive chunks are consecutive in memory, up to the wrap)
    converter_ = std::thread([this] { convert_loop(); });
    worker_ = std::thread([this] { produce(); });
  }
  ~SharedUniforms() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_room_.notify_all();
    if (worker_.joinable()) worker_.join();  // (returns when the conversions it handed out are done)
    {
      std::lock_guard<std::mutex> lk(m_);
      conv_stop_ = true;
    }
    cv_conv_.notify_all();
    if (converter_.joinable()) converter_.join();
  }
  // the 100 uniforms at stream offsets [off, off + 100): a pointer into the ring, or `tmp` when they lie across two chunks
  const double* get100(uint64_t off, double* tmp) {
    const uint64_t c = off / kChunk, pos = off % kChunk;
    const Slot& s = slot(c);
    if (pos + 100 <= kChunk) return s.u + pos;
    const uint64_t k = kChunk - pos;
    std::memcpy(tmp, s.u + pos, k * sizeof(double));
    std::memcpy(tmp + k, slot(c + 1).u, (100 - k) * sizeof(double));
    return tmp;
  }
  double get1(uint64_t off) { return slot(off / kChunk).u[off % kChunk]; }
  const double* chunk(uint64_t c) { return slot(c).u; }  // (waits until it has been generated)
  size_t ring_chunks() const { return ring_.size(); }
  template <typename F>
  void for_each_buffer(F f) {  // (the ring's memory, e.g. to page-lock it for uploads)
    f(mem_.data(), mem_.size() * sizeof(double));
  }
  // the generator as a sequential run holds it after `off` uniforms
  bool state_at(uint64_t off, std::mt19937& g) {
    BulkMt19937 b = slot(off / kChunk).at_start;
    b.discard(2 * (off % kChunk));
    return b.store(g);
  }
  // chunks below `chunk` are no longer needed by anyone
  void release_before(uint64_t chunk) {
    {
      std::lock_guard<std::mutex> lk(m_);
      floor_ = std::max(floor_, chunk);
    }
    cv_room_.notify_all();
  }
  double waited() const { return waited_.load(); }
  double generate_seconds() const { return gen_s_.load(); }
  double convert_seconds() const { return conv_s_.load(); }

 private:
  struct Slot {
    double* u = nullptr;
    BulkMt19937 at_start;
    std::atomic<int64_t> chunk{-1};
  };
  const Slot& slot(uint64_t c) {
    Slot& s = ring_[c % ring_.size()];
    if (s.chunk.load(std::memory_order_acquire) != (int64_t)c) {
      const double t0 = now_s();
      std::unique_lock<std::mutex> lk(m_);
      cv_ready_.wait(lk, [&] { return s.chunk.load(std::memory_order_acquire) == (int64_t)c; });
      double w = waited_.load();
      while (!waited_.compare_exchange_weak(w, w + (now_s() - t0))) {}
    }
    return s;
  }
  // The generator's words are sequential (one std::mt19937); turning two of them into a double is not: the producer only
  // generates -- half of the time a chunk took -- and a second thread converts and publishes the chunk.  Four word buffers go
  // round.  (A thread of its own, not a task of the pool: a worker that waits for a chunk must not be what its conversion waits for.)
  void produce() {
    constexpr int kBufs = 4;
    std::vector<std::vector<uint32_t>> words(kBufs, std::vector<uint32_t>(2 * kChunk));
    for (uint64_t c = 0;; c++) {
      int b = -1;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_room_.wait(lk, [&] { return stop_ || (c < floor_ + ring_.size() && busy_bufs_ < kBufs); });
        if (stop_) break;
        for (int k = 0; k < kBufs; k++)
          if (!(buf_mask_ & (1u << k))) b = k;
        buf_mask_ |= 1u << b;
        busy_bufs_++;
      }
      Slot& s = ring_[c % ring_.size()];
      s.at_start = bulk_;
      uint32_t* w = words[(size_t)b].data();
      const double tg = now_s();
      bulk_.generate(w, 2 * kChunk);
      gen_s_.store(gen_s_.load(std::memory_order_relaxed) + (now_s() - tg), std::memory_order_relaxed);
      auto convert = [this, &s, w, c, b] {
        const double tc = now_s();
        double* u = s.u;
        for (uint64_t i = 0; i < kChunk; i++) u[i] = canonical_from_words(w[2 * i], w[2 * i + 1]);
        conv_s_.store(conv_s_.load(std::memory_order_relaxed) + (now_s() - tc), std::memory_order_relaxed);
        {
          std::lock_guard<std::mutex> lk(m_);
          s.chunk.store((int64_t)c, std::memory_order_release);
          buf_mask_ &= ~(1u << b);
          busy_bufs_--;
        }
        cv_ready_.notify_all();
        cv_room_.notify_all();
      };
      {
        std::lock_guard<std::mutex> lk(m_);
        conv_q_.push_back(convert);
      }
      cv_conv_.notify_one();
    }
    std::unique_lock<std::mutex> lk(m_);  // (the word buffers die with this frame: wait for the conversions still out)
    cv_room_.wait(lk, [&] { return busy_bufs_ == 0; });
  }
  void convert_loop() {
    for (;;) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_conv_.wait(lk, [&] { return conv_stop_ || !conv_q_.empty(); });
        if (conv_q_.empty()) return;
        f = std::move(conv_q_.front());
        conv_q_.pop_front();
      }
      f();
    }
  }
  std::vector<Slot> ring_;
  HugeVector<double> mem_;
  BulkMt19937 bulk_;
  std::thread worker_;
  std::mutex m_;
  std::condition_variable cv_ready_, cv_room_;
  uint64_t floor_ = 0;
  bool stop_ = false, conv_stop_ = false;
  unsigned buf_mask_ = 0;
  int busy_bufs_ = 0;
  std::deque<std::function<void()>> conv_q_;
  std::condition_variable cv_conv_;
  std::thread converter_;
  std::atomic<double> waited_{0.0};
  std::atomic<double> gen_s_{0.0}, conv_s_{0.0};  // (written by the producer / the converter only)
};

// ------------------------------------------------------------------ the age bin of a sampled age, without the logarithm
// bin(x) = max(0, (int)round(log(10 x) * C) + 1) (coal.cpp:2265, 2284) is a step function of x with one step per age bin.
// The steps are located once by bisection over the doubles ON THE LIBRARY EXPRESSION ITSELF (age_bin_index); a sample is
// then classified by a table on its leading bits and one or two comparisons.  The library expression may disagree with the
// exact mathematical step by the last bits of log(): any x within 64 ulps of a located step goes through the library
// expression instead (log's error, under one ulp of a value below 24, moves the step by fewer than 16 ulps of x), so the
// result equals age_bin_index(x) for every x, by construction and by the self-check in the constructor.
class FastBin {
 public:
  FastBin(int A, double C) : A_(A), C_(C), thr_(A + 2), lo_(A + 2), hi_(A + 2) {
    const double inf = std::numeric_limits<double>::infinity();
    thr_[0] = lo_[0] = hi_[0] = -inf;
    thr_[A + 1] = lo_[A + 1] = hi_[A + 1] = inf;
    for (int k = 1; k <= A; k++) {
      uint64_t a = bits(1e-300), b = bits(1e300);  // f(a) < k <= f(b)
      while (b - a > 1) {
        const uint64_t mid = a + (b - a) / 2;
        if (f(from_bits(mid)) >= k) b = mid; else a = mid;
      }
      thr_[k] = from_bits(b);
      lo_[k] = from_bits(b - 64);
      hi_[k] = from_bits(b + 64);
    }
    base_ = bits(thr_[1]) >> kShift;
    const uint64_t top = bits(thr_[A]) >> kShift;
    cell_.resize(top - base_ + 2);
    for (size_t c = 0; c < cell_.size(); c++) {
      const double edge = std::max(from_bits((base_ + c) << kShift), thr_[1]);
      cell_[c] = (uint16_t)std::min(f(edge), A);
    }
    // self-check on the steps, their neighbourhoods and random samples; a failure switches the table off
    ok_ = true;
    std::mt19937_64 g(99);
    for (int k = 1; k <= A && ok_; k++)
      for (int d = -200; d <= 200 && ok_; d++) ok_ = agrees(from_bits(bits(thr_[k]) + d));
    for (int i = 0; i < 200000 && ok_; i++)
      ok_ = agrees(std::exp(std::uniform_real_distribution<double>(std::log(0.01), std::log(2e7))(g)));
    ok_ = ok_ && agrees(0.0) && agrees(1e-310) && agrees(0.05) && agrees(1e9);
  }
  bool ok() const { return ok_; }
  int bins() const { return A_; }
  // lower / upper edge of the guard band around step k (k = 1 .. A; [0] = -inf, [A + 1] = +inf): bin(x) = #{k : hi(k) <= x} for
  // every x outside all bands
  const double* guard_lo() const { return lo_.data(); }
  const double* guard_hi() const { return hi_.data(); }
  // age_bin_index(x, C), with every value >= A returned as A
  int operator()(double x) const {
    if (!ok_) return std::min(f(x), A_);
    if (x < lo_[1]) return 0;
    if (x >= hi_[A_]) return A_;
    if (x < hi_[1]) return x < thr_[1] ? std::min(f(x), A_) : slow_or(x, 1);
    const size_t c = (size_t)((bits(x) >> kShift) - base_);
    const int k = cell_[c];
    if (x < hi_[k]) return std::min(f(x), A_);
    if (x < lo_[k + 1]) return k;
    if (x >= hi_[k + 1]) return std::min(k + 1, A_);
    return std::min(f(x), A_);
  }

 private:
  static constexpr int kShift = 46;  // 11 exponent bits + 6 leading mantissa bits: a cell spans a factor <= 1 + 1/64, a bin e^0.1
  static uint64_t bits(double x) {
    uint64_t u;
    std::memcpy(&u, &x, 8);
    return u;
  }
  static double from_bits(uint64_t u) {
    double x;
    std::memcpy(&x, &u, 8);
    return x;
  }
  int f(double x) const { return age_bin_index(x, C_); }
  int slow_or(double x, int k) const { return x >= hi_[k] ? k : std::min(f(x), A_); }
  bool agrees(double x) const {
    FastBin* self = const_cast<FastBin*>(this);
    const bool keep = self->ok_;
    self->ok_ = true;
    const int fast = (*this)(x);
    self->ok_ = keep;
    return fast == std::min(f(x), A_);
  }
  int A_;
  double C_;
  std::vector<double> thr_, lo_, hi_;
  std::vector<uint16_t> cell_;
  uint64_t base_ = 0;
  bool ok_ = false;
};

// ------------------------------------------------------------------ the 100 sampled ages of one SNP, eight (four) at a time
// All 100 ages of a SNP lie in [age_begin, age_end], a handful of age bins (age_end <= 2.5 age_begin: ten bins of e^0.1): the bin of
// a sample is b_lo + the number of steps k in (b_lo, b_hi + 1] at or below it -- one vector compare per step instead of a table
// walk per sample.  Exact by the same argument as FastBin: the steps are counted twice, against the lower and against the upper
// edge of their guard bands; the counts differ iff some sample lies inside a band, and then (as when the range is too wide, or
// touches the end of the grid) the SNP goes through the scalar code.  x = u * span + begin is formed by a separate multiply and add,
// like the reference's (no fused multiply-add anywhere).  Returns false = "use the scalar path"; else bins[0..99] are set.
using BinSnpFn = bool (*)(const double* u, double span, double begin, const double* lo, const double* hi, int b_lo, int K, int* bins);

__attribute__((target("avx512f"))) bool bin_snp_avx512(const double* u, double span, double begin, const double* lo, const double* hi,
                                                        int b_lo, int K, int* bins) {
  const __m512d vs = _mm512_set1_pd(span), vb = _mm512_set1_pd(begin);
  const __m512i one = _mm512_set1_epi64(1);
  __mmask8 bad = 0;
  for (int i = 0; i < 104; i += 8) {  // (u is padded to 104 values)
    const __m512d x = _mm512_add_pd(_mm512_mul_pd(_mm512_loadu_pd(u + i), vs), vb);
    __m512i c_lo = _mm512_setzero_si512(), c_hi = _mm512_setzero_si512();
    for (int j = 1; j <= K; j++) {
      c_lo = _mm512_mask_add_epi64(c_lo, _mm512_cmp_pd_mask(x, _mm512_set1_pd(lo[b_lo + j]), _CMP_GE_OQ), c_lo, one);
      c_hi = _mm512_mask_add_epi64(c_hi, _mm512_cmp_pd_mask(x, _mm512_set1_pd(hi[b_lo + j]), _CMP_GE_OQ), c_hi, one);
    }
    bad |= _mm512_cmpneq_epi64_mask(c_lo, c_hi);
    bad |= _mm512_cmp_pd_mask(x, _mm512_set1_pd(hi[b_lo]), _CMP_LT_OQ);  // inside (or below) the band of the step the range starts at
    _mm256_storeu_si256(reinterpret_cast<__m256i*>(bins + i), _mm512_cvtepi64_epi32(_mm512_add_epi64(c_hi, _mm512_set1_epi64(b_lo))));
  }
  return bad == 0;
}

__attribute__((target("avx2"))) bool bin_snp_avx2(const double* u, double span, double begin, const double* lo, const double* hi, int b_lo,
                                                  int K, int* bins) {
  const __m256d vs = _mm256_set1_pd(span), vb = _mm256_set1_pd(begin);
  __m256i bad = _mm256_setzero_si256();
  for (int i = 0; i < 100; i += 4) {
    const __m256d x = _mm256_add_pd(_mm256_mul_pd(_mm256_loadu_pd(u + i), vs), vb);
    __m256i c_lo = _mm256_setzero_si256(), c_hi = _mm256_setzero_si256();
    for (int j = 1; j <= K; j++) {  // (a true compare is all ones = -1: subtracting it adds one)
      c_lo = _mm256_sub_epi64(c_lo, _mm256_castpd_si256(_mm256_cmp_pd(x, _mm256_set1_pd(lo[b_lo + j]), _CMP_GE_OQ)));
      c_hi = _mm256_sub_epi64(c_hi, _mm256_castpd_si256(_mm256_cmp_pd(x, _mm256_set1_pd(hi[b_lo + j]), _CMP_GE_OQ)));
    }
    bad = _mm256_or_si256(bad, _mm256_xor_si256(c_lo, c_hi));
    bad = _mm256_or_si256(bad, _mm256_castpd_si256(_mm256_cmp_pd(x, _mm256_set1_pd(hi[b_lo]), _CMP_LT_OQ)));
    alignas(32) long long c[4];
    _mm256_store_si256(reinterpret_cast<__m256i*>(c), c_hi);
    bins[i] = b_lo + (int)c[0], bins[i + 1] = b_lo + (int)c[1], bins[i + 2] = b_lo + (int)c[2], bins[i + 3] = b_lo + (int)c[3];
  }
  return _mm256_testz_si256(bad, bad) != 0;
}

// The additions themselves: bin j of the SNP's range gets cnt[j] additions of the SNP's weight, one after the other (what the
// sample-by-sample loop does to it, in the same order).  Up to sixteen bins side by side in two vectors per table, a masked add per
// step: sixteen chains of dependent additions run at once instead of one after the other.  (Needs b_lo + 16 <= A.)
using AddSnpFn = void (*)(double* sh, double* ns, int b_lo, const int* cnt, double w_sh, double w_ns);

__attribute__((target("avx512f"))) void add_snp_avx512(double* sh, double* ns, int b_lo, const int* cnt, double w_sh, double w_ns) {
  __m512d s0 = _mm512_loadu_pd(sh + b_lo), s1 = _mm512_loadu_pd(sh + b_lo + 8);
  __m512d n0 = _mm512_loadu_pd(ns + b_lo), n1 = _mm512_loadu_pd(ns + b_lo + 8);
  const __m512i c0 = _mm512_cvtepi32_epi64(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(cnt)));
  const __m512i c1 = _mm512_cvtepi32_epi64(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(cnt + 8)));
  const __m512d ws = _mm512_set1_pd(w_sh), wn = _mm512_set1_pd(w_ns);
  int most = 0;
  for (int j = 0; j < 16; j++) most = cnt[j] > most ? cnt[j] : most;
  for (int step = 0; step < most; step++) {
    const __m512i st = _mm512_set1_epi64(step);
    const __mmask8 m0 = _mm512_cmpgt_epi64_mask(c0, st), m1 = _mm512_cmpgt_epi64_mask(c1, st);
    s0 = _mm512_mask_add_pd(s0, m0, s0, ws), s1 = _mm512_mask_add_pd(s1, m1, s1, ws);
    n0 = _mm512_mask_add_pd(n0, m0, n0, wn), n1 = _mm512_mask_add_pd(n1, m1, n1, wn);
  }
  _mm512_storeu_pd(sh + b_lo, s0), _mm512_storeu_pd(sh + b_lo + 8, s1);
  _mm512_storeu_pd(ns + b_lo, n0), _mm512_storeu_pd(ns + b_lo + 8, n1);
}

BinSnpFn pick_bin_snp() {
  if (std::getenv("COLATE_NO_SIMD")) return nullptr;
  __builtin_cpu_init();
  if (__builtin_cpu_supports("avx512f")) return bin_snp_avx512;
  if (__builtin_cpu_supports("avx2")) return bin_snp_avx2;
  return nullptr;
}
AddSnpFn pick_add_snp() {
  if (std::getenv("COLATE_NO_SIMD")) return nullptr;
  __builtin_cpu_init();
  return __builtin_cpu_supports("avx512f") ? add_snp_avx512 : nullptr;
}

// ------------------------------------------------------------------ one pair's tables and its walk through the SNPs
struct UsedSnp {
  double age_begin, age_end, w_sh, w_ns;
  bool emp;  // age_begin <= sample age: the F path (coal.cpp:2245-2275), not-shared weight only, no redraws
};
struct Block {
  std::vector<double> t;  // sh | ns | sh_emp | ns_emp, A values each (emp = row 0 of the reference's A*A tables)
  explicit Block(int A) : t((size_t)4 * A, 0.0) {}
};

struct PairFill {
  // inputs
  size_t index = 0;
  const TmpFile *tgt_file = nullptr, *ref_file = nullptr;
  // walk state (coal.cpp:2071-2321)
  Cursor tgt, ref;
  size_t chr = 0, row = 0;
  bool chr_open = false;
  int current_block_base = 0;
  size_t blk = 0;
  int num_blocks = 0;
  uint64_t off = 0;  // uniforms taken so far
  std::vector<UsedSnp> pending;
  uint64_t pending_off = 0;
  // sampling on the device: the records of the block being walked (handed over when the block is complete) and this pair's tables there
  std::vector<FillRec> dev_cur;
  uint64_t dev_cur_off = 0;
  size_t slot = 0;
  bool walked = false;
  double inline_sample_s = 0;  // seconds this pair's walker spent sampling itself (pool full)
  // results
  std::vector<std::unique_ptr<Block>> blocks;
  std::atomic<bool> redo{false};  // a sample beyond the age grid had to be redrawn: this pair is filled sequentially afterwards
  std::mt19937 rng_end;
  size_t used_snps = 0;
};

// (pair, block) jobs whose block is complete, waiting for the next hand-over to the device (fill_pairs, at the end of a window)
struct DevQueue {
  static constexpr uint32_t kMaxBlocks = 512;  // tables per pair on the device (a pair with more goes back to the host)
  std::mutex m;
  struct Item {
    FillJob job;
    std::vector<FillRec> recs;
  };
  std::vector<Item> ready;
  // record vectors whose job has been handed over, for the next blocks: a fresh vector per block is a fresh mapping of a megabyte --
  // 13 GB of page faults over BASELINE configs[4], whose price (0.3 .. 2 us each, 16 threads on one address space) was the
  // difference between walks of 28 and of 68 thread-seconds from one run to the next
  std::vector<std::vector<FillRec>> spare;
};

struct Engine {
  const std::vector<std::string>& chr_names;
  const std::vector<HugeVector<CompactRow>>& rows;  // per chromosome
  int A;
  double C;
  int num_bases_per_block;
  SharedUniforms& stream;
  const FastBin& fastbin;
  Pool& pool;
  BinSnpFn bin_snp = nullptr;  // the vector form of the 100 bins of a SNP where the CPU has one (and the table passed its self-check)
  AddSnpFn add_snp = nullptr;  // ... and of the additions
  DevQueue* devq = nullptr;    // not null: the sampling runs on the device (fill_device.h)
  bool use_index = true;       // walk through the per-file indices where both files of a pair have one (build_walk_index)

  // the 100 draws of every SNP of one genome-block segment, in order (coal.cpp:2260-2273, 2279-2295)
  void sample(PairFill& pf, Block& b, const std::vector<UsedSnp>& snps, uint64_t off) const {
    struct Timer {
      double t0 = now_s();
      ~Timer() { WorkSeconds::add(g_work.sample, now_s() - t0); }
    } timer;
    double* sh = b.t.data();
    double* ns = sh + A;
    const double age = 0;  // forced, coal.cpp:2074-2075
    double tmp[104];
    const double* const g_lo = fastbin.guard_lo();
    const double* const g_hi = fastbin.guard_hi();
    for (const UsedSnp& s : snps) {
      if (pf.redo.load(std::memory_order_relaxed)) return;
      const double* u = stream.get100(off, tmp);
      const bool last_of_chunk = (off % SharedUniforms::kChunk) + 104 > SharedUniforms::kChunk;  // (the vector code reads 104 values)
      off += 100;
      const double span = s.age_end - s.age_begin;
      if (bin_snp && !s.emp) {
        // the bins the samples can fall into: from that of age_begin to that of the largest possible sample (u < 1)
        const int b_lo = fastbin(s.age_begin), b_hi = fastbin(std::nextafter(span + s.age_begin, std::numeric_limits<double>::infinity()));
        const int K = b_hi + 1 - b_lo;
        if (b_lo >= 1 && K >= 1 && K <= 16 && b_hi + 1 < A) {
          if (last_of_chunk && u != tmp) {  // (never read past the chunk: copy the hundred, pad)
            std::memcpy(tmp, u, 100 * sizeof(double));
            u = tmp;
          }
          if (u == tmp) tmp[100] = tmp[101] = tmp[102] = tmp[103] = 0.0;
          int bins[104];
          if (bin_snp(u, span, s.age_begin, g_lo, g_hi, b_lo, K, bins)) {
            // per bin: as many additions of the SNP's weight as samples fell into it, one after the other -- the same sums as the
            // sample-by-sample loop below (additions to different bins commute; those to one bin are all of the same addend)
            int cnt[18] = {0};
            for (int k = 0; k < 100; k++) cnt[bins[k] - b_lo]++;
            if (add_snp && K < 16 && b_lo + 16 <= A) {
              add_snp(sh, ns, b_lo, cnt, s.w_sh, s.w_ns);
              continue;
            }
            for (int j = 0; j <= K; j++) {
              double a = sh[b_lo + j], r = ns[b_lo + j];
              for (int n = cnt[j]; n > 0; n--) {
                a += s.w_sh;
                r += s.w_ns;
              }
              sh[b_lo + j] = a, ns[b_lo + j] = r;
            }
            continue;
          }
        }
      }
      if (s.emp) {
        for (int k = 0; k < 100; k++) {
          double sampled_age = u[k] * span + s.age_begin;
          if (sampled_age < age) sampled_age = age;
          const int bin = fastbin(sampled_age);
          if (bin < A) ns[bin] += s.w_ns;
        }
      } else {
        for (int k = 0; k < 100; k++) {
          const double sampled_age = u[k] * span + s.age_begin;
          const int bin = fastbin(sampled_age);
          if (sampled_age < age || bin >= A) {  // the reference would draw again: the stream no longer lines up
            pf.redo.store(true);
            return;
          }
          sh[bin] += s.w_sh;
          ns[bin] += s.w_ns;
        }
      }
    }
  }

  void flush(PairFill& pf) const {  // hand the current block's pending SNPs to the pool (or run them here when it is full)
    if (devq) return;  // (on the device a block is one job: its records wait in dev_cur until advance_block)
    if (pf.pending.empty()) return;
    auto snps = std::make_shared<std::vector<UsedSnp>>(std::move(pf.pending));
    pf.pending = std::vector<UsedSnp>();
    Block* b = pf.blocks[pf.blk].get();
    const uint64_t off = pf.pending_off;
    PairFill* p = &pf;
    if (pool.queued() > (size_t)(4 * pool.size() + 8)) {
      const double t0 = now_s();
      sample(*p, *b, *snps, off);
      pf.inline_sample_s += now_s() - t0;
    } else
      pool.submit([this, p, b, snps, off] { sample(*p, *b, *snps, off); });
  }
  void advance_block(PairFill& pf) const {
    flush(pf);
    if (devq && !pf.dev_cur.empty()) {
      const size_t had = pf.dev_cur.size();
      if (pf.blk >= DevQueue::kMaxBlocks || pf.dev_cur.size() > 0xffffffffull) {
        pf.redo.store(true);
      } else {
        DevQueue::Item it;
        it.job = FillJob{0, pf.dev_cur_off, (uint32_t)pf.dev_cur.size(), (uint32_t)(pf.slot * DevQueue::kMaxBlocks + pf.blk)};
        it.recs = std::move(pf.dev_cur);
        std::lock_guard<std::mutex> lk(devq->m);
        devq->ready.push_back(std::move(it));
        pf.dev_cur = std::vector<FillRec>();
        if (!devq->spare.empty()) {
          pf.dev_cur = std::move(devq->spare.back());
          devq->spare.pop_back();
        }
      }
      pf.dev_cur.clear();
      if (pf.dev_cur.capacity() == 0) pf.dev_cur.reserve(had + had / 8);  // (the next block is about as long: no doubling copies on the way)
    }
    pf.blk++;
    pf.num_blocks++;
    if (pf.blk >= pf.blocks.size()) pf.blocks.emplace_back(new Block(A));
  }

  // a SNP the pair uses (coal.cpp:2221-2297): its genome block, the row-0 entries of the F tables, and its 100 draws queued
  void use_snp(PairFill& pf, const CompactRow& m, int tgt_DAF, int tgt_AAF, int DAF_ref, int N_ref) const {
    const float num_samples = 100;
    const double age = 0, ref_age = 0;  // forced, coal.cpp:2074-2075
    const int bp_mut = m.pos;
    const int N_target = tgt_DAF + tgt_AAF;
    double age_begin = m.age_begin;
    if (age_begin < ref_age) age_begin = ref_age;
    while (pf.current_block_base + num_bases_per_block < bp_mut) {  // coal.cpp:2227-2234
      pf.current_block_base += num_bases_per_block;
      advance_block(pf);
    }
    // target genotype rounded to a diploid call, in float (coal.cpp:2236-2242)
    float f_DAF_target = tgt_DAF, f_AAF_target = tgt_AAF;
    f_DAF_target /= N_target / 2.0;
    f_AAF_target /= N_target / 2.0;
    f_DAF_target = std::round(f_DAF_target);
    f_AAF_target = std::round(f_AAF_target);
    if (pf.pending.empty()) pf.pending_off = pf.off;
    if (devq && pf.dev_cur.empty()) pf.dev_cur_off = pf.off;
    if (age_begin <= age) {  // coal.cpp:2245-2275
      const int bin2 = age_bin_index(m.age_end, C);
      if (bin2 < A) {  // row 0 of the A*A table; larger indices land in rows nobody reads
        double* t = pf.blocks[pf.blk]->t.data();
        t[2 * A + bin2] += f_DAF_target * DAF_ref / ((double)N_ref);
        t[3 * A + bin2] += f_AAF_target * DAF_ref / ((double)N_ref);
      }
      const double w_ns = f_AAF_target * DAF_ref / ((double)N_ref * num_samples);
      if (devq) pf.dev_cur.push_back(FillRec{(float)age_begin, m.age_end, 0.0, w_ns});  // (age_begin: a float, or the sample age 0)
      else pf.pending.push_back(UsedSnp{age_begin, (double)m.age_end, 0.0, w_ns, true});
    } else {  // coal.cpp:2277-2297
      const double w_sh = f_DAF_target * DAF_ref / ((double)N_ref * num_samples), w_ns = f_AAF_target * DAF_ref / ((double)N_ref * num_samples);
      if (devq) pf.dev_cur.push_back(FillRec{(float)age_begin, m.age_end, w_sh, w_ns});
      else pf.pending.push_back(UsedSnp{age_begin, (double)m.age_end, w_sh, w_ns, false});
    }
    pf.off += 100;
    pf.used_snps++;
  }

  // Walks on until the pair has taken `limit` uniforms or its SNPs are exhausted.  The test sits in front of a row, before
  // either stream is touched for it, so the walk resumes exactly where it stopped.
  void walk(PairFill& pf, uint64_t limit) const {
    const double t_walk0 = now_s(), sample0 = pf.inline_sample_s;
    walk_impl(pf, limit);
    WorkSeconds::add(g_work.walk, now_s() - t_walk0 - (pf.inline_sample_s - sample0));
  }
  void walk_impl(PairFill& pf, uint64_t limit) const {
    Cursor& tgt = pf.tgt;
    Cursor& ref = pf.ref;
    if (pf.blocks.empty()) pf.blocks.emplace_back(new Block(A));
    while (pf.chr < rows.size()) {
      if (pf.redo.load(std::memory_order_relaxed)) break;
      const bool indexed = use_index && pf.ref_file->indexable && pf.tgt_file->indexable;
      if (!pf.chr_open) {
        pf.current_block_base = 0;
        if (!indexed) {
          ref.set_name(chr_names[pf.chr].c_str());
          tgt.set_name(chr_names[pf.chr].c_str());
          while (!ref.match) {  // skip to this chromosome, coal.cpp:2125-2134
            if (!ref.next()) break;
          }
          while (!tgt.match) {
            if (!tgt.next()) break;
          }
        }
        pf.row = 0;
        pf.chr_open = true;
      }
      const HugeVector<CompactRow>& rr = rows[pf.chr];
      if (indexed) {  // what the two cursors would find, from the two files' indices (build_walk_index)
        const TmpFile::RefIdx* const RI = pf.ref_file->ref_idx[pf.chr].data();
        const TmpFile::TgtIdx* const TI = pf.tgt_file->tgt_idx[pf.chr].data();
        for (; pf.row < rr.size(); pf.row++) {
          if (pf.off >= limit) {
            flush(pf);
            return;
          }
          const TmpFile::RefIdx r = RI[pf.row];
          if (r.N == 0) continue;  // the reference sample does not carry the derived allele here (or its record was read too early)
          const TmpFile::TgtIdx t = TI[pf.row];
          if ((t.DAF | t.AAF) == 0 || r.prev_pass > t.prev_bp) continue;  // no target record here -- or one that an earlier search had reached
          use_snp(pf, rr[pf.row], t.DAF, t.AAF, r.DAF, r.N);
        }
        advance_block(pf);
        pf.chr++;
        pf.chr_open = false;
        continue;
      }
      for (; pf.row < rr.size(); pf.row++) {
        if (pf.off >= limit) {
          flush(pf);
          return;
        }
        const CompactRow& m = rr[pf.row];
        const int bp_mut = m.pos;
        bool use = true;
        // the reference sample must carry the derived allele, coal.cpp:2181-2199
        ref.DAF = 0;
        ref.AAF = 0;
        while (ref.match && ref.bp < bp_mut) {
          if (!ref.next()) break;
        }
        if (!ref.match || ref.bp != bp_mut || ref.anc != m.anc || ref.der != m.der) use = false;
        if (ref.DAF == 0) use = false;
        const int N_ref = ref.DAF + ref.AAF;
        if (use) {  // coal.cpp:2201-2219
          tgt.DAF = 0;
          tgt.AAF = 0;
          while (tgt.match && tgt.bp < bp_mut) {
            if (!tgt.next()) break;
          }
          if (!tgt.match || tgt.bp != bp_mut || tgt.anc != m.anc || tgt.der != m.der) use = false;
        }
        const int N_target = tgt.DAF + tgt.AAF;
        if (N_target == 0) use = false;
        if (!use) continue;

        use_snp(pf, m, tgt.DAF, tgt.AAF, ref.DAF, N_ref);
      }
      advance_block(pf);  // chromosome end, coal.cpp:2306-2310
      pf.chr++;
      pf.chr_open = false;
    }
    flush(pf);
    pf.walked = true;
    pf.blocks.resize((size_t)pf.num_blocks);
    if (!stream.state_at(pf.off, pf.rng_end)) pf.redo.store(true);
  }
};

// "target reference output [target_age reference_age]" per line
bool read_pair_list(const std::string& path, std::vector<PairSpec>& pairs) {
  std::ifstream is(path);
  if (!is) {
    std::cerr << "Error while opening file " << path << std::endl;
    return false;
  }
  std::string line;
  while (std::getline(is, line)) {
    std::istringstream ss(line);
    PairSpec ps;
    if (!(ss >> ps.target >> ps.reference >> ps.output)) continue;
    std::string a1, a2;
    if (ss >> a1) ps.target_age = std::stof(a1);
    if (ss >> a2) ps.ref_age = std::stof(a2);
    pairs.push_back(ps);
  }
  if (pairs.empty()) {
    std::cerr << "Error: no pairs in " << path << std::endl;
    return false;
  }
  return true;
}

struct PairTables {  // flat [nb][A] tables of one pair, as the bootstrap takes them
  int nb = 0;
  std::vector<double> sh, ns, she, nse;
  std::mt19937 rng;  // the run's generator after the table fill
};

// The tables of the pairs listed in `todo` (indices into `pairs`); false after an error message.
bool fill_pairs(const Options& opt, const std::vector<PairSpec>& pairs, const std::vector<size_t>& todo, int seed, int A,
                std::vector<PairTables>& out) {
  const double C = 10;
  const int num_bases_per_block = 30e6;
  std::vector<std::string> names, mut_files;
  chromosome_files(opt, names, mut_files);
  const int T = pairs_threads();
  const double t0 = now_s();
  out.assign(pairs.size(), PairTables());
  if (todo.empty()) return true;

  // the sequential feeder (one pair after the other, files re-read): when the bulk generator does not reproduce this
  // machine's std::mt19937 stream, and for a pair in which a sample beyond the age grid had to be redrawn
  auto fill_sequentially = [&](size_t p) {
    std::mt19937 rng;
    rng.seed(seed);
    BlockTables tab;
    const int nb = fill_tables_from_tmp(names, mut_files, pairs[p].target, pairs[p].reference, {}, {}, C, rng,
                                        num_bases_per_block, A, tab);
    PairTables& pt = out[p];
    pt.nb = nb;
    pt.sh.resize((size_t)std::max(nb, 0) * A), pt.ns.resize(pt.sh.size()), pt.she.resize(pt.sh.size()), pt.nse.resize(pt.sh.size());
    for (int j = 0; j < nb; j++) {
      std::copy(tab.sh[j].begin(), tab.sh[j].end(), pt.sh.begin() + (size_t)j * A);
      std::copy(tab.ns[j].begin(), tab.ns[j].end(), pt.ns.begin() + (size_t)j * A);
      std::copy(tab.sh_emp[j].begin(), tab.sh_emp[j].end(), pt.she.begin() + (size_t)j * A);
      std::copy(tab.ns_emp[j].begin(), tab.ns_emp[j].end(), pt.nse.begin() + (size_t)j * A);
    }
    pt.rng = rng;
  };
  if (!bulk_stream_ok((unsigned)seed)) {
    std::cerr << "Note: the bulk generator does not reproduce this machine's std::mt19937 stream; filling the pairs one by one." << std::endl;
    for (size_t p : todo) fill_sequentially(p);
    return true;
  }

  // ---- the shared uniform stream (its producer starts now), the age-bin table, and -- where there is a GPU -- the sampling on the
  // device (fill_device.h).  The host code is what runs otherwise, and for any pair the device hands back.
  size_t window_mb = 64;
  if (const char* e = std::getenv("COLATE_UNIFORM_WINDOW_MB")) window_mb = (size_t)std::max(4, std::atoi(e));
  const uint64_t W = std::max<uint64_t>(2, window_mb * (1u << 20) / (SharedUniforms::kChunk * sizeof(double)));  // chunks per window
  SharedUniforms stream((unsigned)seed, (size_t)(2 * W + 2));
  FastBin fastbin(A, C);
  if (!fastbin.ok()) std::cerr << "Note: the age-bin table failed its self-check; sampling through log()." << std::endl;
  std::unique_ptr<DeviceFill> dev;
  DevQueue devq;
  std::string dev_note;
  std::thread dev_maker;
  bool dev_pending = false, dev_staging_ok = false;
  int dev_device = 0;
  size_t dev_batch = 0;
  double dev_make_s = 0, dev_staging_s = 0;
  {
    const char* e = std::getenv("COLATE_DEVICE_FILL");
    const bool want = !(e && std::atoi(e) == 0);
    if (!want) dev_note = "COLATE_DEVICE_FILL=0";
    else if (!fastbin.ok()) dev_note = "no age-bin table";
    else if (!DeviceFill::available()) dev_note = "no HIP device";
    else {
      int device = 0;
      try {
        if (opt.has("device")) device = std::stoi(opt.get("device"));
      } catch (...) {
        device = 0;
      }
      if (g_rank.ranked) device += g_rank.rank;
      // a batch: COLATE_DEVICE_FILL_BATCH records (24 bytes each, two pinned buffers; 64 M by default) -- or all there can be: a pair
      // uses a row at most once, and a row of a .mut file is more than four bytes even compressed
      uint64_t rows_bound = 0;
      for (const std::string& f : mut_files) {
        struct stat st;
        if (::stat(f.c_str(), &st) == 0) rows_bound += (uint64_t)st.st_size / 4 + 1;
        else if (::stat((f + ".gz").c_str(), &st) == 0) rows_bound += (uint64_t)st.st_size / 4 + 1;
      }
      size_t batch = (size_t)64 << 20;
      if (const char* b = std::getenv("COLATE_DEVICE_FILL_BATCH")) batch = (size_t)std::max(1024, std::atoi(b));
      batch = std::min<uint64_t>(batch, std::max<uint64_t>(1024, rows_bound * todo.size()));
      dev_pending = true;
      dev_device = device, dev_batch = batch;
    }
  }
  struct JoinMaker {
    std::thread& t;
    ~JoinMaker() {
      if (t.joinable()) t.join();
    }
  } join_maker{dev_maker};

  Pool pool(T);
  // ---- every input file once
  std::vector<HugeVector<CompactRow>> rows(mut_files.size());
  std::vector<size_t> rows_total(mut_files.size(), 0);
  for (size_t c = 0; c < mut_files.size(); c++)
    pool.submit([&, c] {
      const double t0 = now_s();
      HugeVector<CompactRow>& r = rows[c];
      size_t n = 0;
      CompactRow cr;
      for_each_mut_row(mut_files[c], [&](const MutRow& m) {
        n++;
        if (compact_row(m, cr)) r.push_back(cr);
      });
      rows_total[c] = n;
      WorkSeconds::add(g_work.parse_mut, now_s() - t0);
    });
  std::map<std::string, std::unique_ptr<TmpFile>> tmp_files;
  for (size_t p : todo)
    for (const std::string* path : {&pairs[p].target, &pairs[p].reference})
      if (!tmp_files.count(*path)) {
        TmpFile* f = new TmpFile;
        f->path = *path;
        tmp_files[*path].reset(f);
        pool.submit([f] {
          const double t0 = now_s();
          if (load_tmp_file(*f)) {
            decode_tmp_file(*f);
            if (f->decoded && f->data && f->size) {  // (the bytes are no longer needed)
              ::munmap(const_cast<char*>(f->data), f->size);
              f->data = nullptr;
            }
          }
          WorkSeconds::add(g_work.load_tmp, now_s() - t0);
        });
      }
  pool.wait_idle();
  for (auto& kv : tmp_files)
    if (!kv.second->ok) std::cerr << "Failed to open " << kv.first << std::endl;  // (the reference goes on and reads nothing)
  // ---- what the walks find in each file, once per file (build_walk_index)
  const char* e_idx = std::getenv("COLATE_INDEXED_WALK");
  const bool use_index = !(e_idx && std::atoi(e_idx) == 0);
  size_t n_indexed = 0;
  if (use_index) {
    WalkRows wr{&names, &rows, true};
    for (const HugeVector<CompactRow>& r : rows)
      for (size_t i = 1; i < r.size() && wr.rows_ascend; i++) wr.rows_ascend = r[i].pos >= r[i - 1].pos && r[i - 1].pos >= 0;
    for (size_t p : todo) tmp_files[pairs[p].target]->want_tgt = true, tmp_files[pairs[p].reference]->want_ref = true;
    for (auto& kv : tmp_files) {
      TmpFile* f = kv.second.get();
      if (f->ok) pool.submit([f, wr] {
        const double t0 = now_s();
        build_walk_index(*f, wr);
        WorkSeconds::add(g_work.index, now_s() - t0);
      });
    }
    pool.wait_idle();
    for (auto& kv : tmp_files) n_indexed += kv.second->indexable ? 1 : 0;
  }
  const double t1 = now_s();
  size_t n_rows = 0, n_kept = 0, n_rec = 0;
  for (size_t c = 0; c < rows.size(); c++) n_rows += rows_total[c], n_kept += rows[c].size();
  for (auto& kv : tmp_files) n_rec += kv.second->size;

  // ---- the pairs, window by window through the shared uniform stream
  bool dev_failed = false;
  if (dev_pending) {
    const double t0m = now_s();
    const uint64_t max_uniforms = ((uint64_t)n_kept * 100 / SharedUniforms::kChunk + W + 3) * SharedUniforms::kChunk;  // a pair uses at most every kept row
    dev.reset(DeviceFill::create(dev_device, A, fastbin.guard_lo(), fastbin.guard_hi(), todo.size() * DevQueue::kMaxBlocks,
                                 std::min<uint64_t>(dev_batch, std::max<uint64_t>(1024, (uint64_t)n_kept * todo.size())), dev_note));
    if (!dev || !dev->alloc_uniforms(max_uniforms)) {
      std::cerr << "Note: age sampling on the GPU could not be set up (" << (dev ? dev->error() : dev_note) << "); sampling on the host." << std::endl;
      if (dev) dev_note = dev->error();
      dev.reset();
      dev_pending = false;
    } else {
      stream.for_each_buffer([&](double* p, size_t bytes) { dev->pin(p, bytes); });
      dev_make_s = now_s() - t0m;
      // (the record buffers -- gigabytes to page-lock -- beside the first windows: the first hand-over waits for them)
      dev_maker = std::thread([&] {
        const double t1m = now_s();
        dev_staging_ok = dev->alloc_staging();
        dev_staging_s = now_s() - t1m;
      });
    }
  }
  Engine eng{names, rows, A, C, num_bases_per_block, stream, fastbin, pool, fastbin.ok() ? pick_bin_snp() : nullptr, fastbin.ok() ? pick_add_snp() : nullptr,
             dev_pending ? &devq : nullptr, use_index};
  std::vector<std::unique_ptr<PairFill>> fills;
  for (size_t p : todo) {
    fills.emplace_back(new PairFill);
    PairFill& pf = *fills.back();
    pf.index = p;
    pf.slot = fills.size() - 1;
    pf.tgt_file = tmp_files[pairs[p].target].get(), pf.ref_file = tmp_files[pairs[p].reference].get();
    pf.tgt.open(*pf.tgt_file), pf.ref.open(*pf.ref_file);
  }
  int windows = 0;
  uint64_t dev_next_chunk = 0;
  size_t dev_jobs = 0, dev_recs = 0, dev_launches = 0;
  double dev_upload_s = 0, dev_finish_s = 0;
  // Hand-over to the device.  A job is a chain of dependent additions, SNP after SNP, on one wave: what makes the GPU fast is the
  // number of chains in flight.  A window completes ~2 blocks per pair -- 180 jobs for 1024 SIMDs, 145 ms per launch however few
  // they are --, so the completed blocks are kept until half a staging buffer of records has come together (some 2000 jobs at
  // BASELINE configs[4]) and go in one launch: their records side by side into the pinned buffer (the pool copies).
  std::vector<DevQueue::Item> backlog;
  size_t backlog_recs = 0;
  auto hand_over = [&](bool all) {
    while (dev && !dev_failed && !backlog.empty() && (all || backlog_recs >= dev->staging_capacity() / 2)) {
      if (dev_maker.joinable()) {
        dev_maker.join();
        if (!dev_staging_ok) {
          dev_failed = true;
          break;
        }
      }
      std::vector<FillJob> jobs;
      size_t nrecs = 0, taken = 0;
      for (; taken < backlog.size() && jobs.size() < 60000; taken++) {
        DevQueue::Item& it = backlog[taken];
        if (it.recs.size() > dev->staging_capacity()) {  // (a block larger than a batch: the host takes the pair)
          fills[it.job.table / DevQueue::kMaxBlocks]->redo.store(true);
          backlog_recs -= it.recs.size();
          it.recs.clear();
          continue;
        }
        if (nrecs + it.recs.size() > dev->staging_capacity()) break;
        it.job.rec_off = nrecs;
        nrecs += it.recs.size();
        if (!it.recs.empty()) jobs.push_back(it.job);
      }
      FillRec* const dst = dev->staging();
      for (size_t i = 0; i < taken; i++)
        if (!backlog[i].recs.empty()) {
          DevQueue::Item* itp = &backlog[i];
          pool.submit([dst, itp] { std::memcpy(dst + itp->job.rec_off, itp->recs.data(), itp->recs.size() * sizeof(FillRec)); });
        }
      pool.wait_idle();
      dev_jobs += jobs.size(), dev_recs += nrecs, dev_launches += jobs.empty() ? 0 : 1;
      if (!dev->submit(jobs, nrecs)) dev_failed = true;
      backlog_recs -= nrecs;
      {
        std::lock_guard<std::mutex> lk(devq.m);
        for (size_t i = 0; i < taken; i++)
          if (backlog[i].recs.capacity() > 0) {
            backlog[i].recs.clear();
            devq.spare.push_back(std::move(backlog[i].recs));
          }
      }
      backlog.erase(backlog.begin(), backlog.begin() + (long)taken);
    }
  };
  for (uint64_t w = 0;; w++) {
    const uint64_t limit = (w + 1) * W * SharedUniforms::kChunk;
    bool any = false;
    for (auto& pf : fills)
      if (!pf->walked && !pf->redo.load()) {
        any = true;
        PairFill* p = pf.get();
        pool.submit([&eng, p, limit] { eng.walk(*p, limit); });
      }
    if (!any) break;
    pool.wait_idle();
    if (dev && !dev_failed) {
      // the uniforms this window's jobs read (a pair's last SNP of the window may reach 100 into the next chunk): their copies are
      // enqueued and run beside the next window's walks; the ring's chunks are handed back to the producer one window late, when
      // the copies out of them have completed ...
      const double tu = now_s();
      if (!dev->sync_uploads()) dev_failed = true;
      stream.release_before(dev_next_chunk > 0 ? dev_next_chunk - 1 : 0);  // (the overlap chunk is uploaded twice: kept)
      while (dev_next_chunk <= (w + 1) * W) {  // (runs of chunks that are consecutive in the ring's memory: one copy each)
        const uint64_t first = dev_next_chunk;
        uint64_t n = 0;
        const double* p0 = stream.chunk(first);
        while (first + n <= (w + 1) * W && (first + n) % stream.ring_chunks() == first % stream.ring_chunks() + n) {
          (void)stream.chunk(first + n);  // (waits until it has been generated)
          n++;
        }
        if (!dev->upload_uniforms(first * SharedUniforms::kChunk, p0, n * SharedUniforms::kChunk)) dev_failed = true;
        dev_next_chunk += n;
      }
      dev_upload_s += now_s() - tu;
      // ... and the blocks that were completed in it join the backlog
      {
        std::lock_guard<std::mutex> lk(devq.m);
        for (DevQueue::Item& it : devq.ready) {
          backlog_recs += it.recs.size();
          backlog.push_back(std::move(it));
        }
        devq.ready.clear();
      }
      hand_over(false);
    }
    if (!dev) stream.release_before((w + 1) * W);
    if (dev && dev_failed)  // (the device is out: every pair goes through the host's sequential feeder; no walk waits for the stream)
      for (auto& pf : fills) pf->redo.store(true);
    windows++;
  }
  if (dev) {  // the tables of every (pair, block) back from the device; a pair with a flagged block is filled again on the host
    const double tf = now_s();
    hand_over(true);
    std::vector<double> dtab;
    std::vector<int> dflags;
    if (dev_failed || !dev->finish(dtab, dflags)) {
      std::cerr << "Note: age sampling on the GPU failed (" << dev->error() << "); filling the pairs on the host." << std::endl;
      for (auto& pf : fills) pf->redo.store(true);
    } else {
      for (auto& pfp : fills) {
        PairFill& pf = *pfp;
        if (pf.redo.load()) continue;
        for (int j = 0; j < pf.num_blocks && !pf.redo.load(); j++) {
          const size_t t = pf.slot * DevQueue::kMaxBlocks + (size_t)j;
          if (dflags[t]) pf.redo.store(true);
          else std::copy(dtab.begin() + t * 2 * A, dtab.begin() + (t + 1) * 2 * A, pf.blocks[(size_t)j]->t.begin());
        }
      }
    }
    dev_finish_s = now_s() - tf;
  }
  const double t2 = now_s();
  size_t redone = 0, used = 0;
  for (auto& pfp : fills) {
    PairFill& pf = *pfp;
    if (pf.redo.load()) {
      redone++;
      fill_sequentially(pf.index);
      continue;
    }
    PairTables& pt = out[pf.index];
    const int nb = pf.num_blocks;
    pt.nb = nb;
    pt.sh.resize((size_t)nb * A), pt.ns.resize(pt.sh.size()), pt.she.resize(pt.sh.size()), pt.nse.resize(pt.sh.size());
    for (int j = 0; j < nb; j++) {
      const double* t = pf.blocks[(size_t)j]->t.data();
      std::copy(t, t + A, pt.sh.begin() + (size_t)j * A);
      std::copy(t + A, t + 2 * A, pt.ns.begin() + (size_t)j * A);
      std::copy(t + 2 * A, t + 3 * A, pt.she.begin() + (size_t)j * A);
      std::copy(t + 3 * A, t + 4 * A, pt.nse.begin() + (size_t)j * A);
    }
    pt.rng = pf.rng_end;
    used += pf.used_snps;
  }
  if (g_times.on)
    std::cerr << "Timing: pairs front end on " << T << " threads: " << mut_files.size() << " .mut files (" << n_rows << " rows, "
              << n_kept << " kept) and " << tmp_files.size() << " .colate.in files (" << n_rec / 1000000 << " MB) read once in "
              << t1 - t0 << " s; " << todo.size() << " pairs filled in " << t2 - t1 << " s (" << used << " used SNPs, " << windows
              << " stream window(s) of " << window_mb << " MB, waited " << stream.waited() << " thread-s for uniforms (generated in " << stream.generate_seconds() << " s, converted in "
              << stream.convert_seconds() << " s), " << redone
              << " pair(s) redone sequentially in " << now_s() - t2 << " s); thread-seconds: .mut parse " << g_work.parse_mut.load()
              << ", .colate.in decode " << g_work.load_tmp.load() << ", walk indices " << g_work.index.load() << " (" << n_indexed << " of "
              << tmp_files.size() << " files)" << ", SNP walks " << g_work.walk.load() << ", age sampling "
              << g_work.sample.load()
              << (dev ? "; age sampling on the GPU: " + std::to_string(dev_jobs) + " (pair, block) jobs, " + std::to_string(dev_recs) + " SNPs in " + std::to_string(dev_launches) + " launches, " +
                            std::to_string(dev->gpu_seconds()) + " s of copies and kernels, " + std::to_string(dev_upload_s) + " s uploading the uniform stream, " + std::to_string(dev_make_s) + " s setting up, " + std::to_string(dev_staging_s) + " s page-locking the record buffers beside the first windows, " +
                            std::to_string(dev_finish_s) + " s for the last launch and the tables"
                      : "; age sampling on the host (" + dev_note + ")")
              << std::endl;
  g_times.parse_mut = t1 - t0;
  g_times.table_fill = now_s() - t1;
  return true;
}

}  // namespace

int fill_single_pair(const Options& opt, const std::string& target, const std::string& reference, int seed, int A,
                     std::vector<double>& sh, std::vector<double>& ns, std::vector<double>& she, std::vector<double>& nse,
                     std::mt19937& rng) {
  if (!bulk_stream_ok((unsigned)seed)) return -1;  // (fill_pairs would run the single-pair feeder itself: let the caller do it)
  std::vector<PairSpec> one(1);
  one[0].target = target, one[0].reference = reference;
  std::vector<PairTables> tabs;
  if (!fill_pairs(opt, one, {0}, seed, A, tabs)) return -1;
  PairTables& pt = tabs[0];
  sh = std::move(pt.sh), ns = std::move(pt.ns), she = std::move(pt.she), nse = std::move(pt.nse);
  rng = pt.rng;
  return pt.nb;
}

int run_mut_pairs(const Options& opt) {
  if (!opt.has("mut") || !opt.has("bins")) {
    std::cerr << "Error: --pairs needs --mut and --bins (and optionally --chr, --num_bootstraps, --seed)." << std::endl;
    return 1;
  }
  for (const char* o : {"target_mask", "reference_mask", "coal"})
    if (opt.has(o)) {  // per-sample masks / one warm start cannot apply to a whole list of pairs: refuse, do not ignore
      std::cerr << "Error: --" << o << " cannot be combined with --pairs (run such pairs one by one)." << std::endl;
      return 1;
    }
  std::vector<PairSpec> pairs;
  if (!read_pair_list(opt.get("pairs"), pairs)) return 1;
  const bool talk = g_rank.rank == 0;
  if (talk) {
    std::cerr << "---------------------------------------------------------" << std::endl;
    std::cerr << "Calculating coalescence rates for " << pairs.size() << " pairs of (ancient) samples.." << std::endl;
  }
  double years_per_gen = 28.0;
  if (opt.has("years_per_gen")) years_per_gen = std::stof(opt.get("years_per_gen"));
  std::vector<double> age_grid(256);
  const int A = colate_age_grid(age_grid.data(), 256);
  age_grid.resize(A);
  int seed = std::time(0) + getpid();
  if (opt.has("seed")) seed = std::stoi(opt.get("seed"));
  int B = 1;
  if (opt.has("num_bootstraps")) B = std::stoi(opt.get("num_bootstraps"));
  if (B < 1) {
    std::cerr << "Error: --num_bootstraps must be at least 1." << std::endl;
    return 1;
  }
  const size_t P = pairs.size();
  const bool counts_only = opt.has("counts_only");
  const bool want_counts = counts_only || opt.has("counts_out");

  // One process, one GPU: create the HIP context on a second thread while the inputs are read (as run_mut does)
  struct Warm {
    std::thread t;
    ~Warm() {
      if (t.joinable()) t.join();
    }
  } warm;
  if (!g_rank.ranked && !opt.has("devices") && !counts_only) {
    int warm_dev = 0;
    try {
      if (opt.has("device")) warm_dev = std::stoi(opt.get("device"));
    } catch (...) {
      warm_dev = 0;
    }
    warm.t = std::thread([warm_dev] { (void)colate_warm_up(warm_dev); });
  }

  // ---- epochs per pair (coal.cpp:3551-3632): they depend on the ages only, so the launches are known before any file is read
  std::vector<std::vector<double>> epochs(P);
  std::vector<int> ep_null(P, 0);
  std::vector<double> age(P);
  for (size_t p = 0; p < P; p++) {
    age[p] = std::max(pairs[p].target_age, pairs[p].ref_age) / years_per_gen;
    epochs[p].resize(COLATE_MAX_EPOCHS);
    const int E = colate_epochs_from_bins(opt.get("bins").c_str(), age[p], years_per_gen, epochs[p].data(), COLATE_MAX_EPOCHS, &ep_null[p]);
    if (E <= 0) {
      std::cerr << colate_last_error() << std::endl;
      return 1;
    }
    epochs[p].resize(E);
  }
  // classes of pairs with the same number of epochs, in order of first appearance: one launch each
  std::vector<std::vector<size_t>> classes;
  for (size_t p = 0; p < P; p++) {
    size_t c = 0;
    while (c < classes.size() && epochs[classes[c][0]].size() != epochs[p].size()) c++;
    if (c == classes.size()) classes.emplace_back();
    classes[c].push_back(p);
  }
  // --ranks N: this rank's rows [lo, hi) of every class (row = position in the class * B + replicate) and the pairs they belong to
  std::vector<size_t> todo;
  std::vector<int> first_group(classes.size(), 0), group_count(classes.size(), 0);
  for (size_t c = 0; c < classes.size(); c++) {
    const int R = (int)(classes[c].size() * (size_t)B);
    int lo = 0, hi = R;
    if (g_rank.ranked) colate_shard_bounds(R, g_rank.nranks, g_rank.rank, &lo, &hi);
    if (counts_only && g_rank.ranked && g_rank.rank != 0) lo = hi = 0;
    if (hi > lo) {
      first_group[c] = lo / B;
      group_count[c] = (hi - 1) / B - lo / B + 1;
      for (int g = 0; g < group_count[c]; g++) todo.push_back(classes[c][(size_t)(first_group[c] + g)]);
    }
  }
  std::sort(todo.begin(), todo.end());

  std::vector<PairTables> tabs;
  if (!fill_pairs(opt, pairs, todo, seed, A, tabs)) return 1;
  for (size_t p : todo) {
    if (talk) std::cerr << "Pair " << p + 1 << " / " << P << ": " << pairs[p].target << " x " << pairs[p].reference << ": Number of blocks: " << tabs[p].nb << std::endl;
    if (tabs[p].nb < 1) {
      std::cerr << "Error: no genome blocks were read for pair " << p + 1 << "." << std::endl;
      return 1;
    }
  }
  // ---- bootstrap weights from each pair's own generator (coal.cpp:3350-3357)
  std::vector<std::vector<double>> weights(P);
  for (size_t p : todo) {
    weights[p].resize((size_t)B * tabs[p].nb);
    if (int rc = colate_bootstrap_weights(&tabs[p].rng, B, tabs[p].nb, weights[p].data())) {
      std::cerr << "Error: " << colate_last_error() << " (" << rc << ")" << std::endl;
      return 1;
    }
  }
  if (counts_only) {  // no device: the weighted sums and the F redistribution on the host (coal.cpp:3358-3451)
    for (size_t p : todo) {
      std::vector<double> csh((size_t)B * A), cns((size_t)B * A);
      PairTables& pt = tabs[p];
      if (int rc = colate_bootstrap_counts_from_weights(B, pt.nb, A, age_grid.data(), age[p], weights[p].data(), pt.sh.data(),
                                                        pt.ns.data(), pt.she.data(), pt.nse.data(), csh.data(), cns.data())) {
        std::cerr << "Error: " << colate_last_error() << " (" << rc << ")" << std::endl;
        return 1;
      }
      write_counts_file(pairs[p].output + ".counts", B, A, age_grid, csh.data(), cns.data());
    }
    return 0;
  }

  if (talk) std::cerr << "Maximising likelihood using EM.. " << std::endl;
  if (opt.has("device") && !g_rank.ranked) {
    if (int rc = colate_set_device(std::stoi(opt.get("device")))) {
      std::cerr << "Error: " << colate_last_error() << " (" << rc << ")" << std::endl;
      return 1;
    }
  }
  std::vector<int> dev_list;
  if (opt.has("devices")) {
    const int nd = std::stoi(opt.get("devices"));
    if (nd < 1) {
      std::cerr << "Error: --devices must be at least 1." << std::endl;
      return 1;
    }
    for (int d = 0; d < nd; d++) dev_list.push_back(d);
  }
  void* comm = nullptr;
  if (g_rank.ranked) {
    const int ndev = colate_device_count();
    if (ndev < 1) {
      std::cerr << "Error: " << colate_last_error() << std::endl;
      return 1;
    }
    const int dev0 = opt.has("device") ? std::stoi(opt.get("device")) : 0;
    unsigned char id[COLATE_COMM_ID_BYTES];
    int rc = colate_set_device((dev0 + g_rank.rank) % ndev);
    if (!rc) {
      if (g_rank.rank == 0) {
        rc = colate_comm_unique_id(id);
        if (!write_all(g_rank.fd_id_out, id, rc ? 0 : sizeof(id)) && !rc) rc = COLATE_EIO;
        ::close(g_rank.fd_id_out);
      } else if (!read_all(g_rank.fd_id_in, id, sizeof(id))) {
        std::cerr << "Error: rank " << g_rank.rank << " did not receive the communicator id." << std::endl;
        return 1;
      }
    }
    if (!rc) rc = colate_comm_create(id, g_rank.nranks, g_rank.rank, &comm);
    if (rc) {
      std::cerr << "Error: " << colate_last_error() << " (" << rc << ")" << std::endl;
      return 1;
    }
  }
  const double t_em0 = now_s();
  int status = 0;
  for (size_t c = 0; c < classes.size() && status == 0; c++) {
    const std::vector<size_t>& cls = classes[c];
    const int G = (int)cls.size(), E = (int)epochs[cls[0]].size();
    const size_t R = (size_t)G * B;
    const int g0 = first_group[c], gn = group_count[c];
    // this process's groups of the class, concatenated
    std::vector<int> nb(gn);
    std::vector<double> g_age(gn), g_w, g_sh, g_ns, g_she, g_nse, g_ep((size_t)gn * E), g_init((size_t)gn * E, COLATE_DEFAULT_INIT_RATE);
    for (int g = 0; g < gn; g++) {
      const size_t p = cls[(size_t)(g0 + g)];
      const PairTables& pt = tabs[p];
      nb[g] = pt.nb, g_age[g] = age[p];
      g_w.insert(g_w.end(), weights[p].begin(), weights[p].end());
      g_sh.insert(g_sh.end(), pt.sh.begin(), pt.sh.end());
      g_ns.insert(g_ns.end(), pt.ns.begin(), pt.ns.end());
      g_she.insert(g_she.end(), pt.she.begin(), pt.she.end());
      g_nse.insert(g_nse.end(), pt.nse.begin(), pt.nse.end());
      std::copy(epochs[p].begin(), epochs[p].end(), g_ep.begin() + (size_t)g * E);
    }
    std::vector<double> rates(R * E), ll(R), csh, cns;
    std::vector<int> iters(R), flags(R);
    if (want_counts) csh.resize(R * A), cns.resize(R * A);
    int rc;
    if (g_rank.ranked) {
      rc = colate_bootstrap_em_batch_groups_allgather(comm, G, B, g0, gn, E, A, age_grid.data(), nb.data(), g_age.data(), g_w.data(),
                                                      g_sh.data(), g_ns.data(), g_she.data(), g_nse.data(), g_ep.data(), g_init.data(),
                                                      COLATE_DEFAULT_MAX_ITER, COLATE_DEFAULT_MIN_ITER, COLATE_DEFAULT_REL_TOL,
                                                      COLATE_DEFAULT_RATE_FLOOR, rates.data(), iters.data(), ll.data(), flags.data());
    } else if (!dev_list.empty()) {
      // --devices N (one process, several GPUs): counts on the host, the rows sharded over GPUs 0..N-1
      csh.resize(R * A), cns.resize(R * A);
      rc = 0;
      for (int g = 0, wo = 0, bo = 0; g < G && !rc; wo += B * nb[g], bo += nb[g], g++)
        rc = colate_bootstrap_counts_from_weights(B, nb[g], A, age_grid.data(), g_age[g], g_w.data() + wo, g_sh.data() + (size_t)bo * A,
                                                  g_ns.data() + (size_t)bo * A, g_she.data() + (size_t)bo * A, g_nse.data() + (size_t)bo * A,
                                                  csh.data() + (size_t)g * B * A, cns.data() + (size_t)g * B * A);
      std::vector<double> r_ep(R * E), r_init(R * E, COLATE_DEFAULT_INIT_RATE);
      for (size_t r = 0; r < R; r++) std::copy(g_ep.begin() + (r / B) * E, g_ep.begin() + (r / B + 1) * E, r_ep.begin() + r * E);
      if (!rc)
        rc = colate_em_batch_rows_sharded((int)dev_list.size(), dev_list.data(), (int)R, E, A, age_grid.data(), csh.data(), cns.data(),
                                          r_ep.data(), r_init.data(), COLATE_DEFAULT_MAX_ITER, COLATE_DEFAULT_MIN_ITER,
                                          COLATE_DEFAULT_REL_TOL, COLATE_DEFAULT_RATE_FLOOR, rates.data(), iters.data(), ll.data(),
                                          flags.data());
    } else {
      rc = colate_bootstrap_em_batch_groups(G, B, E, A, age_grid.data(), nb.data(), g_age.data(), g_w.data(), g_sh.data(), g_ns.data(),
                                            g_she.data(), g_nse.data(), g_ep.data(), g_init.data(), COLATE_DEFAULT_MAX_ITER,
                                            COLATE_DEFAULT_MIN_ITER, COLATE_DEFAULT_REL_TOL, COLATE_DEFAULT_RATE_FLOOR, rates.data(),
                                            iters.data(), ll.data(), flags.data(), want_counts ? csh.data() : nullptr,
                                            want_counts ? cns.data() : nullptr);
    }
    if (rc) {
      std::cerr << "Error: " << colate_last_error() << " (" << rc << ")" << std::endl;
      status = 1;
      break;
    }
    if (!talk) continue;
    for (int g = 0; g < G; g++) {
      const size_t p = cls[(size_t)g];
      int unresolved_max = 0;
      for (int i = 0; i < B; i++) {
        const size_t r = (size_t)g * B + i;
        std::cerr << "Pair " << p + 1 << " Bootstrap " << i + 1 << ": Total iterations " << iters[r] << std::endl;
        if (flags[r] & (COLATE_FLAG_NAN | COLATE_FLAG_NEG))
          std::cerr << "Warning: pair " << p + 1 << " bootstrap " << i + 1
                    << " produced NaN or negative sufficient statistics (the reference aborts here)." << std::endl;
        unresolved_max = std::max(unresolved_max, COLATE_UNRESOLVED_EPOCHS(flags[r]));
      }
      if (unresolved_max > 0)
        std::cerr << "Note: pair " << p + 1 << ": the last " << unresolved_max << " of " << E
                  << " epochs are older than the data resolve (include/colate_amd.h, COLATE_FLAG_UNRESOLVED)." << std::endl;
      if (want_counts && !csh.empty())
        write_counts_file(pairs[p].output + ".counts", B, A, age_grid, csh.data() + (size_t)g * B * A, cns.data() + (size_t)g * B * A);
      if (colate_write_coal((pairs[p].output + ".coal").c_str(), B, E, epochs[p].data(), rates.data() + (size_t)g * B * E,
                            age[p] > 0.0 ? 1 : 0, ep_null[p])) {
        std::cerr << "Error: " << colate_last_error() << std::endl;
        status = 1;
        break;
      }
    }
  }
  colate_comm_destroy(comm);
  g_times.bootstrap_em = now_s() - t_em0;
  if (g_times.on)
    std::cerr << "Timing: inputs " << g_times.parse_mut << " s, pairs' table fill " << g_times.table_fill << " s, bootstrap_em "
              << g_times.bootstrap_em << " s" << std::endl;
  if (status || !talk) return status;
  print_usage_footer();
  return 0;
}

}  // namespace colate_drv
