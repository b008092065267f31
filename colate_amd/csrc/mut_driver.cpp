// colate_amd/csrc/mut_driver.cpp -- `Colate --mode mut` as a library call
// (colate_mut_main in include/colate_amd.h): the command line of the reference
// (include/coal/Colate.cpp:6-116) for the .colate.in / .colate_mat inputs, the
// feeder that turns two .colate.in streams and the .mut files into per-block
// age-bin tables (include/coal/coal.cpp:2071-2321 with include/src/mutations.cpp:56-283
// and include/src/data.cpp:213-235), and the mut() driver (include/coal/coal.cpp:3071-3863)
// around the GPU EM (colate_em_batch).
//
// Everything here runs once per invocation on the host; the per-replicate EM,
// which is where the reference spends its time, is the HIP kernel.
#include <fcntl.h>
#include <poll.h>
#include <sys/resource.h>
#include <sys/wait.h>
#include <unistd.h>
#include <zlib.h>

#include <signal.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <functional>
#include <iomanip>
#include <iostream>
#include <map>
#include <random>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "colate_amd.h"
#include "colate_internal.h"
#include "mut_feeder.h"

namespace colate_drv {

// ------------------------------------------------------------------ options
// Same option names as Colate.cpp:11-45 (unknown options are an error there too:
// cxxopts throws option_not_exists_exception).  `--num_bootstrap` (README spelling)
// is accepted as an alias of `--num_bootstraps`.  Ours: `--device N` (GPU ordinal), `--devices N`
// (shard the replicates over GPUs 0..N-1 of the node from this one process), `--ranks N` (the same sharding as N
// processes, one per GPU, with one RCCL all-gather of the results: run_ranked below),
// `--counts_out FILE` (write the bootstrap count tables in the reference's .colate_mat layout,
// 17 significant digits), `--counts_only` (stop after that; needs no GPU) and `--write_colate_mat` (write
// <output>.colate_mat exactly as the reference does for BCF/BAM inputs, coal.cpp:3336-3343, 3453-3470).

const char* const kValueOptions[] = {
    "mode", "anc", "mut", "target_bcf", "reference_bcf", "target_mask", "reference_mask",
    "target_table", "target_bam", "reference_bam", "target_tmp", "reference_tmp", "target_age",
    "reference_age", "ref_genome", "anc_genome", "mask", "mask_cutoff", "chr", "bins",
    "lineage_bin", "outgroup_tmrca", "years_per_gen", "coal", "seed", "num_bootstraps", "filters",
    "groups", "poplabels", "map", "input", "output", "device", "devices", "ranks", "counts_out", "pairs"};
const char* const kBoolOptions[] = {"help", "strandfilter", "counts_only", "write_colate_mat"};

bool parse_options(int argc, char** argv, Options& o, std::string& err) {
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    std::string name, value;
    bool have_value = false;
    if (a.rfind("--", 0) == 0) {
      name = a.substr(2);
      size_t eq = name.find('=');
      if (eq != std::string::npos) {
        value = name.substr(eq + 1);
        name = name.substr(0, eq);
        have_value = true;
      }
    } else if (a == "-i") {
      name = "input";
    } else if (a == "-o") {
      name = "output";
    } else {
      err = "Unexpected argument '" + a + "'";
      return false;
    }
    if (name == "num_bootstrap") name = "num_bootstraps";
    bool is_bool = false, known = false;
    for (const char* b : kBoolOptions)
      if (name == b) is_bool = known = true;
    for (const char* v : kValueOptions)
      if (name == v) known = true;
    if (!known) {
      err = "Option '" + name + "' does not exist";
      return false;
    }
    if (is_bool) {
      o.kv[name] = "true";
      continue;
    }
    if (!have_value) {
      if (i + 1 >= argc) {
        err = "Option '" + name + "' is missing an argument";
        return false;
      }
      value = argv[++i];
    }
    o.kv[name] = value;
  }
  return true;
}

void print_help() {
  std::cout << "Usage:\n  Colate [OPTION...]\n\n"
            << "      --help                 Print help.\n"
            << "      --mode arg             Choose which part of the algorithm to run (colate_amd: mut).\n"
            << "      --mut arg              Filename of file containing mut.\n"
            << "      --target_tmp arg       Filename of target tmp file\n"
            << "      --reference_tmp arg    Filename of reference tmp file\n"
            << "      --target_mask arg      Fasta file containing target mask\n"
            << "      --reference_mask arg   Fasta file containing reference mask\n"
            << "      --target_age arg       Target age in years\n"
            << "      --reference_age arg    Reference age in years\n"
            << "      --chr arg              Optional: File specifying chromosomes to use.\n"
            << "      --bins arg             Optional: Epoch boundaries 10^(seq(x,y,stepsize)) [format: x,y,stepsize]. In years.\n"
            << "      --years_per_gen arg    Optional: Years per generation.\n"
            << "      --coal arg             Filename of file containing coalescence rates.\n"
            << "      --seed arg             Optional: Seed for random number generator (int)\n"
            << "      --num_bootstraps arg   Optional: Number of bootstraps.\n"
            << "      --device arg           Optional (colate_amd): GPU ordinal, default 0.\n"
            << "      --devices arg          Optional (colate_amd): shard the bootstrap replicates over GPUs 0..N-1 (one process).\n"
            << "      --ranks arg            Optional (colate_amd): the same as N processes, one per GPU, one RCCL all-gather.\n"
            << "      --pairs arg            Optional (colate_amd): file of `target_tmp reference_tmp output [target_age reference_age]`\n"
            << "                             lines; all pairs share --mut/--chr/--bins/--num_bootstraps/--seed, each .mut is parsed\n"
            << "                             once and all replicates of all pairs run in one GPU launch.\n"
            << "      --counts_out arg       Optional (colate_amd): write the bootstrap count tables (.colate_mat layout).\n"
            << "      --counts_only          Optional (colate_amd): stop after --counts_out (no GPU needed).\n"
            << "      --write_colate_mat     Optional (colate_amd): write <output>.colate_mat as the reference does for BCF/BAM inputs.\n"
            << "      --target_table arg     (--mode make_tmp) Table `chr bp allele` of the target's calls.\n"
            << "      --ref_genome arg       (--mode make_tmp) Reference genome fasta (per chromosome with --chr).\n"
            << "  -o, --output arg           Filename of output.\n"
            << std::endl;
}

// ------------------------------------------------------------------ stage timing (COLATE_TIMING=1: one stderr line at the end)
StageTimes g_times;

// ------------------------------------------------------------------ gz text
// igzstream semantics of the reference: zlib reads gzip and plain files alike.
class GzText {
 public:
  bool open(const std::string& name) {
    close();
    f_ = gzopen(name.c_str(), "rb");
    if (f_) gzbuffer(f_, 1 << 20);
    return f_ != nullptr;
  }
  bool is_open() const { return f_ != nullptr; }
  bool getline(std::string& line) {
    line.clear();
    if (!f_) return false;
    char buf[1 << 14];
    bool got = false;
    while (gzgets(f_, buf, sizeof(buf))) {
      got = true;
      size_t n = std::strlen(buf);
      if (n && buf[n - 1] == '\n') {
        line.append(buf, n - 1);
        return true;
      }
      line.append(buf, n);
    }
    return got;
  }
  void close() {
    if (f_) gzclose(f_);
    f_ = nullptr;
  }
  ~GzText() { close(); }

 private:
  gzFile f_ = nullptr;
};

// ------------------------------------------------------------------ .mut rows

// (the readers run on their own threads: leave without running the static destructors under the other threads' feet)
[[noreturn]] void reader_exit() {
  std::cerr.flush();
  std::cout.flush();
  std::fflush(nullptr);
  std::_Exit(1);
}
[[noreturn]] void mut_line_error(const std::string& line) {
  std::cerr << "Error reading following line in mut file:" << std::endl;
  std::cerr << line << std::endl;
  reader_exit();
}

// std::stoi on the text at p (leading white space, sign, digits; what follows the digits is ignored), without the copy
// and the exceptions: false where std::stoi would throw (no digits, or out of int range)
inline bool parse_stoi(const char* p, const char* end, int& out, const char** after = nullptr) {
  while (p < end && (*p == ' ' || (*p >= '\t' && *p <= '\r'))) p++;
  bool neg = false;
  if (p < end && (*p == '+' || *p == '-')) neg = (*p++ == '-');
  if (p >= end || *p < '0' || *p > '9') return false;
  long long v = 0;
  while (p < end && *p >= '0' && *p <= '9') {
    v = v * 10 + (*p++ - '0');
    if (v > 2147483648LL) return false;
  }
  if (neg) v = -v;
  if (v > 2147483647LL || v < -2147483648LL) return false;
  out = (int)v;
  if (after) *after = p;
  return true;
}
// std::stof: strtof (the field ends at a ';' or at the terminating NUL of the line buffer, where strtof stops by itself)
inline bool parse_stof(const char* p, float& out) {
  char* e = nullptr;
  errno = 0;
  const float v = std::strtof(p, &e);
  if (e == p || errno == ERANGE) return false;
  out = v;
  return true;
}

// One row from the NUL-terminated line [b, e): the fields parse_tmptmp looks at (mutations.cpp:77-246)
inline bool parse_mut_line(char* b, char* e, MutRow& r) {
  // the first ten ';' of the line (snp;pos;dist;rs;tree;branches;is_not_mapping;is_flipped;age_begin;age_end;<rest>)
  char* sep[11];
  int ns = 0;
  for (char* q = b; q < e && ns < 11; q++)
    if (*q == ';') sep[ns++] = q;
  if (ns < 10) return false;  // needs 10 separators
  int tmp;
  if (!parse_stoi(b, sep[0], tmp)) return false;
  if (!parse_stoi(sep[0] + 1, sep[1], r.pos)) return false;
  if (!parse_stoi(sep[1] + 1, sep[2], tmp)) return false;
  if (!parse_stoi(sep[3] + 1, sep[4], tmp)) return false;
  r.num_branches = 0;
  for (const char* q = sep[4] + 1; q < sep[5];) {  // white-space separated branch indices, each through stoi
    while (q < sep[5] && (*q == ' ' || (*q >= '\t' && *q <= '\r'))) q++;
    if (q >= sep[5]) break;
    const char* tok_end = q;
    while (tok_end < sep[5] && !(*tok_end == ' ' || (*tok_end >= '\t' && *tok_end <= '\r'))) tok_end++;
    if (!parse_stoi(q, tok_end, tmp)) return false;
    r.num_branches++;
    q = tok_end;
  }
  if (!parse_stoi(sep[6] + 1, sep[7], r.flipped)) return false;
  *sep[8] = 0;  // (strtof must not read past its field: "1e5;2" is fine, but keep it strict)
  const bool ok1 = parse_stof(sep[7] + 1, r.age_begin);
  *sep[8] = ';';
  *sep[9] = 0;
  const bool ok2 = parse_stof(sep[8] + 1, r.age_end);
  *sep[9] = ';';
  if (!ok1 || !ok2) return false;
  // field 10: up to the next ';' or the end of the line; "NA" is kept when it is empty and the last field
  char* f10_end = (ns >= 11) ? sep[10] : e;
  if (f10_end > sep[9] + 1 || ns >= 11)
    r.mutation_type.assign(sep[9] + 1, f10_end);
  else
    r.mutation_type = "NA";
  return true;
}

bool for_each_mut_row(const std::string& filename, const std::function<void(const MutRow&)>& sink) {
  gzFile f = gzopen(filename.c_str(), "rb");
  if (!f) f = gzopen((filename + ".gz").c_str(), "rb");
  if (!f) {
    std::cerr << "Error while reading " << filename << "(.gz)." << std::endl;
    reader_exit();  // mutations.cpp:265-268: exit(1)
  }
  gzbuffer(f, 1 << 20);
  MutRow row;
  // inflate in 4 MB pieces and cut lines in place (no per-line std::string, no per-field copies)
  std::vector<char> buf((4u << 20) + 1);
  size_t have = 0;
  bool header_done = false, eof = false;
  while (!eof) {
    if (have == buf.size() - 1) buf.resize(buf.size() * 2);  // a line longer than the buffer
    const int got = gzread(f, buf.data() + have, (unsigned)(buf.size() - 1 - have));
    if (got <= 0) eof = true;
    else have += (size_t)got;
    char* b = buf.data();
    char* const end = b + have;
    for (;;) {
      char* nl = static_cast<char*>(std::memchr(b, '\n', (size_t)(end - b)));
      if (!nl) {
        if (!eof || b == end) break;
        nl = end;  // last line without a newline
      }
      *nl = 0;
      if (!header_done) {
        header_done = true;
      } else {
        row.mutation_type = "NA";
        if (!parse_mut_line(b, nl, row)) mut_line_error(std::string(b, nl));
        sink(row);
      }
      b = (nl < end) ? nl + 1 : end;
      if (b >= end) break;
    }
    have = (size_t)(end - b);
    if (have) std::memmove(buf.data(), b, have);
  }
  gzclose(f);
  return true;
}

bool read_mut_file(const std::string& filename, std::vector<MutRow>& rows) {
  rows.clear();
  return for_each_mut_row(filename, [&rows](const MutRow& r) { rows.push_back(r); });
}

// Reader thread: inflates and tokenises the .mut files in order, at most two chromosomes ahead of the table fill.  The
// fill itself has to stay sequential -- every sampled age is a draw from the run's one std::mt19937 (coal.cpp:2262, 2282),
// so the order of the draws is part of the result -- but nothing of the parsing depends on it.
class MutPrefetcher {
 public:
  explicit MutPrefetcher(std::vector<std::string> files) : files_(std::move(files)), slots_(files_.size()) {
    const unsigned hc = std::thread::hardware_concurrency();
    int nthreads = (hc >= 6 && files_.size() > 1) ? (hc >= 12 ? 4 : 2) : 1;
    // COLATE_THREADS=n: at most n threads of this process work on the inputs at a time, the caller's included --
    // n <= 1: no thread is started at all (files are parsed by next() itself); n >= 2: up to min(4, n - 1) readers here,
    // the rest of the n - 1 go to the age sampling (sample_threads(): readers and samplers overlap only while the
    // fill waits for a file)
    if (const char* e = std::getenv("COLATE_THREADS")) nthreads = std::atoi(e) <= 1 ? 0 : std::min(nthreads, std::atoi(e) - 1);
    for (int t = 0; t < nthreads; t++)
      workers_.emplace_back([this] {
        for (;;) {
          size_t i;
          {
            std::unique_lock<std::mutex> lk(m_);
            // at most kAhead files parsed beyond the one the fill is at (memory: ~56 MB per million rows)
            cv_.wait(lk, [this] { return stop_ || next_ >= files_.size() || next_ < consumed_ + kAhead; });
            if (stop_ || next_ >= files_.size()) return;
            i = next_++;
          }
          std::vector<MutRow> rows;
          const double t0 = StageTimes::now();
          read_mut_file(files_[i], rows);
          const double dt = StageTimes::now() - t0;
          std::lock_guard<std::mutex> lk(m_);
          parse_seconds_ += dt;
          slots_[i].rows = std::move(rows);
          slots_[i].ready = true;
          cv_.notify_all();
        }
      });
  }
  ~MutPrefetcher() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (std::thread& t : workers_)
      if (t.joinable()) t.join();
  }
  void next(std::vector<MutRow>& rows) {  // the next file's rows, in the order given
    const double t0 = StageTimes::now();
    if (workers_.empty()) {  // COLATE_THREADS <= 1: on the calling thread
      read_mut_file(files_[consumed_++], rows);
      g_times.parse_mut += StageTimes::now() - t0;
      return;
    }
    std::unique_lock<std::mutex> lk(m_);
    const size_t i = consumed_;
    cv_.wait(lk, [&] { return slots_[i].ready; });
    rows = std::move(slots_[i].rows);
    consumed_++;
    g_times.wait_for_parser += StageTimes::now() - t0;
    g_times.parse_mut = parse_seconds_;
    cv_.notify_all();
  }

 private:
  static constexpr size_t kAhead = 5;
  struct Slot {
    std::vector<MutRow> rows;
    bool ready = false;
  };
  std::vector<std::string> files_;
  std::vector<Slot> slots_;
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_;
  size_t next_ = 0, consumed_ = 0;
  double parse_seconds_ = 0;
  bool stop_ = false;
};

// data.cpp:213-235: sequence = upper-cased lines after the header, concatenated
void read_fasta_mask(const std::string& filename, std::string& seq) {
  GzText is;
  if (!is.open(filename) && !is.open(filename + ".gz")) {
    std::cerr << "Error while opening file " << filename << "." << std::endl;
    std::exit(1);
  }
  std::string line;
  is.getline(line);
  seq.clear();
  while (is.getline(line)) {
    for (char& c : line) c = (char)std::toupper((unsigned char)c);
    seq += line;
  }
}

// ------------------------------------------------------------------ .colate.in
// Record (little-endian, no header), coal.cpp:2505-2514 / 2126-2133:
//   int32 lchrom; char chrom[lchrom]; int32 bp; char anc; char der; int32 AAF; int32 DAF
struct TmpStream {
  FILE* fp = nullptr;
  std::string chrom;  // name of the record last read ("" before any read)
  int bp = 0;
  char anc = 0, der = 0;
  int AAF = 0, DAF = 0;  // the reference resets these two between SNPs (coal.cpp:2182-2183)
  // returns false at end of file, leaving every field as it was (coal.cpp:2126 `break`).  The seven fread calls of the
  // reference per record, served from a buffer of our own (40 M records x 7 library calls were 3 s of the table fill); a
  // field cut short by the end of the file keeps the bytes that were there, as with fread.
  bool next() {
    int lchrom = 0;
    if (!fp || get(&lchrom, sizeof(int)) != sizeof(int)) return false;
    char buf[1024];
    if (lchrom < 0 || lchrom > 1023) lchrom = 0;
    get(buf, (size_t)lchrom);
    chrom.assign(buf, (size_t)lchrom);
    get(&bp, sizeof(int));
    get(&anc, 1);
    get(&der, 1);
    get(&AAF, sizeof(int));
    get(&DAF, sizeof(int));
    return true;
  }

 private:
  size_t get(void* dst, size_t n) {
    if (end_ - pos_ < n) refill();
    const size_t k = std::min(n, end_ - pos_);
    std::memcpy(dst, buf_.data() + pos_, k);
    pos_ += k;
    return k;
  }
  void refill() {
    if (buf_.empty()) buf_.resize(1u << 20);
    const size_t keep = end_ - pos_;
    if (keep) std::memmove(buf_.data(), buf_.data() + pos_, keep);
    pos_ = 0, end_ = keep;
    if (!eof_) {
      const size_t got = std::fread(buf_.data() + end_, 1, buf_.size() - end_, fp);
      end_ += got;
      if (got == 0) eof_ = true;
    }
  }
  std::vector<char> buf_;
  size_t pos_ = 0, end_ = 0;
  bool eof_ = false;
};



// ---- sampling of the mutation ages, off the main thread ---------------------------------------------------------------
// Every used SNP spreads its weight over 100 ages drawn uniformly between age_begin and age_end (coal.cpp:2260-2295), each
// draw one std::uniform_real_distribution<double>(0,1) call on the run's single std::mt19937: the ORDER of the draws is part
// of the result, their evaluation is not.  The main thread therefore walks the SNPs (filters, stream merges: sequential by
// nature), draws the uniforms in the reference's order into a buffer per genome block, and hands the block to a worker, which
// does the log / round / accumulate per draw (two thirds of the time of the whole table fill).  A genome block's tables are
// touched by one job only and the SNPs of a job are taken in order, so every table cell sums the same terms in the same
// order as the reference: bit-identical.  One case breaks the fixed "100 draws per SNP": the reference REdraws a sample whose
// age bin lies beyond the grid (coal.cpp:2286-2287, ages above 9e6 generations); a worker that meets one raises `redo` and
// the whole fill is repeated on the sequential path.
struct UsedSnp {
  double age_begin, age_end, w_sh, w_ns;
  bool emp;  // age_begin <= sample age: the F path (coal.cpp:2245-2275), not-shared weight only, no redraws
};
struct SampleJob {
  std::vector<UsedSnp> snps;
  std::vector<double> u;  // 100 uniforms per SNP, in draw order
  double *sh = nullptr, *ns = nullptr;  // the block's two tables (buffers of tab.sh[blk], tab.ns[blk]: stable while blocks are added)
};
class SamplePool {
 public:
  SamplePool(int nthreads, double C, int A, double age) : C_(C), A_(A), age_(age) {
    for (int i = 0; i < nthreads; i++) workers_.emplace_back([this] { run(); });
  }
  ~SamplePool() { finish(); }
  double waited_ = 0;  // seconds submit() waited for room
  void submit(SampleJob&& j) {
    const double t0 = StageTimes::now();
    std::unique_lock<std::mutex> lk(m_);
    // bounds the uniforms held in memory: a job is a whole genome block (800 bytes of uniforms per used SNP, 80 MB for a
    // dense 100k-SNP block), so the queue is limited by its bytes -- kQueueBytes, or one job whatever its size -- as
    // well as by its length
    const size_t bytes = j.u.size() * sizeof(double);
    cv_room_.wait(lk, [&] { return q_.empty() || (q_.size() < 2 * workers_.size() + 2 && queued_bytes_ + bytes <= kQueueBytes); });
    waited_ += StageTimes::now() - t0;
    queued_bytes_ += bytes;
    q_.push_back(std::move(j));
    cv_work_.notify_one();
  }
  void finish() {  // waits for all jobs
    {
      std::lock_guard<std::mutex> lk(m_);
      done_ = true;
    }
    cv_work_.notify_all();
    for (std::thread& t : workers_)
      if (t.joinable()) t.join();
    workers_.clear();
  }
  bool redo() const { return redo_.load(); }

 private:
  void run() {
    for (;;) {
      SampleJob j;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_work_.wait(lk, [this] { return !q_.empty() || done_; });
        if (q_.empty()) return;
        j = std::move(q_.front());
        q_.pop_front();
        queued_bytes_ -= j.u.size() * sizeof(double);
        cv_room_.notify_one();
      }
      if (redo_.load()) continue;
      double *sh = j.sh, *ns = j.ns;
      const double* u = j.u.data();
      for (const UsedSnp& s : j.snps) {
        const double span = s.age_end - s.age_begin;
        if (s.emp) {
          for (int k = 0; k < 100; k++) {
            double sampled_age = u[k] * span + s.age_begin;
            if (sampled_age < age_) sampled_age = age_;
            const int bin = age_bin_index(sampled_age, C_);
            if (bin < A_) ns[bin] += s.w_ns;
          }
        } else {
          for (int k = 0; k < 100; k++) {
            const double sampled_age = u[k] * span + s.age_begin;
            const int bin = age_bin_index(sampled_age, C_);
            if (sampled_age < age_ || bin >= A_) {  // the reference would draw again: the stream no longer lines up
              redo_.store(true);
              break;
            }
            sh[bin] += s.w_sh;
            ns[bin] += s.w_ns;
          }
          if (redo_.load()) break;
        }
        u += 100;
      }
    }
  }
  const double C_;
  const int A_;
  const double age_;
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_work_, cv_room_;
  std::deque<SampleJob> q_;
  static constexpr size_t kQueueBytes = size_t(256) << 20;
  size_t queued_bytes_ = 0;
  bool done_ = false;
  std::atomic<bool> redo_{false};
};
// std::uniform_real_distribution<double>(0, 1) on std::mt19937 is std::generate_canonical<double, 53>: two 32-bit draws,
// (r1 + r2 * 2^32) / 2^64 in double, capped below 1 (libstdc++ bits/random.tcc).  The same arithmetic spelled out costs a
// third (the library version goes through long double); it is used only after a self-check against the library's own
// distribution on this machine's libstdc++ (the sequence is part of the result), else the library call is.
inline double canonical_fast(std::mt19937& g) {
  const double r1 = (double)g();
  const double r2 = (double)g();
  double ret = (r1 + r2 * 4294967296.0) * 0x1p-64;
  if (ret >= 1.0) ret = std::nextafter(1.0, 0.0);
  return ret;
}
inline bool canonical_fast_ok() {
  static const bool ok = [] {
    for (unsigned seed : {1u, 12345u, 4294967295u}) {
      std::mt19937 a(seed), b(seed);
      std::uniform_real_distribution<double> d(0, 1);
      for (int i = 0; i < 4096; i++)
        if (d(a) != canonical_fast(b)) return false;
      if (a != b) return false;
    }
    return true;
  }();
  return ok;
}


// The uniforms themselves, on a thread of their own: the stream does not depend on the data, only HOW MANY of its values the
// fill takes does.  The thread runs ahead on a copy of the run's generator, filling chunks of kChunk doubles, and keeps the
// generator state at the start of every chunk; when the fill is over, the run's generator is set to the state at the start of
// the last chunk touched and advanced by the few draws taken from it -- exactly where the sequential code would have left it.
class UniformStream {
 public:
  static constexpr size_t kChunk = 1u << 18;  // doubles per chunk (2 MB)
  UniformStream(const std::mt19937& start, bool fast) : gen_(start), first_(start) {
    // the bulk generator only if it reproduces this machine's library on the run's own state
    bulk_ok_ = fast && bulk_.load(start);
    if (bulk_ok_) {
      BulkMt19937 probe = bulk_;
      std::mt19937 lib = start;
      std::uniform_real_distribution<double> d(0, 1);
      uint32_t w[2 * 1300];
      probe.generate(w, 2 * 1300);  // (across two regenerations of the state)
      for (int i = 0; i < 1300 && bulk_ok_; i++) bulk_ok_ = (d(lib) == canonical_from(w[2 * i], w[2 * i + 1]));
      std::mt19937 back;
      bulk_ok_ = bulk_ok_ && probe.store(back) && back == lib;
    }
    fast_ = fast;
    worker_ = std::thread([this] { run(); });
  }
  ~UniformStream() { stop(); }
  void take(double* out, size_t n) {  // the next n uniforms of the stream
    while (n) {
      if (!cur_ || pos_ == kChunk) next_chunk();
      const size_t k = std::min(n, kChunk - pos_);
      std::memcpy(out, cur_->u.data() + pos_, k * sizeof(double));
      out += k, n -= k, pos_ += k;
    }
  }
  // the generator as the sequential code would hold it now (after every uniform handed out so far)
  std::mt19937 state_after_taken() {
    stop();
    if (!cur_) return first_;
    if (bulk_ok_) {
      BulkMt19937 b = cur_->bulk_at_start;
      b.discard(2 * (unsigned long long)pos_);  // two 32-bit draws per uniform (generate_canonical<double, 53>)
      std::mt19937 g;
      if (b.store(g)) return g;
    }
    std::mt19937 g = cur_->at_start;
    g.discard(2 * (unsigned long long)pos_);
    return g;
  }

 private:
  static double canonical_from(uint32_t r1, uint32_t r2) {
    double ret = ((double)r1 + (double)r2 * 4294967296.0) * 0x1p-64;
    if (ret >= 1.0) ret = std::nextafter(1.0, 0.0);
    return ret;
  }
  struct Chunk {
    std::mt19937 at_start;
    BulkMt19937 bulk_at_start;
    std::vector<double> u;
  };
  void run() {
    std::uniform_real_distribution<double> d(0, 1);
    std::vector<uint32_t> words(bulk_ok_ ? 2 * kChunk : 0);
    for (;;) {
      std::unique_ptr<Chunk> c(new Chunk);
      c->u.resize(kChunk);
      if (bulk_ok_) {
        c->bulk_at_start = bulk_;
        bulk_.generate(words.data(), 2 * kChunk);
        for (size_t i = 0; i < kChunk; i++) c->u[i] = canonical_from(words[2 * i], words[2 * i + 1]);
      } else {
        c->at_start = gen_;
        if (fast_)
          for (size_t i = 0; i < kChunk; i++) c->u[i] = canonical_fast(gen_);
        else
          for (size_t i = 0; i < kChunk; i++) c->u[i] = d(gen_);
      }
      std::unique_lock<std::mutex> lk(m_);
      cv_room_.wait(lk, [this] { return ready_.size() < 8 || stop_; });
      if (stop_) return;
      ready_.push_back(std::move(c));
      cv_ready_.notify_one();
    }
  }
  void next_chunk() {
    const double t0 = StageTimes::now();
    std::unique_lock<std::mutex> lk(m_);
    cv_ready_.wait(lk, [this] { return !ready_.empty(); });
    waited_ += StageTimes::now() - t0;
    cur_ = std::move(ready_.front());
    ready_.pop_front();
    pos_ = 0;
    cv_room_.notify_one();
  }
  void stop() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_room_.notify_all();
    if (worker_.joinable()) worker_.join();
  }
 public:
  double waited_ = 0;  // seconds the fill waited for uniforms
  bool bulk() const { return bulk_ok_; }

 private:
  std::mt19937 gen_;
  const std::mt19937 first_;
  BulkMt19937 bulk_;
  bool bulk_ok_ = false, fast_ = false;
  std::thread worker_;
  std::mutex m_;
  std::condition_variable cv_ready_, cv_room_;
  std::deque<std::unique_ptr<Chunk>> ready_;
  std::unique_ptr<Chunk> cur_;
  size_t pos_ = 0;
  bool stop_ = false;
};

inline int sample_threads() {
  if (const char* e = std::getenv("COLATE_THREADS")) return std::max(0, std::atoi(e) - 1);
  const unsigned hc = std::thread::hardware_concurrency();
  return hc >= 4 ? (int)std::min(hc - 2, 12u) : 0;  // 0: sample on the main thread (the sequential path)
}

// coal.cpp:2071-2321.  Returns the number of blocks.
int fill_tables_impl(const std::vector<std::string>& chr_names,
                     const std::vector<std::string>& mut_files, const std::string& target_file,
                     const std::string& ref_file, const std::vector<std::string>& target_masks,
                     const std::vector<std::string>& ref_masks, double C, std::mt19937& rng,
                     int num_bases_per_block, int A, BlockTables& tab,
                     std::map<std::string, std::vector<MutRow>>* mut_cache, SamplePool* pool) {
  const double age = 0, ref_age = 0;  // forced, coal.cpp:2074-2075
  std::uniform_real_distribution<double> dist_unif(0, 1);
  const float num_samples = 100;
  TmpStream tgt, ref;
  tgt.fp = std::fopen(target_file.c_str(), "rb");
  ref.fp = std::fopen(ref_file.c_str(), "rb");
  if (!tgt.fp) std::cerr << "Failed to open " << target_file << std::endl;
  if (!ref.fp) std::cerr << "Failed to open " << ref_file << std::endl;
  const bool has_tar_mask = !target_masks.empty(), has_ref_mask = !ref_masks.empty();

  int num_blocks = 0;
  size_t blk = 0;
  tab.add_block(A);
  SampleJob job;  // (pool) the used SNPs of the current block and their uniforms
  auto flush_job = [&]() {
    if (pool && !job.snps.empty()) {
      job.sh = tab.sh[blk].data();
      job.ns = tab.ns[blk].data();
      pool->submit(std::move(job));
      job = SampleJob();
    }
  };
  std::unique_ptr<UniformStream> stream;  // (pool) the run's uniforms, generated ahead on their own thread
  if (pool) stream.reset(new UniformStream(rng, canonical_fast_ok()));
  auto draw100 = [&]() {  // the SNP's 100 uniforms, in the reference's draw order
    const size_t at = job.u.size();
    job.u.resize(at + 100);
    stream->take(job.u.data() + at, 100);
  };
  auto advance_block = [&]() {
    flush_job();
    blk++;
    num_blocks++;
    if (blk >= tab.sh.size()) tab.add_block(A);
  };

  std::vector<MutRow> rows_local;
  std::string tar_mask, ref_mask;
  std::unique_ptr<MutPrefetcher> prefetch;
  if (!mut_cache) prefetch.reset(new MutPrefetcher(mut_files));
  for (size_t chr = 0; chr < mut_files.size(); chr++) {
    std::cerr << "parsing CHR: " << chr + 1 << " / " << mut_files.size() << std::endl;
    // --pairs: every (target, reference) pair walks the same .mut rows; parse each file once
    const std::vector<MutRow>* rows_p = &rows_local;
    if (mut_cache) {
      auto it = mut_cache->find(mut_files[chr]);
      if (it == mut_cache->end()) {
        read_mut_file(mut_files[chr], (*mut_cache)[mut_files[chr]]);
        it = mut_cache->find(mut_files[chr]);
      }
      rows_p = &it->second;
    } else {
      prefetch->next(rows_local);
    }
    const std::vector<MutRow>& rows = *rows_p;
    const double t_fill0 = StageTimes::now();
    if (has_tar_mask) read_fasta_mask(target_masks[chr], tar_mask);
    if (has_ref_mask) read_fasta_mask(ref_masks[chr], ref_mask);
    int current_block_base = 0;
    const std::string& name = chr_names[chr];
    while (ref.chrom != name) {  // skip to this chromosome, coal.cpp:2125-2134
      if (!ref.next()) break;
    }
    while (tgt.chrom != name) {
      if (!tgt.next()) break;
    }
    for (const MutRow& m : rows) {
      if (!(m.flipped == 0 && m.num_branches == 1 && m.age_begin < m.age_end && m.age_end >= age))
        continue;
      // "anc/der" (mutations.cpp:236-246 splits at the first '/'): both sides as views into the row's string
      const std::string& mt = m.mutation_type;
      const size_t slash = mt.find('/');
      const size_t anc_len = slash == std::string::npos ? mt.size() : slash;
      const char* const anc_p = mt.data();
      const char* const der_p = slash == std::string::npos ? mt.data() + mt.size() : mt.data() + slash + 1;
      const size_t der_len = slash == std::string::npos ? 0 : mt.size() - slash - 1;
      const int bp_mut = m.pos;
      if (anc_len == 0 || der_len == 0) continue;
      const char anc0 = anc_p[0], der0 = der_p[0];

      bool use = true;
      if (has_tar_mask && (size_t)bp_mut < tar_mask.size() && tar_mask[bp_mut - 1] != 'P') use = false;
      if (has_ref_mask && (size_t)bp_mut < ref_mask.size() && ref_mask[bp_mut - 1] != 'P') use = false;
      if (!(anc_len == 1 && (anc0 == 'A' || anc0 == 'C' || anc0 == 'G' || anc0 == 'T' || anc0 == '0'))) use = false;
      if (!(der_len == 1 && (der0 == 'A' || der0 == 'C' || der0 == 'G' || der0 == 'T' || der0 == '1'))) use = false;

      if (use) {  // reference sample must carry the derived allele, coal.cpp:2181-2199
        ref.DAF = 0;
        ref.AAF = 0;
        while (ref.chrom == name && ref.bp < bp_mut) {
          if (!ref.next()) break;
        }
        if (ref.chrom != name || ref.bp != bp_mut || ref.anc != anc0 || ref.der != der0) use = false;
      }
      if (ref.DAF == 0) use = false;
      const int N_ref = ref.DAF + ref.AAF;

      if (use) {  // coal.cpp:2201-2219
        tgt.DAF = 0;
        tgt.AAF = 0;
        while (tgt.chrom == name && tgt.bp < bp_mut) {
          if (!tgt.next()) break;
        }
        if (tgt.chrom != name || tgt.bp != bp_mut || tgt.anc != anc0 || tgt.der != der0) use = false;
      }
      const int N_target = tgt.DAF + tgt.AAF;
      if (N_target == 0) use = false;
      if (!use) continue;

      double age_begin = m.age_begin;
      if (age_begin < ref_age) age_begin = ref_age;
      while (current_block_base + num_bases_per_block < bp_mut) {  // coal.cpp:2227-2234
        current_block_base += num_bases_per_block;
        advance_block();
      }
      // target genotype rounded to a diploid call, in float (coal.cpp:2236-2242)
      float f_DAF_target = tgt.DAF, f_AAF_target = tgt.AAF;
      f_DAF_target /= N_target / 2.0;
      f_AAF_target /= N_target / 2.0;
      f_DAF_target = std::round(f_DAF_target);
      f_AAF_target = std::round(f_AAF_target);

      std::vector<double>& sh = tab.sh[blk];
      std::vector<double>& ns = tab.ns[blk];
      const int DAF_ref = ref.DAF;
      if (age_begin <= age) {  // coal.cpp:2245-2275
        const int bin2 = age_bin_index(m.age_end, C);
        if (bin2 < A) {  // row 0 of the A*A table; larger indices land in rows nobody reads
          tab.sh_emp[blk][bin2] += f_DAF_target * DAF_ref / ((double)N_ref);
          tab.ns_emp[blk][bin2] += f_AAF_target * DAF_ref / ((double)N_ref);
        }
        if (pool) {  // the draws now, in the reference's order; their evaluation on a worker
          job.snps.push_back(UsedSnp{age_begin, (double)m.age_end, 0.0, f_AAF_target * DAF_ref / ((double)N_ref * num_samples), true});
          draw100();
        } else {
          for (int j = 0; j < num_samples; j++) {
            double sampled_age = dist_unif(rng) * (m.age_end - age_begin) + age_begin;
            if (sampled_age < age) sampled_age = age;
            const int bin = age_bin_index(sampled_age, C);
            if (bin < A) ns[bin] += f_AAF_target * DAF_ref / ((double)N_ref * num_samples);
          }
        }
      } else if (pool) {  // coal.cpp:2277-2297
        job.snps.push_back(UsedSnp{age_begin, (double)m.age_end, f_DAF_target * DAF_ref / ((double)N_ref * num_samples),
                                   f_AAF_target * DAF_ref / ((double)N_ref * num_samples), false});
        draw100();
      } else {
        int j = 0;
        while (j < num_samples) {
          const double sampled_age = dist_unif(rng) * (m.age_end - age_begin) + age_begin;
          bool skip = sampled_age < age;
          const int bin = age_bin_index(sampled_age, C);
          if (bin >= A) skip = true;
          if (!skip) {
            sh[bin] += f_DAF_target * DAF_ref / ((double)N_ref * num_samples);
            ns[bin] += f_AAF_target * DAF_ref / ((double)N_ref * num_samples);
            j++;
          }
        }
      }
      if (pool && pool->redo()) break;  // (a redraw was needed somewhere: this pass is void)
    }
    advance_block();  // chromosome end, coal.cpp:2306-2310
    g_times.table_fill += StageTimes::now() - t_fill0;
    if (pool && pool->redo()) break;  // (this pass is void: do not parse the rest for nothing)
  }
  if (tgt.fp) std::fclose(tgt.fp);
  if (ref.fp) std::fclose(ref.fp);
  if (pool) {
    const double t0 = StageTimes::now();
    pool->finish();  // (before the tables are trimmed)
    rng = stream->state_after_taken();
    if (g_times.on)
      std::cerr << "Timing: table fill waited " << stream->waited_ << " s for uniforms (bulk generator " << (stream->bulk() ? "on" : "off")
                << "), " << pool->waited_ << " s for room in the sampling queue, " << StageTimes::now() - t0 << " s for the last jobs" << std::endl;
  }
  tab.sh.resize(num_blocks);
  tab.ns.resize(num_blocks);
  tab.sh_emp.resize(num_blocks);
  tab.ns_emp.resize(num_blocks);
  return num_blocks;
}

// coal.cpp:2071-2321 with the sampling on worker threads where the machine has them.  COLATE_THREADS=n: 1 = everything on the
// calling thread, no thread is started (MutPrefetcher parses inline, the sequential path samples); n >= 2: n - 1 sampling
// workers + the uniform-stream thread next to the caller, and up to min(4, n - 1) .mut readers that run ahead of the fill.
// Returns the number of blocks.
int fill_tables_from_tmp(const std::vector<std::string>& chr_names,
                         const std::vector<std::string>& mut_files, const std::string& target_file,
                         const std::string& ref_file, const std::vector<std::string>& target_masks,
                         const std::vector<std::string>& ref_masks, double C, std::mt19937& rng,
                         int num_bases_per_block, int A, BlockTables& tab,
                         std::map<std::string, std::vector<MutRow>>* mut_cache) {
  const int nt = sample_threads();
  if (nt > 0) {
    const std::mt19937 rng0 = rng;
    BlockTables t2;
    SamplePool pool(nt, C, A, /*age=*/0.0);
    const int nb = fill_tables_impl(chr_names, mut_files, target_file, ref_file, target_masks, ref_masks, C, rng,
                                    num_bases_per_block, A, t2, mut_cache, &pool);
    pool.finish();
    if (!pool.redo()) {
      tab = std::move(t2);
      return nb;
    }
    rng = rng0;  // a sample beyond the age grid had to be redrawn: once more, on the sequential path
    if (g_times.on) std::cerr << "Timing: a sample beyond the age grid was redrawn; the table fill is repeated sequentially" << std::endl;
  }
  return fill_tables_impl(chr_names, mut_files, target_file, ref_file, target_masks, ref_masks, C, rng, num_bases_per_block, A,
                          tab, mut_cache, nullptr);
}

// OUT.colate_mat (coal.cpp:3471-3499): 185 grid values, then per replicate 185 shared
// and 185 not-shared counts, read with operator>> (a failed extraction leaves zeros).
bool load_colate_mat(const std::string& path, int B, int A, std::vector<double>& grid,
                     std::vector<double>& csh, std::vector<double>& cns) {
  GzText is;
  if (!is.open(path)) return false;
  std::string all, line;
  while (is.getline(line)) {
    all += line;
    all += '\n';
  }
  std::istringstream ss(all);
  for (int b = 0; b < A; b++) ss >> grid[b];
  csh.assign((size_t)B * A, 0.0);
  cns.assign((size_t)B * A, 0.0);
  for (int i = 0; i < B; i++) {
    for (int b = 0; b < A; b++) ss >> csh[(size_t)i * A + b];
    for (int b = 0; b < A; b++) ss >> cns[(size_t)i * A + b];
  }
  return true;
}

bool file_exists(const std::string& p) {
  FILE* f = std::fopen(p.c_str(), "rb");
  if (!f) return false;
  std::fclose(f);
  return true;
}

RankCtx g_rank;

bool write_all(int fd, const void* buf, size_t n) {
  const char* p = static_cast<const char*>(buf);
  while (n) {
    ssize_t k = ::write(fd, p, n);
    if (k <= 0) return false;
    p += k, n -= (size_t)k;
  }
  return true;
}
bool read_all(int fd, void* buf, size_t n) {
  char* p = static_cast<char*>(buf);
  while (n) {
    ssize_t k = ::read(fd, p, n);
    if (k <= 0) return false;
    p += k, n -= (size_t)k;
  }
  return true;
}

void print_usage_footer() {  // coal.cpp:3852-3861
  rusage usage;
  getrusage(RUSAGE_SELF, &usage);
  std::cerr << "CPU Time spent: " << usage.ru_utime.tv_sec << "." << std::setfill('0') << std::setw(6)
            << usage.ru_utime.tv_usec << "s; Max Memory usage: " << usage.ru_maxrss / 1000.0 << "Mb." << std::endl;
  std::cerr << "---------------------------------------------------------" << std::endl << std::endl;
}

// coal.cpp:3295-3313: with --chr one .mut per listed chromosome (<mut>_chr<name>.mut), else the --mut path verbatim and
// the chromosome name ""
void chromosome_files(const Options& opt, std::vector<std::string>& names, std::vector<std::string>& mut_files) {
  if (opt.has("chr")) {
    GzText is_chr;
    if (!is_chr.open(opt.get("chr"))) std::cerr << "Error while opening file " << opt.get("chr") << std::endl;
    std::string line;
    while (is_chr.getline(line)) {
      names.push_back(line);
      mut_files.push_back(opt.get("mut") + "_chr" + line + ".mut");
    }
  } else {
    names.push_back("");
    mut_files.push_back(opt.get("mut"));
  }
}

int run_mut(const Options& opt) {
  if (!opt.has("mut") || !opt.has("output")) {  // coal.cpp:3077-3087
    std::cout << "Not enough arguments supplied." << std::endl;
    std::cout << "Needed: mut, bins, output. Optional: target_tmp, reference_tmp, target_age, "
                 "reference_age, target_mask, reference_mask, coal, num_bootstrap."
              << std::endl;
    print_help();
    return 0;
  }
  std::cerr << "---------------------------------------------------------" << std::endl;
  std::cerr << "Calculating coalescence rates for (ancient) samples.." << std::endl;

  // One process, one GPU: create the HIP context on a second thread while this one parses the input files (a fresh
  // process pays a few hundred ms for it; end to end 0.63 -> see profiles/r02/bench/e2e.txt).  Not with --ranks (every
  // rank picks its own device after the fork) or --devices (several contexts), not when no device is needed.
  struct Warm {
    std::thread t;
    ~Warm() {
      if (t.joinable()) t.join();
    }
  } warm;
  if (!g_rank.ranked && !opt.has("devices") && !opt.has("counts_only")) {
    int warm_dev = 0;  // the device the run will use (--device N)
    try {
      if (opt.has("device")) warm_dev = std::stoi(opt.get("device"));
    } catch (...) {
      warm_dev = 0;  // (reported where the option is used)
    }
    warm.t = std::thread([warm_dev] { (void)colate_warm_up(warm_dev); });
  }

  double target_age = 0, ref_age = 0;
  try {
    if (opt.has("target_age")) target_age = std::stof(opt.get("target_age"));
    if (opt.has("reference_age")) ref_age = std::stof(opt.get("reference_age"));
  } catch (...) {
    std::cerr << "Error: sample ages must be numbers." << std::endl;
    return 1;
  }
  if (!(target_age >= 0.0) || !(ref_age >= 0.0)) {
    std::cerr << "Error: sample ages must be non-negative." << std::endl;
    return 1;
  }
  double years_per_gen = 28.0;
  if (opt.has("years_per_gen")) years_per_gen = std::stof(opt.get("years_per_gen"));
  const double age = std::max(target_age, ref_age) / years_per_gen;
  std::cerr << age << std::endl;
  const bool is_ancient = age > 0.0;

  const double C = 10;
  std::vector<double> age_grid(256);
  const int A = colate_age_grid(age_grid.data(), 256);
  age_grid.resize(A);
  std::cerr << "num_bins: " << A << std::endl;

  const int num_bases_per_block = 30e6;
  std::mt19937 rng;
  int seed = std::time(0) + getpid();  // coal.cpp:3158
  if (opt.has("seed")) seed = std::stoi(opt.get("seed"));
  rng.seed(seed);
  int B = 1;
  if (opt.has("num_bootstraps")) B = std::stoi(opt.get("num_bootstraps"));
  if (B < 1) {
    std::cerr << "Error: --num_bootstraps must be at least 1." << std::endl;
    return 1;
  }
  const std::string out = opt.get("output");

  std::vector<double> csh, cns, fsh, fns, fshe, fnse, weights;
  int num_blocks = 0;
  bool gpu_bootstrap = false;
  const std::string mat = out + ".colate_mat";
  if (file_exists(mat)) {  // coal.cpp:3169-3170, 3471-3499
    std::cerr << "Loading precomputed file " << mat << std::endl;
    load_colate_mat(mat, B, A, age_grid, csh, cns);
  } else if (opt.has("target_tmp") && opt.has("reference_tmp")) {
    std::vector<std::string> mut_files, tmask, rmask, names;
    if (opt.has("chr")) {  // coal.cpp:3295-3310
      GzText is_chr;
      if (!is_chr.open(opt.get("chr")))
        std::cerr << "Error while opening file " << opt.get("chr") << std::endl;
      std::string line;
      while (is_chr.getline(line)) {
        names.push_back(line);
        mut_files.push_back(opt.get("mut") + "_chr" + line + ".mut");
        if (opt.has("target_mask")) tmask.push_back(opt.get("target_mask") + "_chr" + line + ".fa");
        if (opt.has("reference_mask")) rmask.push_back(opt.get("reference_mask") + "_chr" + line + ".fa");
      }
    } else {
      names.push_back("");
      mut_files.push_back(opt.get("mut"));
      if (opt.has("target_mask")) tmask.push_back(opt.get("target_mask"));
      if (opt.has("reference_mask")) rmask.push_back(opt.get("reference_mask"));
    }
    // Without masks the pair goes through the engine of the batched front end (mut_pairs.cpp) as a list of one: every .mut file
    // parsed in parallel, the .colate.in files mapped, the SNP walk on one thread and the age sampling of the genome blocks on
    // all the others, exact table-driven age bins -- the same tables bit for bit (22 x 1M rows: 3.3 -> 0.x s of table fill,
    // profiles/r04/bench/e2e_large.txt).  With masks (and with COLATE_THREADS=1 or COLATE_SINGLE_FEEDER=1) the single-pair
    // feeder below, which is also what the engine falls back to.
    int nb = -1;
    const char* thr_env = std::getenv("COLATE_THREADS");
    if (tmask.empty() && rmask.empty() && !std::getenv("COLATE_SINGLE_FEEDER") && !(thr_env && std::atoi(thr_env) <= 1)) {
      for (size_t chr = 0; chr < mut_files.size(); chr++) std::cerr << "parsing CHR: " << chr + 1 << " / " << mut_files.size() << std::endl;
      nb = fill_single_pair(opt, opt.get("target_tmp"), opt.get("reference_tmp"), seed, A, fsh, fns, fshe, fnse, rng);
    }
    if (nb < 0) {
      BlockTables tab;
      nb = fill_tables_from_tmp(names, mut_files, opt.get("target_tmp"), opt.get("reference_tmp"), tmask, rmask, C, rng,
                                num_bases_per_block, A, tab);
      // block bootstrap + F redistribution (coal.cpp:3326-3451) on flat [nb][A] tables
      const int nbp = nb > 0 ? nb : 0;
      fsh.resize((size_t)nbp * A), fns.resize((size_t)nbp * A), fshe.resize((size_t)nbp * A), fnse.resize((size_t)nbp * A);
      for (int j = 0; j < nbp; j++) {
        std::copy(tab.sh[j].begin(), tab.sh[j].end(), fsh.begin() + (size_t)j * A);
        std::copy(tab.ns[j].begin(), tab.ns[j].end(), fns.begin() + (size_t)j * A);
        std::copy(tab.sh_emp[j].begin(), tab.sh_emp[j].end(), fshe.begin() + (size_t)j * A);
        std::copy(tab.ns_emp[j].begin(), tab.ns_emp[j].end(), fnse.begin() + (size_t)j * A);
      }
    }
    std::cerr << "Number of blocks: " << nb << std::endl;
    if (nb < 1) {
      std::cerr << "Error: no genome blocks were read." << std::endl;
      return 1;
    }
    num_blocks = nb;
    csh.assign((size_t)B * A, 0.0);
    cns.assign((size_t)B * A, 0.0);
    // the weights come from the run's mt19937 either way (coal.cpp:3350-3357); the weighted sums and
    // the F redistribution run on the GPU together with the EM unless only the counts are wanted
    // (--counts_only, no device needed) or the replicates are sharded over several GPUs
    gpu_bootstrap = !opt.has("counts_only") && !opt.has("devices") && !(g_rank.ranked && opt.has("counts_out")) &&
                    !opt.has("write_colate_mat");
    if (gpu_bootstrap) {
      weights.resize((size_t)B * nb);
      if (int rc = colate_bootstrap_weights(&rng, B, nb, weights.data())) {
        std::cerr << "Error: " << colate_last_error() << " (" << rc << ")" << std::endl;
        return 1;
      }
    } else if (int rc = colate_bootstrap_counts(&rng, B, nb, A, age_grid.data(), age, fsh.data(), fns.data(),
                                                fshe.data(), fnse.data(), csh.data(), cns.data())) {
      std::cerr << "Error: " << colate_last_error() << " (" << rc << ")" << std::endl;
      return 1;
    }
  } else {
    std::cerr << "Error: colate_amd reads --target_tmp/--reference_tmp (.colate.in) inputs or an "
                 "existing <output>.colate_mat; BCF/BAM inputs go through `Colate --mode make_tmp` first."
              << std::endl;
    return 1;
  }

  auto write_counts = [&]() {  // same layout as the reference's .colate_mat (coal.cpp:3336-3343, 3453-3469)
    write_counts_file(opt.get("counts_out"), B, A, age_grid, csh.data(), cns.data());
  };
  if (opt.has("write_colate_mat") && num_blocks > 0) {
    // coal.cpp:3336-3343, 3453-3470 (what the reference does when its inputs are BCF/BAM files): the counts are divided
    // by 1e3 IN PLACE -- the EM then runs on the scaled tables -- and written with the stream's default 6 significant
    // digits: grid line, then per replicate a line of shared and a line of not-shared counts.  The file is what a later
    // run (ours or the reference's) picks up as "precomputed file" (coal.cpp:3471-3499).
    const double norm = 1e3;
    for (double& v : csh) v /= norm;
    for (double& v : cns) v /= norm;
    if (g_rank.rank == 0) {
      std::ofstream os_mat(mat);
      for (int b = 0; b < A; b++) os_mat << age_grid[b] << " ";
      os_mat << "\n";
      for (int i = 0; i < B; i++) {
        for (int b = 0; b < A; b++) os_mat << csh[(size_t)i * A + b] << " ";
        os_mat << "\n";
        for (int b = 0; b < A; b++) os_mat << cns[(size_t)i * A + b] << " ";
        os_mat << "\n";
      }
    }
  }
  if (opt.has("counts_out") && !gpu_bootstrap) {
    if (g_rank.rank == 0) write_counts();
  }
  auto report_times = [&]() {
    if (g_times.on)
      std::cerr << "Timing: parse_mut " << g_times.parse_mut << " s (on the reader thread when pipelined), table_fill "
                << g_times.table_fill << " s, waited_for_parser " << g_times.wait_for_parser << " s, bootstrap_em "
                << g_times.bootstrap_em << " s" << std::endl;
  };
  if (opt.has("counts_only")) {
    report_times();
    return 0;
  }

  // ---- epochs (coal.cpp:3501-3646)
  std::vector<double> epochs(COLATE_MAX_EPOCHS), init_rates(COLATE_MAX_EPOCHS, COLATE_DEFAULT_INIT_RATE);
  int E = 0, ep_null = 0;
  if (opt.has("coal")) {
    E = colate_epochs_from_coal(opt.get("coal").c_str(), age, epochs.data(), init_rates.data(), COLATE_MAX_EPOCHS);
    if (E > 0) {
      for (int e = 0; e < E; e++) std::cerr << init_rates[e] << " ";
      std::cerr << std::endl;
    }
  } else if (opt.has("bins")) {
    E = colate_epochs_from_bins(opt.get("bins").c_str(), age, years_per_gen, epochs.data(), COLATE_MAX_EPOCHS, &ep_null);
  } else {
    std::cerr << "Error: need --bins or --coal." << std::endl;
    return 1;
  }
  if (E <= 0) {
    std::cerr << colate_last_error() << std::endl;
    return 1;
  }
  epochs.resize(E);
  init_rates.resize(E);

  std::cerr << "Maximising likelihood using EM.. " << std::endl;
  if (opt.has("device") && !g_rank.ranked) {
    if (int rc = colate_set_device(std::stoi(opt.get("device")))) {
      std::cerr << "Error: " << colate_last_error() << " (" << rc << ")" << std::endl;
      return 1;
    }
  }
  std::vector<double> rates((size_t)B * E), ll(B);
  std::vector<int> iters(B), flags(B);
  int rc;
  const double t_em0 = StageTimes::now();
  if (g_rank.ranked) {
    // one process per GPU: this rank's contiguous replicate range on its own device, then ONE RCCL all-gather
#ifdef COLATE_TEST_HOOKS  // (only in lib/testhooks/libcolate_amd.so, which the tests of the launcher load: never in the product library)
    if (const char* h = std::getenv("COLATE_TEST_HANG_RANK")) {  // a rank stuck as if inside a collective
      if (std::atoi(h) == g_rank.rank)
        for (;;) ::pause();
    }
#endif
    const int ndev = colate_device_count();
    if (ndev < 1) {
      std::cerr << "Error: " << colate_last_error() << std::endl;
      return 1;
    }
    const int dev0 = opt.has("device") ? std::stoi(opt.get("device")) : 0;
    unsigned char id[COLATE_COMM_ID_BYTES];
    void* comm = nullptr;
    rc = colate_set_device((dev0 + g_rank.rank) % ndev);
    if (!rc) {
      if (g_rank.rank == 0) {
        rc = colate_comm_unique_id(id);
        if (!write_all(g_rank.fd_id_out, id, rc ? 0 : sizeof(id)) && !rc) rc = COLATE_EIO;
        ::close(g_rank.fd_id_out);  // (on failure the launcher sees end-of-file and tells the others)
      } else if (!read_all(g_rank.fd_id_in, id, sizeof(id))) {
        std::cerr << "Error: rank " << g_rank.rank << " did not receive the communicator id." << std::endl;
        return 1;
      }
    }
    if (!rc) rc = colate_comm_create(id, g_rank.nranks, g_rank.rank, &comm);
    if (!rc) {
      if (gpu_bootstrap)
        rc = colate_bootstrap_em_batch_allgather(comm, B, num_blocks, E, A, age_grid.data(), age, weights.data(), fsh.data(),
                                                 fns.data(), fshe.data(), fnse.data(), epochs.data(), init_rates.data(),
                                                 COLATE_DEFAULT_MAX_ITER, COLATE_DEFAULT_MIN_ITER, COLATE_DEFAULT_REL_TOL,
                                                 COLATE_DEFAULT_RATE_FLOOR, rates.data(), iters.data(), ll.data(), flags.data());
      else
        rc = colate_em_batch_allgather(comm, B, E, A, age_grid.data(), csh.data(), cns.data(), epochs.data(),
                                       init_rates.data(), COLATE_DEFAULT_MAX_ITER, COLATE_DEFAULT_MIN_ITER,
                                       COLATE_DEFAULT_REL_TOL, COLATE_DEFAULT_RATE_FLOOR, rates.data(), iters.data(),
                                       ll.data(), flags.data());
    }
    std::string msg = rc ? colate_last_error() : "";
    colate_comm_destroy(comm);
    if (rc) {
      std::cerr << "Error: " << msg << " (" << rc << ")" << std::endl;
      return 1;
    }
    if (g_rank.rank != 0) return 0;  // every rank holds all results; rank 0 reports and writes them
  } else if (opt.has("devices")) {
    const int nd = std::stoi(opt.get("devices"));
    if (nd < 1) {
      std::cerr << "Error: --devices must be at least 1." << std::endl;
      return 1;
    }
    std::vector<int> devs(nd);
    for (int d = 0; d < nd; d++) devs[d] = d;
    rc = colate_em_batch_sharded(nd, devs.data(), B, E, A, age_grid.data(), csh.data(), cns.data(),
                                 epochs.data(), init_rates.data(), COLATE_DEFAULT_MAX_ITER,
                                 COLATE_DEFAULT_MIN_ITER, COLATE_DEFAULT_REL_TOL, COLATE_DEFAULT_RATE_FLOOR,
                                 rates.data(), iters.data(), ll.data(), flags.data());
  } else if (gpu_bootstrap) {
    rc = colate_bootstrap_em_batch(B, num_blocks, E, A, age_grid.data(), age, weights.data(), fsh.data(), fns.data(),
                                   fshe.data(), fnse.data(), epochs.data(), init_rates.data(), COLATE_DEFAULT_MAX_ITER,
                                   COLATE_DEFAULT_MIN_ITER, COLATE_DEFAULT_REL_TOL, COLATE_DEFAULT_RATE_FLOOR,
                                   rates.data(), iters.data(), ll.data(), flags.data(), csh.data(), cns.data());
    if (rc == 0 && opt.has("counts_out")) write_counts();
  } else {
    rc = colate_em_batch(B, E, A, age_grid.data(), csh.data(), cns.data(), epochs.data(),
                         init_rates.data(), COLATE_DEFAULT_MAX_ITER, COLATE_DEFAULT_MIN_ITER,
                         COLATE_DEFAULT_REL_TOL, COLATE_DEFAULT_RATE_FLOOR, rates.data(),
                         iters.data(), ll.data(), flags.data());
  }
  g_times.bootstrap_em = StageTimes::now() - t_em0;
  report_times();
  if (rc) {
    std::cerr << "Error: " << colate_last_error() << " (" << rc << ")" << std::endl;
    return 1;
  }
  int unresolved_max = 0;
  for (int i = 0; i < B; i++) {
    std::cerr << "Bootstrap " << i + 1 << ": Total iterations " << iters[i] << std::endl;
    if (flags[i] & (COLATE_FLAG_NAN | COLATE_FLAG_NEG))
      std::cerr << "Warning: bootstrap " << i + 1
                << " produced NaN or negative sufficient statistics (the reference aborts here)."
                << std::endl;
    unresolved_max = std::max(unresolved_max, COLATE_UNRESOLVED_EPOCHS(flags[i]));
  }
  if (unresolved_max > 0)  // (include/colate_amd.h, COLATE_FLAG_UNRESOLVED)
    std::cerr << "Note: the last " << unresolved_max << " of " << E << " epochs are older than the data resolve: "
              << "their printed rates depend on rounding residue (in the reference build too) and are not reproducible."
              << std::endl;
  if (colate_write_coal((out + ".coal").c_str(), B, E, epochs.data(), rates.data(), is_ancient ? 1 : 0, ep_null)) {
    std::cerr << "Error: " << colate_last_error() << std::endl;
    return 1;
  }

  print_usage_footer();
  return 0;
}


void write_counts_file(const std::string& path, int B, int A, const std::vector<double>& grid,
                       const double* csh, const double* cns) {
  FILE* f = std::fopen(path.c_str(), "w");
  if (!f) {
    std::cerr << "Error: cannot write " << path << std::endl;
    std::exit(1);
  }
  for (int b = 0; b < A; b++) std::fprintf(f, "%.17g ", grid[b]);
  std::fprintf(f, "\n");
  for (int i = 0; i < B; i++) {
    for (int b = 0; b < A; b++) std::fprintf(f, "%.17g ", csh[(size_t)i * A + b]);
    std::fprintf(f, "\n");
    for (int b = 0; b < A; b++) std::fprintf(f, "%.17g ", cns[(size_t)i * A + b]);
    std::fprintf(f, "\n");
  }
  std::fclose(f);
}

// ------------------------------------------------------------------ --mode make_tmp --target_table
// The htslib-free input of the reference's make_tmp (coal.cpp:2923-3069 -> maketmp_table, coal.cpp:2682-2808): a text
// table `chr bp allele` of the target's haploid calls becomes the binary .colate.in stream `--mode mut` reads
// (record layout coal.cpp:2505-2514).  BCF and BAM inputs (maketmp_vcf / maketmp_bam) need htslib and stay with the
// reference build.  The table is walked with formatted extraction exactly like the reference's igzstream (a failed
// read at the end of the file leaves the last record in place).
int run_make_tmp(const Options& opt) {
  if (!opt.has("mut") || !opt.has("output")) {  // coal.cpp:2929-2939
    std::cout << "Not enough arguments supplied." << std::endl;
    std::cout << "Needed: mut, ref_genome, output, either of target_bcf or target_bam. Optional: filters, target_mask, "
                 "strandfilter, anc_genome."
              << std::endl;
    print_help();
    std::cout << "Calculate coalescence rates for sample." << std::endl;
    return 0;
  }
  std::cerr << "---------------------------------------------------------" << std::endl;
  std::cerr << "Calculating Colate tmp input file for ";
  if (opt.has("target_bcf") || opt.has("target_bam")) {
    std::cerr << std::endl
              << "Error: colate_amd's make_tmp reads --target_table only; BCF/BAM inputs need htslib (reference build)."
              << std::endl;
    return 1;
  }
  if (!opt.has("target_table")) {  // (the reference falls through and writes nothing)
    std::cerr << std::endl << "Error: --mode make_tmp needs --target_table." << std::endl;
    return 1;
  }
  if (!opt.has("ref_genome")) {  // cxxopts throws on options["ref_genome"].as<std::string>() (coal.cpp:3027, 3037)
    std::cerr << std::endl << "Error: --mode make_tmp --target_table needs --ref_genome." << std::endl;
    return 1;
  }
  std::cerr << opt.get("target_table") << ".." << std::endl;
  std::vector<std::string> names, mut_files, ref_genomes, tmasks;
  if (opt.has("chr")) {  // coal.cpp:3018-3032
    GzText is_chr;
    if (!is_chr.open(opt.get("chr"))) std::cerr << "Error while opening file " << opt.get("chr") << std::endl;
    std::string line;
    while (is_chr.getline(line)) {
      names.push_back(line);
      mut_files.push_back(opt.get("mut") + "_chr" + line + ".mut");
      ref_genomes.push_back(opt.get("ref_genome") + "_chr" + line + ".fa");
      if (opt.has("target_mask")) tmasks.push_back(opt.get("target_mask") + "_chr" + line + ".fa");
    }
  } else {
    names.push_back("");
    mut_files.push_back(opt.get("mut"));
    ref_genomes.push_back(opt.get("ref_genome"));
    if (opt.has("target_mask")) tmasks.push_back(opt.get("target_mask"));
  }
  const std::string out_name = opt.get("output") + ".colate.in";
  FILE* fp = std::fopen(out_name.c_str(), "wb");
  if (!fp) {
    std::cerr << "Error: cannot write " << out_name << std::endl;
    return 1;
  }
  std::istringstream is;
  {
    GzText table;
    if (!table.open(opt.get("target_table"))) {  // coal.cpp:2699-2702
      std::cerr << "Error while opening file " << opt.get("target_table") << std::endl;
      return 1;
    }
    std::string all, line;
    while (table.getline(line)) {
      all += line;
      all += '\n';
    }
    is.str(all);
  }
  const bool has_tar_mask = !tmasks.empty();
  std::string chr_table, allele, ancestral, derived;
  int bp_target = -1;
  const int N_target = 1;
  for (size_t chr = 0; chr < mut_files.size(); chr++) {
    std::cerr << "parsing CHR: " << chr + 1 << " / " << mut_files.size() << std::endl;
    std::string tar_mask, ref_genome;
    if (has_tar_mask) read_fasta_mask(tmasks[chr], tar_mask);
    std::vector<MutRow> rows;
    read_mut_file(mut_files[chr], rows);
    read_fasta_mask(ref_genomes[chr], ref_genome);  // read (and required to exist) as in the reference; only its presence matters
    if (bp_target == -1) is >> chr_table >> bp_target >> allele;
    while (chr_table != names[chr]) {
      if (!(is >> chr_table >> bp_target >> allele)) break;
    }
    for (const MutRow& m : rows) {
      if (m.flipped != 0 || m.num_branches != 1) continue;
      size_t i = 0;
      ancestral.clear();
      derived.clear();
      while (i < m.mutation_type.size() && m.mutation_type[i] != '/') ancestral.push_back(m.mutation_type[i++]);
      i++;
      while (i < m.mutation_type.size()) derived.push_back(m.mutation_type[i++]);
      const int bp_mut = m.pos;
      if (ancestral.empty() || derived.empty()) continue;
      bool use = true;
      if (ancestral != "A" && ancestral != "C" && ancestral != "G" && ancestral != "T" && ancestral != "0") use = false;
      if (derived != "A" && derived != "C" && derived != "G" && derived != "T" && derived != "1") use = false;
      if (has_tar_mask) {  // coal.cpp:2749-2755: sites beyond the mask are dropped here (unlike in parse_tmptmp)
        if (bp_mut >= (int)tar_mask.size())
          use = false;
        else if (bp_mut < 1 || tar_mask[bp_mut - 1] != 'P')
          use = false;
      }
      if (!use) continue;
      if (chr_table == names[chr] && bp_target < bp_mut) {
        while (!is.eof() && chr_table == names[chr] && bp_target < bp_mut) is >> chr_table >> bp_target >> allele;
      }
      int DAF_target = 0;
      if (chr_table == names[chr] && bp_target == bp_mut) {  // the target has a call here (coal.cpp:2768-2782)
        if (allele == derived || allele == ancestral) {
          if (allele == derived) DAF_target = 1;
        } else {
          use = false;
        }
      } else {
        use = false;
      }
      if (!use) continue;
      const int lchrom = (int)names[chr].size();
      const int AAF_target = N_target - DAF_target;
      std::fwrite(&lchrom, sizeof(int), 1, fp);
      std::fwrite(names[chr].c_str(), sizeof(char), (size_t)lchrom, fp);
      std::fwrite(&bp_mut, sizeof(int), 1, fp);
      std::fwrite(&ancestral[0], sizeof(char), 1, fp);
      std::fwrite(&derived[0], sizeof(char), 1, fp);
      std::fwrite(&AAF_target, sizeof(int), 1, fp);
      std::fwrite(&DAF_target, sizeof(int), 1, fp);
    }
  }
  std::fclose(fp);
  print_usage_footer();  // coal.cpp:3055-3067
  return 0;
}

// `--ranks N`: fork N processes BEFORE anything touches the GPU (this process never does), one per GPU; each runs the
// whole `--mode mut` pipeline (same inputs, same --seed, hence the same tables and bootstrap weights), computes its
// contiguous range of replicates and takes part in one RCCL all-gather (colate_comm.cpp); rank 0 writes the outputs.
// The launcher only relays rank 0's 128-byte communicator id to the other ranks and collects the exit codes.
int run_ranked(const Options& opt, int nranks) {
  if (colate_device_touched()) {
    // fork() after the HIP runtime is up gives children with a half-copied runtime (its threads and device queues are
    // not duplicated): they hang or fault.  The command-line `Colate` never gets here; a host process that has already
    // computed through this library (or that shares it with torch) must start the ranks as fresh processes instead.
    std::cerr << "Error: --ranks forks one process per GPU and must run before this process first uses a GPU through "
                 "libcolate_amd; start `Colate --ranks N` as its own process (never re-exec from here)." << std::endl;
    return 1;
  }
  if (!opt.has("seed")) {
    std::cerr << "Error: --ranks needs --seed (every rank must draw the same bootstrap weights)." << std::endl;
    return 1;
  }
  if (opt.has("devices")) {
    std::cerr << "Error: --ranks cannot be combined with --devices." << std::endl;
    return 1;
  }
  // where the ranks other than 0 keep their progress lines: next to the output (with --pairs: next to the list of pairs)
  const std::string log_prefix = opt.has("pairs") ? opt.get("pairs") : opt.get("output");
  int up[2];
  if (::pipe(up) != 0) {
    std::perror("pipe");
    return 1;
  }
  std::vector<int> down_r(nranks, -1), down_w(nranks, -1);
  for (int r = 1; r < nranks; r++) {
    int fd[2];
    if (::pipe(fd) != 0) {
      std::perror("pipe");
      return 1;
    }
    down_r[r] = fd[0], down_w[r] = fd[1];
  }
  std::cerr.flush();
  std::cout.flush();
  std::vector<pid_t> pids(nranks, -1);
  for (int r = 0; r < nranks; r++) {
    pid_t pid = ::fork();
    if (pid < 0) {
      std::perror("fork");
      return 1;
    }
    if (pid == 0) {  // rank r
      g_rank.ranked = true, g_rank.rank = r, g_rank.nranks = nranks;
      ::close(up[0]);
      for (int q = 1; q < nranks; q++) {
        ::close(down_w[q]);
        if (q != r) ::close(down_r[q]);
      }
      if (r == 0) {
        g_rank.fd_id_out = up[1];
      } else {
        ::close(up[1]);
        g_rank.fd_id_in = down_r[r];
        // only rank 0 talks on the terminal; the others keep their progress lines in a file that a clean exit removes
        const std::string log = log_prefix + ".rank" + std::to_string(r) + ".stderr";
        if (!std::freopen(log.c_str(), "w", stderr)) std::perror("freopen");
      }
      int code = 1;
      try {
        code = opt.has("pairs") ? run_mut_pairs(opt) : run_mut(opt);
      } catch (const std::exception& e) {
        std::cerr << "Error: " << e.what() << std::endl;
      }
      std::cerr.flush();
      if (r != 0 && code == 0) std::remove((log_prefix + ".rank" + std::to_string(r) + ".stderr").c_str());
      std::fflush(nullptr);
      ::_exit(code);
    }
    pids[r] = pid;
  }
  ::close(up[1]);
  for (int r = 1; r < nranks; r++) ::close(down_r[r]);
  // One loop relays rank 0's communicator id to the other ranks AND reaps the ranks as they end (our own pids only: this
  // function is also reachable through the library ABI, whose host may have children of its own).  Nothing here blocks:
  //  * a rank that fails before or outside the collective leaves the others waiting in ncclCommInitRank / ncclAllGather
  //    for ever, so the first failure -- or any exit while the id is still outstanding -- starts a grace period
  //    (COLATE_RANK_GRACE_SEC, default 15 s: ranks that are merely finishing get there) after which the rest are killed;
  //  * rank 0 ending without having published the id closes the other ranks' pipes (they report and exit);
  //  * a rank 0 that neither publishes the id nor ends (stuck in HIP initialisation, ncclGetUniqueId or its table fill)
  //    is given COLATE_RANK_ID_TIMEOUT_SEC (default 3600 s: the id follows the table fill, which may be long), then
  //    every rank is killed and the launcher returns non-zero.
  double grace_s = 15.0, id_timeout_s = 3600.0;
  if (const char* g = std::getenv("COLATE_RANK_GRACE_SEC")) grace_s = std::atof(g);
  if (const char* g = std::getenv("COLATE_RANK_ID_TIMEOUT_SEC")) id_timeout_s = std::atof(g);
  unsigned char id[COLATE_COMM_ID_BYTES];
  size_t id_have = 0;
  bool id_open = true;
  auto close_id_pipes = [&]() {
    if (!id_open) return;
    ::close(up[0]);
    for (int r = 1; r < nranks; r++) ::close(down_w[r]);
    id_open = false;
  };
  ::fcntl(up[0], F_SETFL, ::fcntl(up[0], F_GETFL, 0) | O_NONBLOCK);
  const auto t_start = std::chrono::steady_clock::now();
  auto seconds_since = [](std::chrono::steady_clock::time_point t) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count();
  };
  int worst = 0, remaining = nranks;
  std::vector<char> done(nranks, 0);
  bool failing = false, killed = false;
  auto t_fail = std::chrono::steady_clock::now();
  auto kill_rest = [&]() {
    for (int r = 0; r < nranks; r++)
      if (!done[r]) ::kill(pids[r], SIGKILL);
    killed = true;
  };
  while (remaining > 0) {
    bool progressed = false;
    if (id_open) {  // (waits up to 20 ms for bytes of the id: this is also the loop's pause)
      pollfd pfd{up[0], POLLIN, 0};
      if (::poll(&pfd, 1, 20) > 0) {
        const ssize_t k = ::read(up[0], id + id_have, sizeof(id) - id_have);
        if (k > 0) {
          id_have += (size_t)k;
          if (id_have == sizeof(id)) {
            for (int r = 1; r < nranks; r++) write_all(down_w[r], id, sizeof(id));
            close_id_pipes();
          }
          progressed = true;
        } else if (k == 0) {  // rank 0 ended (or closed its end) without an id: end-of-file for the others, too
          close_id_pipes();
        }
      }
      if (id_open && !killed && seconds_since(t_start) > id_timeout_s) {
        std::cerr << "Error: rank 0 has not published the communicator id after " << id_timeout_s
                  << " s (COLATE_RANK_ID_TIMEOUT_SEC); ending all ranks." << std::endl;
        close_id_pipes();
        kill_rest();
        worst = 1;
      }
    }
    for (int r = 0; r < nranks; r++) {
      if (done[r]) continue;
      int st = 0;
      const pid_t w = ::waitpid(pids[r], &st, WNOHANG);
      if (w == 0) continue;
      done[r] = 1, remaining--, progressed = true;
      const bool ok = (w == pids[r]) && WIFEXITED(st) && WEXITSTATUS(st) == 0;
      if (!ok) {
        std::cerr << "Error: rank " << r << (killed ? " was ended by the launcher" : " failed");
        if (r > 0) std::cerr << " (see " << log_prefix << ".rank" << r << ".stderr)";
        std::cerr << std::endl;
        worst = 1;
      }
      // a failure -- or any exit while the id is outstanding (nobody can finish properly without it) -- starts the clock
      if ((!ok || (id_open && nranks > 1)) && !failing) failing = true, t_fail = std::chrono::steady_clock::now();
    }
    if (remaining == 0) break;
    if (failing && !killed && seconds_since(t_fail) > grace_s) {
      std::cerr << "Error: a rank failed; ending the " << remaining << " rank(s) still waiting after " << grace_s << " s."
                << std::endl;
      close_id_pipes();
      kill_rest();
      worst = 1;
    }
    if (!progressed && !id_open) ::usleep(20000);
  }
  close_id_pipes();
  return worst;
}

}  // namespace colate_drv

using namespace colate_drv;

extern "C" int colate_mut_main(int argc, char** argv) {
  Options opt;
  std::string err;
  if (!parse_options(argc, argv, opt, err)) {
    std::cerr << err << std::endl;
    return 1;
  }
  if (!opt.has("mode")) {  // Colate.cpp:104-112
    std::cout << "Not enough arguments supplied." << std::endl;
    print_help();
    return 0;
  }
  const std::string& mode = opt.get("mode");
  if (mode == "mut") {
    if (opt.has("help")) {
      print_help();
      std::cout << "Calculate coalescence rates for sample." << std::endl;
      return 0;
    }
    try {
      if (opt.has("ranks")) {
        const int nranks = std::stoi(opt.get("ranks"));
        if (nranks < 1 || nranks > 64) {
          std::cerr << "Error: --ranks must be between 1 and 64." << std::endl;
          return 1;
        }
        if (opt.has("mut") && (opt.has("output") || opt.has("pairs"))) return run_ranked(opt, nranks);  // (also for N = 1: same code path)
      }
      if (opt.has("pairs")) return run_mut_pairs(opt);
      return run_mut(opt);
    } catch (const std::exception& e) {
      std::cerr << "Error: " << e.what() << std::endl;
      return 1;
    }
  }
  if (mode == "make_tmp") {
    try {
      return run_make_tmp(opt);
    } catch (const std::exception& e) {
      std::cerr << "Error: " << e.what() << std::endl;
      return 1;
    }
  }
  std::cout << "####### error #######" << std::endl;
  std::cout << "colate_amd implements --mode mut and --mode make_tmp --target_table (preprocess_mut, make_tmp from "
               "BCF/BAM, calc_depth, print_tmp, CondCoalRates stay with the reference build)."
            << std::endl;
  return 1;
}
