// colate_amd/csrc/mut_feeder.h -- what the two host-side translation units of the `Colate --mode mut` driver share:
// mut_driver.cpp (command line, readers, the single-pair feeder of include/coal/coal.cpp:2071-2321, mut() driver, --ranks launcher)
// and mut_pairs.cpp (the batched all-pairs front end, SURVEY.md section 8 f2 / BASELINE configs[4]).
#pragma once
#include <cmath>
#include <cstdint>
#include <functional>
#include <map>
#include <random>
#include <sstream>
#include <string>
#include <vector>

namespace colate_drv {

struct Options {
  std::map<std::string, std::string> kv;
  bool has(const std::string& k) const { return kv.count(k) > 0; }
  const std::string& get(const std::string& k) const { return kv.at(k); }
};

// stage timing (COLATE_TIMING=1: one stderr line at the end)
struct StageTimes {
  double parse_mut = 0, table_fill = 0, wait_for_parser = 0, bootstrap_em = 0;
  bool on = std::getenv("COLATE_TIMING") != nullptr;
  static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
};
extern StageTimes g_times;

// Only the columns parse_tmptmp looks at (mutations.cpp:77-246):
// snp;pos;dist;rs;tree;branches;is_not_mapping;is_flipped;age_begin;age_end;anc/der;...
struct MutRow {
  int pos = 0;
  int num_branches = 0;
  int flipped = 0;
  float age_begin = 0.0f, age_end = 0.0f;  // stored as float in the reference (mutations.hpp:21)
  std::string mutation_type = "NA";
};
// the rows of one .mut(.gz) file (mutations.cpp:56-283); exits like the reference when the file cannot be read
bool read_mut_file(const std::string& filename, std::vector<MutRow>& rows);
// the same, row by row (nothing is kept)
bool for_each_mut_row(const std::string& filename, const std::function<void(const MutRow&)>& sink);

struct BlockTables {  // one entry per genome block; emp = row 0 of the reference's A*A tables
  std::vector<std::vector<double>> sh, ns, sh_emp, ns_emp;
  void add_block(int A) {
    sh.emplace_back(A, 0.0);
    ns.emplace_back(A, 0.0);
    sh_emp.emplace_back(A, 0.0);
    ns_emp.emplace_back(A, 0.0);
  }
};

inline int age_bin_index(double x, double C) {  // coal.cpp:2265, 2284
  const double v = std::round(std::log(10 * x) * C);
  if (!(v > -2e9)) return 0;  // log(0) = -inf: the reference's (int) cast yields INT_MIN -> max(0, .) = 0
  return std::max(0, (int)v + 1);
}

// std::mt19937's recurrence with the state regenerated 624 words at a time in loops the compiler vectorises (the library's
// operator() does the same work word by word: 7.5 ns per word on the build container, against ~2 here).  Same sequence by
// construction; UniformStream checks it against the library's generator before it trusts it.  State goes in and out of a
// std::mt19937 through its textual form (the 624 words and the position, [rand.eng.mers]).
class BulkMt19937 {
 public:
  bool load(const std::mt19937& g) {
    std::ostringstream os;
    os << g;
    std::istringstream is(os.str());
    for (int i = 0; i < 624; i++)
      if (!(is >> x_[i])) return false;
    if (!(is >> p_) || p_ > 624) return false;
    return true;
  }
  bool store(std::mt19937& g) const {
    std::ostringstream os;
    for (int i = 0; i < 624; i++) os << x_[i] << ' ';
    os << p_;
    std::istringstream is(os.str());
    return static_cast<bool>(is >> g);
  }
  // the next n 32-bit outputs
  void generate(uint32_t* out, size_t n) {
    while (n) {
      if (p_ >= 624) twist();
      const size_t k = std::min(n, (size_t)(624 - p_));
      const uint32_t* x = x_ + p_;
      for (size_t i = 0; i < k; i++) {  // tempering
        uint32_t y = x[i];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        out[i] = y;
      }
      out += k, n -= k, p_ += (uint32_t)k;
    }
  }
  void discard(unsigned long long n) {
    uint32_t tmp[624];
    while (n) {
      const size_t k = (size_t)std::min<unsigned long long>(n, 624);
      generate(tmp, k);
      n -= k;
    }
  }

 private:
  static uint32_t mix(uint32_t a, uint32_t b) {
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  }
  void twist() {
    for (int i = 0; i < 227; i++) x_[i] = x_[i + 397] ^ mix(x_[i], x_[i + 1]);          // (old words only)
    for (int i = 227; i < 454; i++) x_[i] = x_[i - 227] ^ mix(x_[i], x_[i + 1]);        // (new words of the first loop)
    for (int i = 454; i < 623; i++) x_[i] = x_[i - 227] ^ mix(x_[i], x_[i + 1]);        // (new words of the second)
    x_[623] = x_[396] ^ mix(x_[623], x_[0]);
    p_ = 0;
  }
  uint32_t x_[624];
  uint32_t p_ = 624;
};

// `--ranks N`: this process is rank `rank` of `nranks` (run_ranked forks them); the 128-byte RCCL id travels from
// rank 0 to the others through the launcher's pipes.
struct RankCtx {
  bool ranked = false;  // launched by run_ranked (also with one rank: the RCCL path with a communicator of one)
  int rank = 0, nranks = 1;
  int fd_id_out = -1;  // rank 0: writes the id here
  int fd_id_in = -1;   // ranks > 0: read it here
};
extern RankCtx g_rank;
bool write_all(int fd, const void* buf, size_t n);
bool read_all(int fd, void* buf, size_t n);

struct PairSpec {
  std::string target, reference, output;
  double target_age = 0, ref_age = 0;
};

// coal.cpp:2071-2321 for one (target, reference) pair from the files themselves: the single-pair feeder (sampling on worker
// threads where the machine has them).  Returns the number of genome blocks.
int fill_tables_from_tmp(const std::vector<std::string>& chr_names, const std::vector<std::string>& mut_files,
                         const std::string& target_file, const std::string& ref_file,
                         const std::vector<std::string>& target_masks, const std::vector<std::string>& ref_masks, double C,
                         std::mt19937& rng, int num_bases_per_block, int A, BlockTables& tab,
                         std::map<std::string, std::vector<MutRow>>* mut_cache = nullptr);

// the chromosome list of --chr (coal.cpp:3295-3310): names and <mut>_chr<name>.mut paths; without --chr one unnamed chromosome
void chromosome_files(const Options& opt, std::vector<std::string>& names, std::vector<std::string>& mut_files);

void write_counts_file(const std::string& path, int B, int A, const std::vector<double>& grid, const double* csh,
                       const double* cns);
void print_usage_footer();  // "CPU Time spent: ...; Max Memory usage: ..." (coal.cpp:3852-3861)

// mut_pairs.cpp
int run_mut_pairs(const Options& opt);
// One pair through the engine of the batched front end: the flat [nb][A] tables and the generator as the fill leaves it.
// Returns the number of genome blocks, or -1 if the engine cannot be used here (the caller then runs fill_tables_from_tmp).
int fill_single_pair(const Options& opt, const std::string& target, const std::string& reference, int seed, int A,
                     std::vector<double>& sh, std::vector<double>& ns, std::vector<double>& she, std::vector<double>& nse,
                     std::mt19937& rng);

}  // namespace colate_drv
