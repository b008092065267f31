// tools/gen_wg_inputs.cpp -- TEST / BENCH INFRASTRUCTURE: synthetic `Colate --mode mut` inputs at realistic size, fast.
// Same shape as tests/synth_files.py (the formats of include/src/mutations.cpp:77-246 and include/coal/coal.cpp:2505-2514):
// per chromosome P_chr<c>.mut(.gz), plus T.colate.in, R.colate.in and chr.txt in <outdir>.
//   g++ -O2 -std=c++17 tools/gen_wg_inputs.cpp -lz -o /tmp/gen_wg_inputs && /tmp/gen_wg_inputs OUT 22 1000000 [gz|plain [NT NR]]
// With NT / NR (BASELINE configs[4]: 10 x 10): additionally T1..T<NT-1>.colate.in and R1..R<NR-1>.colate.in, further samples
// over the same .mut files (each from a generator of its own, so T.colate.in / R.colate.in = sample 0 do not depend on
// NT / NR), and pairs.txt listing all NT x NR pairs as `T<i>.colate.in R<j>.colate.in out_<i>_<j>`.
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

static void rec(FILE* f, const std::string& chrom, int bp, char anc, char der, int aaf, int daf) {
  int l = (int)chrom.size();
  fwrite(&l, 4, 1, f);
  fwrite(chrom.data(), 1, chrom.size(), f);
  fwrite(&bp, 4, 1, f);
  fwrite(&anc, 1, 1, f);
  fwrite(&der, 1, 1, f);
  fwrite(&aaf, 4, 1, f);
  fwrite(&daf, 4, 1, f);
}

int main(int argc, char** argv) {
  if (argc < 4) return 1;
  const std::string out = argv[1];
  const int nchr = atoi(argv[2]), snps = atoi(argv[3]);
  const bool gz = argc > 4 && !strcmp(argv[4], "gz");
  const int NT = argc > 6 ? atoi(argv[5]) : 1, NR = argc > 6 ? atoi(argv[6]) : 1;
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> U(0, 1);
  FILE* tgt = fopen((out + "/T.colate.in").c_str(), "wb");
  FILE* ref = fopen((out + "/R.colate.in").c_str(), "wb");
  // further samples: own files, own generators; their coverage and sharing differ a little from sample to sample
  std::vector<FILE*> xt, xr;
  std::vector<std::mt19937_64> gt, gr;
  for (int k = 1; k < NT; k++) xt.push_back(fopen((out + "/T" + std::to_string(k) + ".colate.in").c_str(), "wb")), gt.emplace_back(1000 + k);
  for (int k = 1; k < NR; k++) xr.push_back(fopen((out + "/R" + std::to_string(k) + ".colate.in").c_str(), "wb")), gr.emplace_back(2000 + k);
  if (NT > 1 || NR > 1) {
    FILE* pf = fopen((out + "/pairs.txt").c_str(), "w");
    for (int i = 0; i < NT; i++)
      for (int j = 0; j < NR; j++)
        fprintf(pf, "T%s.colate.in R%s.colate.in out_%d_%d\n", i ? std::to_string(i).c_str() : "", j ? std::to_string(j).c_str() : "", i, j);
    fclose(pf);
  }
  FILE* chrf = fopen((out + "/chr.txt").c_str(), "w");
  const char bases[] = "ACGT";
  const long span = 240000000;
  std::string buf;
  for (int c = 1; c <= nchr; c++) {
    const std::string name = std::to_string(c);
    fprintf(chrf, "%s\n", name.c_str());
    std::vector<int> pos(snps);
    {  // increasing positions: jittered grid
      const double step = (double)(span - 2000) / snps;
      for (int i = 0; i < snps; i++) pos[i] = 1000 + (int)(i * step + U(rng) * (step > 2 ? step - 1 : 1));
    }
    const std::string path = out + "/P_chr" + name + ".mut" + (gz ? ".gz" : "");
    gzFile gf = gz ? gzopen(path.c_str(), "wb1") : nullptr;
    FILE* pf = gz ? nullptr : fopen(path.c_str(), "w");
    auto put = [&](const std::string& s) {
      if (gz) gzwrite(gf, s.data(), (unsigned)s.size()); else fwrite(s.data(), 1, s.size(), pf);
    };
    put("snp;pos_of_snp;dist;rs-id;tree_index;branch_indices;is_not_mapping;is_flipped;age_begin;age_end;"
        "ancestral_allele/alternative_allele;upstream_allele;downstream_allele;\n");
    buf.clear();
    char line[512];
    for (int i = 0; i < snps; i++) {
      const int bp = pos[i];
      double age_begin = U(rng) < 0.08 ? 0.0 : std::pow(10.0, 1 + 4.2 * U(rng));
      double age_end = age_begin == 0 ? 30.0 * (1 + 1.5 * U(rng)) : age_begin * (1 + 1.5 * U(rng));
      const int ai = (int)(U(rng) * 4) & 3, di = (ai + 1 + ((int)(U(rng) * 3) % 3)) & 3;
      const char a = bases[ai], d = bases[di];
      const int flipped = U(rng) < 0.03;
      const char* branches = U(rng) > 0.04 ? "7" : "7 12";
      const bool odd = U(rng) < 0.02;
      if (U(rng) < 0.01) age_end = age_begin;
      const int dist = i + 1 < snps ? pos[i + 1] - bp : 1;
      int n = odd ? snprintf(line, sizeof line, "%d;%d;%d;rs%d;%d;%s;0;%d;%.6g;%.6g;%c%c/%c;%c;%c;\n", i, bp, dist, i, i / 10, branches,
                             flipped, age_begin, age_end, a, a, d, a, d)
                  : snprintf(line, sizeof line, "%d;%d;%d;rs%d;%d;%s;0;%d;%.6g;%.6g;%c/%c;%c;%c;\n", i, bp, dist, i, i / 10, branches,
                             flipped, age_begin, age_end, a, d, a, d);
      buf.append(line, n);
      if (buf.size() > (1u << 22)) { put(buf); buf.clear(); }
      if (U(rng) < 0.9) {
        const int daf = (int)(U(rng) * 3) % 3;
        const bool sw = U(rng) < 0.03;
        rec(ref, name, bp, sw ? d : a, sw ? a : d, 2 - daf, daf);
      }
      if (U(rng) < 0.1) rec(ref, name, bp + 1, 'A', 'G', 1, 1);
      if (U(rng) < 0.9) {
        const int nr = (int)(U(rng) * 5) % 5;
        const double age_mid = 0.5 * (age_begin + age_end);
        const bool shares = U(rng) < 0.8 * (1.0 - std::exp(-age_mid / 12000.0));
        int daf = U(rng) > 0.05 ? (shares ? nr : 0) : (nr ? (int)(U(rng) * (nr + 1)) % (nr + 1) : 0);
        rec(tgt, name, bp, a, d, nr - daf, daf);
      }
      for (size_t k = 0; k < xr.size(); k++) {  // further reference samples
        std::mt19937_64& g = gr[k];
        if (U(g) < 0.88 + 0.01 * k) {
          const int daf = (int)(U(g) * 3) % 3;
          const bool sw = U(g) < 0.03;
          rec(xr[k], name, bp, sw ? d : a, sw ? a : d, 2 - daf, daf);
        }
        if (U(g) < 0.1) rec(xr[k], name, bp + 1, 'A', 'G', 1, 1);
      }
      for (size_t k = 0; k < xt.size(); k++) {  // further targets: own coverage, own pairwise Ne
        std::mt19937_64& g = gt[k];
        if (U(g) < 0.6 + 0.03 * k) {
          const int nr = (int)(U(g) * 5) % 5;
          const double age_mid = 0.5 * (age_begin + age_end);
          const bool shares = U(g) < 0.8 * (1.0 - std::exp(-age_mid / (8000.0 + 1500.0 * k)));
          int daf = U(g) > 0.05 ? (shares ? nr : 0) : (nr ? (int)(U(g) * (nr + 1)) % (nr + 1) : 0);
          rec(xt[k], name, bp, a, d, nr - daf, daf);
        }
      }
    }
    put(buf);
    if (gz) gzclose(gf); else fclose(pf);
  }
  fclose(tgt); fclose(ref); fclose(chrf);
  for (FILE* f : xt) fclose(f);
  for (FILE* f : xr) fclose(f);
  return 0;
}
