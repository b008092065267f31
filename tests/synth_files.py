"""Synthetic `Colate --mode mut` input files (TEST INFRASTRUCTURE): per-chromosome .mut files plus a
target and a reference .colate.in, in the formats the reference reads
(.mut: include/src/mutations.cpp:77-246; .colate.in: include/coal/coal.cpp:2505-2514).

SURVEY.md §8(d) file-level workload: SNPs at increasing positions, age_begin = 10^U(1,5.2) (a
fraction with age_begin = 0 to exercise the F-redistribution path, coal.cpp:2245-2275),
age_end = age_begin*(1+1.5U), transitions among A/C/G/T, reference DAF in {0,1,2}, target DAF in {0,1,2};
some rows flipped / multi-branch / non-SNP so that every filter of coal.cpp:2150-2219 fires."""
import gzip
import os
import struct

import numpy as np


def write_inputs(outdir, chroms=("1", "2"), snps_per_chr=1500, seed=7, span=95_000_000, gz=False,
                 with_chr_file=True, extra_targets=0, extra_refs=0):
    """extra_targets / extra_refs: further samples over the same .mut files (T1.colate.in, ..., R1.colate.in, ...), each drawn
    from a generator of its own so that T.colate.in / R.colate.in do not depend on them (batched all-pairs fixtures)."""
    rng = np.random.default_rng(seed)
    os.makedirs(outdir, exist_ok=True)
    xt = [(open(os.path.join(outdir, f"T{k + 1}.colate.in"), "wb"), np.random.default_rng(seed * 1000 + 1 + k))
          for k in range(extra_targets)]
    xr = [(open(os.path.join(outdir, f"R{k + 1}.colate.in"), "wb"), np.random.default_rng(seed * 1000 + 501 + k))
          for k in range(extra_refs)]
    bases = "ACGT"
    tgt = open(os.path.join(outdir, "T.colate.in"), "wb")
    ref = open(os.path.join(outdir, "R.colate.in"), "wb")

    def rec(f, chrom, bp, anc, der, aaf, daf):
        c = chrom.encode()
        f.write(struct.pack("<i", len(c)) + c + struct.pack("<i", bp) + anc.encode() + der.encode()
                + struct.pack("<ii", aaf, daf))

    for chrom in chroms:
        name = chrom if with_chr_file else ""
        pos = np.sort(rng.choice(np.arange(1000, span), size=snps_per_chr, replace=False))
        path = os.path.join(outdir, f"P_chr{chrom}.mut" if with_chr_file else "P.mut")
        opener = (lambda p: gzip.open(p + ".gz", "wt")) if gz else (lambda p: open(p, "w"))
        with opener(path) as f:
            f.write("snp;pos_of_snp;dist;rs-id;tree_index;branch_indices;is_not_mapping;is_flipped;age_begin;age_end;"
                    "ancestral_allele/alternative_allele;upstream_allele;downstream_allele;\n")
            for i, bp in enumerate(pos):
                if rng.uniform() < 0.08:
                    age_begin = 0.0
                else:
                    age_begin = 10 ** rng.uniform(1, 5.2)
                age_end = max(age_begin, 30.0) * (1 + 1.5 * rng.uniform()) if age_begin == 0 else age_begin * (1 + 1.5 * rng.uniform())
                a = bases[rng.integers(4)]
                d = bases[(bases.index(a) + rng.integers(1, 4)) % 4]
                flipped = int(rng.uniform() < 0.03)
                branches = "7" if rng.uniform() > 0.04 else "7 12"
                mtype = f"{a}/{d}" if rng.uniform() > 0.02 else f"{a}{a}/{d}"
                if rng.uniform() < 0.01:
                    age_end = age_begin  # filtered: age_begin < age_end fails
                dist = int(pos[i + 1] - bp) if i + 1 < len(pos) else 1
                f.write(f"{i};{bp};{dist};rs{i};{i // 10};{branches};0;{flipped};{age_begin:.6g};{age_end:.6g};"
                        f"{mtype};{a};{d};\n")
                # reference sample: mostly present with DAF in {0,1,2}; sometimes absent or allele mismatch
                u = rng.uniform()
                if u < 0.9:
                    daf = int(rng.integers(0, 3))
                    ra, rd = (a, d) if rng.uniform() > 0.03 else (d, a)
                    rec(ref, name, int(bp), ra, rd, 2 - daf, daf)
                if rng.uniform() < 0.1:  # extra reference record at a position without a mutation
                    rec(ref, name, int(bp) + 1, "A", "G", 1, 1)
                # target sample: low-coverage style counts (0..4 reads), sometimes absent; it shares the
                # mutation with probability 1 - exp(-age/12000) (pairwise Ne ~ 6000), plus some read noise
                if rng.uniform() < 0.9:
                    n = int(rng.integers(0, 5))
                    age_mid = 0.5 * (age_begin + age_end)
                    shares = rng.uniform() < 0.8 * (1.0 - np.exp(-age_mid / 12000.0))
                    daf = (n if shares else 0) if rng.uniform() > 0.05 else (int(rng.integers(0, n + 1)) if n else 0)
                    rec(tgt, name, int(bp), a, d, n - daf, daf)
                for k, (f2, g2) in enumerate(xr):  # further reference samples
                    if g2.uniform() < 0.85:
                        daf = int(g2.integers(0, 3))
                        ra, rd = (a, d) if g2.uniform() > 0.03 else (d, a)
                        rec(f2, name, int(bp), ra, rd, 2 - daf, daf)
                    if g2.uniform() < 0.1:
                        rec(f2, name, int(bp) + 1, "A", "G", 1, 1)
                for k, (f2, g2) in enumerate(xt):  # further targets: own coverage, own pairwise Ne
                    if g2.uniform() < 0.8:
                        n = int(g2.integers(0, 5))
                        age_mid = 0.5 * (age_begin + age_end)
                        shares = g2.uniform() < 0.8 * (1.0 - np.exp(-age_mid / (7000.0 + 3000.0 * k)))
                        daf = (n if shares else 0) if g2.uniform() > 0.05 else (int(g2.integers(0, n + 1)) if n else 0)
                        rec(f2, name, int(bp), a, d, n - daf, daf)
    tgt.close()
    ref.close()
    for f2, _ in xt + xr:
        f2.close()
    if with_chr_file:
        with open(os.path.join(outdir, "chr.txt"), "w") as f:
            for c in chroms:
                f.write(c + "\n")
    return outdir


if __name__ == "__main__":
    import sys

    write_inputs(sys.argv[1])
