#!/bin/bash
# End-to-end at realistic input size (GPU box): 22 chromosomes x N .mut.gz rows (tools/gen_wg_inputs.cpp; N = 1e6 gives
# ~540 MB of .mut.gz, the size of the reference's example data, README.md:52), the same command line through the reference
# binary and through this repo's Colate, with the stage times of ours (COLATE_TIMING=1) and a comparison of the .coal files.
#   gpurun --timeout 1100 -- 'tools/e2e_large.sh 1000000 > gpurun_out/e2e_large.txt 2>&1'
set -euo pipefail
N=${1:-1000000}; B=${2:-100}
R="$(cd "$(dirname "$0")/.." && pwd)"
d=$(mktemp -d /tmp/e2e_large.XXXX)
g++ -O2 -std=c++17 "$R/tools/gen_wg_inputs.cpp" -lz -o "$d/gen"
t0=$(date +%s.%N); "$d/gen" "$d" 22 "$N" gz; t1=$(date +%s.%N)
echo "inputs: 22 x $N rows, $(du -sh "$d" | cut -f1) on disk, generated in $(awk "BEGIN {printf \"%.2f\", $t1 - $t0}") s"
args="--mode mut --mut P --target_tmp T.colate.in --reference_tmp R.colate.in --chr chr.txt --bins 3,7,0.2 --seed 1 --num_bootstraps $B"
cd "$d"
for run in 1 2; do
  t0=$(date +%s.%N)
  COLATE_TIMING=1 "$R/colate_amd/bin/Colate" $args -o ours > ours.out 2> ours.err
  t1=$(date +%s.%N)
  echo "colate_amd run $run: $(awk "BEGIN {printf \"%.2f\", $t1 - $t0}") s wall; $(grep '^Timing' ours.err)"
done
grep -c '^Bootstrap' ours.err | sed 's/^/colate_amd: bootstrap lines /'
if [ -x "$R/oracle/_ref/Colate_ref" ]; then
  t0=$(date +%s.%N)
  "$R/oracle/_ref/Colate_ref" $args -o ref > ref.out 2> ref.err
  t1=$(date +%s.%N)
  echo "reference: $(awk "BEGIN {printf \"%.2f\", $t1 - $t0}") s wall"
  python3 - <<'PY'
import numpy as np
a = open("ours.coal").read().split("\n"); b = open("ref.coal").read().split("\n")
print("header lines identical:", a[:2] == b[:2], "| rows:", len(a), len(b))
ra = np.array([[float(x) for x in l.split()[2:]] for l in a[2:] if l]); rb = np.array([[float(x) for x in l.split()[2:]] for l in b[2:] if l])
same = (ra == rb)
rel = np.abs(ra - rb) / np.maximum(np.abs(rb), 1e-300)
print(f"6-digit rates identical in {same.mean() * 100:.3f} % of {same.size} entries; max relative difference {rel.max():.2e}")
PY
  ia=$(grep '^Bootstrap' ours.err | awk '{print $NF}' | tr '\n' ' '); ib=$(tr '\r' '\n' < ref.err | grep '^Bootstrap' | awk '{k=$2; v[k]=$NF} END {for (i=1;i<=length(v);i++) printf "%s ", v[i":"]}')
  [ "$ia" = "$ib" ] && echo "iteration counts identical" || echo "iteration counts differ"
fi
rm -rf "$d"
