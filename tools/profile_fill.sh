#!/bin/bash
# rocprofv3 kernel trace of `Colate --pairs` at BASELINE configs[4] size with the age sampling on the device (GPU box):
#   gpurun --timeout 900 -- 'tools/profile_fill.sh'      then   cp gpurun_out/prof_fill/summary.txt gpurun_out/prof_fill/kt/runc_kernel_stats.csv profiles/r04/fill_pairs100/
set -euo pipefail
N=${1:-1000000}; NT=${2:-10}; NR=${3:-10}; B=${4:-20}
R="$(cd "$(dirname "$0")/.." && pwd)"
out=$R/gpurun_out/prof_fill
mkdir -p $out
d=$(mktemp -d /tmp/prof_fill.XXXX)
trap 'rm -rf "$d"' EXIT
g++ -O2 -std=c++17 "$R/tools/gen_wg_inputs.cpp" -lz -o "$d/gen"
"$d/gen" "$d" 22 "$N" gz "$NT" "$NR" > /dev/null
cd "$d"
export TMPDIR=/tmp COLATE_TIMING=1
rocprofv3 --kernel-trace --stats -f csv -d $out/kt -o runc -- "$R/colate_amd/bin/Colate" --mode mut --mut P --chr chr.txt --bins 3,7,0.2 --seed 1 --num_bootstraps $B --pairs pairs.txt > $out/run.out 2> $out/run.err
( echo "# rocprofv3 --kernel-trace --stats -- Colate --mode mut --pairs pairs.txt ($NT x $NR pairs, 22 x $N rows, $B replicates per pair)"
  grep '^Timing' $out/run.err
  echo "## kernel stats (kt/runc_kernel_stats.csv)"
  python3 - "$out/kt/runc_kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-90s calls=%6s avg_ns=%12s min_ns=%12s max_ns=%12s total_ns=%14s pct=%s" % (r["Name"][:90], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["TotalDurationNs"], r["Percentage"]))
PY
) > $out/summary.txt
cat $out/summary.txt
