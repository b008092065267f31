#!/bin/bash
# BASELINE configs[4] end to end (GPU box): NT targets x NR references over 22 chromosomes x N .mut.gz rows
# (tools/gen_wg_inputs.cpp; N = 1e6 is the size of the reference's example data), B bootstrap replicates per pair.
#   1. `Colate --pairs` of this repo: all pairs in one process -- inputs read once, pairs filled in parallel, one bootstrap
#      launch + one EM launch (COLATE_TIMING=1 gives the stage times);
#   2. the reference binary, once per pair (it has no list mode), PAR runs side by side;
#   3. every pair's .coal compared byte for byte, and the iteration counts;
#   4. round 3's --pairs loop (colate_amd/lib_r03: .colate.in re-read per pair, pairs filled one after the other, host
#      bootstrap) on the first OLD pairs, for the before/after of the host part.
#   gpurun --timeout 1150 -- 'tools/e2e_pairs.sh 1000000 10 10 20 16 8 > gpurun_out/pairs100.txt 2>&1'
set -euo pipefail
N=${1:-1000000}; NT=${2:-10}; NR=${3:-10}; B=${4:-20}; PAR=${5:-16}; OLD=${6:-8}
R="$(cd "$(dirname "$0")/.." && pwd)"
d=$(mktemp -d /tmp/e2e_pairs.XXXX)
trap 'rm -rf "$d"' EXIT
now() { date +%s.%N; }
since() { awk "BEGIN {printf \"%.2f\", $(now) - $1}"; }
g++ -O2 -std=c++17 "$R/tools/gen_wg_inputs.cpp" -lz -o "$d/gen"
t0=$(now); "$d/gen" "$d" 22 "$N" gz "$NT" "$NR"
P=$(wc -l < "$d/pairs.txt")
echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null || echo n/a); affinity: $(python3 -c 'import os; print(len(os.sched_getaffinity(0)))')"
echo "host: $(nproc) cores, $(awk '/MemTotal/ {printf "%.0f GB", $2 / 1e6}' /proc/meminfo); inputs: 22 x $N rows, $NT targets x $NR references = $P pairs, $(du -sh "$d" | cut -f1) on disk, generated in $(since $t0) s"
common="--mode mut --mut P --chr chr.txt --bins 3,7,0.2 --seed 1 --num_bootstraps $B"
cd "$d"

# (round 4, last: the age sampling of the table fill runs on the GPU by default -- fill_device.h --; `host` = COLATE_DEVICE_FILL=0.
#  The last run is the one whose .coal files are compared with the reference's.)
for run in ${RUNS:-host:default device:32 device:default}; do
  fill=${run%%:*}; threads=${run##*:}
  t0=$(now)
  if [ "$threads" = default ]; then unset COLATE_THREADS; else export COLATE_THREADS=$threads; fi
  if [ "$fill" = host ]; then export COLATE_DEVICE_FILL=0; else unset COLATE_DEVICE_FILL; fi
  case "$fill" in device[0-9]*) export COLATE_UNIFORM_WINDOW_MB=${fill#device};; *) unset COLATE_UNIFORM_WINDOW_MB;; esac  # (deviceNNN: stream windows of NNN MB)
  COLATE_TIMING=1 ${PIN:+taskset -c $PIN} "$R/colate_amd/bin/Colate" $common --pairs pairs.txt > ours.out 2> ours.err
  echo "colate_amd --pairs (age sampling on the $fill, COLATE_THREADS=$threads${PIN:+, taskset -c $PIN}): $(since $t0) s wall for $P pairs x $B replicates"
  grep '^Timing' ours.err | sed 's/^/    /'
done
unset COLATE_THREADS COLATE_DEVICE_FILL COLATE_UNIFORM_WINDOW_MB
grep '^CPU Time' ours.err | sed 's/^/    /'

if [ -x "$R/oracle/_ref/Colate_ref" ] && [ -z "${SKIP_REF:-}" ]; then
  t0=$(now)
  run_ref() {  # one reference run: target reference output
    local s=$(date +%s.%N)
    "$R/oracle/_ref/Colate_ref" $common --target_tmp "$1" --reference_tmp "$2" -o "ref_$3" > "ref_$3.out" 2> "ref_$3.err"
    echo "$3 $(awk "BEGIN {printf \"%.2f\", $(date +%s.%N) - $s}")" >> ref_times.txt
  }
  export -f run_ref; export R common
  : > ref_times.txt
  xargs -P "$PAR" -L 1 bash -c 'run_ref "$0" "$1" "$2"' < pairs.txt
  wall=$(since $t0)
  echo "reference, one run per pair, $PAR side by side: $wall s wall; sum of the runs $(awk '{s += $2} END {printf "%.1f", s}' ref_times.txt) s (mean $(awk '{s += $2} END {printf "%.2f", s / NR}' ref_times.txt) s per pair; sequentially that sum is the wall time)"
  same=0; diff=0; itsame=0
  while read -r t r o; do
    if cmp -s "$o.coal" "ref_$o.coal"; then same=$((same + 1)); else diff=$((diff + 1)); python3 - "$o" <<'PY'
import sys
o = sys.argv[1]
a = open(o + ".coal").read().split("\n"); b = open("ref_" + o + ".coal").read().split("\n")
bad = []
for i, (x, y) in enumerate(zip(a, b)):
    for j, (u, v) in enumerate(zip(x.split(), y.split())):
        if u != v:
            bad.append(f"line {i} token {j}: ours {u} reference {v} (rel {abs(float(u) - float(v)) / max(abs(float(v)), 1e-300):.1e})")
print(f"    DIFFERENT: {o}: {len(bad)} token(s) of {sum(len(x.split()) for x in b)}: " + "; ".join(bad[:6]), "| lines", len(a), len(b))
PY
    fi
    k=$(awk -v o="$o" '$1 == o {print NR}' <(awk '{print $3}' pairs.txt))
    a=$(grep "^Pair $k Bootstrap" ours.err | awk '{print $NF}' | tr '\n' ' ')
    b=$(tr '\r' '\n' < "ref_$o.err" | grep '^Bootstrap.*Total iterations' | awk '{k = $2; v[k] = $NF} END {for (i = 1; i <= length(v); i++) printf "%s ", v[i":"]}')
    [ "$a" = "$b" ] && itsame=$((itsame + 1))
  done < pairs.txt
  echo ".coal files byte-identical to the reference's: $same of $P (different: $diff); iteration counts identical: $itsame of $P"
  echo "iterations over all pairs and replicates: min $(grep ' Bootstrap ' ours.err | awk '{print $NF}' | sort -n | head -1), max $(grep ' Bootstrap ' ours.err | awk '{print $NF}' | sort -n | tail -1)"
fi

if [ -f "$R/colate_amd/lib_r03/libcolate_amd.so" ] && [ "$OLD" -gt 0 ]; then
  head -n "$OLD" pairs.txt | awk '{print $1, $2, "old_" $3}' > old_pairs.txt
  t0=$(now)
  LD_LIBRARY_PATH="$R/colate_amd/lib_r03" COLATE_TIMING=1 "$R/colate_amd/bin/Colate" $common --pairs old_pairs.txt > old.out 2> old.err || tail -5 old.err
  w=$(since $t0)
  echo "round 3's --pairs loop (lib_r03) on the first $OLD pairs: $w s wall = $(awk "BEGIN {printf \"%.2f\", $w / $OLD}") s per pair -> $(awk "BEGIN {printf \"%.0f\", $w / $OLD * $P}") s for $P pairs"
  ok=0; while read -r t r o; do cmp -s "$o.coal" "${o#old_}.coal" && ok=$((ok + 1)); done < old_pairs.txt
  echo "    its .coal files equal this round's: $ok of $OLD"
fi
