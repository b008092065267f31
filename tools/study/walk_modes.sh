#!/bin/bash
# Why do the SNP walks of `Colate --pairs` take 28 thread-seconds in one run and 68 in the next (same inputs, same box)?  Kernel counters
# around each of several runs: NUMA hinting faults / migrations, THP allocations, page faults.   gpurun -- 'tools/study/walk_modes.sh'
set -euo pipefail
R="$(cd "$(dirname "$0")/../.." && pwd)"
d=$(mktemp -d /tmp/walk_modes.XXXX); trap 'rm -rf "$d"' EXIT
g++ -O2 -std=c++17 "$R/tools/gen_wg_inputs.cpp" -lz -o "$d/gen"; "$d/gen" "$d" 22 ${1:-1000000} gz 10 10 > /dev/null
echo "THP: $(cat /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null); defrag: $(cat /sys/kernel/mm/transparent_hugepage/defrag 2>/dev/null); numa_balancing: $(cat /proc/sys/kernel/numa_balancing 2>/dev/null)"
cd "$d"
snap() { grep -E "^(numa_hint_faults|numa_hint_faults_local|numa_pages_migrated|thp_fault_alloc|thp_fault_fallback|pgfault|pgmajfault|numa_pte_updates) " /proc/vmstat | tr '\n' ' '; }
for i in 1 2 3 4 5 6; do
  a=$(snap)
  COLATE_TIMING=1 "$R/colate_amd/bin/Colate" --mode mut --mut P --chr chr.txt --bins 3,7,0.2 --seed 1 --num_bootstraps 20 --pairs pairs.txt > o.out 2> o.err
  b=$(snap)
  echo "run $i: $(grep 'Timing: pairs' o.err | sed 's/.*SNP walks \([0-9.]*\).*/walks \1 thread-s/'); $(grep '^CPU Time' o.err)"
  python3 - "$a" "$b" <<'PY'
import sys
a = sys.argv[1].split(); b = sys.argv[2].split()
A = dict(zip(a[0::2], map(int, a[1::2]))); B = dict(zip(b[0::2], map(int, b[1::2])))
print("    deltas (whole host):", {k: B[k] - A[k] for k in A})
PY
done
