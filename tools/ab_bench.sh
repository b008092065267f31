#!/bin/bash
# A/B two builds of libcolate_amd.so on the same box, alternating, N rounds:
#   tools/ab_bench.sh colate_amd/lib_a/libcolate_amd.so colate_amd/lib_b/libcolate_amd.so [bench args...]
set -e
A=$1; B=$2; shift 2
cp colate_amd/lib/libcolate_amd.so /tmp/orig.so
for r in 1 2 3 4; do
  for v in A B; do
    if [ $v = A ]; then cp $A colate_amd/lib/libcolate_amd.so; else cp $B colate_amd/lib/libcolate_amd.so; fi
    python bench.py --no-cpu-baseline --no-host-path --steps 30 --warmup 5 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), 'rep/s', round(d['roofline']['kernel_ms'],4),'ms')"
  done
done
cp /tmp/orig.so colate_amd/lib/libcolate_amd.so
