#!/bin/bash
# like ab_bench.sh for any number of builds: tools/abn_bench.sh lib1.so lib2.so ... -- [bench args]
set -e
libs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done; [ "${1:-}" = "--" ] && shift
cp colate_amd/lib/libcolate_amd.so /tmp/orig.so
for r in 1 2 3; do
  for l in "${libs[@]}"; do
    cp $l colate_amd/lib/libcolate_amd.so
    python bench.py --no-cpu-baseline --no-host-path --steps 30 --warmup 5 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', round(d['value']), 'rep/s', round(d['roofline']['kernel_ms'],4),'ms')"
  done
done
cp /tmp/orig.so colate_amd/lib/libcolate_amd.so
