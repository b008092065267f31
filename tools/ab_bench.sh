#!/bin/bash
# A/B any number of builds of libcolate_amd.so on the same GPU box, alternating, ROUNDS (default 3) rounds (device clocks differ by
# several per cent between boxes, so only same-box comparisons mean anything):
#   tools/ab_bench.sh lib1.so lib2.so ... -- [bench args]
# The builds are selected through COLATE_AMD_LIB (colate_amd/_lib.py); the product library is never overwritten.
set -euo pipefail
libs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done; [ "${1:-}" = "--" ] && shift
for r in $(seq 1 ${ROUNDS:-3}); do
  for l in "${libs[@]}"; do
    COLATE_AMD_LIB="$(realpath "$l")" python3 bench.py --no-cpu-baseline --no-host-path --no-cxx-rccl-check --steps 30 --warmup 5 --passes-per-step 20 --no-other-configs "$@" |
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', round(d['value']), 'rep/s incl. copy+sync;', round(d['roofline']['kernel_ms'],4),'ms kernel;', d['roofline']['kernel_build'])"
  done
done
