#!/bin/bash
# Kernel time of each build of the EM kernel (COLATE_EM_VARIANT) over batch sizes, to place the thresholds of
# colate_em_variant() (colate_amd/csrc/em_kernels.hip):   gpurun -- 'tools/variant_sweep.sh [bins]'
set -uo pipefail
cd "$(dirname "$0")/.."
bins=${1:-3,7,0.2}
for b in 128 256 320 400 512 640 768 1024 1536 2048 4096; do
  for v in latency-ilp latency throughput; do
    COLATE_EM_VARIANT=$v python3 bench.py --no-cpu-baseline --no-host-path --no-cxx-rccl-check --steps 20 --warmup 3 --replicates $b --bins $bins 2>/dev/null |
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$b', '$v', round(d['roofline']['kernel_ms'],4), 'ms', round(d['value']), 'rep/s')"
  done
done
