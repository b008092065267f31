#!/bin/bash
# Builds libcolate_amd.so eight times with the EM loop shifted by 0..7 dwords behind its 64-byte boundary
# (-DCOLATE_LOOP_PAD=n for every instantiation) into colate_amd/lib_<prefix><n>/, for tools/ab_bench.sh on the GPU box:
#   tools/pad_sweep.sh [prefix [extra hipcc flags]] && gpurun -- 'tools/ab_bench.sh colate_amd/lib_pad*/libcolate_amd.so -- --replicates 4096'
# The winners go into em_loop_pad() (colate_amd/csrc/em_kernel_impl.hpp).
set -euo pipefail
prefix=${1:-pad}; shift || true
cd "$(dirname "$0")/../colate_amd/csrc"
F="-O3 -std=c++17 -fPIC -ffp-contract=off -I$(cd ../..; pwd)/include --offload-arch=gfx950 -mllvm -force-precise-rotation-cost=true"
make > /dev/null
for n in 0 1 2 3 4 5 6 7; do
  /opt/rocm/bin/hipcc $F "$@" -mllvm -amdgpu-sched-strategy=max-ilp -DCOLATE_LOOP_PAD=$n -c em_kernels_ilp.hip -o /tmp/ilp_$prefix$n.o &
  /opt/rocm/bin/hipcc $F "$@" -DCOLATE_LOOP_PAD=$n -c em_kernels.hip -o /tmp/em_$prefix$n.o &
  wait
  mkdir -p ../lib_$prefix$n
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_$prefix$n/libcolate_amd.so /tmp/em_$prefix$n.o /tmp/ilp_$prefix$n.o em_kernels_big.o \
    bootstrap_kernel.o colate_api.o colate_comm.o mut_host.o mut_driver.o mut_pairs.o -lz -ldl
  echo "built lib_$prefix$n"
done
