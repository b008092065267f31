#!/bin/bash
# Timing-only ablations of the EM iteration (csrc/tools/em_phase_probe.hip, -DCOLATE_ABL=<mask>: each bit removes one piece,
# results are garbage, the kernel time shows what the piece really costs on the chain).  Builds here (no GPU needed):
#   tools/ablate_build.sh            -> colate_amd/bin/abl_<name>
# and on the GPU box:  for f in colate_amd/bin/abl_*; do echo $f; $f 100 23; $f 100 122; done
set -euo pipefail
cd "$(dirname "$0")/../colate_amd/csrc"
F="-O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -mllvm -force-precise-rotation-cost=true -mllvm -amdgpu-sched-strategy=max-ilp -I../../include -I. -Wno-unused-value -DCOLATE_NO_STAMPS"
build() { /opt/rocm/bin/hipcc $F -DCOLATE_ABL=$2 tools/em_phase_probe.hip -o ../bin/abl_$1 2>/dev/null & }
build base 0
build nodiv_mstep $((1<<10))
build no_mstep $((1<<14))
build nodiv_p1B $((1<<7))
wait
build nodiv_binA $((1<<8))
build noscan $((1<<9))
build nosegred $((1<<2))
build notail $((1<<1))
wait
build nocsscan $((1<<11))
build nobinmath $((1<<12))
build noP3 $((1<<13))
build noP1 $((1<<15))
wait
build noexp_all $(( (1<<3)|(1<<4)|(1<<5)|(1<<6) ))
wait
ls -la ../bin/abl_*
