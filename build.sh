#!/bin/bash
# builds the product library, the CLI and the diagnostic probe; stops at the first error
set -euo pipefail
cd "$(dirname "$0")"
make -C colate_amd/csrc -j4
( cd colate_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 \
    -mllvm -force-precise-rotation-cost=true -mllvm -amdgpu-sched-strategy=max-ilp \
    -I../../include -I. -Wno-unused-value tools/em_phase_probe.hip -o ../bin/em_phase_probe
  /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 \
    -I../../include -I. -Wno-unused-value tools/residency_probe.hip -o ../bin/residency_probe
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value tools/fetch_calib.hip -o ../bin/fetch_calib
  for u in ubench ubench_branch ubench_ldsatomic ubench_mfma ubench_exec ubench_clock ubench_fetch ubench_fetch2 ubench_fetch3; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-value tools/$u.hip -o ../bin/$u
  done )
make -C oracle oracle
# the loops of the EM kernels where the pads of em_loop_pad() were tuned (tools/loop_offsets.py; a mismatch = re-run tools/pad_sweep.sh)
python3 tools/loop_offsets.py --check profiles/r04_loop_offsets.json | tail -2
echo BUILD OK
