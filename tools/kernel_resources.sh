#!/bin/bash
# Prints VGPRs / spills / scratch / occupancy of every kernel in a .hip file, compiled with the given extra flags.
#   tools/kernel_resources.sh colate_amd/csrc/em_kernels.hip [extra hipcc flags...]
set -euo pipefail
src="$1"; shift
dir="$(cd "$(dirname "$src")" && pwd)"
root="$(cd "$(dirname "$0")/.." && pwd)"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -I"$root/include" -I"$dir" --offload-arch=gfx950 "$@" \
  -Rpass-analysis=kernel-resource-usage -c "$src" -o /dev/null 2>&1 |
  awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name)}
       /    VGPRs:/ {v=$(NF-1)} /AGPRs:/ {a=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /Occupancy/ {o=$(NF-1)}
       /SGPRs Spill/ {ss=$(NF-1)} /VGPRs Spill/ {vs=$(NF-1)}
       /LDS Size/ {cmd="c++filt " name; cmd | getline d; close(cmd);
                   printf "%-60s VGPR %3s AGPR %3s vspill %3s sspill %3s scratch %4s B  occupancy %s waves/SIMD\n", d, v, a, vs, ss, s, o}'
