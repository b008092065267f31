#!/bin/bash
# pad sweep on the GPU: kernel ms per pad for the shapes given as SHAPES (default: the four of round 4), the library in
# colate_amd/lib_r04base (the committed build) and round 3's beside them
cd /root/repo
IFS='|' read -ra shapes <<< "${SHAPES:-|--bins 2,7.95,0.05|--replicates 400|--replicates 4096}"
for shape in "${shapes[@]}"; do
  echo "== shape: ${shape:-default (B=100, E=23)}"
  ROUNDS=${ROUNDS:-2} tools/ab_bench.sh colate_amd/lib_pad0/libcolate_amd.so colate_amd/lib_pad1/libcolate_amd.so colate_amd/lib_pad2/libcolate_amd.so colate_amd/lib_pad3/libcolate_amd.so colate_amd/lib_pad4/libcolate_amd.so colate_amd/lib_pad5/libcolate_amd.so colate_amd/lib_pad6/libcolate_amd.so colate_amd/lib_pad7/libcolate_amd.so $(ls colate_amd/lib_r04base/libcolate_amd.so 2>/dev/null) colate_amd/lib_r03/libcolate_amd.so -- $shape 2>&1 | grep rep/s
done
