#!/bin/bash
# pad sweep on the GPU: kernel ms per pad for four shapes
cd /root/repo
for shape in "" "--bins 2,7.95,0.05" "--replicates 400" "--replicates 4096"; do
  echo "== shape: ${shape:-default (B=100, E=23)}"
  ROUNDS=2 tools/ab_bench.sh colate_amd/lib_pad0/libcolate_amd.so colate_amd/lib_pad1/libcolate_amd.so colate_amd/lib_pad2/libcolate_amd.so colate_amd/lib_pad3/libcolate_amd.so colate_amd/lib_pad4/libcolate_amd.so colate_amd/lib_pad5/libcolate_amd.so colate_amd/lib_pad6/libcolate_amd.so colate_amd/lib_pad7/libcolate_amd.so colate_amd/lib_r03/libcolate_amd.so -- $shape 2>&1 | grep rep/s
done
