#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box: kernel trace (+ optionally three PMC passes, each on its own);
# output under gpurun_out/<name>/, summarised by profiles/summarize.py.
#   gpurun -- 'tools/profile_bench.sh r02_b100_e23 pmc'                       # BASELINE configs[1]
#   gpurun -- 'tools/profile_bench.sh r02_b1024_e23 nopmc --replicates 1024'  # throughput variant
set -euo pipefail
name=${1:-prof}; pmc=${2:-nopmc}; shift $(( $# > 2 ? 2 : $# ))
R=/root/repo
out=$R/gpurun_out/$name
mkdir -p "$out"
cd /tmp
export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-host-path --no-cxx-rccl-check --no-other-configs $*"
rocprofv3 --kernel-trace --stats -f csv -d "$out/kt" -o runc -- python3 $B > "$out/bench.json" 2> "$out/kt.log"
if [ "$pmc" = pmc ]; then
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    -f csv -d "$out/pmc1" -o runc -- python3 $B --steps 3 --warmup 1 --passes-per-step 1 > /dev/null 2> "$out/pmc1.log"
  rocprofv3 --pmc FETCH_SIZE -f csv -d "$out/pmc2" -o runc -- python3 $B --steps 3 --warmup 1 --passes-per-step 1 > /dev/null 2> "$out/pmc2.log"
  rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY \
    -f csv -d "$out/pmc3" -o runc -- python3 $B --steps 3 --warmup 1 --passes-per-step 1 > /dev/null 2> "$out/pmc3.log"
fi
python3 $R/profiles/summarize.py "$out" "$out/summary.txt" > /dev/null
cut -c1-240 "$out/bench.json"
head -4 "$out/summary.txt"
