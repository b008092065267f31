#!/bin/bash
# Round-4 profile set (GPU box): kernel traces + PMC passes for the four single-GPU shapes of BASELINE.json (configs[1], [3], [2] on
# one GPU, [4] = 100 pairs x 20 replicates), kernel traces for two more, the FETCH_SIZE calibration for 8-B-per-lane reads and the
# bootstrap kernel.  Output under gpurun_out/prof_r04/.
#   gpurun --timeout 1100 -- 'tools/profile_all.sh'    then   tools/collect_profiles.sh
set -euo pipefail
R=/root/repo
cd $R
P=prof_r04
tools/profile_bench.sh $P/b100_e23 pmc
tools/profile_bench.sh $P/b100_e122 pmc --bins 2,7.95,0.05
tools/profile_bench.sh $P/b1000_e23 pmc --total-replicates 1000
tools/profile_bench.sh $P/pairs100x20_e23 pmc --pairs 100 --replicates 20
tools/profile_bench.sh $P/b400_e23 nopmc --replicates 400
tools/profile_bench.sh $P/b4096_e23 nopmc --replicates 4096 --steps 10
out=$R/gpurun_out/$P
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -f csv -d $out/calib -o runc -- $R/colate_amd/bin/fetch_calib > $out/calib.log 2>&1
rocprofv3 --kernel-trace --stats -f csv -d $out/bootstrap -o runc -- python3 $R/tools/bench_bootstrap.py 1000 115 > $out/bootstrap.json 2> $out/bootstrap.log
cat $out/bootstrap.json
grep -h read8 $out/calib/*/*counter_collection.csv | head -3
