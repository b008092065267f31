#!/bin/bash
# The bench records committed under profiles/r04/bench/ (GPU box, one GPU):
#   gpurun --timeout 1100 -- 'tools/bench_records.sh'   then   cp gpurun_out/bench_records/* profiles/r04/bench/
# bench_default.json    python bench.py                                  (the driver's N = 1 line: configs[1], cpu_baseline = reference binary)
# bench_e122.json       --bins 2,7.95,0.05                               (configs[3]; differing .coal tokens per epoch against the reference binary)
# bench_strong1000.json --total-replicates 1000                          (configs[2] on one GPU)
# bench_pairs100x20.json --pairs 100 --replicates 20                     (configs[4] on one GPU: 2000 rows, per-row epochs, bootstrap + EM per pass)
# bench_gloo2.json      two ranks on the one GPU through torch.distributed.run, gloo (the launcher path of --gpus N)
# e2e.txt               tools/e2e_compare.py: Colate of this repo against the reference binary on the same input files
# ab_final.txt          same-box A/B against round 3's library, every shape DESIGN.md quotes
set -uo pipefail
cd "$(dirname "$0")/.."
out=gpurun_out/bench_records
mkdir -p $out
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err && echo default ok
python3 bench.py --bins 2,7.95,0.05 --no-other-configs > $out/bench_e122.json 2> $out/bench_e122.err && echo e122 ok
python3 bench.py --total-replicates 1000 --no-cpu-baseline --no-other-configs > $out/bench_strong1000.json 2> $out/bench_strong1000.err && echo strong ok
python3 bench.py --pairs 100 --replicates 20 --no-other-configs > $out/bench_pairs100x20.json 2> $out/bench_pairs100x20.err && echo pairs ok
COLATE_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --no-cpu-baseline --passes-per-step 50 2> $out/bench_gloo2.err | grep '^{' > $out/bench_gloo2.json && echo gloo2 ok
if [ -x oracle/_ref/Colate_ref ]; then python3 tools/e2e_compare.py > $out/e2e.txt 2> $out/e2e.err && echo e2e ok; fi
L="colate_amd/lib_r03/libcolate_amd.so colate_amd/lib/libcolate_amd.so"
( echo "== B=100"; tools/ab_bench.sh $L --
  for b in 400 512 1024 4096; do echo "== B=$b"; tools/ab_bench.sh $L -- --replicates $b; done
  echo "== E=122"; tools/ab_bench.sh $L -- --bins 2,7.95,0.05 ) 2>/dev/null | grep -v amdgpu.ids > $out/ab_final.txt
tail -4 $out/ab_final.txt
