#!/bin/bash
# The parity evidence committed under profiles/parity/ (GPU box; ~6 min of oracle time on 14 host processes):
#   gpurun --timeout 1100 -- 'tools/parity_all.sh'     then   cp gpurun_out/parity/*.json profiles/parity/
set -euo pipefail
cd "$(dirname "$0")/.."
out=gpurun_out/parity
mkdir -p $out
S="python3 tools/parity_sweep.py"
$S $out/wg_e23_modern.json      256 11 3,7,0.2                 # BASELINE configs[1]-shaped
$S $out/chr1_e23_modern.json    128 1  3,7,0.2                 # configs[0]-shaped
$S $out/wg_e23_ancient7000.json 128 11 3,7,0.2 7000            # ancient sample (LBK-like age)
$S $out/wg_e122_modern.json     64  11 2,7.95,0.05             # configs[3]
$S $out/wg_e23_smallne.json     64  11 3,7,0.2 0 2000          # small Ne: everything coalesces early
$S $out/sparse_e23.json         192 0  3,7,0.2                 # low-coverage-like tables (1001 .. 70000 iterations)
$S $out/sparse_e43.json         96  0  3,7,0.1
$S $out/sparse_e122.json        48  0  2,7.95,0.05
$S $out/sparse_e23_ancient30000.json 96 0 3,7,0.2 30000
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/parity/*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], d["iterations"], "checker", d["checker"]["stable_fraction"], d["checker"]["max_rel_diff_on_stable"],
          "kernel", d["kernel"]["resolved_fraction"], d["kernel"]["max_rel_diff_on_resolved"], d["kernel"]["entries_beyond_1e-6_on_resolved"],
          d["kernel"]["replicates_flagging_fewer_epochs_than_checker"], d["kernel"]["max_extra_epochs_flagged_vs_checker"])
PY
