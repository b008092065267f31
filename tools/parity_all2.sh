#!/bin/bash
# A second, larger parity sweep with other replicate counts (= other seeds of colate_amd/workloads.py) than
# tools/parity_all.sh:   gpurun --timeout 1190 -- 'tools/parity_all2.sh'   then   cp gpurun_out/parity2/*.json profiles/parity/
set -euo pipefail
cd "$(dirname "$0")/.."
out=gpurun_out/parity2
mkdir -p $out
S="python3 tools/parity_sweep.py"
$S $out/sweep2_wg_e23_modern.json      512 11 3,7,0.2
$S $out/sweep2_chr1_e23_modern.json    256 1  3,7,0.2
$S $out/sweep2_wg_e23_ancient7000.json 256 11 3,7,0.2 7000
$S $out/sweep2_wg_e122_modern.json     96  11 2,7.95,0.05
$S $out/sweep2_wg_e43_largene.json     128 11 3,7,0.1 0 40000
$S $out/sweep2_sparse_e23.json         512 0  3,7,0.2
$S $out/sweep2_sparse_e43.json         256 0  3,7,0.1
$S $out/sweep2_sparse_e23_ancient30000.json 256 0 3,7,0.2 30000
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/parity2/*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], d["config"]["replicates"], d["iterations"], "checker", round(d["checker"]["stable_fraction"], 4), d["checker"]["max_rel_diff_on_stable"],
          "kernel", round(d["kernel"]["resolved_fraction"], 4), d["kernel"]["max_rel_diff_on_resolved"], d["kernel"]["entries_beyond_1e-6_on_resolved"],
          d["kernel"]["replicates_flagging_fewer_epochs_than_checker"], d["kernel"]["max_extra_epochs_flagged_vs_checker"])
PY
