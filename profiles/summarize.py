#!/usr/bin/env python3
"""Summarise rocprofv3 output directories (kernel-trace --stats and --pmc passes, CSV format)
into one small text file per profile, for committing under profiles/.

    python profiles/summarize.py gpurun_out/prof_X profiles/r01_X/summary.txt
"""
import collections
import csv
import glob
import os
import sys


def main(src, dst):
    out = []
    for f in sorted(glob.glob(os.path.join(src, "**", "*_kernel_stats.csv"), recursive=True)):
        out.append(f"## kernel stats ({os.path.relpath(f, src)})")
        for r in csv.DictReader(open(f)):
            out.append(f"{r['Name'][:80]:80s} calls={r['Calls']:>4s} avg_ns={float(r['AverageNs']):.0f} "
                       f"min_ns={r['MinNs']} max_ns={r['MaxNs']} pct={r['Percentage']}")
    for f in sorted(glob.glob(os.path.join(src, "**", "*_counter_collection.csv"), recursive=True)):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(list)
        meta = {}
        for r in rows:
            if "em_kernel" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
                meta[r["Kernel_Name"][:60]] = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count",
                                                                      "LDS_Block_Size", "Workgroup_Size", "Grid_Size")}
        out.append(f"## PMC ({os.path.relpath(f, src)}) -- mean per dispatch of the EM kernel")
        for (k, c), v in sorted(agg.items()):
            out.append(f"{k:60s} {c:24s} n={len(v):3d} mean={sum(v) / len(v):.6g}")
        for k, m in meta.items():
            out.append(f"{k:60s} {m}")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    open(dst, "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
