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
    return out


def _kernel_source_sha16():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    return bench.kernel_source_sha16()


def pmc_entry(src, fetch_factor, note):
    """One entry of profiles/pmc.json (what bench.py puts into `roofline.traffic` / `roofline.latency.pmc`) from a
    profile directory that has the three PMC passes.  fetch_factor = bytes per FETCH_SIZE-KB/1024 measured by
    csrc/tools/fetch_calib.hip for the kernel's 8-B-per-lane reads."""
    import json

    bench = json.load(open(os.path.join(src, "bench.json")))
    c = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, "pmc*", "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "em_kernel" in r["Kernel_Name"]:
                c[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in c.items()}
    cfg = bench["config"]
    iters = cfg["em_iterations_mean"] + 1
    waves = m["SQ_WAVES"]
    workload = {"replicates": cfg["replicates_rank0"], "epochs": cfg["epochs"], "age_bins": cfg["age_bins"]}
    if "pairs" in cfg:  # (bench.py --pairs: the EM launch has pairs x replicates rows with per-row epochs)
        workload = {"pairs": cfg["pairs"], "replicates_per_pair": cfg["replicates_per_pair"], "rows": cfg["replicates_rank0"],
                    "epochs": cfg["epochs"], "age_bins": cfg["age_bins"]}
    return {
        "workload": workload,
        # the kernel sources the counters were taken on (bench.py prices the VALU line with this record only while they are the tree's)
        # (recomputed from the tree at collection time: tools/collect_profiles.sh runs on the sources the profile was taken on)
        "kernel_source_sha16": _kernel_source_sha16(),
        "kernel": bench["roofline"]["kernel"], "source": note,
        "hbm_bytes_per_launch": m["FETCH_SIZE"] * 1024 * fetch_factor + m["WRITE_SIZE"] * 1024,
        "fetch_size_kb": m["FETCH_SIZE"], "write_size_kb": m["WRITE_SIZE"], "fetch_factor_8B_per_lane": fetch_factor,
        "valu_active_frac_of_wave_cycles": m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"],
        "wait_any_frac_of_wave_cycles": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
        "wait_inst_any_frac_of_wave_cycles": m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"],
        "waves_per_launch": waves,
        # what really bounds the kernel (bench.py's top-level `roofline`): the share of the chip's vector-issue cycles in
        # which a VALU instruction was executing.  SQ_ACTIVE_INST_VALU counts quad-cycles summed over waves; the kernel
        # lasted GRBM_GUI_ACTIVE / 8 clocks (the counter is summed over the 8 XCDs); 256 CUs x 4 SIMDs can be busy.
        "sq_active_inst_valu": m["SQ_ACTIVE_INST_VALU"], "sq_wave_cycles": m["SQ_WAVE_CYCLES"],
        "grbm_gui_active": m["GRBM_GUI_ACTIVE"],
        "valu_busy_frac_chip": 4.0 * m["SQ_ACTIVE_INST_VALU"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0),
        "valu_insts_per_wave_per_em_iteration": m["SQ_INSTS_VALU"] / waves / iters,
        "salu_insts_per_wave_per_em_iteration": m["SQ_INSTS_SALU"] / waves / iters,
        "lds_insts_per_wave_per_em_iteration": m["SQ_INSTS_LDS"] / waves / iters,
        "units": "SQ_* cycle counters are quad-cycles (guide: s_memtime tick vs SQ PMC units); fractions are ratios of like units",
    }


if __name__ == "__main__":
    if sys.argv[1] == "--pmc-entry":  # summarize.py --pmc-entry SRC FETCH_FACTOR NOTE  -> JSON on stdout
        import json

        print(json.dumps(pmc_entry(sys.argv[2], float(sys.argv[3]), sys.argv[4]), indent=1))
    else:
        main(sys.argv[1], sys.argv[2])
