#!/usr/bin/env python3
"""Per shape and library: the kernel times of tools/ab_bench.sh runs (sections start with '== <shape>').
   tools/pad_summary.py gpurun_out/<dir>/padsweep.txt"""
import collections
import sys

sec = None
acc = collections.OrderedDict()
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("=="):
        sec = line
        continue
    p = line.split()
    if "ms" not in p:
        continue
    acc.setdefault((sec, p[0].split("/")[1], p[-1]), []).append(float(p[p.index("ms") - 1]))
for (s, lib, b), v in acc.items():
    print(s, lib, b, " ".join("%.4f" % x for x in v))
