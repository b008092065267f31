#!/usr/bin/env python3
"""Condenses tools/ab_bench.sh output (sections `== name`, one line per run) into kernel ms per library and section."""
import collections
import re
import sys

sec, d = None, collections.defaultdict(list)
for l in open(sys.argv[1]):
    if l.startswith("=="):
        sec = l.strip()
        continue
    m = re.match(r"colate_amd/(\S+?)/\S+ (\d+) rep/s.*?; ([\d.]+) ms", l)
    if m:
        d[(sec, m.group(1))].append(float(m.group(3)))
for k in sorted(d, key=lambda k: (k[0], min(d[k]))):
    print(k[0], k[1], " ".join("%.4f" % x for x in d[k]))
