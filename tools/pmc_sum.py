#!/usr/bin/env python3
"""Average counter values per (kernel, grid) from a rocprofv3 --pmc counter_collection.csv, for kernels whose name contains a substring:
    python3 tools/pmc_sum.py <counter_collection.csv> <substring>"""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if sys.argv[2] not in n:
        continue
    k = (n.split("(")[0][-60:], r["Grid_Size"])
    e = acc[k][r["Counter_Name"]]
    e[0] += 1
    e[1] += float(r["Counter_Value"])
for k, d in acc.items():
    print(k, {c: round(v[1] / v[0]) for c, v in d.items()})
