#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per dispatch for kernels whose name contains a substring.
usage: pmc_summary.py <dir> <kernel-substring>"""
import collections, csv, glob, os, sys
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-32s %16.0f  (n=%d)" % (k, sum(v) / len(v), len(v)))
