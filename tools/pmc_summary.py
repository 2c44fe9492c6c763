#!/usr/bin/env python
"""Average rocprofv3 PMC counters per dispatch of one kernel.  python tools/pmc_summary.py <dir> <kernel substring>"""
import collections
import csv
import glob
import os
import sys

d, key = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (disp, name), v in per.items():
        acc[name].append(v)
for name, vs in sorted(acc.items()):
    vs = vs[1:] if len(vs) > 1 else vs          # drop the first (cold) dispatch
    print(f"{name:32s} {sum(vs) / len(vs):18.1f}   (n={len(vs)})")
