#!/usr/bin/env python
"""Condense a rocprofv3 `--kernel-trace --stats --output-format csv` run into a short text table
(kernel names truncated) for profiles/.   python tools/prof_summary.py <dir> [prefix]"""
import csv
import glob
import os
import sys


def short(name: str, n: int = 72) -> str:
    name = name.split("(")[0] if not name.startswith("void ") else name[5:].split("<")[0]
    return name[:n]


def main():
    d = sys.argv[1]
    out = []
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
        out.append(f"# {os.path.basename(f)}")
        out.append(f"{'kernel':72s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>7s}")
        for r in csv.DictReader(open(f)):
            out.append(f"{short(r['Name']):72s} {r['Calls']:>6s} {float(r['AverageNs']) / 1e3:10.2f} "
                       f"{float(r['MinNs']) / 1e3:10.2f} {float(r['MaxNs']) / 1e3:10.2f} {float(r['Percentage']):7.2f}")
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
        seen = {}
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k not in seen:
                seen[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"],
                           r["Workgroup_Size_X"], r["Grid_Size_X"])
        out.append(f"# {os.path.basename(f)}: kernel  vgpr agpr sgpr lds_bytes wg grid")
        for k, v in seen.items():
            out.append(f"{k:72s} " + " ".join(v))
    print("\n".join(out))


if __name__ == "__main__":
    main()
