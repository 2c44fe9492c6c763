#!/usr/bin/env python
"""Split the melspec400_kernel dispatches of a `rocprofv3 --kernel-trace` run of tools/mfcc_launch_ab.py by the role they
played: a dispatch directly followed by mfcc_fix_list_kernel is a PASS 0, the one behind the list kernel the FIX-UP launch,
any other the ONE-LAUNCH form.  Prints count / average / min duration per role, and the gap to the next kernel.
    python tools/mfcc_trace_split.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        print("no kernel trace under", d)
        return
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    roles = {}
    for i, (s, e, name) in enumerate(rows):
        if "melspec400_kernel" not in name:
            key = name.split("(")[0][-60:]
        else:
            nxt = rows[i + 1][2] if i + 1 < len(rows) else ""
            prv = rows[i - 1][2] if i > 0 else ""
            if "mfcc_fix_list_kernel" in nxt:
                key = "melspec400 PASS 0 (three launches)"
            elif "mfcc_fix_list_kernel" in prv:
                key = "melspec400 FIX-UP launch (three launches)"
            else:
                key = "melspec400 ONE LAUNCH"
        gap = (rows[i + 1][0] - e) if i + 1 < len(rows) else 0
        roles.setdefault(key, []).append(((e - s) / 1e3, gap / 1e3))
    print(f"{'role':60s} {'n':>6s} {'avg us':>9s} {'min us':>9s} {'median':>9s} {'gap to next (median us)':>24s}")
    for key, v in sorted(roles.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
        du = sorted(x[0] for x in v)
        gp = sorted(x[1] for x in v)
        print(f"{key:60s} {len(v):6d} {sum(du) / len(du):9.2f} {du[0]:9.2f} {du[len(du) // 2]:9.2f} {gp[len(gp) // 2]:24.2f}")


if __name__ == "__main__":
    main()
