#!/usr/bin/env python
"""Instruction histogram of one kernel in a hipcc -save-temps .s file (optionally per basic block).
  python tools/isa_hist.py file.s kernel_substring [--blocks]"""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    blocks = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + key + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    tot = collections.Counter()
    cur, curname = collections.Counter(), "entry"
    per = []
    for l in lines[start + 1:end]:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            if re.match(r"^\.LBB\d+_\d+:", t):
                per.append((curname, cur)); cur, curname = collections.Counter(), t.rstrip(":")
            continue
        if t.endswith(":"):
            continue
        op = t.split()[0]
        tot[op] += 1
        cur[op] += 1
    per.append((curname, cur))

    def cls(c):
        g = collections.Counter()
        for k, v in c.items():
            if k.startswith("v_"): g["valu"] += v
            elif k.startswith("ds_"): g["lds"] += v
            elif k.startswith("global_") or k.startswith("buffer_") or k.startswith("scratch_") or k.startswith("flat_"): g["vmem"] += v
            elif k.startswith("s_"): g["salu"] += v
            else: g["other"] += v
        return dict(g)
    if blocks:
        for name, c in per:
            n = sum(c.values())
            if n >= 20:
                print(f"{name:12s} n={n:5d} {cls(c)}")
                print("    " + ", ".join(f"{k}:{v}" for k, v in c.most_common(14)))
    print("TOTAL", sum(tot.values()), cls(tot))
    for k, v in tot.most_common(50):
        print(f"  {k:30s}{v}")


if __name__ == "__main__":
    main()
