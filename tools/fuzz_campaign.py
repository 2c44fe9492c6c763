#!/usr/bin/env python
"""Extended differential fuzzing on the device: the seeded fuzz tests of tests/test_gpu_fuzz.py called with seeds the test
suite does not use (the suite pins 8-40 seeds per family; this sweeps a few hundred more).  Failures are collected, not
raised: one line per failing (test, seed) with the assertion message.
    python tools/fuzz_campaign.py [--first 100 --count 150] > profiles/rNN_fuzz_campaign.txt"""
import argparse
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=100)
    ap.add_argument("--count", type=int, default=150)
    ap.add_argument("--budget-s", type=float, default=240.0)
    ap.add_argument("--only", default="", help="substring of the family names to run")
    args = ap.parse_args()
    import torch
    import test_gpu_fuzz as tf
    families = [n for n in dir(tf) if n.startswith("test_fuzz_") and "seed" in getattr(tf, n).__code__.co_varnames[:1]
                and args.only in n]
    print(f"# {torch.cuda.get_device_name(0)}; seeds {args.first} .. {args.first + args.count - 1} per family; families: {len(families)}")
    t_start = time.time()
    total = fails = 0
    for name in families:
        fn = getattr(tf, name)
        ok = bad = 0
        t0 = time.time()
        for seed in range(args.first, args.first + args.count):
            if time.time() - t_start > args.budget_s:
                break
            try:
                fn(seed)
                ok += 1
            except AssertionError as e:
                bad += 1
                print(f"FAIL {name} seed {seed}: {str(e)[:400]}", flush=True)
            except Exception as e:                            # noqa: BLE001 -- a crash of one case must not end the sweep
                bad += 1
                print(f"ERROR {name} seed {seed}: {type(e).__name__}: {str(e)[:300]}", flush=True)
                traceback.print_exc(limit=3, file=sys.stdout)
        total += ok + bad
        fails += bad
        print(f"{name:60s} {ok + bad:4d} cases, {bad} failed, {time.time() - t0:.1f} s", flush=True)
    print(f"# total {total} cases, {fails} failed")


if __name__ == "__main__":
    main()
